"""Conv gradient on ONE 65 536-point cloud (ScanNet-shape line, B = 1): in-degree statistics of the level-0 graph and the kernel's time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sph3d_gcn_amd import _lib, _tgraph, tf_nnquery
from sph3d_gcn_amd.harness import synth
dev = torch.device("cuda:0"); l = _lib.lib()
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for B, N, ext in ((1, 65536, (6.0, 6.0, 3.0)), (16, 8192, None)):
    xyz = synth.s3dis_batch(7000, B, N, extent=ext)[0] if ext else synth.s3dis_batch(1000, B, N)[0]
    xyz = torch.from_numpy(xyz[:, :, :3].copy()).to(dev)
    K, F = 64, 33
    nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, 0.1, K, [8, 2, 2], with_transpose=False)
    offsets, ent_key, ent_scale, active = _tgraph.transpose(nidx, cnt, N, bin_index=filt, num_bins=F)
    order = _tgraph.source_order(nidx)
    off = offsets.view(B, N * F + 1)[:, ::F].long()           # bound of every source's run
    deg = (off[:, 1:] - off[:, :-1]).float()
    edges = int(cnt.sum())
    qs = torch.quantile(deg.flatten(), torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)).tolist()
    print("B %d N %d: edges %d, nn_count mean %.1f; in-degree mean %.1f, median %.0f, p90 %.0f, p99 %.0f, p99.9 %.0f, max %.0f; active bins %d"
          % (B, N, edges, float(cnt.float().mean()), float(deg.mean()), *qs, float(deg.max()), int(active[0])), flush=True)
    for C, r in ((128, 2), (64, 2)):
        x = torch.randn(B, N, C, device=dev); w = torch.randn(F, C, r, device=dev); go = torch.randn(B, N, C * r, device=dev)
        gi = torch.empty_like(x); gf = torch.empty_like(w)
        wsb = l.sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r); ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
        for name, od in (("balanced order", order), ("index order", None)):
            def call():
                _lib.check(l.sph3d_depthwise_conv3d_grad_t(B, N, N, F, C, r, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale), _lib.ptr(od),
                                                           _lib.ptr(active), _lib.ptr(x), _lib.ptr(w), _lib.ptr(go), _lib.ptr(gi), _lib.ptr(gf), _lib.ptr(ws), wsb, _lib.stream_ptr()))
            t = timeit(call)
            print("   C %3d r %d %-15s: %8.1f us  %.1f ps/edge" % (C, r, name, t, t * 1e6 / edges), flush=True)
        if B == 1:
            # spatial orders (round 6): one cloud's grad_out (64 MB at C = 128) is far beyond an XCD's 4-MB L2, so WHERE consecutive
            # sources lie decides whether a row is fetched once or by every source that lists it
            mort = torch.empty((B, N), dtype=torch.int32, device=dev)
            _lib.check(l.sph3d_spatial_order(B, N, _lib.ptr(xyz), _lib.ptr(mort), _lib.stream_ptr()))
            # the kernel deals position p to XCD p % 8 (parts = 8 at B = 1): give XCD x the x-th eighth of the Morton sequence
            reg = mort.view(8, N // 8).t().contiguous().view(1, N)
            for name, od in (("morton order", mort), ("8 morton regions", reg)):
                def call():
                    _lib.check(l.sph3d_depthwise_conv3d_grad_t(B, N, N, F, C, r, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale), _lib.ptr(od),
                                                               _lib.ptr(active), _lib.ptr(x), _lib.ptr(w), _lib.ptr(go), _lib.ptr(gi), _lib.ptr(gf), _lib.ptr(ws), wsb, _lib.stream_ptr()))
                t = timeit(call)
                print("   C %3d r %d %-15s: %8.1f us  %.1f ps/edge" % (C, r, name, t, t * 1e6 / edges), flush=True)
