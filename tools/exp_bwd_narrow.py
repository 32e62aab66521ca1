"""Conv gradient on the ModelNet plan's narrow level-0 shapes (rows of 64-136 outputs: the half / quarter-wave forms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
B, N, K, F = 32, 10000, 64, 33
xyz = torch.from_numpy(synth.modelnet_batch(1000, B, N)[:, :, :3].copy()).to(dev)
nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, 0.1, K, [8, 2, 2], with_transpose=False)
offsets, ent_key, ent_scale, active = _tgraph.transpose(nidx, cnt, N, bin_index=filt, num_bins=F)
order = _tgraph.source_order(nidx)
edges = int(cnt.sum())
for C, r in ((36, 2), (64, 1), (64, 2), (128, 1), (32, 1)):
    x = torch.randn(B, N, C, device=dev); w = torch.randn(F, C, r, device=dev); go = torch.randn(B, N, C * r, device=dev)
    gi = torch.empty_like(x); gf = torch.empty_like(w)
    wsb = l.sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r); ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
    def call():
        _lib.check(l.sph3d_depthwise_conv3d_grad_t(B, N, N, F, C, r, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale), _lib.ptr(order),
                                                   _lib.ptr(active), _lib.ptr(x), _lib.ptr(w), _lib.ptr(go), _lib.ptr(gi), _lib.ptr(gf), _lib.ptr(ws), wsb, _lib.stream_ptr()))
    t = timeit(call)
    print("%s C %3d r %d (rows of %3d outputs): %8.1f us  %.1f ps/edge  gathered %.2f TB/s" % (os.environ.get("TAG", ""), C, r, C * r, t, t * 1e6 / edges, edges * C * r * 4 / t / 1e6), flush=True)
