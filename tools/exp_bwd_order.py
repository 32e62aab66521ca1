"""Conv gradient (dwconv_bwd_t_vec) at the S3DIS level shapes under different SOURCE PROCESSING ORDERS: index order, the library's
degree-balanced windows of the index order, Morton order, Morton order with degree-balanced windows of several sizes.  The
gradient gathers ~48 grad_out rows of 1 KB per source; in index order (the blocks' points are randomly permuted) a cloud's 8 MB of
grad_out rows cycle through the XCD's 4 MB L2 (TCC hit 75 %, 5x the algorithmic HBM traffic: profiles/r04_pmc_bwd_*)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, _tgraph, tf_nnquery, tf_sample
from sph3d_gcn_amd.harness import synth

dev = torch.device("cuda:0"); l = _lib.lib()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def morton(xyz, bits):
    lo = xyz.amin(dim=1, keepdim=True); hi = xyz.amax(dim=1, keepdim=True)
    ext = (hi - lo).amax(dim=2, keepdim=True).clamp(min=1e-9)
    q = ((xyz - lo) / ext * (1 << bits)).long().clamp(0, (1 << bits) - 1)
    code = torch.zeros(xyz.shape[:2], dtype=torch.long, device=xyz.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, :, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, dim=1, stable=True)


def balance(order, deg, window):
    """inside every window of `window` consecutive positions: sources by in-degree, descending in even windows, ascending in odd"""
    B, N = order.shape
    out = order.clone()
    d = torch.gather(deg, 1, order)
    for w0 in range(0, N, window):
        w1 = min(N, w0 + window)
        idx = torch.argsort(d[:, w0:w1], dim=1, descending=((w0 // window) % 2 == 0), stable=True)
        out[:, w0:w1] = torch.gather(order[:, w0:w1], 1, idx)
    return out


def run(N, radius, C, xyz):
    B, K, F, r = xyz.shape[0], 64, 33, 2
    nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, radius, K, [8, 2, 2], with_transpose=False)
    offsets, ent_key, ent_scale, active = _tgraph.transpose(nidx, cnt, N, bin_index=filt, num_bins=F)
    lib_order = _tgraph.source_order(nidx)
    off = offsets.view(B, N * F + 1).long()
    deg = off[:, F::F] - off[:, 0:N * F:F]
    x = torch.randn(B, N, C, device=dev); w = torch.randn(F, C, r, device=dev); go = torch.randn(B, N, C * r, device=dev)
    gi = torch.empty_like(x); gf = torch.empty_like(w)
    wsb = l.sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)

    def call(order):
        _lib.check(l.sph3d_depthwise_conv3d_grad_t(B, N, N, F, C, r, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale),
                                                   _lib.ptr(order), _lib.ptr(active), _lib.ptr(x), _lib.ptr(w), _lib.ptr(go),
                                                   _lib.ptr(gi), _lib.ptr(gf), _lib.ptr(ws), wsb, _lib.stream_ptr()))
    ref_gi = None
    variants = [("index", None), ("library (balanced 2048-windows of the index order)", lib_order)]
    for bits in (4, 6):
        mo = morton(xyz, bits)
        variants.append(("morton%d" % bits, mo.int().contiguous()))
        for win in (128, 256, 512, 1024, 2048):
            variants.append(("morton%d + balanced %d" % (bits, win), balance(mo, deg, win).int().contiguous()))
    for name, order in variants:
        t = timeit(lambda: call(order))
        call(order); torch.cuda.synchronize()
        if ref_gi is None:
            ref_gi = gi.clone()
        err = float((gi - ref_gi).abs().max())
        print("N %5d C %4d  %-52s %8.1f us   (max |d grad_in| vs index order %.2e)" % (N, C, name, t, err), flush=True)


xyz0 = torch.from_numpy(synth.s3dis_batch(1000, 16, 8192)[0][:, :, :3].copy()).to(dev)
which = sys.argv[1:] or ["l0c128", "l0c64", "l1c256"]
if "l0c128" in which:
    run(8192, 0.1, 128, xyz0)
if "l0c64" in which:
    run(8192, 0.1, 64, xyz0)
if "l1c256" in which:
    idx = tf_sample.farthest_point_sample(2048, xyz0).long()
    xyz1 = torch.gather(xyz0, 1, idx.unsqueeze(2).expand(-1, -1, 3)).contiguous()
    run(2048, 0.2, 256, xyz1)
