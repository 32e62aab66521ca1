"""Conv gradient at level 0 with a degree-balanced processing order: inside every window of 512 consecutive source points
(= one position per wave of an XCD) the sources are sorted by in-degree, descending in even windows and ascending in odd
ones, so that every wave's share of edges is about the same while the waves still sweep the cloud front to back."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, _tgraph
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
B, K, N, rad = 16, 64, 8192, 0.1
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, N)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, rad, [8, 2, 2])
offsets = _tgraph.transpose(nidx, cnt, N, filt, None, 33)[0]
def balanced_order(win):
    o = offsets.view(B, N * 33 + 1)
    deg = (o[:, 33::33] - o[:, 0:-1:33]).view(B, N // win, win)
    idx = torch.argsort(deg, dim=-1, descending=True)
    flip = (torch.arange(N // win, device=dev) % 2 == 1).view(1, -1, 1)
    idx = torch.where(flip, idx.flip(-1), idx)
    base = (torch.arange(N // win, device=dev) * win).view(1, -1, 1)
    return (idx + base).view(B, N).to(torch.int32).contiguous()
t_ord = timeit(lambda: balanced_order(512))
for C in (128, 64):
    x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev); go = torch.randn(B, N, C * 2, device=dev)
    res = []
    builtin = torch.empty((B, N), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().sph3d_graph_balanced_order(B, N, 33, _lib.ptr(offsets), _lib.ptr(builtin), _lib.stream_ptr()))
    for win in (0, 512, 1024, 2048, -1):
        _tgraph._orders.clear()
        if win > 0: _tgraph.set_source_order(nidx, balanced_order(win))
        if win < 0: _tgraph.set_source_order(nidx, builtin)
        gi, gf = tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
        res.append((win, timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)), gi, gf))
    d = max(float((res[0][2] - r[2]).abs().max()) for r in res[1:])
    print("C=%d: " % C + "  ".join("%s %.3f ms" % ("index order" if w_ == 0 else ("built-in" if w_ < 0 else "window %d" % w_), t) for w_, t, _, _ in res) + "  (max diff %.1e; order by torch ops %.3f ms)" % (d, t_ord), flush=True)
