"""Level-0 conv gradient (16 x 8192, K = 64, 6.45 M edges) at row widths C*r = 64 .. 512: time and gathered bytes per second — does the
gather rate depend on whether one cloud's grad_out (8192 rows) fits the 4-MB L2 of the XCD that sweeps it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, _tgraph
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K, N = 16, 64, 8192
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, N)[0]).to(dev)[:, :, :3].contiguous()
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, K)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, 0.1, [8, 2, 2])
edges = int(cnt.sum())
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("edges %d" % edges)
for C, r in ((32, 1), (32, 2), (64, 1), (64, 2), (128, 1), (128, 2), (256, 1), (256, 2)):
    x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, r, device=dev); go = torch.randn(B, N, C * r, device=dev)
    tb = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt))
    tf = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
    print("C %4d r %d row %5d B cloud %5.1f MB | grad %7.1f us  %5.1f TB/s gathered | fwd %7.1f us %5.1f TB/s (row %4d B)" % (
        C, r, C * r * 4, N * C * r * 4 / 1e6, tb, edges * C * r * 4 / tb / 1e6, tf, edges * C * 4 / tf / 1e6, C * 4))
