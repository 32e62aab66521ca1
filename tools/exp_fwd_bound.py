"""What bounds the conv forward gather: level-0 kernels with the neighbour ids folded onto fewer distinct rows per cloud
(ids modulo R: same instruction stream and edge lists, shrinking memory footprint)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_nnquery, tf_buildkernel, tf_conv3d
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
for C in (128, 64):
    x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev)
    for R in (8192, 2048, 512, 64, 1):
        idx = (nidx % R).contiguous()
        t = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt))
        print("C=%d distinct input rows per cloud %5d: %.3f ms" % (C, R, t), flush=True)
    one_bin = torch.zeros_like(filt)
    t = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, one_bin))
    print("C=%d real rows, every edge in bin 0 (one filter row): %.3f ms" % (C, t), flush=True)
