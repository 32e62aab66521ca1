"""Per-layer conv fwd / bwd times at the S3DIS level shapes (B=16), outside the training step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, tf_sample, _tgraph
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K = 16, 64
xyz0 = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
levels = [(8192, 0.1, (64, 128)), (2048, 0.2, (128, 256)), (768, 0.4, (256,)), (384, 0.8, (256, 512))]
xyz = xyz0
tf_tot = tb_tot = 0.0
for N, rad, Cs in levels:
    if xyz.shape[1] != N:
        idx = tf_sample.farthest_point_sample(N, xyz)
        xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
    filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, rad, [8, 2, 2])
    for C in Cs:
        x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev); go = torch.randn(B, N, C * 2, device=dev)
        tf = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
        tb = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt))
        tf_tot += tf; tb_tot += tb
        print("N=%5d C=%4d nnz=%8d  fwd %.3f ms  bwd %.3f ms" % (N, C, int(cnt.sum()), tf, tb))
print("sum fwd %.3f  bwd %.3f" % (tf_tot, tb_tot))
