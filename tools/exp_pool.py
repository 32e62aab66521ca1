"""Pooling / un-pooling kernels at the S3DIS level shapes (B=16): max pool fwd + bwd, mean interpolate fwd + bwd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_nnquery, tf_pool3d, tf_unpool3d, tf_sample, sph3gcn_util as u
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
B, K = 16, 64
xyz0 = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = [0, 0, 0, 0]
xyz = xyz0
for N, M, rad, C in ((8192, 2048, 0.1, 128), (2048, 768, 0.2, 256), (768, 384, 0.4, 256), (384, 128, 0.8, 512)):
    si = tf_sample.farthest_point_sample(M, xyz)
    pairs = torch.stack([torch.arange(B, device=dev).view(B, 1).expand(B, M).int(), si], dim=-1).contiguous()
    xyz_c = u.gather_nd(xyz, pairs)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
    pidx, pcnt = u.gather_nd(idx, pairs), u.gather_nd(cnt, pairs)
    x = torch.randn(B, N, C, device=dev, requires_grad=True)
    y = tf_pool3d.max_pool3d(x, pidx, pcnt)[0] if isinstance(tf_pool3d.max_pool3d(x, pidx, pcnt), tuple) else tf_pool3d.max_pool3d(x, pidx, pcnt)
    go = torch.randn_like(y)
    a = timeit(lambda: tf_pool3d.max_pool3d(x.detach(), pidx, pcnt))
    def bw():
        yy = tf_pool3d.max_pool3d(x, pidx, pcnt); yy = yy[0] if isinstance(yy, tuple) else yy
        return torch.autograd.grad(yy, x, go)
    b = timeit(bw) - a
    iidx, icnt, idst = tf_nnquery.build_sphere_neighbor(xyz_c, xyz, rad, None, K)
    xc = torch.randn(B, M, C, device=dev, requires_grad=True)
    z = tf_unpool3d.mean_interpolate(xc, iidx, icnt); gz = torch.randn_like(z)
    c = timeit(lambda: tf_unpool3d.mean_interpolate(xc.detach(), iidx, icnt))
    d = timeit(lambda: torch.autograd.grad(tf_unpool3d.mean_interpolate(xc, iidx, icnt), xc, gz)) - c
    print("N=%5d->%5d C=%3d: max pool fwd %.3f bwd %.3f | mean interpolate fwd %.3f bwd %.3f ms" % (N, M, C, a, b, c, d), flush=True)
    for i, v in enumerate((a, b, c, d)): tot[i] += v
    xyz = xyz_c
print("sum: max pool fwd %.3f bwd %.3f | mean interpolate fwd %.3f bwd %.3f ms" % tuple(tot))
