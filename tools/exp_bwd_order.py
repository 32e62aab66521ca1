"""Conv gradient per S3DIS level with and without a spatial (Morton-cell) processing order of the source points (B=16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, tf_sample, _tgraph, _plan
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); l = _lib.lib()
B, K = 16, 64
xyz0 = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
levels = [(8192, 0.1, (64, 128)), (2048, 0.2, (128, 256, 512)), (768, 0.4, (256, 512)), (384, 0.8, (256, 512, 1024)), (128, 1.6, (512,))]
xyz = xyz0
tot = [0.0, 0.0]
for N, rad, Cs in levels:
    if xyz.shape[1] != N:
        idx = tf_sample.farthest_point_sample(N, xyz)
        xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
    filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, rad, [8, 2, 2])
    t_order = timeit(lambda: (_plan._orders.clear(), _plan.spatial_order(xyz)))
    order = _plan.spatial_order(xyz)
    for C in Cs:
        x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev); go = torch.randn(B, N, C * 2, device=dev)
        res = []
        for v in (0, 1):
            _tgraph._orders.clear()
            if v: _tgraph.set_source_order(nidx, order)
            gi, gf = tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
            t = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt))
            res.append((t, gi, gf)); tot[v] += t
        d = max(float((res[0][1] - res[1][1]).abs().max()), float((res[0][2] - res[1][2]).abs().max() / res[0][2].abs().max()))
        print("N=%5d C=%4d  index order %.3f ms  spatial order %.3f ms  (diff %.1e; order kernel %.3f ms)" % (N, C, res[0][0], res[1][0], d, t_order), flush=True)
print("sum index %.3f spatial %.3f" % tuple(tot))
