"""Tiled vs gather depthwise convolution forward at the S3DIS level shapes (B=16): times, plan build time, max diff."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, tf_sample, _plan
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K = 16, 64
LEVELS = os.environ.get("LEVELS", "0,1,2,3")
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
for li, (N, rad, Cs) in enumerate(levels):
    if xyz.shape[1] != N:
        idx = tf_sample.farthest_point_sample(N, xyz)
        xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    if str(li) not in LEVELS.split(","):
        continue
    nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
    filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, rad, [8, 2, 2])
    def plan_f():
        _plan._fwd.clear(); _plan._orders.clear()
        _plan.forward_plan(nidx, cnt, filt, 33)
    print("N=%5d nnz=%8d  plan %.3f ms" % (N, int(cnt.sum()), timeit(plan_f, 3)))
    p = _plan.forward_plan(nidx, cnt, filt, 33)
    h = p[0].view(-1, 2)
    print("   first sub-tiles: mean targets %.1f rows %.1f | extra steps %d" % (float(h[:, 0].float().mean()), float(h[:, 1].float().mean()), int(p[6][1:].sum())))
    for C in Cs:
        x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev)
        _plan.set_mode("gather")
        ref = tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
        tg = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
        _plan.set_mode("tiled")
        out = tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
        tt = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
        print("  C=%4d gather %.3f ms | tiled %.3f ms (max diff %.1e)" % (C, tg, tt, float((out - ref).abs().max())), flush=True)
    _plan.set_mode("gather")
