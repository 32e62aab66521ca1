"""Tiled vs gather depthwise convolution at the S3DIS level shapes (B=16): times per variant, plan build times, max diff."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, tf_sample, _tgraph, _plan
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K = 16, 64
LEVELS = os.environ.get("LEVELS", "0,1,2,3")
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "208,216,408,416").split(",")]
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
    def plan_b():
        _plan._bwd.clear(); _plan._orders.clear()
        _plan.backward_plan(nidx, cnt, filt, 33, N)
    _tgraph.transpose(nidx, cnt, N, bin_index=filt, num_bins=33)
    print("N=%5d nnz=%8d  plan fwd %.3f ms  plan bwd %.3f ms" % (N, int(cnt.sum()), timeit(plan_f, 3), timeit(plan_b, 3)))
    p = _plan.forward_plan(nidx, cnt, filt, 33)
    d = p[1].view(B, -1, 33)
    print("   fwd tiles: g hist", torch.bincount(d[:, :, 0].flatten(), minlength=17).tolist()[1:], " direct subtiles", int((d[:, :, 1::2] < 0).sum()),
          " mean U of g=16", float(d[:, :, 1][d[:, :, 0] == 16].float().mean()))
    p = _plan.backward_plan(nidx, cnt, filt, 33, N)
    d = p[1].view(B, -1, 33)
    print("   bwd tiles: g hist", torch.bincount(d[:, :, 0].flatten(), minlength=17).tolist()[1:], " direct subtiles", int((d[:, :, 1::2] < 0).sum()))
    for C in Cs:
        x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev); go = torch.randn(B, N, C * 2, device=dev)
        _plan.set_mode("direct")
        ref = tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
        rgi, rgf = tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
        tf = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
        tb = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt))
        line = "  C=%4d gather: fwd %.3f bwd %.3f |" % (C, tf, tb)
        for v in VARIANTS:
            _plan.set_mode("auto", v)
            try:
                out = tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
                df = float((out - ref).abs().max())
                tfv = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
            except Exception as e:
                df, tfv = -1, -1
            try:
                gi, gf = tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
                db = max(float((gi - rgi).abs().max()), float((gf - rgf).abs().max() / rgf.abs().max()))
                tbv = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt))
            except Exception as e:
                db, tbv = -1, -1
            line += " v%d: fwd %.3f (d %.1e) bwd %.3f (d %.1e) |" % (v, tfv, df, tbv, db)
        print(line, flush=True)
    _plan.set_mode("auto")
