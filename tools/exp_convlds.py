"""LDS-tile forward (csrc/convlds.hip) vs the gather kernel at the level shapes of the S3DIS plan: isolated kernel times,
plan / spatial-order build times.  SPH3D_LC_DBG=1: no staging, 2: no gather (phase timing, results wrong)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_conv3d, tf_nnquery, tf_sample, _plan, _lib
from sph3d_gcn_amd.harness import synth
dev = torch.device("cuda:0")
B = 16
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
levels = [(8192, 0.1, (64, 128)), (2048, 0.2, (128, 256, 512)), (768, 0.4, (256, 512)), (384, 0.8, (256, 512, 1024))]
only0 = len(sys.argv) > 1 and sys.argv[1] == "l0"
cur = xyz
print("dbg", os.environ.get("SPH3D_LC_DBG", "0"), "waves", os.environ.get("SPH3D_LC_WAVES", "16"))
for n, rad, cs in levels:
    while cur.shape[1] > n:
        nxt = {8192: 2048, 2048: 768, 768: 384}[cur.shape[1]]
        idx = tf_sample.farthest_point_sample(nxt, cur)
        cur = torch.gather(cur, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    _plan.set_mode("lds")
    nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(cur, rad, 64, [8, 2, 2], with_transpose=False)
    def plan_only():
        _plan._plans.clear()
        _plan.conv_plan(nidx, cnt, filt, 33, n)
    def order_only():
        _plan._orders.clear()
        _plan.spatial_order(cur)
    t_plan = timeit(plan_only, 10)
    t_order = timeit(order_only, 10)
    pl = _plan.conv_plan(nidx, cnt, filt, 33, n)
    hdr = pl[0].cpu().numpy().reshape(B, -1, 132)
    mt = pl[2].cpu().numpy().reshape(-1, 2)[:, 1]
    nt = int(hdr[:, :, 0].sum())
    a = hdr[:, :, 1::2]
    T = ((a >> 8) & 0xff)[a != 0]; U = (a >> 16)[a != 0]
    print("N=%d: plan %.1f us, order %.1f us, tiles %d, targets/tile %.1f, rows/tile %.1f, avg cnt %.1f, avg pairs %.1f" %
          (n, t_plan, t_order, nt, T.mean(), U.mean(), float(cnt.float().mean()), float((mt >> 8).sum()) / (B * n)))
    for C in cs:
        x = torch.randn(B, n, C, device=dev); w = torch.randn(33, C, 2, device=dev)
        _plan.set_mode("lds")
        tl = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
        o1 = tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
        _plan.set_mode("gather")
        tg = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))
        o2 = tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
        print("  %5d x C=%4d: lds %7.1f us   gather %7.1f us   max|diff| %.2e" % (n, C, tl, tg, float((o1 - o2).abs().max())))
    if only0:
        break
