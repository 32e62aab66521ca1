"""Inference separable layer at EVERY separable layer shape of the S3DIS plan (16 blocks x 8192 points): the one-kernel layer
(csrc/sepconv.hip: register-resident W for C <= 128 / Cout <= 128, the general kernel otherwise) vs depthwise kernel + GEMM
with bias/ELU epilogue + batch-norm affine; then the whole net's forward in eval mode, fused vs layer by layer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_conv3d, tf_gemm, tf_sample
from sph3d_gcn_amd import sph3gcn_util as s3g_util
from sph3d_gcn_amd.harness import synth, s3dis_net
dev = torch.device('cuda:0'); _lib.lib()
B = 16
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)
levels = [(8192, 0.1, [(64, 128), (128, 128)]), (2048, 0.2, [(128, 256), (256, 256), (512, 256)]),
          (768, 0.4, [(256, 256), (512, 256)]), (384, 0.8, [(256, 512), (512, 512), (1024, 512)])]
cur = xyz
tot_u = tot_f = 0.0
with torch.no_grad():
    for n, rad, shapes in levels:
        while cur.shape[1] > n:
            nxt = {8192: 2048, 2048: 768, 768: 384}[cur.shape[1]]
            idx = tf_sample.farthest_point_sample(nxt, cur)
            cur = torch.gather(cur, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(cur, rad, 64, [8, 2, 2], with_transpose=False)
        for C, Cout in shapes:
            r = 2
            x = torch.randn(B, n, C, device=dev); dw = torch.randn(33, C, r, device=dev)
            w = torch.randn(C * r, Cout, device=dev) / (C * r) ** 0.5
            bias = torch.randn(Cout, device=dev); sc = torch.rand(Cout, device=dev) + 0.5; sh = torch.randn(Cout, device=dev)
            def unfused():
                d = tf_conv3d._depthwise_conv3d_impl(x, dw, nidx, cnt, filt)
                y = tf_gemm._gemm_bias_act_impl(d.view(-1, C * r), w, bias, 1)
                return torch.addcmul(sh, y, sc).view(B, n, Cout)
            def fused():
                return tf_conv3d.separable_conv3d_fused(x, dw, w, nidx, cnt, filt, bias=bias, elu=True, scale=sc, shift=sh)
            tu, tf_ = timeit(unfused), timeit(fused)
            a, b = unfused(), fused()
            tot_u += tu; tot_f += tf_
            print("N=%5d C=%4d -> %3d: unfused %7.1f us, fused %7.1f us, max |diff| %.2e (scale %.1f)"
                  % (n, C, Cout, tu, tf_, float((a - b).abs().max()), float(a.abs().max())))
    print("sum over the layer shapes: unfused %.1f us, fused %.1f us" % (tot_u, tot_f))
    # whole net, eval mode
    model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(8192), device=dev)
    pts = xyz
    with torch.enable_grad():
        model(pts, is_training=True)[0].sum().backward()      # creates the variables, moves the statistics
    for fuse in ("auto", True, False):
        s3g_util.FUSE_SEPARABLE_INFERENCE = fuse
        t = timeit(lambda: model(pts, is_training=False)[0], 5)
        print("eval forward (graph build + 16 x 8192-pt blocks), FUSE_SEPARABLE_INFERENCE=%s: %.2f ms" % (fuse, t / 1e3))
    s3g_util.FUSE_SEPARABLE_INFERENCE = "auto"
