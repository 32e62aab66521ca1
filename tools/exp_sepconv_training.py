"""Training-mode separable layer, per layer shape of the S3DIS plan the ring kernel covers (16 blocks x 8192 points):
separate ops (depthwise kernel + GEMM with statistics epilogue) vs the one-kernel layer (csrc/sepring.hip), forward only (the
backward passes are identical), isolated on the GPU; then the inference layer: barrier kernel (sepconv.hip) vs ring kernel
(run the script with SPH3D_SC_RING=0 / 1).  usage: python tools/exp_sepconv_training.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_conv3d, tf_gemm, tf_norm, tf_sample
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B = 16
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)
levels = [(8192, 0.1, [(64, 128), (128, 128)]), (2048, 0.2, [(128, 256), (128, 128)])]
cur = xyz
print("SPH3D_SC_RING=%s SPH3D_SR_NB=%s" % (os.environ.get("SPH3D_SC_RING", "(default 1)"), os.environ.get("SPH3D_SR_NB", "(auto)")))
with torch.no_grad():
    for n, rad, shapes in levels:
        while cur.shape[1] > n:
            idx = tf_sample.farthest_point_sample(n, cur)
            cur = torch.gather(cur, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(cur, rad, 64, [8, 2, 2], with_transpose=False)
        for C, Cout in shapes:
            r = 2
            x = torch.randn(B, n, C, device=dev); dw = torch.randn(33, C, r, device=dev)
            w = torch.randn(C * r, Cout, device=dev) / (C * r) ** 0.5
            bias = torch.randn(Cout, device=dev); sc = torch.rand(Cout, device=dev) + 0.5; sh = torch.randn(Cout, device=dev)
            def conv_only():
                return tf_conv3d._depthwise_conv3d_impl(x, dw, nidx, cnt, filt)
            d0 = conv_only()
            def gemm_only():
                return tf_norm._gemm_bnstats_impl(d0.view(-1, C * r), w, None)
            def separate():
                d = tf_conv3d._depthwise_conv3d_impl(x, dw, nidx, cnt, filt)
                return tf_norm._gemm_bnstats_impl(d.view(-1, C * r), w, None)
            def fused():
                return tf_conv3d._separable_conv3d_train_impl(x, dw, w, None, nidx, cnt, filt)
            def infer():
                return tf_conv3d.separable_conv3d_fused(x, dw, w, nidx, cnt, filt, bias=bias, elu=True, scale=sc, shift=sh)
            tc, tg, ts, tf_, ti = timeit(conv_only), timeit(gemm_only), timeit(separate), timeit(fused), timeit(infer)
            y_s = separate()[0]; y_f = fused()[1].view(-1, Cout)
            print("rows=%6d C=%3d r=%d -> %3d | conv %6.1f + gemm(stats) %6.1f = separate %6.1f us | fused-train %6.1f us (%+.1f) | inference layer %6.1f us | max|dy| %.1e"
                  % (B * n, C, r, Cout, tc, tg, ts, tf_, tf_ - ts, ti, float((y_s - y_f).abs().max())))
print("ring give-ups:", _lib.lib().sph3d_separable_conv3d_ring_failures())
