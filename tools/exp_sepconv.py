"""Inference separable layer at the S3DIS level-0 shapes (16 x 8192, K = 64): one fused kernel (csrc/sepconv.hip) vs the
depthwise kernel + the GEMM with the bias / ELU epilogue + the batch-norm affine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, tf_gemm
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K, N = 16, 64, 8192
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, N)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, K)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, 0.1, [8, 2, 2])
with torch.no_grad():
    for C, r, Cout in ((64, 2, 128), (128, 2, 128), (64, 2, 64), (128, 1, 128)):
        x = torch.randn(B, N, C, device=dev); dw = torch.randn(33, C, r, device=dev)
        w = torch.randn(C * r, Cout, device=dev) / (C * r) ** 0.5
        bias = torch.randn(Cout, device=dev); sc = torch.rand(Cout, device=dev) + 0.5; sh = torch.randn(Cout, device=dev)
        def unfused():
            d = tf_conv3d._depthwise_conv3d_impl(x, dw, nidx, cnt, filt)
            y = tf_gemm._gemm_bias_act_impl(d.view(-1, C * r), w, bias, 1)
            return torch.addcmul(sh, y, sc).view(B, N, Cout)
        def fused():
            return tf_conv3d.separable_conv3d_fused(x, dw, w, nidx, cnt, filt, bias=bias, elu=True, scale=sc, shift=sh)
        tdw = timeit(lambda: tf_conv3d._depthwise_conv3d_impl(x, dw, nidx, cnt, filt))
        tu, tf_ = timeit(unfused), timeit(fused)
        err = float((unfused() - fused()).abs().max())
        print("C=%3d r=%d Cout=%3d: depthwise alone %.3f ms, layer unfused %.3f ms, fused %.3f ms, max |diff| %.2e"
              % (C, r, Cout, tdw, tu, tf_, err))
