import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_gemm
dev=torch.device('cuda:0')
def timeit(fn,n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
shapes=[(131072,3,64),(131072,128,128),(131072,256,128),(32768,256,256),(32768,512,256),(12288,512,256),(6144,512,512),(6144,1024,512),(2048,1024,512),(6144,2048,256),(12288,1024,256),(32768,1024,128),(131072,256,13)]
tot_h=tot_b=0
import os
for R,Ci,Co in shapes:
    x=torch.randn(R,Ci,device=dev); w=torch.randn(Ci,Co,device=dev); dy=torch.randn(R,Co,device=dev)
    fl=2*R*Ci*Co/1e9
    if Co <= 16 and Ci % 32 == 0 and tf_gemm.skinny_supported(R, Ci // 2, Ci // 2, Co):
        # the logits layer as the step runs it (csrc/skinny.hip: two operand halves, no concatenation); NT = the two half-width input gradients
        a1, a2 = x[:, :Ci // 2].contiguous(), x[:, Ci // 2:].contiguous()
        def nt():
            tf_gemm._pointwise_gemm_impl(dy, w[:Ci // 2], True); tf_gemm._pointwise_gemm_impl(dy, w[Ci // 2:], True)
        h=[timeit(lambda: tf_gemm._skinny_impl(a1,a2,w,None)), timeit(nt), timeit(lambda: tf_gemm._skinny_tn_impl(a1,a2,dy))]
    else:
        h=[timeit(lambda: tf_gemm._pointwise_gemm(x,w,False)), timeit(lambda: tf_gemm._pointwise_gemm(dy,w,True)), timeit(lambda: tf_gemm._pointwise_gemm_tn(x,dy))]
    b=[timeit(lambda: x@w), timeit(lambda: dy@w.t()), timeit(lambda: x.t()@dy)]
    tot_h+=sum(h); tot_b+=sum(b)
    print("R%6d Cin%5d Cout%4d  hip NN/NT/TN ms %.3f %.3f %.3f (%.0f/%.0f/%.0f TF) | blas %.3f %.3f %.3f (%.0f/%.0f/%.0f TF)"%(R,Ci,Co,h[0],h[1],h[2],fl/h[0],fl/h[1],fl/h[2],b[0],b[1],b[2],fl/b[0],fl/b[1],fl/b[2]))
print("total hip %.3f ms blas %.3f ms"%(tot_h,tot_b))
