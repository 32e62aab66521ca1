"""One GEMM shape for PMC passes: SHAPE=R,Cin,Cout KIND=nn|nt|tn"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_gemm
dev = torch.device('cuda:0'); _lib.lib()
R, Ci, Co = [int(v) for v in os.environ.get("SHAPE", "131072,256,128").split(",")]
kind = os.environ.get("KIND", "nn")
x = torch.randn(R, Ci, device=dev); w = torch.randn(Ci, Co, device=dev); dy = torch.randn(R, Co, device=dev)
for _ in range(5):
    if kind == "nn": tf_gemm._pointwise_gemm_impl(x, w, False)
    elif kind == "nt": tf_gemm._pointwise_gemm_impl(dy, w, True)
    else: tf_gemm._pointwise_gemm_tn_impl(x, dy)
torch.cuda.synchronize()
