import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_gemm
dev=torch.device('cuda:0')
R,Ci,Co=131072,256,128
x=torch.randn(R,Ci,device=dev); w=torch.randn(Ci,Co,device=dev)
for _ in range(5): y=tf_gemm._pointwise_gemm_impl(x,w,False)
torch.cuda.synchronize()
