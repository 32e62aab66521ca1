import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, _tgraph
from sph3d_gcn_amd.harness import synth
dev=torch.device('cuda:0')
B,N,K=16,8192,64
xyz=torch.from_numpy(synth.s3dis_batch(1000,B,N)[0]).to(dev)
idx,cnt,dst=tf_nnquery.build_sphere_neighbor(xyz,xyz,0.1,None,K)
filt=tf_buildkernel.spherical_kernel(xyz,xyz,idx,cnt,dst,0.1,[8,2,2])
print("nnz",int(cnt.sum()))
l=_lib.lib()
def timeit(fn,n=5):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
def morton_order(xyz, bits=10):
    lo = xyz.amin(dim=1, keepdim=True); hi = xyz.amax(dim=1, keepdim=True)
    q = ((xyz - lo) / (hi - lo).clamp(min=1e-9) * (2**bits - 1)).long()
    code = torch.zeros(xyz.shape[:2], dtype=torch.long, device=xyz.device)
    for i in range(bits):
        for a in range(3):
            code |= ((q[..., a] >> i) & 1) << (3 * i + a)
    return torch.argsort(code, dim=1).to(torch.int32).contiguous()
MODE = os.environ.get("ORDER", "morton")
if MODE == "morton":
    _tgraph.set_source_order(idx, morton_order(xyz))
print("order mode", MODE)
for C in (128,64):
    x=torch.randn(B,N,C,device=dev); w=torch.randn(33,C,2,device=dev); go=torch.randn(B,N,C*2,device=dev)
    print("C",C,"fwd ms",timeit(lambda: tf_conv3d.depthwise_conv3d(x,w,idx,cnt,filt)))
    print("  bwd_t ms",timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x,w,go,idx,cnt,filt)))
