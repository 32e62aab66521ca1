import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d
from sph3d_gcn_amd.harness import synth
dev=torch.device('cuda:0')
B,N,K=16,8192,64
xyz=torch.from_numpy(synth.s3dis_batch(1000,B,N)[0]).to(dev)
idx,cnt,dst=tf_nnquery.build_sphere_neighbor(xyz,xyz,0.1,None,K)
filt=tf_buildkernel.spherical_kernel(xyz,xyz,idx,cnt,dst,0.1,[8,2,2])
l=_lib.lib()
def timeit(fn,n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
def morton(xyz, bits=10):
    mn=xyz.min(1,keepdim=True)[0]; mx=xyz.max(1,keepdim=True)[0]
    q=((xyz-mn)/(mx-mn+1e-9)*((1<<bits)-1)).long()
    def spread(v):
        v=(v|(v<<16))&0x030000FF; v=(v|(v<<8))&0x0300F00F; v=(v|(v<<4))&0x030C30C3; v=(v|(v<<2))&0x09249249; return v
    return spread(q[...,0])|(spread(q[...,1])<<1)|(spread(q[...,2])<<2)
code=morton(xyz); order=torch.argsort(code,dim=1).int().contiguous()
C=128
x=torch.randn(B,N,C,device=dev); w=torch.randn(33,C,2,device=dev)
ref=tf_conv3d.depthwise_conv3d(x,w,idx,cnt,filt)
print("natural order ms",timeit(lambda: tf_conv3d.depthwise_conv3d(x,w,idx,cnt,filt)))
l._cdll.sph3d_debug_order.argtypes=[ctypes.c_void_p]
l._cdll.sph3d_debug_order(ctypes.c_void_p(order.data_ptr()))
out=tf_conv3d.depthwise_conv3d(x,w,idx,cnt,filt)
print("max diff",(out-ref).abs().max().item())
print("morton order ms",timeit(lambda: tf_conv3d.depthwise_conv3d(x,w,idx,cnt,filt)))
l._cdll.sph3d_debug_order(ctypes.c_void_p(0))
xyz2=torch.gather(xyz,1,order.long().unsqueeze(-1).expand(-1,-1,3)).contiguous()
idx2,cnt2,dst2=tf_nnquery.build_sphere_neighbor(xyz2,xyz2,0.1,None,K)
filt2=tf_buildkernel.spherical_kernel(xyz2,xyz2,idx2,cnt2,dst2,0.1,[8,2,2])
print("spatially sorted cloud: nnz",int(cnt2.sum()),"ms",timeit(lambda: tf_conv3d.depthwise_conv3d(x,w,idx2,cnt2,filt2)))
