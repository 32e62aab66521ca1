import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, _tgraph
from sph3d_gcn_amd.harness import synth
dev=torch.device('cuda:0')
B,N,K=16,8192,64
xyz=torch.from_numpy(synth.s3dis_batch(1000,B,N)[0]).to(dev)
idx,cnt,dst=tf_nnquery.build_sphere_neighbor(xyz,xyz,0.1,None,K)
filt=tf_buildkernel.spherical_kernel(xyz,xyz,idx,cnt,dst,0.1,[8,2,2])
off,key,sc,_=_tgraph.transpose(idx,cnt,N,bin_index=filt,num_bins=33)
o=off.view(B,N*33+1).cpu().numpy()
seg=np.diff(o,axis=1).reshape(B,N,33)
print("segment length histogram (fraction):", np.bincount(seg.ravel(),minlength=10)[:12]/seg.size)
deg=seg.sum(2)
print("in-degree mean %.1f max %d  frac>64 %.4f"%(deg.mean(),deg.max(),(deg>64).mean()))
print("edges beyond 1 slot per seg: %.3f of edges; beyond 2: %.3f; beyond 3: %.3f"%tuple(np.maximum(seg-j,0).sum()/seg.sum() for j in (1,2,3)))
print("nonempty segs per source %.1f"%( (seg>0).sum(2).mean()))
print("per-bin mean len:", np.round(seg.mean((0,1)),2))
