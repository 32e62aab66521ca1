import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist
dev=torch.device('cuda:0'); _lib.lib()
pts,label,inner=bench.make_batch(0,dev)
torch.cuda.synchronize(); ev=torch.cuda.Event(); ev.record(); bench._PTS_READY[pts.data_ptr()]=ev
model=s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(8192),device=dev)
pred,_=model(pts,True); model.loss(pred,label,inner).backward()
flat=hdist.FlatGradAllReduce(model.parameters()); opt=torch.optim.Adam([flat.flat_param],lr=1e-3,eps=1e-4)
torch.cuda.synchronize()
ts=[]
import gc
gc.callbacks.append(lambda phase,info: print('GC',phase,info) if phase=='stop' and info['generation']==2 else None)
for i in range(100):
    t0=time.perf_counter(); bench.train_step(model,flat,opt,pts,label,inner); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print("per-step (sync each) ms:", " ".join("%.1f"%t for t in ts)); print("spikes at", [i for i,t in enumerate(ts) if t>25])
print("mem allocated GB %.2f reserved GB %.2f"%(torch.cuda.memory_allocated()/1e9, torch.cuda.memory_reserved()/1e9))
st=torch.cuda.memory_stats(); print("num_alloc_retries",st.get("num_alloc_retries"),"segments",st.get("segment.all.current"), "cudaMalloc calls", st.get("num_device_alloc"))
# unsynced windows
for W,K in [(0,8),(0,10),(0,10)]:
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(K): bench.train_step(model,flat,opt,pts,label,inner)
    torch.cuda.synchronize(); print("window K=%d: %.2f ms/step"%(K,(time.perf_counter()-t0)/K*1e3), "device allocs", torch.cuda.memory_stats().get("num_device_alloc"))
