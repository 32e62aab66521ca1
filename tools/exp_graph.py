import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, _tgraph, tf_gemm
from sph3d_gcn_amd.harness import s3dis_net, synth, dist as hdist
dev=torch.device('cuda:0')
mode=sys.argv[1] if len(sys.argv)>1 else "fwdbwd"
overlap = (len(sys.argv)<=2 or sys.argv[2]!="nooverlap")
cfg=s3dis_net.small_config(1024)
xyz,label,inner=synth.s3dis_batch(0,2,1024,extent=(1.0,1.0,1.5))
pts=torch.from_numpy(xyz).to(dev); label=torch.from_numpy(label).to(dev); inner=torch.from_numpy(inner).to(dev)
model=s3dis_net.SPH3DS3DIS(cfg,device=dev)
def fwd():
    plan=s3dis_net.GraphPlan(pts,model.config,overlap=overlap)
    pred,_=model(pts,True,graphs=plan)
    return model.loss(pred,label,inner)
loss=fwd(); loss.backward()
flat=hdist.FlatGradAllReduce(model.parameters())
def step():
    flat.zero()
    l=fwd()
    if mode=="fwdbwd": l.backward()
    return l
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
_tgraph.clear()
print("capturing", mode, "overlap", overlap, flush=True)
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode=os.environ.get("CAPMODE","global")):
    l=step()
print("captured", flush=True)
_tgraph.clear()
torch.cuda.synchronize()
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay",i,float(l), float(flat.flat.abs().sum()), flush=True)
ref=step(); torch.cuda.synchronize(); print("eager",float(ref), float(flat.flat.abs().sum()))
