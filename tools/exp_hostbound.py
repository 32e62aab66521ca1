"""is the step host-bound?  host time to ISSUE a step (no sync) vs the step's wall time, with the allocator warm"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import dist as hdist, s3dis_net
dev = torch.device("cuda:0")
_lib.lib()
NB = int(os.environ.get("NB", "2"))
batches = [bench.make_batch(0, dev, w) for w in range(NB)]
torch.cuda.synchronize()
ev = torch.cuda.Event(); ev.record()
for bt in batches: bench._PTS_READY[bt[0].data_ptr()] = ev
pts, label, inner = batches[0]
model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(8192), device=dev)
graphs = s3dis_net.build_graphs(pts, model.config)
pred, _ = model(pts, is_training=True, graphs=graphs)
model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters())
opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4, fused=True)
n = [0]
def step():
    p, l, i = batches[n[0] % NB]; n[0] += 1
    return bench.train_step(model, flat, opt, p, l, i)
for _ in range(25): step()
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# host alone: sync before every step so the GPU is idle when issue starts (pure issue time per step)
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t1 = time.perf_counter(); step(); ts.append(time.perf_counter() - t1)
torch.cuda.synchronize()
print("per step: wall %.3f ms, host issue (running ahead) %.3f ms, host issue from idle %.3f ms" % (t_all / K * 1e3, t_issue / K * 1e3, sorted(ts)[len(ts)//2] * 1e3))
