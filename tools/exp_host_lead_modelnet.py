"""exp_host_lead.py for the ModelNet line (32 x 10 000 points): host lead and device time of every step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import dist as hdist, optim as hoptim, modelnet_net, synth
dev = torch.device("cuda:0"); _lib.lib()
per_gpu, npts = 32, 10000
cfg = modelnet_net.modelnet_config(npts)
model = modelnet_net.SPH3DModelNet(cfg, device=dev)
rng = np.random.RandomState(17)
batches = [(torch.from_numpy(synth.modelnet_batch(1000 + w * 64 * per_gpu, per_gpu, npts)).to(dev),
            torch.from_numpy(rng.randint(0, 40, (per_gpu,))).to(dev)) for w in range(3)]
torch.cuda.synchronize()
ready = torch.cuda.Event(); ready.record()
fwd = lambda b: model.loss(model(b[0], is_training=True, points_ready=ready)[0], b[1])
fwd(batches[0]).backward()
flat = hdist.FlatGradAllReduce(model.parameters())
opt = hoptim.FlatAdam(flat.flat_param, lr=1e-3, eps=1e-4)
n = [0]
def step():
    b = batches[n[0] % 3]; n[0] += 1
    loss = fwd(b); flat.backward(loss); flat.all_reduce(); opt.step()
for _ in range(20):
    step()
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True); ref.record()
t0 = time.perf_counter()
marks, host = [], []
for _ in range(int(os.environ.get("STEPS", "60"))):
    step()
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
dev_t = [ref.elapsed_time(e) for e in marks]
d = [dev_t[0]] + [b - a for a, b in zip(dev_t[:-1], dev_t[1:])]
h = [host[0]] + [b - a for a, b in zip(host[:-1], host[1:])]
print("device step ms:", ["%.1f" % t for t in d])
print("host step ms  :", ["%.1f" % t for t in h])
print("lead          :", ["%.0f" % (a - b) for a, b in zip(dev_t, host)])
