"""How far the issuing thread runs ahead of the device over a run of headline steps, and the device time of every step: with the
sampling / graph streams busy a step takes 9-10 ms, alone on the device 7.2 (the last steps of a run, whose plans were built earlier)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import dist as hdist, optim as hoptim, s3dis_net, synth
dev = torch.device("cuda:0"); _lib.lib()
if os.environ.get("MAINSTREAM") == "1":          # experiment: the feature path on a created stream instead of the null stream
    torch.cuda.set_stream(torch.cuda.Stream())
B = 16
batches = []
for w in range(3):
    xyz, label, inner = synth.s3dis_batch(1000 + 64 * w, B, 8192)
    batches.append((torch.from_numpy(xyz).to(dev), torch.from_numpy(label).to(dev), torch.from_numpy(inner).to(dev)))
torch.cuda.synchronize()
for bt in batches:
    ev = torch.cuda.Event(); ev.record(); bench._PTS_READY[bt[0].data_ptr()] = ev
model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(8192), device=dev)
pred, _ = model(batches[0][0], is_training=True)
model.loss(pred, batches[0][1], batches[0][2]).backward()
flat = hdist.FlatGradAllReduce(model.parameters())
opt = hoptim.FlatAdam(flat.flat_param, lr=1e-3, eps=1e-4)
n = [0]
def step():
    p, l, i = batches[n[0] % 3]; n[0] += 1
    return bench.train_step(model, flat, opt, p, l, i)
for _ in range(30):
    step()
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True); ref.record()
t0 = time.perf_counter()
marks, host = [], []
for _ in range(int(os.environ.get("STEPS", "16"))):
    step()
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
dev_t = [ref.elapsed_time(e) for e in marks]
print("host issued by:", ["%.1f" % t for t in host])
print("device end at :", ["%.1f" % t for t in dev_t])
print("lead          :", ["%.1f" % (d - h) for h, d in zip(host, dev_t)])
d = [dev_t[0]] + [b - a for a, b in zip(dev_t[:-1], dev_t[1:])]
print("device step ms:", ["%.1f" % t for t in d])
h = [host[0]] + [b - a for a, b in zip(host[:-1], host[1:])]
print("host step ms  :", ["%.1f" % t for t in h])
