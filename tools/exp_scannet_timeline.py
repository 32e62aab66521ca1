"""Timeline of the ScanNet-shape line (one 65 536-point block per step): when do the sampling chains of successive steps run, relative
to the steps' ends on the main stream?  HIP events of every C-ABI call (bench.py's event pass) against one reference event."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import dist as hdist, optim as hoptim, s3dis_net, synth
dev = torch.device("cuda:0"); _lib.lib()
s3dis_net.SAMPLING_STREAMS = int(os.environ.get("STREAMS", "2"))
NB = int(os.environ.get("BATCHES", "2"))
npts = 65536
cfg = s3dis_net.scannet_config(npts)
model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
rng = np.random.RandomState(17)
batches = []
for w in range(NB):
    xyz, label, inner = synth.s3dis_batch(7000 + w * 64, 1, npts, extent=(6.0, 6.0, 3.0))
    pts = np.concatenate([xyz, rng.rand(1, npts, 6).astype(np.float32)], axis=2)
    batches.append((torch.from_numpy(pts).to(dev), torch.from_numpy(rng.randint(0, cfg.num_cls, (1, npts))).to(dev), torch.from_numpy(inner).to(dev)))
torch.cuda.synchronize()
ready = torch.cuda.Event(); ready.record()
fwd = lambda b: model.loss(model(b[0], is_training=True, points_ready=ready)[0], b[1], b[2])
fwd(batches[0]).backward()
flat = hdist.FlatGradAllReduce(model.parameters())
opt = hoptim.FlatAdam(flat.flat_param, lr=1e-3, eps=1e-4)
n = [0]
def step():
    b = batches[n[0] % NB]; n[0] += 1
    loss = fwd(b); flat.backward(loss); flat.all_reduce(); opt.step()
for _ in range(20):
    step()
torch.cuda.synchronize()
import time
ref = torch.cuda.Event(enable_timing=True); ref.record()
_lib.timing_start()
ends, host = [], []
t0 = time.perf_counter()
for _ in range(int(os.environ.get("STEPS", "8"))):
    step()
    e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e); host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
ev = _lib.timing_stop()
print("streams %d, resident batches %d" % (s3dis_net.SAMPLING_STREAMS, NB))
print("step ends (device, ms):", ["%.1f" % ref.elapsed_time(e) for e in ends])
print("step issued (host, ms):", ["%.1f" % h for h in host])
for name, ints, e0, e1 in ev:
    if name == "sph3d_farthest_point_sample" and ints[1] == npts:
        print("   FPS 65536: start %.1f  end %.1f" % (ref.elapsed_time(e0), ref.elapsed_time(e1)))
