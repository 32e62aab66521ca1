"""Gaps between consecutive kernels of the feature path when it runs ALONE (the graph plan prebuilt and reused: no side-stream work):
torch profiler kernel timestamps of a few steps — what the main stream's queue costs beyond its kernels."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import dist as hdist, optim as hoptim, s3dis_net, synth
dev = torch.device("cuda:0"); _lib.lib()
B = 16
xyz, label, inner = synth.s3dis_batch(1000, B, 8192)
pts, label, inner = torch.from_numpy(xyz).to(dev), torch.from_numpy(label).to(dev), torch.from_numpy(inner).to(dev)
model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(8192), device=dev)
reuse = os.environ.get("REUSE", "1") == "1"
plan = s3dis_net.build_graphs(pts, model.config) if reuse else None
pred, _ = model(pts, is_training=True, graphs=plan)
model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters())
opt = hoptim.FlatAdam(flat.flat_param, lr=1e-3, eps=1e-4)
def step():
    pred, _ = model(pts, is_training=True, graphs=plan)
    loss = model.loss(pred, label, inner)
    flat.backward(loss); flat.all_reduce(); opt.step()
for _ in range(20):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
prof.export_chrome_trace("/tmp/trace_main.json")
tr = json.load(open("/tmp/trace_main.json"))
evs = [e for e in tr["traceEvents"] if e.get("cat") in ("kernel", "gpu_memset", "gpu_memcpy")]
evs.sort(key=lambda e: e["ts"])
adam = [i for i, e in enumerate(evs) if "adam" in e["name"]]
a, b = adam[1], adam[2]
seg = evs[a + 1:b + 1]
span = seg[-1]["ts"] + seg[-1]["dur"] - (evs[a]["ts"] + evs[a]["dur"])
busy = sum(e["dur"] for e in seg)
gaps = []
prev = evs[a]
for e in seg:
    gaps.append((e["ts"] - (prev["ts"] + prev["dur"]), prev["name"][:50], e["name"][:50]))
    prev = e
print("plan %s: one step = %d kernels, span %.1f us, busy %.1f us, gaps %.1f us" % ("reused (main stream alone)" if reuse else "built per step", len(seg), span, busy, span - busy))
hist = [sum(1 for g in gaps if lo <= g[0] < hi) for lo, hi in ((-1e9, 0.5), (0.5, 2), (2, 5), (5, 10), (10, 50), (50, 1e9))]
print("gap histogram (<0.5, 0.5-2, 2-5, 5-10, 10-50, >50 us):", hist)
for g in sorted(gaps, reverse=True)[:14]:
    print("   %7.1f us  after %-50s before %s" % g)
