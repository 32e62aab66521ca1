"""Unprofiled three-stream timeline of the bench step (timing events on every stream, no rocprof)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist
dev = torch.device('cuda:0'); _lib.lib()
pts, label, inner = bench.make_batch(0, dev)
torch.cuda.synchronize(); ev = torch.cuda.Event(); ev.record(); bench._PTS_READY[pts.data_ptr()] = ev
model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(8192), device=dev)
pred, _ = model(pts, True); model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4)
torch.cuda.synchronize()
for _ in range(20): bench.train_step(model, flat, opt, pts, label, inner)
torch.cuda.synchronize()

TR = []   # (label, event)
def mark(label, stream=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else torch.cuda.current_stream())
    TR.append((label, e))

GP = s3dis_net.GraphPlan
o_chain, o_build, o_sync = GP._sampling_chain, GP._build_all, GP._sync
def chain(self, side):
    if side is not None: mark("fps.begin", side)
    o_chain(self, side)
    if side is not None: mark("fps.end", side)
def build(self, stream):
    mark("graph.begin", stream); o_build(self, stream); mark("graph.end", stream)
def sync(self, key, ev, tensors):
    first = key not in self._synced
    if first: mark("main.before_%s%d" % key)
    o_sync(self, key, ev, tensors)
    if first: mark("main.after_%s%d" % key)
GP._sampling_chain, GP._build_all, GP._sync = chain, build, sync

t0 = time.perf_counter()
host = []
for i in range(6):
    mark("step%d.begin" % i)
    h0 = time.perf_counter()
    loss = bench.fwd_bwd(model, flat, pts, label, inner)
    mark("step%d.bwd_end" % i)
    flat.all_reduce(); opt.step()
    mark("step%d.end" % i)
    host.append((h0 - t0, time.perf_counter() - t0))
torch.cuda.synchronize()
print("wall ms/step", (time.perf_counter() - t0) / 6 * 1e3)
ref = TR[0][1]
for (lab, e) in TR:
    print("%9.3f  %s" % (ref.elapsed_time(e), lab))
print("host issue windows (ms):", ["%.2f-%.2f" % (a * 1e3, b * 1e3) for a, b in host])
