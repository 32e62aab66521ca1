"""Graph construction issued ONE STEP AHEAD: the GraphPlan of batch t+1 is queued on the side streams before the feature path
of batch t is issued (the hipLaunchKernel back-pressure keeps the host only a few hundred launches ahead of the GPU, so side
work issued at the top of its own step reaches the GPU too late to hide the 2.9-ms sampling chain)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib, _tgraph
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist
dev = torch.device('cuda:0'); _lib.lib()
_tgraph._MAX_ENTRIES = 40
batches = [bench.make_batch(0, dev, w) for w in range(2)]
torch.cuda.synchronize(); ev = torch.cuda.Event(); ev.record()
for b in batches: bench._PTS_READY[b[0].data_ptr()] = ev
cfg = s3dis_net.s3dis_config(8192)
model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
pts, label, inner = batches[0]
pred, _ = model(pts, True); model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4, fused=True)
def run(fn, n=60, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
k = [0]
def normal():
    b = batches[k[0] % 2]; k[0] += 1
    bench.train_step(model, flat, opt, *b)
print("normal step: %.2f ms" % run(normal), flush=True)
state = {"plan": None}
def prefetched(where):
    def fn():
        b = batches[k[0] % 2]; nb = batches[(k[0] + 1) % 2]; k[0] += 1
        plan = state["plan"]
        if plan is None:
            plan = s3dis_net.GraphPlan(b[0], cfg, points_ready=ev)
        if where == "top":
            state["plan"] = s3dis_net.GraphPlan(nb[0], cfg, points_ready=ev)
        pred, _ = model(b[0], is_training=True, graphs=plan)
        loss = model.loss(pred, b[1], b[2])
        if where == "mid":
            state["plan"] = s3dis_net.GraphPlan(nb[0], cfg, points_ready=ev)
        flat.backward(loss); flat.all_reduce(); opt.step()
    return fn
for where in ("top", "mid"):
    state["plan"] = None
    print("plan of batch t+1 issued at the %s of step t: %.2f ms" % (where, run(prefetched(where))), flush=True)
