"""What the side streams cost the step: (a) the normal step (sampling + graph construction on two side streams, one
step ahead), (b) the same step with a prebuilt GraphPlan reused (main-stream feature path only: no FPS, no neighbour
search, no transposes), (c) graph construction alone (a fresh GraphPlan per iteration, including the transposed graphs
the backward pass would ask for, nothing on the main stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist
dev = torch.device('cuda:0'); _lib.lib()
pts, label, inner = bench.make_batch(0, dev)
cfg = s3dis_net.s3dis_config(8192)
model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
pred, _ = model(pts, True); model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4)
def run(fn, n=30, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a0 = run(lambda: bench.train_step(model, flat, opt, pts, label, inner))
torch.cuda.synchronize(); ev = torch.cuda.Event(); ev.record(); bench._PTS_READY[pts.data_ptr()] = ev      # as bench.py: side streams wait for the INPUT only
a = run(lambda: bench.train_step(model, flat, opt, pts, label, inner))
bench._PTS_READY.clear()
plan = s3dis_net.GraphPlan(pts, cfg)
def step_reuse():
    pred, _ = model(pts, is_training=True, graphs=plan)
    loss = model.loss(pred, label, inner)
    flat.backward(loss); flat.all_reduce(); opt.step()
b = run(step_reuse)
def graphs_only():
    s3dis_net.GraphPlan(pts, cfg)
c = run(graphs_only)
print("step with the side streams forked from the main stream %.2f ms" % a0)
print("normal step %.2f ms | feature path alone (plan reused) %.2f ms | graph construction alone %.2f ms" % (a, b, c))
