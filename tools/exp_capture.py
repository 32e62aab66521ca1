"""Whole-step HIP-graph capture experiment: eager three-stream step vs replay of a captured 1-step / 2-step graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib, _tgraph
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist

dev = torch.device("cuda", 0); torch.cuda.set_device(0); _lib.lib()
WORK = torch.cuda.Stream(); torch.cuda.set_stream(WORK)     # not the legacy default stream: autograd nodes remember their stream
batches = [bench.make_batch(0, dev, w) for w in range(2)]
torch.cuda.synchronize()
pts, label, inner = batches[0]
model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(bench.NUM_POINT), device=dev)
graphs = s3dis_net.build_graphs(pts, model.config)
pred, _ = model(pts, is_training=True, graphs=graphs)
model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters())
opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4, fused=True, capturable=True)

def step(p, l, i, ready=None):
    pred, _ = model(p, is_training=True, points_ready=ready)
    loss = model.loss(pred, l, i)
    flat.backward(loss)
    flat.all_reduce()
    opt.step()
    return loss

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

k = [0]
def eager():
    b = batches[k[0] % 2]; k[0] += 1
    return step(*b)
for _ in range(20): eager()
print("eager: %.3f ms/step" % timed(eager, 50), flush=True)

def capture(nsteps):
    static = [tuple(t.clone() for t in batches[j]) for j in range(nsteps)]
    g = torch.cuda.CUDAGraph()
    _tgraph.clear()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=WORK):
        cur = torch.cuda.current_stream()
        root = torch.cuda.Event(); root.record(cur)
        losses = [step(*static[j], ready=root) for j in range(nsteps)]
        for s in s3dis_net._side_stream[dev]:
            cur.wait_stream(s)
    _tgraph.clear()
    torch.cuda.synchronize()
    return g, static, losses

for nsteps in (1, 2):
    t0 = time.perf_counter()
    g, static, losses = capture(nsteps)
    print("captured %d-step graph in %.1f s" % (nsteps, time.perf_counter() - t0), flush=True)
    kk = [0]
    def replay():
        for j in range(nsteps):
            b = batches[(kk[0] + j) % 2]
            for d, s in zip(static[j], b): d.copy_(s)
        kk[0] += nsteps
        g.replay()
    for _ in range(5): replay()
    ms = timed(replay, 30) / nsteps
    print("%d-step graph replay: %.3f ms/step  loss %s" % (nsteps, ms, [float(x) for x in losses]), flush=True)
    del g
# ---- feature path only (prebuilt graph plan): eager vs replay ----
plan = s3dis_net.build_graphs(batches[0][0], model.config)
def feat_step():
    pred, _ = model(batches[0][0], is_training=True, graphs=plan)
    loss = model.loss(pred, batches[0][1], batches[0][2])
    flat.backward(loss); flat.all_reduce(); opt.step()
    return loss
for _ in range(5): feat_step()
print("feature path only, eager: %.3f ms/step" % timed(feat_step, 30), flush=True)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=WORK):
    feat_step()
torch.cuda.synchronize()
for _ in range(5): g.replay()
print("feature path only, graph replay: %.3f ms/step" % timed(g.replay, 30), flush=True)
del g
l = eager(); torch.cuda.synchronize(); print("eager after: loss %.4f" % float(l))
