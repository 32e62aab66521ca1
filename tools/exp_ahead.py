"""How far does the host run ahead of the GPU in the training loop?  Host loop time (no sync) vs total time, and the lag
between issuing the end of step t and the GPU finishing it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist
dev = torch.device('cuda:0'); _lib.lib()
batches = [bench.make_batch(0, dev, w) for w in range(2)]
torch.cuda.synchronize(); ev = torch.cuda.Event(); ev.record()
for b in batches: bench._PTS_READY[b[0].data_ptr()] = ev
cfg = s3dis_net.s3dis_config(8192)
model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
pts, label, inner = batches[0]
pred, _ = model(pts, True); model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4, fused=True)
k = [0]
def step():
    b = batches[k[0] % 2]; k[0] += 1
    bench.train_step(model, flat, opt, *b)
for _ in range(30): step()
torch.cuda.synchronize()
N = 60
evs = []; th = []
t0 = time.perf_counter()
for i in range(N):
    step()
    e = torch.cuda.Event(enable_timing=False); e.record(); evs.append(e)
    th.append(time.perf_counter() - t0)
t_host = time.perf_counter() - t0
# when does the GPU finish each step?  poll (host is done issuing)
tg = [None] * N
while any(x is None for x in tg):
    now = time.perf_counter() - t0
    for i, e in enumerate(evs):
        if tg[i] is None and e.query(): tg[i] = now
t_total = time.perf_counter() - t0
print("host loop %.1f ms (%.2f ms/step), total %.1f ms (%.2f ms/step)" % (t_host * 1e3, t_host / N * 1e3, t_total * 1e3, t_total / N * 1e3))
print("host step times (ms):", " ".join("%.1f" % ((th[i] - (th[i - 1] if i else 0)) * 1e3) for i in range(N)))
