"""Host-side cost of issuing one training step: wall time to issue vs to finish, and a cProfile of the issuing thread."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist
dev = torch.device('cuda:0'); _lib.lib()
pts, label, inner = bench.make_batch(0, dev)
model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(8192), device=dev)
pred, _ = model(pts, True); model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4)
for _ in range(20): bench.train_step(model, flat, opt, pts, label, inner)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): bench.train_step(model, flat, opt, pts, label, inner)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue ms/step %.2f   total ms/step %.2f" % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
# issue cost with an idle queue: synchronize before every step so the host never blocks on a full queue
ts = []
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter()
    bench.train_step(model, flat, opt, pts, label, inner)
    ts.append(time.perf_counter() - a)
print("issue ms/step with an empty queue: median %.2f min %.2f" % (sorted(ts)[5] * 1e3, min(ts) * 1e3))
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5): bench.train_step(model, flat, opt, pts, label, inner)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
