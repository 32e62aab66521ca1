"""How long one rank's Python needs to ISSUE a training step: the headline step with ONE block per step (the same ~450 launches,
a sixteenth of the device work) — and where that time goes (cProfile of the issuing thread; autograd on the same thread)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import dist as hdist, optim as hoptim, s3dis_net, synth

dev = torch.device("cuda:0"); _lib.lib()
B = int(os.environ.get("BLOCKS", "1"))
xyz, label, inner = synth.s3dis_batch(1000, B, 8192)
batches = []
for w in range(2):
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
    p, l, i = batches[n[0] % 2]; n[0] += 1
    return bench.train_step(model, flat, opt, p, l, i)
for _ in range(20):
    step()
torch.cuda.synchronize()
def timed(k):
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    t1 = time.perf_counter()            # host done issuing
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3
iss, tot = timed(40)
print("blocks/step %d: host issue %.3f ms/step, step %.3f ms (two autograd threads)" % (B, iss, tot))
torch.autograd.set_multithreading_enabled(False)
for _ in range(5):
    step()
torch.cuda.synchronize()
iss, tot = timed(40)
print("blocks/step %d: host issue %.3f ms/step, step %.3f ms (backward on the issuing thread)" % (B, iss, tot))
if os.environ.get("PROFILE", "1") != "0":
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
    txt = s.getvalue()
    print(txt[:9000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("sph3d_gcn_amd|bench", 60)
    print(s.getvalue()[:12000])
