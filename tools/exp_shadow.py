"""What the side streams' work costs the feature path (headline step, 16 x 8192): device ms per step with
  full       the plan of every step built beside the previous step (the bench's loop),
  idle       ONE plan built ahead and reused: nothing on the side streams,
  fps_only   the sampling chain of a fresh plan every step (its graphs not built), the feature path on the reused plan,
  graph_only the graphs + transposed graphs of a fresh plan every step from the reused plan's samples (no FPS launches).
usage: python tools/exp_shadow.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import dist as hdist, optim as hoptim, s3dis_net, synth
dev = torch.device("cuda:0"); _lib.lib()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 16
xyz, label, inner = synth.s3dis_batch(1000, B, 8192)
pts, label, inner = torch.from_numpy(xyz).to(dev), torch.from_numpy(label).to(dev), torch.from_numpy(inner).to(dev)
torch.cuda.synchronize()
ready = torch.cuda.Event(); ready.record()
cfg = s3dis_net.s3dis_config(8192)
model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
pred, _ = model(pts, is_training=True)
model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters())
opt = hoptim.FlatAdam(flat.flat_param, lr=1e-3, eps=1e-4)
ref_plan = s3dis_net.GraphPlan(pts, model.config, points_ready=ready)
torch.cuda.synchronize()


class FpsOnly(s3dis_net.GraphPlan):
    def _build_all(self, stream):
        pass


class GraphOnly(s3dis_net.GraphPlan):
    def _sampling_chain(self, side):
        for l in range(len(self.config.radius)):
            self.indices.append(ref_plan.indices[l])
            if ref_plan.indices[l] is not None:
                self.xyz_layers.append(ref_plan.xyz_layers[l + 1])
            ev = torch.cuda.Event(); ev.record(side); self.events.append(ev)


def step(mode):
    if mode == "full":
        plan = None
    elif mode == "idle":
        plan = ref_plan
    elif mode == "fps_only":
        FpsOnly(pts, model.config, points_ready=ready)
        plan = ref_plan
    else:
        plan = GraphOnly(pts, model.config, points_ready=ready)
    pred, _ = model(pts, is_training=True, graphs=plan, points_ready=ready)
    loss = model.loss(pred, label, inner)
    flat.backward(loss)
    flat.all_reduce()
    opt.step()


for mode in ("full", "idle", "fps_only", "graph_only", "full", "idle"):
    for _ in range(12):
        step(mode)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(STEPS):
        step(mode)
    e1.record()
    host = (time.perf_counter() - t0) * 1e3 / STEPS
    torch.cuda.synchronize()
    print("%-10s device %.3f ms/step   (host issue %.2f ms/step)" % (mode, e0.elapsed_time(e1) / STEPS, host))

# ---- which calls of the feature path pay: per-call device time (HIP events around every C-ABI call) by mode ----------------
import collections
res = {}
for mode in ("idle", "fps_only", "graph_only", "full"):
    for _ in range(6):
        step(mode)
    torch.cuda.synchronize()
    _lib.timing_start()
    main_raw = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        step(mode)
    recs = _lib.timing_stop()
    torch.cuda.synchronize()
    acc = collections.defaultdict(float)
    for name, args, e0, e1 in recs:
        if getattr(e0, "raw_stream", None) != main_raw:
            continue
        acc[(name, args[:7])] += e0.elapsed_time(e1) * 1e3 / 10
    res[mode] = acc
keys = sorted(res["idle"], key=lambda k: -(res["full"].get(k, 0) - res["idle"][k]))
print("\nmain-stream calls, us per step (idle | fps_only | graph_only | full), sorted by full - idle; top 25 and totals")
for k in keys[:25]:
    print("  %-44s %-34s %8.1f %8.1f %8.1f %8.1f" % (k[0], str(k[1]), res["idle"][k], res["fps_only"].get(k, 0), res["graph_only"].get(k, 0), res["full"].get(k, 0)))
fam = collections.defaultdict(lambda: [0.0] * 4)
for i, mode in enumerate(("idle", "fps_only", "graph_only", "full")):
    for k, v in res[mode].items():
        n = k[0]
        f = ("gemm" if "gemm" in n else "conv grad" if "conv3d_grad" in n else "conv fwd" if "depthwise_conv3d" in n or "separable" in n
             else "elu_bn" if "elu_bn" in n else "pool/unpool" if any(s in n for s in ("pool", "interpolate", "scatter", "gather")) else "other")
        fam[f][i] += v
for f, v in sorted(fam.items(), key=lambda kv: -kv[1][3]):
    print("  family %-12s %8.1f %8.1f %8.1f %8.1f" % ((f,) + tuple(v)))
