"""Every C-ABI call of one feature-path step (graphs prebuilt, main stream only), with its dimensions and device time:
which calls make up the 7.5 ms.  usage: python tools/exp_calls.py [name substring]"""
import sys, os, collections
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
flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4, fused=True)
plan = s3dis_net.build_graphs(pts, cfg)
def step():
    pred, _ = model(pts, is_training=True, graphs=plan)
    loss = model.loss(pred, label, inner)
    flat.backward(loss); flat.all_reduce(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
N = 10
_lib.timing_start()
for _ in range(N): step()
torch.cuda.synchronize()
acc = collections.OrderedDict()
for name, ints, e0, e1 in _lib.timing_stop():
    d = acc.setdefault((name, ints[:8]), [0.0, 0])
    d[0] += e0.elapsed_time(e1); d[1] += 1
pat = sys.argv[1] if len(sys.argv) > 1 else ""
tot = collections.defaultdict(float)
for (name, ints), (ms, cnt) in acc.items():
    tot[name] += ms / N
    if pat in name:
        print("%-36s %-46s x%.0f  %7.1f us each" % (name.replace("sph3d_", ""), str(list(ints)), cnt / N, ms / cnt * 1e3))
print("per family (ms/step):", {k.replace("sph3d_", ""): round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
