"""Forward+backward step time of the ModelNet and ShapeNet harness nets at their full configurations (secondary numbers)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import modelnet_net, shapenet_net, synth, dist as hdist
dev = torch.device('cuda:0'); _lib.lib()
def run(name, model, pts, loss_fn, steps=8):
    pred, _ = model(pts, is_training=True); loss_fn(pred).backward()
    flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4)
    def step():
        pred, _ = model(pts, is_training=True); flat.backward(loss_fn(pred)); opt.step()
    for _ in range(6): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print("%s: %.2f ms/step, %.1f clouds/s, params %d" % (name, dt * 1e3, pts.shape[0] / dt, sum(p.numel() for p in model.parameters())))
B = 32
pts = torch.from_numpy(synth.modelnet_batch(0, B, 10000)).to(dev)
label = torch.randint(0, 40, (B,), device=dev)
m = modelnet_net.SPH3DModelNet(modelnet_net.modelnet_config(10000), device=dev)
run("ModelNet cls B=32 N=10000", m, pts, lambda pred: m.loss(pred, label))
B = 64
pts = torch.from_numpy(synth.modelnet_batch(100, B, 2048)).to(dev)
label = torch.randint(0, 3, (B, 2048), device=dev)
s = shapenet_net.SPH3DShapeNet(3, shapenet_net.shapenet_config(2048), device=dev)
run("ShapeNet seg B=64 N=2048", s, pts, lambda pred: s.loss(pred, label))
