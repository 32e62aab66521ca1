"""Which part of the side-stream work costs the feature path its 2 ms?  The main stream always runs the feature path on a
FIXED prebuilt plan; the side streams run, per step, a discarded copy of: nothing / the whole GraphPlan / the FPS chain only /
the graph kernels only (FPS indices served from a cache) / the graph kernels without the transposed graphs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sph3d_gcn_amd import _lib, _tgraph
from sph3d_gcn_amd import sph3gcn_util as s3g_util
from sph3d_gcn_amd.harness import s3dis_net, dist as hdist
dev = torch.device('cuda:0'); _lib.lib()
_tgraph._MAX_ENTRIES = 64
batches = [bench.make_batch(0, dev, w) for w in range(2)]
torch.cuda.synchronize(); ev = torch.cuda.Event(); ev.record()
cfg = s3dis_net.s3dis_config(8192)
model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
pts, label, inner = batches[0]
pred, _ = model(pts, True); model.loss(pred, label, inner).backward()
flat = hdist.FlatGradAllReduce(model.parameters()); opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4, fused=True)
plan0 = s3dis_net.build_graphs(pts, cfg)
torch.cuda.synchronize()
def run(fn, n=50, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def feature():
    pred, _ = model(pts, is_training=True, graphs=plan0)
    loss = model.loss(pred, label, inner)
    flat.backward(loss); flat.all_reduce(); opt.step()
real_fps = s3g_util.farthest_point_sample
cache = {}
def cached_fps(m, xyz):
    key = (m, xyz.shape[1])
    if key not in cache: cache[key] = real_fps(m, xyz)
    return cache[key]
real_pre = s3dis_net.GraphPlan._pretranspose
keep = []
def side(kind):
    other = batches[1][0]
    if kind == "none": return
    if kind == "fps":
        s_fps = s3dis_net._side_stream[dev][0]
        s_fps.wait_event(ev)
        with torch.cuda.stream(s_fps):
            cur = other[:, :, :3].contiguous()
            for m in cfg.num_sample:
                if m > 1:
                    idx = real_fps(m, cur)
                    cur = torch.gather(cur, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        return
    if kind == "graph_cachedfps": s3g_util.farthest_point_sample = cached_fps
    if kind == "graph_notranspose": s3dis_net.GraphPlan._pretranspose = lambda self, g, a, n_src_unpool=None: None
    try:
        keep.append(s3dis_net.GraphPlan(other, cfg, points_ready=ev))
        del keep[:-2]
    finally:
        s3g_util.farthest_point_sample = real_fps
        s3dis_net.GraphPlan._pretranspose = real_pre
import collections
for kind in ("none", "all", "fps", "graph_cachedfps", "graph_notranspose", "none"):
    def step():
        side(kind)
        feature()
    ms = run(step)
    _lib.timing_start()
    for _ in range(10): step()
    torch.cuda.synchronize()
    fam = collections.defaultdict(float)
    for name, ints, e0, e1 in _lib.timing_stop():
        fam["gemm" if "gemm" in name else name.replace("sph3d_", "")] += e0.elapsed_time(e1) / 10
    top = sorted(fam.items(), key=lambda kv: -kv[1])[:9]
    print("side work per step = %-18s: %.2f ms/step | " % (kind, ms) + "  ".join("%s %.2f" % (k[:22], v) for k, v in top), flush=True)
