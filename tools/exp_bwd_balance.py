"""Load balance of the conv gradient's static sweep at level 0: in-edges per wave under the kernel's assignment
(wave g of an XCD takes positions g, g + 512, ... of its clouds) against a dynamic hand-out in processing order."""
import sys, os, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from sph3d_gcn_amd import tf_nnquery
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
B, K, N, rad = 16, 64, 8192, 0.1
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, N)[0]).to(dev)[:, :, :3].contiguous()
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
valid = torch.arange(K, device=dev)[None, None, :] < cnt[:, :, None]
deg = torch.zeros(B, N, dtype=torch.long, device=dev)
for b in range(B):
    deg[b] = torch.bincount(nidx[b][valid[b]].long(), minlength=N)
deg = deg.cpu().numpy()
print("in-degree mean %.1f std %.1f max %d" % (deg.mean(), deg.std(), deg.max()))
for over in (0, 30, 60):        # per-source fixed cost in edge-equivalents
    cost = deg + over
    waves = 512
    stat = []; dyn = []
    for xcd in range(8):
        load = np.zeros(waves)
        for b in range(xcd, B, 8):
            for g in range(waves):
                load[g] += cost[b, g::waves].sum()
        stat.append(load.max() / load.mean())
        h = [(0.0, g) for g in range(waves)]; heapq.heapify(h)
        for b in range(xcd, B, 8):
            for n in range(N):
                l, g = heapq.heappop(h); heapq.heappush(h, (l + cost[b, n], g))
        l = np.array([x[0] for x in h]); dyn.append(l.max() / l.mean())
    print("fixed cost %2d: static max/mean %.2f (worst XCD %.2f)   dynamic %.3f" % (over, np.mean(stat), np.max(stat), np.mean(dyn)))
