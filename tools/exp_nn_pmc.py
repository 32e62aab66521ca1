"""One level-0 graph build (16 x 8192, K = 64, kernel [8,2,2]) for a counter pass over the search kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
xyz = torch.from_numpy(synth.s3dis_batch(1000, 16, 8192)[0]).to(dev)[:, :, :3].contiguous()
mode = os.environ.get("NN", "fused")
for _ in range(3):
    if mode == "fused":
        tf_nnquery.build_sphere_graph(xyz, 0.1, 64, [8, 2, 2], with_transpose=True)
    else:
        tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, 64)
torch.cuda.synchronize()
