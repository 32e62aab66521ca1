"""Isolated timing of the fused graph construction (search + bins + transposed counts) on the bench's level shapes, and a
bit-for-bit comparison of its outputs between the cell-grid search and the chain kernel (run once with SPH3D_NNGRID=0 to dump
the reference outputs, once without to compare).  usage: python tools/exp_nngrid.py [dump|check]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph3d_gcn_amd import _lib, tf_nnquery  # noqa: E402
from sph3d_gcn_amd.harness import synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "time"
dev = torch.device("cuda:0")
xyz, _, _ = synth.s3dis_batch(1000, 16, 8192)
pts0 = torch.from_numpy(xyz[:, :, :3]).to(dev).contiguous()
levels = [(8192, 0.1), (2048, 0.2), (768, 0.4), (384, 0.8)]
out = {}
for n, radius in levels:
    pts = pts0[:, :n].contiguous()
    for _ in range(3):
        r = tf_nnquery.build_sphere_graph(pts, radius, 64, (8, 2, 2), with_transpose=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        r = tf_nnquery.build_sphere_graph(pts, radius, 64, (8, 2, 2), with_transpose=False)
    e1.record()
    torch.cuda.synchronize()
    cnt = r[1].float().mean().item()
    print("N=%5d r=%.1f  %.1f us per call   mean count %.1f  max %d" % (n, radius, e0.elapsed_time(e1) * 100, cnt, int(r[1].max())), flush=True)
    out[n] = [t.cpu() for t in r]
    # pooling-style graph: queries = first quarter of the points
    q = pts[:, :n // 4].contiguous()
    for _ in range(3):
        r2 = tf_nnquery.build_sphere_neighbor_counted(pts, q, radius, 64)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        r2 = tf_nnquery.build_sphere_neighbor_counted(pts, q, radius, 64)
    e1.record()
    torch.cuda.synchronize()
    print("   pooling graph %d -> %d: %.1f us per call (with the transposed graph's finish)  mean count %.1f" % (n, n // 4, e0.elapsed_time(e1) * 100, r2[1].float().mean().item()), flush=True)
    out[(n, "q")] = [t.cpu() for t in r2]
path = "/tmp/nngrid_ref.pt"
if mode == "dump":
    torch.save(out, path)
elif mode == "check":
    ref = torch.load(path)
    for k in out:
        for a, b in zip(ref[k], out[k]):
            assert torch.equal(a, b), (k, (a != b).sum().item())
    print("outputs identical to the chain kernel's")
