"""Cycle counts of the pruned FPS kernel's round loop (library built with -DSPH3D_FPS_PROF: sph3d_gcn_amd/csrc/libsph3d_prof.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPH3D_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sph3d_gcn_amd", "csrc", "libsph3d_prof.so")
import numpy as np, torch
from sph3d_gcn_amd import tf_sample, _lib
from sph3d_gcn_amd.harness import synth
dev = torch.device("cuda:0"); l = _lib.lib()
raw = ctypes.CDLL(os.environ["SPH3D_LIB"])
for N, m in ((8192, 2048), (2048, 768)):
    x = torch.from_numpy(synth.s3dis_batch(1000, 16, N)[0][:, :, :3].copy()).to(dev)
    for _ in range(2):
        tf_sample.farthest_point_sample(m, x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); tf_sample.farthest_point_sample(m, x); e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 128)()
    assert raw.sph3d_debug_fps_prof(out) == 0
    a = np.array(list(out), dtype=np.float64).reshape(16, 8)
    us = e0.elapsed_time(e1) * 1e3
    print("N %d -> %d: %.1f us, %.3f us/round" % (N, m, us, us / (m - 1)))
    print(" wave | active rounds: n, own cycles (with a rescan: n, cycles; without: cycles) | idle rounds: n, own cycles | publish+barrier | post")
    for w in range(16):
        r = a[w]
        if r[2] + r[3] == 0:
            continue
        nores = max(r[2] - r[6], 1)
        print("  %2d  | %5d %7.0f (%5d %7.0f; %7.0f) | %5d %7.0f | %7.0f | %6.0f" % (
            w, r[2], r[0] / max(r[2], 1), r[6], r[7] / max(r[6], 1), (r[0] - r[7]) / nores, r[3], r[1] / max(r[3], 1),
            r[4] / (r[2] + r[3]), r[5] / (r[2] + r[3])))
    tot = a[:, 0] + a[:, 1] + a[:, 4] + a[:, 5]
    print(" cycles per round (wave 0): %.0f  -> clock %.2f GHz if the loop is the whole kernel" % (tot[0] / (m - 1), tot[0] / (m - 1) / (us / (m - 1)) / 1e3))
