"""Generate golden vectors from the REFERENCE ITSELF: the reference's tf_ops/*/tf_*_gpu.cu compiled unmodified
with hipcc for gfx950 (oracle/Makefile target `ref` -> oracle/_ref/libsph3d_ref_gfx950.so) and run on an MI355X.

Run on the GPU box (needs the prebuilt oracle/_ref library, which travels with the snapshot):
    python tests/golden/make_golden.py gpurun_out/golden
then copy gpurun_out/golden/* into tests/golden/ and commit.  The reference has no tests or fixtures of its
own, so these files are the pin between the CPU oracle and the reference's kernels.

Small cases store full arrays; large cases store SHA-256 digests of the raw output bytes (integer outputs and
nn_dist are deterministic; atomically-accumulated gradients are only stored for small cases, as arrays).
Inputs are regenerated from seeds by sph3d_gcn_amd/harness/synth.py; their digests are stored to detect drift.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_gpu  # noqa: E402
from sph3d_gcn_amd.harness import synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cases():
    """name -> dict(inputs...) ; shared with tests/test_golden.py"""
    c = {}
    c["tiny"] = dict(db=synth.uniform_cloud(1, 1, 16), q=None, r=0.3, K=4, full=True)
    c["chain"] = dict(db=synth.uniform_cloud(2, 33, 1100), q=None, r=0.08, K=4, full=False)
    c["decoder"] = dict(db=synth.uniform_cloud(3, 2, 300), q=synth.uniform_cloud(4, 2, 500), r=0.03, K=8, full=True)
    c["s3dis2048"] = dict(db=synth.s3dis_batch(7, 2, 2048)[0], q=None, r=0.1, K=64, full=False)
    c["modelnet1024"] = dict(db=synth.modelnet_batch(0, 2, 1024), q=None, r=0.1, K=32, full=False)
    return c


def fps_cases():
    return {"fps_s3dis2048": (synth.s3dis_batch(7, 2, 2048)[0], 512),
            "fps_modelnet10000": (synth.modelnet_batch(3, 1, 10000), 300),
            "fps_small": (synth.uniform_cloud(5, 3, 100), 40)}


def feature_case():
    rng = np.random.RandomState(42)
    B, N, M, C, r, K = 2, 200, 100, 8, 2, 16
    db = synth.uniform_cloud(6, B, N)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(33, C, r).astype(np.float32)
    go = rng.randn(B, M, C * r).astype(np.float32)
    gp = rng.randn(B, M, C).astype(np.float32)
    return dict(db=db, M=M, K=K, r=0.25, x=x, w=w, go=go, gp=gp)


def main(outdir):
    assert ref_gpu.available(), "needs oracle/_ref/libsph3d_ref_gfx950.so and a GPU"
    os.makedirs(outdir, exist_ok=True)
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    N_ = lambda t: t.cpu().numpy()
    arrays, digests = {}, {}
    for name, c in cases().items():
        db, q = c["db"], c["db"] if c["q"] is None else c["q"]
        idx, cnt, dst = ref_gpu.build_sphere_neighbor(T(db), T(q), c["r"], None, c["K"])
        filt = ref_gpu.spherical_kernel(T(db), T(q), idx, cnt, dst, c["r"], [8, 2, 2])
        idx, cnt, dst, filt = N_(idx), N_(cnt), N_(dst), N_(filt)
        digests[name] = dict(db=sha(db), q=sha(q), nn_index=sha(idx), nn_count=sha(cnt), nn_dist=sha(dst),
                             filt_index_ocml=sha(filt), count_sum=int(cnt.sum()))
        arrays[name + "/nn_count"] = cnt.astype(np.int16)
        arrays[name + "/filt_index_ocml"] = filt.astype(np.int8)
        if c["full"]:
            arrays[name + "/nn_index"] = idx
            arrays[name + "/nn_dist"] = dst
    for name, (pts, m) in fps_cases().items():
        out = N_(ref_gpu.farthest_point_sample(m, T(pts)))
        digests[name] = dict(pts=sha(pts), out=sha(out))
        arrays[name + "/out"] = out.astype(np.int16)
    # cube
    db, q = synth.uniform_cloud(8, 2, 300), synth.uniform_cloud(9, 2, 100)
    cidx, ccnt = ref_gpu.build_cube_neighbor(T(db), T(q), 0.3, None, 8, 3)
    arrays["cube/nn_index"], arrays["cube/nn_count"] = N_(cidx).astype(np.int16), N_(ccnt).astype(np.int16)
    # feature ops on a small graph (full arrays; gradients are atomically accumulated -> compare with tolerance)
    f = feature_case()
    db, M, K = f["db"], f["M"], f["K"]
    q = db[:, :M].copy()
    idx, cnt, dst = ref_gpu.build_sphere_neighbor(T(db), T(q), f["r"], None, K)
    filt = ref_gpu.spherical_kernel(T(db), T(q), idx, cnt, dst, f["r"], [8, 2, 2])
    arrays["feat/nn_index"], arrays["feat/nn_count"], arrays["feat/filt"] = N_(idx), N_(cnt), N_(filt)
    arrays["feat/conv"] = N_(ref_gpu.depthwise_conv3d(T(f["x"]), T(f["w"]), idx, cnt, filt))
    gi, gf = ref_gpu.depthwise_conv3d_grad(T(f["x"]), T(f["w"]), T(f["go"]), idx, cnt, filt)
    arrays["feat/conv_gi"], arrays["feat/conv_gf"] = N_(gi), N_(gf)
    mo, mi = ref_gpu.max_pool3d(T(f["x"]), idx, cnt)
    arrays["feat/maxpool"], arrays["feat/maxpool_idx"] = N_(mo), N_(mi)
    arrays["feat/maxpool_grad"] = N_(ref_gpu.max_pool3d_grad(T(f["x"]), T(f["gp"]), mi))
    arrays["feat/avgpool"] = N_(ref_gpu.avg_pool3d(T(f["x"]), idx, cnt))
    arrays["feat/avgpool_grad"] = N_(ref_gpu.avg_pool3d_grad(T(f["x"]), T(f["gp"]), idx, cnt))
    # un-pooling graph: db = coarse (first M points), query = all N fine points
    uidx, ucnt, udst = ref_gpu.build_sphere_neighbor(T(q), T(db), 0.3, None, K)
    feat = f["x"][:, :M].copy()
    gu = np.random.RandomState(43).randn(*f["x"].shape).astype(np.float32)
    wgt = N_(udst)
    wgt = ((wgt + 1e-7) / (wgt.sum(-1, keepdims=True) + 1e-7)).astype(np.float32)
    arrays["feat/un_index"], arrays["feat/un_count"], arrays["feat/un_weight"] = N_(uidx), N_(ucnt), wgt
    arrays["feat/mean"] = N_(ref_gpu.mean_interpolate(T(feat), uidx, ucnt))
    arrays["feat/mean_grad"] = N_(ref_gpu.mean_interpolate_grad(T(feat), T(gu), uidx, ucnt))
    arrays["feat/weighted"] = N_(ref_gpu.weighted_interpolate(T(feat), T(wgt), uidx, ucnt))
    arrays["feat/weighted_grad"] = N_(ref_gpu.weighted_interpolate_grad(T(feat), T(gu), T(wgt), uidx, ucnt))
    np.savez_compressed(os.path.join(outdir, "ref_gfx950.npz"), **arrays)
    meta = dict(generator="tests/golden/make_golden.py", source="reference tf_ops/*/tf_*_gpu.cu compiled with hipcc "
                "-O3 -ffp-contract=off --offload-arch=gfx950, run on " + torch.cuda.get_device_name(0),
                digests=digests)
    json.dump(meta, open(os.path.join(outdir, "ref_gfx950.json"), "w"), indent=1, sort_keys=True)
    print("golden written:", outdir, {k: v.shape for k, v in list(arrays.items())[:4]})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/golden")
