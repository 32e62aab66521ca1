"""Generate tests/golden/blockio_ref.npz: inputs and outputs of the REFERENCE's own numpy input-side code, so that
``sph3d_gcn_amd/harness/blockio.py`` can be pinned against it bit for bit.

Runs in the build container only (it reads /root/reference at run time; nothing of it is copied here):
  * utils/data_util.py is pure numpy and is imported as it is: rotate_point_cloud, rotate_perturbation_point_cloud,
    jitter_point_cloud (data_util.py:47-61,140-176);
  * s3dis_seg/train_s3dis.py imports TensorFlow at the top, so the module cannot be imported; its ``augment_fn``
    (train_s3dis.py:114-142) is pure numpy + data_util: the FunctionDef node is taken out of the file's AST, compiled and
    executed with ``np`` and ``data_util`` in scope — the reference's own statements, run here, not restated;
  * the block sampling rule of the training loop (train_s3dis.py:343-347) is two np.random.choice calls, restated in
    ``sample_like_reference`` below (three lines; it lives inside a TF session loop and cannot be extracted).

    python tests/golden/make_blockio_golden.py          # writes tests/golden/blockio_ref.npz
"""
import ast
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.path.insert(0, os.path.join(REF, "utils"))
    import data_util                                           # the reference's module, unmodified
    src = open(os.path.join(REF, "s3dis_seg", "train_s3dis.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "augment_fn"][0]
    ns = {"np": np, "data_util": data_util}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train_s3dis.py:augment_fn", "exec"), ns)
    return data_util, ns["augment_fn"]


def sample_like_reference(num, num_point):
    # train_s3dis.py:343-347
    if num < num_point:
        return np.random.choice(num, num_point, replace=True)
    return np.random.choice(num, num_point, replace=False)


def main():
    data_util, augment_fn = load_reference()
    rng = np.random.RandomState(2024)
    out = {}
    # --- the three data_util functions on a [4, 50, 3] float32 batch, each under its own seed of the GLOBAL stream ---
    xyz = (rng.rand(4, 50, 3) * 3.0).astype(np.float32)
    out["xyz"] = xyz
    for name, seed in (("rotate_point_cloud", 11), ("rotate_perturbation_point_cloud", 12), ("jitter_point_cloud", 13)):
        np.random.seed(seed)
        out[name] = np.asarray(getattr(data_util, name)(xyz.copy()))
        out[name + "_seed"] = np.int64(seed)
    # --- augment_fn on the arrays the training loop hands it: float64 input [bsize, n, 6], int32 labels ---
    bsize, n = 7, 40
    binp = np.zeros((bsize, n, 6))                                       # float64, as train_s3dis.py:328
    binp[...] = np.concatenate(((rng.rand(bsize, n, 3) * 2.0).astype(np.float32), rng.rand(bsize, n, 3).astype(np.float32)), 2)
    blab = rng.randint(0, 13, (bsize, n)).astype(np.int32)
    binn = rng.randint(0, 2, (bsize, n)).astype(np.int32)
    out["aug_in_input"], out["aug_in_label"], out["aug_in_inner"] = binp.copy(), blab.copy(), binn.copy()
    np.random.seed(21)
    a, b, c = augment_fn(binp.copy(), blab.copy(), binn.copy())
    out["aug_seed"] = np.int64(21)
    out["aug_out_input"], out["aug_out_label"], out["aug_out_inner"] = np.asarray(a), np.asarray(b), np.asarray(c)
    # --- block sampling: enough points (without replacement) and too few (with replacement) ---
    np.random.seed(31)
    out["sample_seed"] = np.int64(31)
    out["sample_300_of_1000"] = sample_like_reference(1000, 300)
    out["sample_300_of_120"] = sample_like_reference(120, 300)
    path = os.path.join(HERE, "blockio_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape, str(v.dtype)) for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
