"""Pin the exact set of bin ids on which the default (shared, correctly rounded atan2f) binning differs from the
reference kernel as built for gfx950 (ocml atan2f), on the golden clouds.

Inputs: tests/golden/ref_gfx950.npz (outputs of the reference build, made on the GPU box by make_golden.py) and the CPU
oracle.  Output: the "bin_mismatches" table of tests/golden/ref_gfx950.json — per case a list of
[b, m, k, ours, reference, which ('azimuth' | 'elevation'), distance of the exact angle to the bin boundary in radians].
tests/test_golden.py asserts the list entry by entry and that every entry is a neighbour whose exact angle lies within
rounding distance of a boundary.  Run here (CPU only):  python tests/golden/pin_bin_mismatches.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import oracle  # noqa: E402
import make_golden as gen  # noqa: E402

KERNEL = [8, 2, 2]


def boundary_distance(db, q, idx, b, m, k, ours, ref):
    """-> (which, radians between the exact (float64) angle of the neighbour and the nearest bin boundary)"""
    n, p = KERNEL[0], KERNEL[1]
    pt, qp = db[b, idx[b, m, k]], q[b, m]
    dx, dy, dz = (np.float32(pt[i]) - np.float32(qp[i]) for i in range(3))
    d2 = np.sqrt(np.float32(dx * dx + dy * dy), dtype=np.float32)
    a, r_ = ours - 1, ref - 1
    if a % n != r_ % n:                                      # azimuth cell moved
        theta = np.arctan2(np.float64(dy), np.float64(dx)) + np.pi
        cell = 2 * np.pi / n
        return "azimuth", float(abs(theta - np.round(theta / cell) * cell))
    phi = np.arctan2(np.float64(dz), np.float64(d2)) + np.pi / 2
    cell = np.pi / p
    return "elevation", float(abs(phi - np.round(phi / cell) * cell))


def mismatches():
    arr = np.load(os.path.join(HERE, "ref_gfx950.npz"))
    out = {}
    for name, c in gen.cases().items():
        db = c["db"]
        q = db if c["q"] is None else c["q"]
        idx, cnt, dst = oracle.build_sphere_neighbor(db, q, c["r"], None, c["K"])
        filt = oracle.spherical_kernel(db, q, idx, cnt, dst, c["r"], KERNEL)
        ref = arr[name + "/filt_index_ocml"].astype(np.int32)
        rows = []
        for b, m, k in np.argwhere(filt != ref):
            which, dist = boundary_distance(db, q, idx, b, m, k, int(filt[b, m, k]), int(ref[b, m, k]))
            rows.append([int(b), int(m), int(k), int(filt[b, m, k]), int(ref[b, m, k]), which, dist])
        out[name] = rows
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "ref_gfx950.json")
    meta = json.load(open(path))
    meta["bin_mismatches"] = mismatches()
    meta["bin_mismatches_note"] = ("default binning (shared correctly rounded atan2f) vs the reference kernel built for gfx950 "
                                   "(ocml atan2f), kernel [8,2,2]; made by tests/golden/pin_bin_mismatches.py")
    json.dump(meta, open(path, "w"), indent=1, sort_keys=True)
    for k, v in meta["bin_mismatches"].items():
        print(k, len(v), "max boundary distance %.3g rad" % max([r[6] for r in v], default=0.0))
        for r in v:
            print("   ", r)
