"""The cell-grid neighbour search (csrc/nngrid.hip): the rows of the early chain positions come from a spatial grid, the
rest — and the whole call when a query needs the reference's radius growth — from the chain kernel.  Every case is checked bit
for bit against the oracle (tf_nnquery_gpu.cu:15-65 restated in C), and the launch counter shows which path ran."""
import numpy as np
import pytest
import torch

import oracle
from sph3d_gcn_amd import _lib, tf_buildkernel, tf_nnquery
from sph3d_gcn_amd.harness import synth

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _n(t):
    return t.detach().cpu().numpy()


def _cloud(kind, B, N, seed):
    if kind == "s3dis":
        return synth.s3dis_batch(seed, B, N)[0][:, :, :3].copy()
    if kind == "modelnet":
        return synth.modelnet_batch(seed, B, N)[:, :, :3].copy()
    return synth.uniform_cloud(seed, B, N, 1.0)


def _check(dev, db, q, radius, K, fixed=False, expect_grid=True):
    idx_o, cnt_o, dst_o = oracle.build_sphere_neighbor(db, q, radius, None, K, fixed=fixed)
    before = _lib.lib().sph3d_nngrid_launches()
    if fixed:
        tf_nnquery.set_radius_mode("fixed")
    try:
        d = _t(db, dev)
        idx, cnt, dst = tf_nnquery.build_sphere_neighbor(d, d if q is db else _t(q, dev), radius, None, K)
    finally:
        tf_nnquery.set_radius_mode("compat")
    took_grid = _lib.lib().sph3d_nngrid_launches() - before
    assert took_grid == (1 if expect_grid else 0)
    np.testing.assert_array_equal(_n(cnt), cnt_o)
    np.testing.assert_array_equal(_n(idx), idx_o)
    np.testing.assert_array_equal(_n(dst).view(np.int32), dst_o.view(np.int32))   # bit pattern
    return cnt_o


# (kind, B, N, M (None = the cloud itself, int = its first M points), radius, K)
GRID_CASES = [
    ("s3dis", 2, 8192, None, 0.1, 64),         # the bench's level 0: positions 0-2 from the grid, 3-7 from the chain kernel
    ("s3dis", 3, 8192, 2048, 0.1, 64),         # its first pooling graph: two positions, all from the grid
    ("s3dis", 16, 2048, None, 0.2, 64),        # level 1
    ("uniform", 3, 4096, None, 0.05, 16),
    ("uniform", 2, 4096, None, 0.12, 8),       # far more hits than K: the bitmap's first K bits
    ("modelnet", 2, 10000, None, 0.1, 64),     # ten positions, cloud in [-1, 1]
    ("uniform", 2, 2100, None, 0.06, 100),     # K > 64
    ("uniform", 1, 65536, 128, 0.02, 32),      # the largest cloud the bitmaps take
    ("uniform", 32, 2048, None, 0.07, 16),     # one cloud per reference block: the last batch size without cloud-to-cloud carries
]


@pytest.mark.parametrize("case", GRID_CASES, ids=lambda c: "%s-B%d-N%d-M%s-r%g-K%d" % c)
def test_grid_rows_equal_the_oracle(dev, case):
    kind, B, N, M, radius, K = case
    db = _cloud(kind, B, N, seed=11)
    q = db if M is None else np.ascontiguousarray(db[:, :M])
    cnt = _check(dev, db, q, radius, K)
    assert cnt.min() >= 1


def test_queries_without_neighbours_hand_the_call_to_the_chain_kernel(dev):
    """decoder-like: queries that are not points of the cloud, a third of them outside its bounding box — they find nothing inside
    the nominal radius, the reference grows the radius for them and for the rest of their chains: the grid raises its flag and
    the chain kernel recomputes the call (fused counters included: checked through the fused op below)"""
    db = _cloud("uniform", 2, 2048, seed=3)
    q = _cloud("uniform", 2, 2500, seed=4) * 1.5 - 0.25
    cnt = _check(dev, db, q, 0.05, 16)
    assert (cnt >= 1).all()
    # the same through the fused graph op with the transposed graph's counts: equal to the separate ops
    d, qq = _t(db, dev), _t(q, dev)
    i1, c1, d1 = tf_nnquery.build_sphere_neighbor_counted(d, qq, 0.05, 16)
    i0, c0, d0 = tf_nnquery.build_sphere_neighbor(d, qq, 0.05, None, 16)
    assert torch.equal(i0, i1) and torch.equal(c0, c1) and torch.equal(d0, d1)


def test_more_than_32_clouds_carry_the_chain_position_from_cloud_to_cloud(dev):
    """clouds b and b + 32 share a reference block: the second one's queries sit at the positions behind the first one's"""
    db = _cloud("uniform", 40, 2048, seed=5)            # positions 0-1 in clouds 0-31, 2-3 in clouds 32-39
    _check(dev, db, db, 0.08, 8)
    db = _cloud("s3dis", 70, 2048, seed=5)              # three clouds per block: up to position 5 (r = 0.45 at r0 = 0.2)
    _check(dev, db, db, 0.2, 64)
    db = _cloud("s3dis", 40, 4096, seed=6)
    q = np.ascontiguousarray(db[:, :1024])              # pooling-like, one query per thread and cloud: position = b / 32
    _check(dev, db, q, 0.1, 32)


def test_shapes_the_grid_does_not_take_keep_the_chain_kernel(dev):
    db = _cloud("uniform", 33, 2100, seed=5)            # > 32 clouds with a ragged last slice: positions differ per thread
    _check(dev, db, db, 0.08, 8, expect_grid=False)
    db = _cloud("uniform", 2, 1000, seed=6)             # fewer than 1024 points
    _check(dev, db, db, 0.1, 16, expect_grid=False)


def test_a_radius_comparable_to_the_extent_falls_back(dev):
    """fewer than 512 cells: the build kernel raises the flag, the chain kernel computes the call"""
    db = _cloud("uniform", 2, 2048, seed=7)
    _check(dev, db, db, 0.3, 32)


def test_fixed_radius_mode_takes_every_query_from_the_grid(dev):
    db = _cloud("s3dis", 2, 4096, seed=8)
    _check(dev, db, db, 0.1, 32, fixed=True)
    q = _cloud("uniform", 2, 3000, seed=9) * 2.0         # isolated queries: per-query growth -> chain kernel
    _check(dev, db, q, 0.1, 32, fixed=True)


def test_far_from_the_origin_and_degenerate_extents(dev):
    db = _cloud("uniform", 2, 4096, seed=10)
    _check(dev, db + np.float32(1000.0), db + np.float32(1000.0), 0.05, 16)        # coordinates with 6e-5 resolution
    flat = db.copy()
    flat[:, :, 2] = 0.25                                 # a plane: one layer of cells
    _check(dev, flat, flat, 0.03, 16)
    dup = db.copy()
    dup[:, 1::2] = dup[:, 0::2]                          # every point twice
    _check(dev, dup, dup, 0.05, 16)


def test_fused_graph_with_bins_and_counts_equals_the_separate_ops(dev):
    """sph3d_build_sphere_graph over the grid: indices, counts, distances, bins — and a transposed graph that the convolution
    gradient can use (exercised by the model tests); here the four tensors against the separate ops"""
    db = _t(_cloud("s3dis", 4, 4096, seed=12), dev)
    for mode in ("shared", "ocml"):
        tf_buildkernel.set_atan2(mode)
        try:
            i1, c1, d1, f1 = tf_nnquery.build_sphere_graph(db, 0.1, 64, (8, 2, 2), with_transpose=True)
            i0, c0, d0 = tf_nnquery.build_sphere_neighbor(db, db, 0.1, None, 64)
            f0 = tf_buildkernel.spherical_kernel(db, db, i0, c0, d0, 0.1, [8, 2, 2])
        finally:
            tf_buildkernel.set_atan2("shared")
        assert torch.equal(i0, i1) and torch.equal(c0, c1) and torch.equal(d0, d1) and torch.equal(f0, f1)


def test_two_streams_search_at_the_same_time_with_their_own_grids(dev):
    """the grid lives in a library-owned buffer PER STREAM: searches queued on two streams at once must not share it"""
    a = _t(_cloud("s3dis", 4, 4096, seed=21), dev)
    b = _t(_cloud("uniform", 4, 4096, seed=22), dev)
    want_a = tf_nnquery.build_sphere_neighbor(a, a, 0.1, None, 32)
    want_b = tf_nnquery.build_sphere_neighbor(b, b, 0.07, None, 32)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(3):
        got = []
        for _rep in range(4):                               # interleaved launches, several in flight per stream
            with torch.cuda.stream(s1):
                got.append((tf_nnquery.build_sphere_neighbor(a, a, 0.1, None, 32), want_a))
            with torch.cuda.stream(s2):
                got.append((tf_nnquery.build_sphere_neighbor(b, b, 0.07, None, 32), want_b))
        torch.cuda.synchronize()
        for g, w in got:
            assert all(torch.equal(x, y) for x, y in zip(g, w))
