"""The pruned farthest-point sampling kernel (csrc/sample.hip: fps_prune_kernel, clouds of 1025 .. 16384 points): spatially
sorted slots whose distance update is skipped when the new sample cannot change them, per-lane arg-max caches that are
rescanned only when they went stale.  The samples must be the reference's
(tf_sample_gpu.cu:7-73 restated in oracle/) bit for bit: same indices in the same order, including every tie."""
import numpy as np
import pytest
import torch

import oracle
from sph3d_gcn_amd import tf_sample
from sph3d_gcn_amd.harness import synth

pytestmark = pytest.mark.gpu


def _fps(dev, pts, m):
    return tf_sample.farthest_point_sample(m, torch.from_numpy(np.ascontiguousarray(pts)).to(dev)).cpu().numpy()


def _cloud(kind, B, N, seed):
    if kind == "s3dis":
        return synth.s3dis_batch(seed, B, N)[0][:, :, :3].copy()
    if kind == "modelnet":
        return synth.modelnet_batch(seed, B, N)[:, :, :3].copy()
    return synth.uniform_cloud(seed, B, N, 1.0)


# every (points per lane, waves) instantiation, ragged last slots / waves, the bench's own shapes
CASES = [("uniform", 2, 6144, 700), ("uniform", 1, 6145, 300), ("uniform", 1, 13000, 400), ("modelnet", 1, 16384, 900),
         ("uniform", 1, 16385, 100),
         ("uniform", 3, 1025, 300), ("uniform", 2, 2048, 2048), ("s3dis", 4, 2048, 768), ("uniform", 2, 3000, 700),
         ("s3dis", 2, 4096, 1024), ("uniform", 2, 5000, 1200), ("s3dis", 3, 8192, 2048), ("modelnet", 2, 10000, 2500),
         ("uniform", 1, 12288, 512), ("modelnet", 3, 2500, 625)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-B%d-N%d-m%d" % c)
def test_pruned_fps_equals_the_oracle(dev, case):
    kind, B, N, m = case
    pts = _cloud(kind, B, N, seed=77)
    np.testing.assert_array_equal(_fps(dev, pts, m), oracle.farthest_point_sample(m, pts))


def test_pruned_fps_ties_duplicates_and_clusters(dev):
    rng = np.random.RandomState(3)
    # every point twice (exact zero distances and equal running distances everywhere), shuffled
    base = rng.rand(2, 1500, 3).astype(np.float32)
    dup = np.concatenate([base, base], axis=1)
    for b in range(2):
        dup[b] = dup[b][rng.permutation(3000)]
    np.testing.assert_array_equal(_fps(dev, dup, 1700), oracle.farthest_point_sample(1700, dup))
    # a lattice: many exactly equal distances, ties decided by (k mod 1024, k) across waves and slots
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(12), indexing="ij"), -1).reshape(1, -1, 3)
    g = (g * 0.05).astype(np.float32)
    g = g[:, rng.permutation(g.shape[1])]
    np.testing.assert_array_equal(_fps(dev, g, 800), oracle.farthest_point_sample(800, g))
    # two distant clusters and a few stragglers: whole waves are skipped round after round
    c = np.concatenate([rng.randn(1, 3000, 3) * 0.05, rng.randn(1, 3000, 3) * 0.05 + 10.0, rng.rand(1, 144, 3) * 10.0], axis=1)
    c = c[:, rng.permutation(c.shape[1])].astype(np.float32)
    np.testing.assert_array_equal(_fps(dev, c, 1500), oracle.farthest_point_sample(1500, c))
    # a degenerate cloud: all points on a line (two zero extents of the sorting grid), and all points identical
    line = np.zeros((1, 2000, 3), np.float32)
    line[0, :, 1] = rng.rand(2000)
    np.testing.assert_array_equal(_fps(dev, line, 500), oracle.farthest_point_sample(500, line))
    same = np.full((1, 1500, 3), 0.25, np.float32)
    np.testing.assert_array_equal(_fps(dev, same, 20), oracle.farthest_point_sample(20, same))


def test_pruned_fps_with_non_finite_coordinates(dev):
    """NaN / Inf coordinates: the box tests turn conservative (a NaN never lets a slot be skipped), the samples stay the oracle's"""
    rng = np.random.RandomState(9)
    pts = rng.rand(2, 4096, 3).astype(np.float32)
    pts[0, 100] = (np.nan, 0.5, 0.5)
    pts[0, 2000, 2] = np.inf
    pts[1, 7, 0] = -np.inf
    with np.errstate(invalid="ignore", over="ignore"):
        want = oracle.farthest_point_sample(300, pts)
    np.testing.assert_array_equal(_fps(dev, pts, 300), want)


def test_pruned_fps_asked_for_more_samples_than_points(dev):
    """m > n: once every point is a sample all running distances are +0 and the reference keeps returning index 0"""
    rng = np.random.RandomState(1)
    pts = rng.rand(1, 2100, 3).astype(np.float32)
    want = oracle.farthest_point_sample(2105, pts)
    assert want[0, -5:].tolist() == [0, 0, 0, 0, 0]
    np.testing.assert_array_equal(_fps(dev, pts, 2105), want)
