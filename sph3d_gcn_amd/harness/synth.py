"""Seeded synthetic point clouds shaped like the reference's datasets (no datasets / network here).

Geometry follows SURVEY §8d: S3DIS-like blocks are 2.1 x 2.1 x 3.0 m room slabs (1.5 m block + 0.3 m
context each side, io/make_tfrecord_s3dis.py:205-209,250-251) with floor, ceiling, two walls and a few
boxes, snapped to a 3 cm grid and de-duplicated (preprocesing/s3dis_prepare_data.m), then
8192 points drawn without replacement and shuffled (s3dis_seg/train_s3dis.py:343-347,124-129).
ModelNet-like clouds are surfaces of random boxes/ellipsoids scaled to the unit sphere
(io/make_tfrecord_modelnet.py:93-95).
"""
import numpy as np


def _surface_points(rng, n, ext):
    """n points on floor, ceiling, two walls and three boxes inside [0,ext] (x,y,z)."""
    ex, ey, ez = ext
    parts = []
    k = n // 8
    u = rng.rand(k * 2, 2)
    floor = np.stack([u[:k, 0] * ex, u[:k, 1] * ey, np.zeros(k)], 1)
    ceil = np.stack([u[k:, 0] * ex, u[k:, 1] * ey, np.full(k, ez)], 1)
    parts += [floor, ceil]
    u = rng.rand(k * 2, 2)
    wall1 = np.stack([np.zeros(k), u[:k, 0] * ey, u[:k, 1] * ez], 1)
    wall2 = np.stack([u[k:, 0] * ex, np.full(k, ey), u[k:, 1] * ez], 1)
    parts += [wall1, wall2]
    rest = n - 4 * k
    per = rest // 3
    for b in range(3):
        m = per if b < 2 else rest - 2 * per
        size = 0.3 + 0.7 * rng.rand(3)
        org = rng.rand(3) * (np.array([ex, ey, ez * 0.5]) - size).clip(min=0.05)
        face = rng.randint(0, 6, m)
        p = rng.rand(m, 3) * size
        for ax in range(3):
            p[face == 2 * ax, ax] = 0.0
            p[face == 2 * ax + 1, ax] = size[ax]
        parts.append(p + org)
    return np.concatenate(parts, 0)


def s3dis_block(block_id, num_point=8192, extent=(2.1, 2.1, 3.0), voxel=0.03):
    """One S3DIS-like block: (xyz float32 [num_point,3], label int64 [num_point], inner float32 [num_point])."""
    rng = np.random.RandomState(1234 + block_id)
    want = int(num_point * 1.25) + 256
    pts = np.zeros((0, 3))
    n_raw = want * 2
    while pts.shape[0] < want:
        raw = _surface_points(rng, n_raw, extent)
        raw = np.round(raw / voxel) * voxel
        pts = np.unique(np.concatenate([pts, raw], 0), axis=0)
        n_raw *= 2
    sel = rng.choice(pts.shape[0], num_point, replace=False)
    xyz = pts[sel]
    xyz = xyz[rng.permutation(num_point)]
    label = rng.randint(0, 13, num_point)
    cx, cy = extent[0] / 2, extent[1] / 2
    inner = ((np.abs(xyz[:, 0] - cx) <= 0.75) & (np.abs(xyz[:, 1] - cy) <= 0.75)).astype(np.float32)
    return xyz.astype(np.float32), label.astype(np.int64), inner


def s3dis_batch(first_block, batch, num_point=8192, extent=(2.1, 2.1, 3.0)):
    xs, ls, ins = [], [], []
    for b in range(batch):
        x, l, i = s3dis_block(first_block + b, num_point, extent)
        xs.append(x)
        ls.append(l)
        ins.append(i)
    return np.stack(xs), np.stack(ls), np.stack(ins)


def modelnet_cloud(cloud_id, num_point=1024):
    """Points on a random box or ellipsoid surface, centred, scaled to unit radius."""
    rng = np.random.RandomState(100 + cloud_id)
    if cloud_id % 2 == 0:
        v = rng.randn(num_point, 3)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        pts = v * (0.3 + 0.7 * rng.rand(3))
    else:
        size = 0.3 + 0.7 * rng.rand(3)
        face = rng.randint(0, 6, num_point)
        pts = (rng.rand(num_point, 3) - 0.5) * size
        for ax in range(3):
            pts[face == 2 * ax, ax] = -size[ax] / 2
            pts[face == 2 * ax + 1, ax] = size[ax] / 2
    pts = pts - pts.mean(0, keepdims=True)
    pts = pts / np.max(np.linalg.norm(pts, axis=1))
    return pts.astype(np.float32)


def modelnet_batch(first, batch, num_point=1024):
    return np.stack([modelnet_cloud(first + b, num_point) for b in range(batch)])


def uniform_cloud(seed, B, N, scale=1.0):
    rng = np.random.RandomState(seed)
    return (rng.rand(B, N, 3).astype(np.float32) * scale).astype(np.float32)
