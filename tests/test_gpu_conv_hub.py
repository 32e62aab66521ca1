"""The conv gradient's hub path (csrc/conv3d.hip: dwconv_bwd_t_vec<..., HUB>): sources with more in-edges than a threshold are left
out of the persistent sweep and shared among the waves of a second launch.  Forced on at small sizes through the per-call
switches SPH3D_BWD_HUB_MIN_N / SPH3D_BWD_HUB_T and compared with the oracle (tf_ops/convolution/tf_conv3d_gpu.cu:32-101 restated)
and with the ordinary path; graphs with a few sources of hundreds to thousands of in-edges spread over several bins, with and
without the compact bin table, full / half / quarter-wave forms, and the concatenated-input variant."""
import os

import numpy as np
import pytest
import torch

import oracle
from sph3d_gcn_amd import tf_conv3d

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _n(t):
    return t.detach().cpu().numpy()


def _hub_graph(B, N, K, nbins, hubs, seed):
    """every query lists up to K ascending neighbours; the first `hubs` points are neighbours of most queries (in random bins)"""
    rng = np.random.RandomState(seed)
    cnt = rng.randint(max(1, K // 2), K + 1, size=(B, N)).astype(np.int32)
    idx = np.zeros((B, N, K), np.int32)
    filt = np.zeros((B, N, K), np.int32)
    for b in range(B):
        for m in range(N):
            c = int(cnt[b, m])
            h = [i for i in range(hubs) if rng.rand() < 0.9][:c]
            rest = rng.permutation(np.arange(hubs, N))[:c - len(h)]
            idx[b, m, :c] = np.sort(np.concatenate([np.array(h, np.int64), rest]))[:c]
            filt[b, m, :c] = rng.randint(0, nbins, size=c)
    return idx, cnt, filt


@pytest.fixture
def hub_env():
    old = {k: os.environ.get(k) for k in ("SPH3D_BWD_HUB_MIN_N", "SPH3D_BWD_HUB_T")}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("nbins", [33, 12], ids=["bins33", "bins12-compact"])
@pytest.mark.parametrize("C,r", [(128, 2), (64, 2), (32, 2), (128, 1), (64, 1)])
def test_hub_path_equals_oracle_and_ordinary_path(dev, hub_env, C, r, nbins):
    B, N, K = 2, 1536, 16
    idx, cnt, filt = _hub_graph(B, N, K, nbins, hubs=5, seed=C * 3 + r + nbins)
    indeg = np.bincount(idx[0][np.arange(K)[None, :] < cnt[0][:, None]], minlength=N)
    assert indeg[:5].min() > 900 and indeg[5:].max() < 200          # five hubs per cloud, far above the threshold used below
    rng = np.random.RandomState(C + r)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(33, C, r).astype(np.float32)
    go = rng.randn(B, N, C * r).astype(np.float32)
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    args = (_t(x, dev), _t(w, dev), _t(go, dev), _t(idx, dev), _t(cnt, dev), _t(filt, dev))
    os.environ["SPH3D_BWD_HUB_MIN_N"] = "1000000"                     # ordinary path
    gi0, gf0 = tf_conv3d.depthwise_conv3d_grad(*args)
    os.environ["SPH3D_BWD_HUB_MIN_N"] = "1024"                        # hub path: sources with more than 256 in-edges
    os.environ["SPH3D_BWD_HUB_T"] = "256"
    gi1, gf1 = tf_conv3d.depthwise_conv3d_grad(*args)
    torch.cuda.synchronize()
    s_i = max(1.0, float(np.abs(gi_o).max()))
    s_f = max(1.0, float(np.abs(gf_o).max()))
    for gi, gf in ((gi0, gf0), (gi1, gf1)):
        np.testing.assert_allclose(_n(gi) / s_i, gi_o / s_i, **TOL)
        np.testing.assert_allclose(_n(gf) / s_f, gf_o / s_f, **TOL)
    # non-hub rows do not depend on the path at all
    assert torch.equal(gi0[:, 5:], gi1[:, 5:])


def test_hub_path_without_any_hub_and_with_every_source_a_hub(dev, hub_env):
    B, N, K, C, r = 1, 2048, 8, 64, 2
    idx, cnt, filt = _hub_graph(B, N, K, 33, hubs=0, seed=3)
    rng = np.random.RandomState(1)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(33, C, r).astype(np.float32)
    go = rng.randn(B, N, C * r).astype(np.float32)
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    args = (_t(x, dev), _t(w, dev), _t(go, dev), _t(idx, dev), _t(cnt, dev), _t(filt, dev))
    os.environ["SPH3D_BWD_HUB_MIN_N"] = "1024"
    for T in ("100000", "1"):                                         # no source is a hub / every source with 2+ in-edges is one
        os.environ["SPH3D_BWD_HUB_T"] = T
        gi, gf = tf_conv3d.depthwise_conv3d_grad(*args)
        np.testing.assert_allclose(_n(gi), gi_o, **TOL)
        s_f = max(1.0, float(np.abs(gf_o).max()))
        np.testing.assert_allclose(_n(gf) / s_f, gf_o / s_f, **TOL)


def test_hub_path_concatenated_inputs(dev, hub_env):
    from sph3d_gcn_amd.tf_conv3d import _depthwise_conv3d_cat_grad_impl, concat_supported
    B, N, K, Ca, Cb, r = 1, 1536, 16, 128, 128, 2
    idx, cnt, filt = _hub_graph(B, N, K, 33, hubs=4, seed=11)
    rng = np.random.RandomState(2)
    xa = rng.randn(B, N, Ca).astype(np.float32)
    xb = rng.randn(B, N, Cb).astype(np.float32)
    w = rng.randn(33, Ca + Cb, r).astype(np.float32)
    go = rng.randn(B, N, (Ca + Cb) * r).astype(np.float32)
    gi_o, gf_o = oracle.depthwise_conv3d_grad(np.concatenate([xa, xb], 2), w, go, idx, cnt, filt)
    assert concat_supported(_t(xa, dev), _t(xb, dev), _t(w, dev))
    os.environ["SPH3D_BWD_HUB_MIN_N"] = "1024"
    os.environ["SPH3D_BWD_HUB_T"] = "256"
    ga, gb, gf = _depthwise_conv3d_cat_grad_impl(_t(xa, dev), _t(xb, dev), _t(w, dev), _t(go, dev), _t(idx, dev), _t(cnt, dev), _t(filt, dev))
    s_i = max(1.0, float(np.abs(gi_o).max()))
    np.testing.assert_allclose(_n(torch.cat((ga, gb), 2)) / s_i, gi_o / s_i, **TOL)
    s_f = max(1.0, float(np.abs(gf_o).max()))
    np.testing.assert_allclose(_n(gf) / s_f, gf_o / s_f, **TOL)
