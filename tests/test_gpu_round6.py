"""Round-6 graph-stream pieces: the pooling graph (both gathers + the counting pass of its transposed graph) in one launch, and
packed entries of the transposed graph, against the separate ops."""
import numpy as np
import pytest
import torch

from sph3d_gcn_amd import _lib, _tgraph, tf_nnquery, tf_pool3d
from sph3d_gcn_amd import sph3gcn_util as s3g_util
from sph3d_gcn_amd.harness import synth

pytestmark = pytest.mark.gpu


def _segments(tg, B, L):
    off = tg[0].cpu().numpy()
    key, scale = (t.cpu().numpy() for t in _tgraph.entries(tg))
    out = []
    for b in range(B):
        o = off[b * (L + 1):(b + 1) * (L + 1)]
        out.append([sorted(zip(key[o[i]:o[i + 1]].tolist(), scale[o[i]:o[i + 1]].tolist())) for i in range(L)])
    return off, out


@pytest.mark.parametrize("case", [(2, 2048, 512, 0.12, 64), (3, 700, 100, 0.2, 16), (16, 1024, 256, 0.15, 32)], ids=lambda c: "B%d-N%d-S%d" % c[:3])
def test_pooling_graph_one_launch_equals_two_gathers_and_a_transpose(dev, case):
    B, N, S, radius, K = case
    xyz = torch.from_numpy(synth.s3dis_batch(3, B, N)[0][:, :, :3].copy()).to(dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, radius, None, K)
    g = torch.Generator().manual_seed(S)
    pick = torch.stack([torch.randperm(N, generator=g)[:S] for _ in range(B)]).to(torch.int32).to(dev)
    pairs = torch.stack([torch.arange(B, dtype=torch.int32, device=dev).view(B, 1).expand(B, S), pick], dim=2).contiguous()
    want_idx, want_cnt = s3g_util.gather_nd(idx, pairs), s3g_util.gather_nd(cnt, pairs)
    _tgraph.clear()
    got_idx, got_cnt = s3g_util.gather_pooling_graph(idx, cnt, pairs, with_transpose=True)
    assert torch.equal(got_idx, want_idx) and torch.equal(got_cnt, want_cnt)
    tg_fused = _tgraph.peek(got_idx, got_cnt, N, need_unique_rows=True)            # cached by the fused call
    assert tg_fused is not None
    tg_sep = _tgraph.transpose(want_idx, want_cnt, N, unique_rows=True)
    off_f, seg_f = _segments(tg_fused, B, N)
    off_s, seg_s = _segments(tg_sep, B, N)
    np.testing.assert_array_equal(off_f, off_s)
    assert seg_f == seg_s
    # and the max-pool gradient through it equals the scatter formulation
    feat = torch.randn(B, N, 32, device=dev, requires_grad=True)
    out, _mi = tf_pool3d.max_pool3d(feat, got_idx, got_cnt)
    go = torch.randn_like(out)
    (gi,) = torch.autograd.grad(out, feat, go)
    _tgraph.clear()
    feat2 = feat.detach().clone().requires_grad_(True)
    out2, _ = tf_pool3d.max_pool3d(feat2, want_idx.clone(), want_cnt.clone())       # no transpose cached: the scatter path
    (gi2,) = torch.autograd.grad(out2, feat2, go)
    np.testing.assert_allclose(gi.cpu().numpy(), gi2.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_packed_and_unpacked_entries_describe_the_same_graph(dev):
    B, N, K, F = 2, 1500, 48, 33
    xyz = torch.from_numpy(synth.s3dis_batch(5, B, N)[0][:, :, :3].copy()).to(dev)
    idx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, 0.15, K, [8, 2, 2], with_transpose=False)
    old = _tgraph.PACK_ENTRIES
    try:
        _tgraph.clear(); _tgraph.PACK_ENTRIES = True
        tg_p = _tgraph.transpose(idx, cnt, N, bin_index=filt, num_bins=F)
        assert tg_p[2] is None                                                    # no scale array
        _tgraph.clear(); _tgraph.PACK_ENTRIES = False
        tg_u = _tgraph.transpose(idx, cnt, N, bin_index=filt, num_bins=F)
        assert tg_u[2] is not None
    finally:
        _tgraph.PACK_ENTRIES = old
        _tgraph.clear()
    off_p, seg_p = _segments(tg_p, B, N * F)
    off_u, seg_u = _segments(tg_u, B, N * F)
    np.testing.assert_array_equal(off_p, off_u)
    assert seg_p == seg_u                                                         # same rows, bit-identical 1 / count
