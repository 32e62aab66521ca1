"""The neighbour-search entry points at the C-ABI boundary (include/sph3d.h): the `_ws` forms take the cell grid's
memory from the caller and never allocate; the convenience forms (the reference launcher's signature) keep one library buffer
per (device, stream).  Same rows bit for bit through every door; streams do not share state; a released / recycled stream
handle starts from nothing."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import synth

pytestmark = pytest.mark.gpu


def _search(l, dev, db, K, radius, how, stream=None, ws=None):
    B, N, _ = db.shape
    idx = torch.empty((B, N, K), dtype=torch.int32, device=dev)
    cnt = torch.empty((B, N), dtype=torch.int32, device=dev)
    dst = torch.empty((B, N, K), dtype=torch.float32, device=dev)
    st = stream.cuda_stream if stream is not None else _lib.stream_ptr()
    if how == "convenience":
        rc = l.sph3d_build_sphere_neighbor(B, N, N, K, radius, _lib.ptr(db), _lib.ptr(db), _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(dst), st)
    else:
        rc = l.sph3d_build_sphere_neighbor_ws(B, N, N, K, radius, _lib.ptr(db), _lib.ptr(db), _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(dst),
                                              _lib.ptr(ws), 0 if ws is None else ws.numel(), st)
    return rc, idx, cnt, dst


def test_workspace_forms_equal_the_convenience_forms_and_the_oracle(dev):
    l = _lib.lib()
    l.sph3d_release_all_scratch()
    xyz = synth.s3dis_batch(21, 2, 4096)[0][:, :, :3].copy()
    db = torch.from_numpy(xyz).to(dev)
    K, radius = 32, 0.1
    idx_o, cnt_o, dst_o = oracle.build_sphere_neighbor(xyz, xyz, radius, None, K)
    need = l.sph3d_build_sphere_neighbor_workspace(2, 4096, 4096)
    assert need > 0
    results = []
    before = l.sph3d_nngrid_launches()
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    results.append(_search(l, dev, db, K, radius, "ws", ws=ws))
    assert l.sph3d_nngrid_launches() == before + 1               # the grid ran out of the caller's memory
    results.append(_search(l, dev, db, K, radius, "ws", ws=None))
    assert l.sph3d_nngrid_launches() == before + 1               # no workspace: no grid (and no allocation): the chain kernel
    results.append(_search(l, dev, db, K, radius, "convenience"))
    assert l.sph3d_nngrid_launches() == before + 2
    torch.cuda.synchronize()
    for rc, idx, cnt, dst in results:
        assert rc == 0
        np.testing.assert_array_equal(cnt.cpu().numpy(), cnt_o)
        np.testing.assert_array_equal(idx.cpu().numpy(), idx_o)
        np.testing.assert_array_equal(dst.cpu().numpy().view(np.int32), dst_o.view(np.int32))
    # a workspace that is too small is refused, not silently replaced
    small = torch.empty((need // 2,), dtype=torch.uint8, device=dev)
    rc, *_ = _search(l, dev, db, K, radius, "ws", ws=small)
    assert rc == -2 and b"workspace" in l.sph3d_last_error()
    # exactly one library buffer exists (the convenience call's), and the hook frees it
    assert l.sph3d_release_stream_scratch(ctypes.c_void_p(_lib.stream_ptr())) == 1
    assert l.sph3d_release_all_scratch() == 0


def test_two_streams_and_a_recycled_stream_handle(dev):
    """Two streams search different clouds at the same time through the convenience entry point: each gets its own grid buffer
    (a shared one would be overwritten mid-search).  Then a stream is released, destroyed and a new one created — whatever handle
    it gets, it starts without a buffer and its search is right."""
    l = _lib.lib()
    l.sph3d_release_all_scratch()
    K, radius = 32, 0.1
    clouds = [synth.s3dis_batch(31 + i, 4, 8192)[0][:, :, :3].copy() for i in range(2)]
    dbs = [torch.from_numpy(c).to(dev) for c in clouds]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):                                          # interleaved issue: the searches overlap on the device
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                outs.append((i, _search(l, dev, dbs[i], K, radius, "convenience", stream=streams[i])))
    torch.cuda.synchronize()
    refs = [oracle.build_sphere_neighbor(c, c, radius, None, K) for c in clouds]
    for i, (rc, idx, cnt, dst) in outs:
        assert rc == 0
        np.testing.assert_array_equal(cnt.cpu().numpy(), refs[i][1])
        np.testing.assert_array_equal(idx.cpu().numpy(), refs[i][0])
    for s in streams:
        assert l.sph3d_release_stream_scratch(ctypes.c_void_p(s.cuda_stream)) == 1
    del streams, outs
    fresh = torch.cuda.Stream(device=dev)                          # may or may not reuse a destroyed handle
    assert l.sph3d_release_stream_scratch(ctypes.c_void_p(fresh.cuda_stream)) == 0      # nothing inherited
    with torch.cuda.stream(fresh):
        rc, idx, cnt, dst = _search(l, dev, dbs[1], K, radius, "convenience", stream=fresh)
    torch.cuda.synchronize()
    assert rc == 0
    np.testing.assert_array_equal(idx.cpu().numpy(), refs[1][0])
    assert l.sph3d_release_all_scratch() == 1


def test_workspace_form_runs_under_stream_capture(dev):
    """nothing in the `_ws` form allocates or synchronises: the whole search can be captured into a HIP graph and replayed"""
    l = _lib.lib()
    xyz = synth.s3dis_batch(41, 2, 4096)[0][:, :, :3].copy()
    db = torch.from_numpy(xyz).to(dev)
    K, radius = 32, 0.1
    need = l.sph3d_build_sphere_neighbor_workspace(2, 4096, 4096)
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    B, N = 2, 4096
    idx = torch.zeros((B, N, K), dtype=torch.int32, device=dev)
    cnt = torch.zeros((B, N), dtype=torch.int32, device=dev)
    dst = torch.zeros((B, N, K), dtype=torch.float32, device=dev)
    s = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            rc = l.sph3d_build_sphere_neighbor_ws(B, N, N, K, radius, _lib.ptr(db), _lib.ptr(db), _lib.ptr(idx), _lib.ptr(cnt),
                                                  _lib.ptr(dst), _lib.ptr(ws), need, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert int(cnt.abs().sum()) == 0                               # captured, not run
    g.replay()
    torch.cuda.synchronize()
    idx_o, cnt_o, dst_o = oracle.build_sphere_neighbor(xyz, xyz, radius, None, K)
    np.testing.assert_array_equal(cnt.cpu().numpy(), cnt_o)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_o)


def test_graph_plan_tensors_share_one_block_per_stream_and_equal_the_plain_plan(dev):
    """harness/s3dis_net.GraphPlan carves the outputs of the graph-building ops from one block per producing stream (_lib.Arena), sized
    by the previous plan of the same shapes: the second plan's index / graph tensors share a storage per stream, and every
    tensor equals the one a plan without arenas builds (integers: bit for bit)"""
    from sph3d_gcn_amd import _tgraph
    from sph3d_gcn_amd.harness import s3dis_net
    cfg = s3dis_net.s3dis_config(2048)
    xyz, _, _ = synth.s3dis_batch(77, 2, 2048)
    pts = torch.from_numpy(xyz).to(dev)
    torch.cuda.synchronize()

    def tensors(plan):
        out = {}
        for l, g in sorted(plan._enc.items()):
            for k, v in g.items():
                if torch.is_tensor(v):
                    out["enc%d.%s" % (l, k)] = v
        for l, g in sorted(plan._dec.items()):
            for k, v in g.items():
                if torch.is_tensor(v):
                    out["dec%d.%s" % (l, k)] = v
        for i, t in enumerate(plan.indices):
            if torch.is_tensor(t):
                out["indices%d" % i] = t
        for i, t in enumerate(plan.xyz_layers):
            out["xyz%d" % i] = t
        return out

    old = s3dis_net._USE_ARENA
    try:
        s3dis_net._USE_ARENA = False
        _tgraph.clear()
        plain = s3dis_net.GraphPlan(pts, cfg)
        torch.cuda.synchronize()
        ref = {k: v.clone() for k, v in tensors(plain).items()}
        s3dis_net._USE_ARENA = True
        s3dis_net._ARENA_NEED.clear()
        _tgraph.clear()
        first = s3dis_net.GraphPlan(pts, cfg)              # measures
        torch.cuda.synchronize()
        assert len(s3dis_net._ARENA_NEED) == 1 and min(next(iter(s3dis_net._ARENA_NEED.values()))) > 0
        _tgraph.clear()
        second = s3dis_net.GraphPlan(pts, cfg)             # carves
        torch.cuda.synchronize()
    finally:
        s3dis_net._USE_ARENA = old
    got = tensors(second)
    assert set(got) == set(ref)
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    graph_store = {got[k].untyped_storage().data_ptr() for k in got if k.startswith(("enc", "dec")) and got[k].dtype == torch.int32}
    assert len(graph_store) == 1, "the graph stream's index tensors of a plan live in one block"
    plain_store = {v.untyped_storage().data_ptr() for k, v in tensors(plain).items() if k.startswith(("enc", "dec"))}
    assert len(plain_store) > 8
    _tgraph.clear()
