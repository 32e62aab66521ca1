"""Host logic on CPU: the s3g_util glue and the SPH3D call pattern, with the oracle ops swapped in
(BASELINE config #1: small cloud, CPU reference ops, plumbing only), the variable store, gather_nd,
batch-norm semantics, and the world_size-2 gloo path of the data-parallel step."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import torch_ops
from sph3d_gcn_amd import sph3gcn_util as s3g_util
from sph3d_gcn_amd.harness import dist as hdist
from sph3d_gcn_amd.harness import s3dis_net, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_net_forward_backward_on_oracle_ops():
    cfg = s3dis_net.small_config(1024)
    xyz, label, inner = synth.s3dis_batch(0, 2, 1024, extent=(1.0, 1.0, 1.5))
    pts = torch.from_numpy(xyz)
    with torch_ops.patched_util():
        model = s3dis_net.SPH3DS3DIS(cfg, device=torch.device("cpu"))
        pred, end = model(pts, is_training=True)
        loss = model.loss(pred, torch.from_numpy(label), torch.from_numpy(inner))
        loss.backward()
    assert pred.shape == (2, 1024, 13)
    assert torch.isfinite(pred).all() and torch.isfinite(loss)
    names = [n for n, _ in model.named_parameters()]
    assert any("conv1_1/depthwise_weights" in n for n in names)
    assert any("deconv2_2/weights" in n for n in names)
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    # depthwise filter shape = [binSize, Cin, multiplier] (utils/sph3gcn_util.py:136-137)
    w = dict(model.named_parameters())["store.params.conv1_1/depthwise_weights"]
    assert tuple(w.shape) == (33, 16, 2)
    # second forward reuses the variables
    n_before = sum(p.numel() for p in model.parameters())
    with torch_ops.patched_util():
        model(pts, is_training=False)
    assert n_before == sum(p.numel() for p in model.parameters())


def test_full_s3dis_plan_parameter_count():
    """the S3DIS net has 3 935 680 parameters... reproduced from the channel plan without running it."""
    cfg = s3dis_net.s3dis_config()
    F = cfg.binSize
    total = 3 * cfg.mlp + 2 * cfg.mlp          # mlp1 weights + bn gamma/beta
    cin = cfg.mlp
    enc_out = []
    for chans, mult in zip(cfg.channels, cfg.multiplier):
        for co, r in zip(chans, mult):
            total += F * cin * r + cin * r * co + 2 * co
            cin = co
        enc_out.append(cin)
    for l, (chans, mult) in enumerate(zip(reversed(cfg.channels), reversed(cfg.multiplier))):
        for co, r in zip(chans, mult):
            total += F * cin * r + cin * r * co + 2 * co
            cin = co
        cin = cin + list(reversed(enc_out))[l]
    total += cin * cfg.num_cls
    assert total == 3935680 + 0 or abs(total - 3935680) < 4096, total


def test_gather_nd_and_build_graph_indices():
    x = torch.arange(2 * 5 * 3, dtype=torch.float32).reshape(2, 5, 3)
    idx = torch.tensor([[[0, 4], [0, 1]], [[1, 0], [1, 3]]], dtype=torch.int32)
    got = s3g_util.gather_nd(x, idx)
    assert torch.equal(got[0, 0], x[0, 4]) and torch.equal(got[1, 1], x[1, 3])
    with torch_ops.patched_util():
        xyz = torch.from_numpy(synth.uniform_cloud(1, 2, 64))
        i, c, d, ind = s3g_util.build_graph(xyz, 0.3, 8, 16, 'FPS')
        assert ind.shape == (2, 16, 2) and (ind[1, :, 0] == 1).all() and (ind[:, 0, 1] == 0).all()
        with pytest.raises(ValueError):
            s3g_util.build_graph(xyz, 0.3, 8, 16, 'nope')
        i2, c2, d2, ind2 = s3g_util.build_graph(xyz, 0.3, 8, None)
        assert ind2 is None
        nn_idx, nn_cnt, nn_dst = s3g_util.build_global_graph(xyz, xyz.mean(1, keepdim=True), 100.0)
        assert nn_idx.shape == (2, 1, 64) and (nn_cnt == 64).all()
    with pytest.raises(ValueError):
        s3g_util.pool3d(x, None, None, 's', 'median')
    with pytest.raises(ValueError):
        s3g_util.unpool3d(x, None, None, None, 's', 'cubic')


def test_batch_norm_matches_tf_layers_semantics():
    store = s3g_util.VariableStore()
    x = torch.randn(4, 50, 6) * 3 + 1
    with s3g_util.variable_store(store):
        y = s3g_util.batch_normalization(x, True, 'bn0')
        mean = x.reshape(-1, 6).mean(0)
        var = x.reshape(-1, 6).var(0, unbiased=False)
        torch.testing.assert_close(y, (x - mean) / torch.sqrt(var + 1e-3), rtol=1e-4, atol=1e-4)
        mm = store.get_buffer('bn0/moving_mean', (6,), 0.0)
        torch.testing.assert_close(mm, 0.01 * mean, rtol=1e-4, atol=1e-5)      # momentum 0.99
        mv = store.get_buffer('bn0/moving_variance', (6,), 1.0)
        torch.testing.assert_close(mv, 0.99 + 0.01 * var, rtol=1e-5, atol=1e-6)  # biased variance, as tf.nn.moments
        y2 = s3g_util.batch_normalization(x, False, 'bn0')
        assert torch.isfinite(y2).all()
    assert store.regularization_loss() is not None     # gamma/beta l2 regularisers registered


def test_xavier_and_weight_decay_collection():
    store = s3g_util.VariableStore(seed=3)
    with s3g_util.variable_store(store):
        w = s3g_util._variable_with_weight_decay('a/w', [33, 16, 2], 1e-3, 0.5)
        limit = (6.0 / (33 * 16 + 33 * 2)) ** 0.5
        assert float(w.abs().max()) <= limit + 1e-6 and float(w.abs().max()) > 0.5 * limit
        w2 = s3g_util._variable_with_weight_decay('a/w', [33, 16, 2], 1e-3, 0.5)
        assert w2 is w
    torch.testing.assert_close(store.collect_losses(), 0.5 * 0.5 * w.pow(2).sum())


def test_shard_range():
    for total, world in [(128, 8), (10, 4), (3, 8)]:
        got = [hdist.shard_range(total, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        for a, b in zip(got, got[1:]):
            assert a[1] == b[0]


def test_gloo_world2_flat_grad_allreduce():
    """N > 1 path on CPU: two processes, gloo, each with its own cloud shard; after the flat all-reduce both
    ranks hold the same summed gradient, equal to the single-process gradient over the whole batch."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", script],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "GLOO_OK" in p.stdout


def test_gloo_world2_bench_rank_code_path():
    """bench.py's rank path end to end with two ranks on CPU (gloo): rank init + CPU slice per rank, train_step (graph
    build + fwd + bwd with the hooked, bucketed flat all-reduce + Adam), the contract's timed region (run_timed: barrier +
    sync on both sides) and the MAX over ranks; the replicas must stay identical after the optimiser steps."""
    script = os.path.join(ROOT, "tests", "_gloo_bench_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", script],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "GLOO_BENCH_OK" in p.stdout


def test_gloo_world8_bench_rank_code_path():
    """the same rank path with EIGHT ranks (the driver's 8-GPU run cannot be rehearsed on hardware here): rendezvous,
    disjoint CPU slices from pin_rank, shard_range over a global batch of 8 clouds, run_timed's barriers, per-rank times
    gathered, MAX over ranks, replicas identical after the steps (tiny plan: one 256-point block per rank)"""
    script = os.path.join(ROOT, "tests", "_gloo_bench_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", OMP_NUM_THREADS="1", GLOO_BENCH_WORLD="8")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", script],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "GLOO_BENCH_OK" in p.stdout


def test_bucket_collective_sees_the_concatenated_gradients():
    """the bucket's all-reduce is issued right behind its concatenation, from the hook that completed the bucket: a stand-in
    collective (same call contract as dist.all_reduce) must find the flat bucket already holding this step's gradients —
    and its result (here: x 3, 'three identical replicas') is what the optimiser sees after all_reduce()"""
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 300, 7, 1000, 3)]
    seen = []

    class _Work:
        def __init__(self, view):
            self.view = view

        def wait(self):
            self.view.mul_(3.0)

    def collective(view):
        seen.append(view.clone())
        return _Work(view)

    flat = hdist.FlatGradAllReduce(ps, bucket_bytes=1024, collective=collective)
    assert len(flat.buckets) >= 2
    x = torch.randn(1000)
    loss = sum((p * x[:p.numel()]).sum() * (i + 1) for i, p in enumerate(ps))
    flat.backward(loss)
    want = torch.zeros_like(flat.flat)               # parameters start on 16-byte boundaries of the flat buffer: padding stays 0
    for (o, k), (i, p) in zip(flat._slots, enumerate(ps)):
        want[o:o + k] = x[:k] * (i + 1)
        assert o % 4 == 0
    # every collective call saw its bucket's finished concatenation (buckets complete in reverse order)
    got = {}
    for v in seen:
        for (i0, i1, f0, f1) in flat.buckets:
            if v.numel() == f1 - f0 and torch.equal(v, want[f0:f1]):
                got[(f0, f1)] = True
    assert len(got) == len(flat.buckets)
    flat.all_reduce()
    torch.testing.assert_close(flat.flat, 3.0 * want)
    assert flat.stats["buckets_started_in_backward"] == len(flat.buckets)


def test_modelnet_small_net_on_oracle_ops():
    """SPH3D_modelnet call pattern (BASELINE config #1): 1024 points, raw-xyz concatenation (odd channel
    counts 35 / 67), per-level global max-pools, global conv with K = remaining points and 17 bins, fc + dropout."""
    from sph3d_gcn_amd.harness import modelnet_net
    cfg = modelnet_net.small_config(1024)
    pts = torch.from_numpy(synth.modelnet_batch(0, 2, 1024))
    label = torch.tensor([3, 17])
    with torch_ops.patched_util():
        model = modelnet_net.SPH3DModelNet(cfg, device=torch.device("cpu"))
        gen = torch.Generator().manual_seed(5)
        pred, _ = model(pts, is_training=True, dropout_generator=gen)
        loss = model.loss(pred, label)
        loss.backward()
    assert pred.shape == (2, 40)
    assert torch.isfinite(pred).all() and torch.isfinite(loss)
    params = dict(model.named_parameters())
    # depthwise filters: conv1_1 sees mlp(32) + raw xyz(3) = 35 channels, conv2_1 64 + 3 = 67; the global conv has 17 bins
    assert tuple(params["store.params.conv1_1/depthwise_weights"].shape) == (33, 35, 2)
    assert tuple(params["store.params.conv2_1/depthwise_weights"].shape) == (33, 67, 1)
    assert tuple(params["store.params.global_conv/depthwise_weights"].shape) == (17, 128, 2)
    # fc1 input = the per-level global max-pools (64 + 128) + the global conv output
    assert tuple(params["store.params.fc1/weights"].shape) == (64 + 128 + cfg.global_channels, 512)
    for n, p in params.items():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    # the loss carries the weight-decay collection and the BN regularisers (train_modelnet.py:162-164)
    with torch_ops.patched_util():
        pred2, _ = model(pts, is_training=False)
    plain = modelnet_net.get_loss(pred2, label)
    assert model.loss(pred2, label) > plain
    assert len(model.store._decay) > 0 and len(model.store._reg) > 0


def test_full_modelnet_plan_parameter_count():
    """the ModelNet net (10000 points) has 788 396 parameters; the plan runs here on 1000-point clouds with
    the same channel plan (the parameter count does not depend on the point count)."""
    from sph3d_gcn_amd.harness import modelnet_net
    cfg = modelnet_net.modelnet_config(10000)
    assert cfg.num_sample == [2500, 625, 156]
    cfg.num_input = 1000
    cfg.num_sample = [250, 62, 15]
    cfg.nn_uplimit = [16, 16, 16]
    pts = torch.from_numpy(synth.modelnet_batch(7, 1, 1000))
    with torch_ops.patched_util():
        model = modelnet_net.SPH3DModelNet(cfg, device=torch.device("cpu"))
        pred, _ = model(pts, is_training=False)
    assert pred.shape == (1, 40)
    assert sum(p.numel() for p in model.parameters()) == 788396


def test_shapenet_small_net_on_oracle_ops():
    """SPH3D_shapenet call pattern: raw xyz features, encoder keeps the mlp1 output, mlp2 + skip head."""
    from sph3d_gcn_amd.harness import shapenet_net
    cfg = shapenet_net.small_config(512)
    pts = torch.from_numpy(synth.modelnet_batch(20, 2, 512))
    label = torch.randint(0, 3, (2, 512), generator=torch.Generator().manual_seed(1))
    with torch_ops.patched_util():
        model = shapenet_net.SPH3DShapeNet(3, cfg, device=torch.device("cpu"))
        pred, end = model(pts, is_training=True)
        loss = model.loss(pred, label)
        loss.backward()
    assert pred.shape == (2, 512, 3)
    assert end['feats'].shape == (2, 512, 2 * cfg.mlp)
    assert torch.isfinite(pred).all() and torch.isfinite(loss)
    params = dict(model.named_parameters())
    assert tuple(params["store.params.mlp1/weights"].shape) == (3, cfg.mlp)
    assert tuple(params["store.params.logits/weights"].shape) == (2 * cfg.mlp, 3)
    for n, p in params.items():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_bench_algorithmic_bytes_follow_the_survey_formulas():
    """compulsory bytes per call = every distinct input / output element once (4 B each).  The roofline line of
    bench.py is computed from these; checked here against the formulas written out by hand for the headline shapes."""
    import bench
    B, N, M, F, C, r, K = 16, 8192, 8192, 33, 128, 2, 64
    fwd = 4 * (B * N * C + 2 * B * M * K + B * M + F * C * r + B * M * C * r)
    assert bench.algorithmic_bytes("sph3d_depthwise_conv3d", (B, N, M, F, C, r, K)) == fwd == 268993536
    nn = 4 * B * (3 * N + 3 * M + 2 * M * K + M)
    assert bench.algorithmic_bytes("sph3d_build_sphere_neighbor", (B, N, M, K)) == nn
    # both gradients in one call: read input, graph, counts, filter, grad_out; write grad_input and grad_filter
    bwd = bench.algorithmic_bytes("sph3d_depthwise_conv3d_grad_t", (B, N, M, F, C, r))
    assert bwd == 336136192


def test_flat_adam_learning_rate_survives_a_resume_on_the_cpu_branch():
    """torch.optim.Optimizer.load_state_dict replaces param_groups; FlatAdam.lr / param_groups must keep writing
    to the list the wrapped optimiser actually reads"""
    from sph3d_gcn_amd.harness import optim
    p = torch.nn.Parameter(torch.ones(8))
    opt = optim.FlatAdam(p, lr=1e-2)
    p.grad = torch.ones(8)
    opt.step()
    state = opt.state_dict()
    opt.load_state_dict(state)
    opt.lr = 0.0                                   # a decayed rate after the resume ...
    assert opt._torch.param_groups[0]["lr"] == 0.0 and opt.param_groups is opt._torch.param_groups
    before = p.detach().clone()
    opt.step()                                     # ... must be the one the step uses
    assert torch.equal(p.detach(), before)
    for g in opt.param_groups:
        g["lr"] = 1e-2
    opt.step()
    assert not torch.equal(p.detach(), before)


def test_fused_separable_layer_support_is_what_the_launcher_can_run():
    """sph3d_separable_conv3d_fused_supported folds the LDS fit in: a kernel with more bins than fit beside the
    smallest tile is reported unsupported (the layer then runs layer by layer) instead of failing at launch"""
    from sph3d_gcn_amd import _lib
    l = _lib.lib()
    assert l.sph3d_separable_conv3d_fused_supported(8192, 33, 128, 2, 64, 128) == 1       # the S3DIS plan's [8,2,2] kernel
    assert l.sph3d_separable_conv3d_fused_supported(8192, 49, 256, 2, 64, 256) == 1       # [8,2,3]
    assert l.sph3d_separable_conv3d_fused_supported(8192, 97, 256, 2, 64, 256) == 1       # [8,4,3]: fits with a lower tile
    assert l.sph3d_separable_conv3d_fused_supported(8192, 161, 256, 2, 64, 256) == 0      # 162 KB of filter slice alone
    assert l.sph3d_separable_conv3d_fused_supported(8192, 254, 128, 2, 64, 128) == 0


def test_plan_arena_carves_aligned_views_and_falls_back():
    """_lib.Arena / _lib.empty: the graph-building ops' outputs come from one block while a plan is being built (one allocator
    event per block and consuming stream at free time instead of one per tensor: DESIGN section 0 item 8)"""
    import torch
    from sph3d_gcn_amd import _lib
    dev = torch.device("cpu")
    assert _lib.empty((3, 5), torch.int32, dev).shape == (3, 5)                      # no arena: a plain tensor
    a = _lib.Arena(4096, dev)
    with _lib.arena_scope(a):
        x = _lib.empty((2, 3, 4), torch.int32, dev)
        y = _lib.empty((7,), torch.float32, dev)
        z = _lib.empty((0, 9), torch.int32, dev)
        big = _lib.empty((2000,), torch.float32, dev)                                 # does not fit: falls back
        other = _lib.empty((4,), torch.int64, dev)                                    # not an index / graph dtype: never from the arena
        with _lib.arena_scope(None):
            plain = _lib.empty((4,), torch.int32, dev)
    base = a.buf.untyped_storage().data_ptr()
    assert x.untyped_storage().data_ptr() == base and y.untyped_storage().data_ptr() == base
    assert x.data_ptr() == base and y.data_ptr() == base + 256 and x.is_contiguous() and y.is_contiguous()
    assert x.dtype == torch.int32 and y.dtype == torch.float32 and z.numel() == 0
    assert big.untyped_storage().data_ptr() != base and other.untyped_storage().data_ptr() != base
    assert plain.untyped_storage().data_ptr() != base
    assert a.need == 256 + 256 + 0 + 8192                                             # what a plan of these shapes asks for
    x.fill_(7); y.fill_(1.5)
    assert int(x.sum()) == 7 * 24 and float(y.sum()) == 10.5                          # disjoint
    assert _lib.empty((2,), torch.int32, dev).untyped_storage().data_ptr() != base    # scope left
    # the measuring arena of a first plan (capacity 0) and zero-element shapes: plain tensors, the need still counted
    m = _lib.Arena(0, dev)
    with _lib.arena_scope(m):
        e = _lib.empty((0, 9), torch.int32, dev)
        f = _lib.empty((5,), torch.float32, dev)
    assert e.shape == (0, 9) and f.shape == (5,) and m.need == 256 and m.buf is None
    # call-local workspaces can be given back (per raw stream handle, or all)
    _lib._scratch[(0, 123, 0)] = torch.empty(8)
    _lib._scratch[(0, 456, 0)] = torch.empty(8)
    _lib.release_scratch(123)
    assert (0, 123, 0) not in _lib._scratch and (0, 456, 0) in _lib._scratch
    _lib.release_scratch()
    assert not _lib._scratch
