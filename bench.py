#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json: point-cloud blocks/s, forward + backward (+ Adam step),
SPH3D_s3dis-shaped network on 8192-point S3DIS-like blocks, 16 blocks per GPU (weak scaling).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank
   per GPU, or run plainly — without WORLD_SIZE in the environment the script re-launches itself that way)

A "step" = graph construction (nnquery, FPS, buildkernel: they depend on the input xyz, so they are part
of every step) + forward + loss + backward + gradient all-reduce (N > 1) + optimiser update, on one batch
of synthetic blocks already resident in HBM (two different batches alternate, so nothing a step computes can be
left over from the step before).  Rank 0 prints ONE JSON line (contract in the task brief), with two extra objects:
  roofline     — the dominant libsph3d op family of the step (GEMM NN / NT / TN count as one family): algorithmic
                 bytes or flops of its largest call (SURVEY §8d formulas) / that call's mean device time, measured
                 with HIP events on the launching stream (avg_us: inside the running step, other streams busy;
                 isolated_us: the same call alone on an idle GPU; trace_us: the kernel's duration in the committed
                 rocprofv3 trace of the same command, when profiles/ holds one);
  cpu_baseline — the same harness step on the CPU oracle (oracle/, OpenMP) on a bounded sample of the same
                 workload, at all host threads (median of 10 steps) and at one thread; rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # cpu_baseline leg: two OpenMP pools (torch, oracle) must not spin against each other

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from sph3d_gcn_amd import _lib
from sph3d_gcn_amd.harness import optim as hoptim
from sph3d_gcn_amd.harness import dist as hdist
from sph3d_gcn_amd.harness import s3dis_net, synth

BLOCKS_PER_GPU = 16
PRIME_STEPS = 16
NUM_POINT = 8192
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense (MI355X_MICROARCH.md); the split products execute 6 bf16 piece products per fp32 product
SPLIT_PRODUCTS = 6


def algorithmic_bytes(name, a):
    """Compulsory HBM bytes of one C-ABI call (every distinct input/output element once, 4 B each; gathered
    re-reads are NOT counted) — SURVEY §8(d).  `a` = the call's integer arguments in ABI order."""
    if name == "sph3d_build_sphere_neighbor":
        B, N, M, K = a[:4]
        return 4 * B * (3 * N + 3 * M + 2 * M * K + M)
    if name == "sph3d_spherical_kernel":
        B, N, M, K = a[:4]
        return 4 * B * (3 * N + 3 * M + 3 * M * K + M)
    if name == "sph3d_depthwise_conv3d":
        B, N, M, F, C, r, K = a[:7]
        return 4 * (B * N * C + 2 * B * M * K + B * M + F * C * r + B * M * C * r)
    if name == "sph3d_depthwise_conv3d_grad":
        B, N, M, F, C, r, K = a[:7]
        return 4 * (B * N * C + 2 * B * M * K + B * M + F * C * r + B * M * C * r) + 4 * (B * N * C + F * C * r)
    if name == "sph3d_depthwise_conv3d_grad_t":      # same op through the transposed graph (K = 64 rows of the path)
        B, N, M, F, C, r = a[:6]
        K = 64
        return 4 * (B * N * C + 2 * B * M * K + B * M + F * C * r + B * M * C * r) + 4 * (B * N * C + F * C * r)
    if name == "sph3d_scatter_grad_t":
        B, Nin, Mout, C = a[:4]
        return 4 * B * (Nin * C + Mout * 64 + Mout + Mout * C)
    if name == "sph3d_graph_transpose":
        B, N, M, K, F = a[:5]
        return 4 * B * (2 * M * K + M + N * F + 2 * M * K)
    if name == "sph3d_farthest_point_sample":
        b, n, m = a[:3]
        return 4 * b * (3 * n + m)
    if name in ("sph3d_max_pool3d", "sph3d_avg_pool3d"):
        B, N, M, C, K = a[:5]
        return 4 * B * (N * C + M * K + M + M * C + (M * C if name == "sph3d_max_pool3d" else 0))
    if name in ("sph3d_mean_interpolate", "sph3d_weighted_interpolate"):
        B, Nf, Mc, C, K = a[:5]
        return 4 * B * (Mc * C + Nf * K + Nf + Nf * C + (Nf * K if name == "sph3d_weighted_interpolate" else 0))
    if name in ("sph3d_avg_pool3d_grad", "sph3d_mean_interpolate_grad", "sph3d_weighted_interpolate_grad"):
        B, N, M, C, K = a[:5]
        return 4 * B * (N * C + M * K + M + M * C)
    if name == "sph3d_max_pool3d_grad":
        B, N, M, C = a[:4]
        return 4 * B * (N * C + 2 * M * C)
    if name == "sph3d_pointwise_gemm":
        R, Cin, Cout = a[:3]
        return 4 * (R * Cin + Cin * Cout + R * Cout)
    if name == "sph3d_pointwise_gemm_tn":
        R, Cin, Cout = a[:3]
        return 4 * (R * Cin + Cin * Cout + R * Cout)
    return 0


NUM_BATCHES = 2            # distinct resident batches, used in turn


def make_batch(rank, dev, which=0):
    first = 1000 + (which * 64 + rank) * BLOCKS_PER_GPU
    xyz, label, inner = synth.s3dis_batch(first, BLOCKS_PER_GPU, NUM_POINT)
    return (torch.from_numpy(xyz).to(dev), torch.from_numpy(label).to(dev), torch.from_numpy(inner).to(dev))


_PTS_READY = {}


def fwd_bwd(model, flat, pts, label, inner):
    # graphs are built inside (GraphPlan) on two side streams that wait only for the INPUT batch (resident in HBM
    # since before the timed region), so a step's sampling / graph construction overlaps the previous step's backward
    pred, _ = model(pts, is_training=True, points_ready=_PTS_READY.get(pts.data_ptr()))
    loss = model.loss(pred, label, inner)
    flat.backward(loss)          # all parameter gradients -> the flat fp32 buffer (one concatenation, no per-parameter adds)
    return loss


def train_step(model, flat, opt, pts, label, inner):
    loss = fwd_bwd(model, flat, pts, label, inner)
    flat.all_reduce()
    opt.step()
    return loss


def run_timed(one_step, steps, warmup, world, sync):
    """The contract's timed region: W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides;
    -> (seconds, MAX over the ranks; last loss).  `sync` = torch.cuda.synchronize on the GPU (a no-op for the CPU/gloo test
    that drives this same function with world 2)."""
    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(warmup):
        one_step()
    barrier()
    sync()
    t0 = time.perf_counter()
    loss = None
    for _ in range(steps):
        loss = one_step()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, loss


def probe_host_issue(model, flat, opt, batch, dev, steps=12):
    """ms the host needs to issue one training step: the step on ONE block of the batch (wall clock per step with the device
    nearly idle: it finishes each step long before the host has issued the next)"""
    pts, label, inner = (t[:1].contiguous() for t in batch)
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    ev.record()
    _PTS_READY[pts.data_ptr()] = ev

    def step():
        return train_step(model, flat, opt, pts, label, inner)      # (N > 1: with its collective — every rank runs the probe)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    _PTS_READY.pop(pts.data_ptr(), None)
    return (t1 - t0) / steps * 1e3


def probe_fps_chain(pts, config, reps=3):
    """ms of the sampling chain (the plan's farthest-point samplings, level after level) of one batch, alone on the device"""
    from sph3d_gcn_amd import sph3gcn_util as s3g
    xyz = pts[:, :, 0:3].contiguous()

    def chain():
        cur = xyz
        for m in config.num_sample:
            if m > 1:
                idx = s3g.farthest_point_sample(m, cur)
                cur = torch.gather(cur, 1, idx.long().unsqueeze(2).expand(-1, -1, 3)).contiguous()
    chain()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        chain()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def reduce_max_seconds(elapsed, world, dev):
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_blocks=BLOCKS_PER_GPU, max_timed=5, warm_steps=1, budget_s=45.0):
    """Same harness step (graph build + fwd + bwd + Adam) on the CPU oracle — kind = "port": oracle/ is the C restatement of
    the reference's kernels, OpenMP across independent work items; GEMM / BN / ELU run in torch-CPU (MKL/oneDNN) so the
    baseline is not handicapped.  Bounded sample: `sample_blocks` S3DIS-like 8192-point blocks per step — the SAME 16 blocks
    per step the GPU processes (round 2 timed 2 blocks per step: the oracle parallelises over clouds and rows, two clouds
    starve it — FPS is one thread per cloud — and its multi-thread figure was only 1.7x the one-thread one).  Protocol
    (SURVEY §8d): one step at each candidate thread count {physical cores, hardware threads} picks the faster; then
    `warm_steps` untimed and up to `max_timed` timed steps at that count inside `budget_s` seconds (at least 2), MEDIAN
    reported; then ONE block at ONE thread; then the four level-0 oracle ops alone at one thread and at the chosen count
    (per-op thread scaling)."""
    import oracle  # noqa: F401  (cpu_baseline leg: the oracle is the thing timed here, by design)
    from oracle import torch_ops
    hw = os.cpu_count() or 1
    phys = hw // 2 if hw >= 16 else hw
    cands = [hw] if phys == hw else [phys, hw]
    xyz, label, inner = synth.s3dis_batch(5000, sample_blocks, NUM_POINT)
    pts, label, inner = torch.from_numpy(xyz), torch.from_numpy(label), torch.from_numpy(inner)
    t_start = time.perf_counter()

    def set_threads(c):
        oracle.set_num_threads(c)
        torch.set_num_threads(c)

    with torch_ops.patched_util():
        model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(NUM_POINT), device=torch.device("cpu"))
        set_threads(cands[0])
        p1, l1, i1 = pts[:1], label[:1], inner[:1]
        pred, _ = model(p1, is_training=True)
        model.loss(pred, l1, i1).backward()                             # cold step on one block: creates the variables
        flat = hdist.FlatGradAllReduce(model.parameters())
        opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4)

        def timed(p, l, i):
            t0 = time.perf_counter()
            train_step(model, flat, opt, p, l, i)
            return time.perf_counter() - t0

        probe = {}
        for c in cands:
            set_threads(c)
            probe[c] = timed(pts, label, inner)
        best_c = min(probe, key=probe.get)
        set_threads(best_c)
        for _ in range(warm_steps):
            timed(pts, label, inner)
        times = []
        while len(times) < max_timed and (len(times) < 2 or (time.perf_counter() - t_start) + float(np.median(times)) < budget_s):
            times.append(timed(pts, label, inner))
        med = float(np.median(times))
        # one thread, one block
        set_threads(1)
        timed(p1, l1, i1)
        t1 = [timed(p1, l1, i1) for _ in range(2)]
        med1 = float(np.median(t1))
    # per-op thread scaling of the oracle's level-0 ops (4 blocks; one thread vs the chosen count)
    scaling = {}
    try:
        nb = min(4, sample_blocks)
        x4 = np.ascontiguousarray(xyz[:nb])
        rng = np.random.RandomState(0)
        feat = rng.randn(nb, NUM_POINT, 128).astype(np.float32)
        filt = rng.randn(33, 128, 2).astype(np.float32)
        gout = rng.randn(nb, NUM_POINT, 256).astype(np.float32)
        oracle.set_num_threads(best_c)
        idx, cnt, dst = oracle.build_sphere_neighbor(x4, x4, 0.1, None, 64)
        bins = oracle.spherical_kernel(x4, x4, idx, cnt, dst, 0.1, [8, 2, 2])
        ops = {"build_sphere_neighbor": lambda: oracle.build_sphere_neighbor(x4, x4, 0.1, None, 64),
               "farthest_point_sample": lambda: oracle.farthest_point_sample(2048, x4),
               "depthwise_conv3d": lambda: oracle.depthwise_conv3d(feat, filt, idx, cnt, bins),
               "depthwise_conv3d_grad": lambda: oracle.depthwise_conv3d_grad(feat, filt, gout, idx, cnt, bins)}
        for name, fn in ops.items():
            r = []
            for c in (1, best_c):
                oracle.set_num_threads(c)
                fn()
                t0 = time.perf_counter()
                fn()
                r.append(time.perf_counter() - t0)
            scaling[name] = {"s_1_thread": round(r[0], 4), "s_%d_threads" % best_c: round(r[1], 4), "speedup": round(r[0] / r[1], 1)}
    except Exception as e:                       # a reporting aid, never a reason to lose the bench line
        scaling = {"error": str(e)}
    return {"value": round(sample_blocks / med, 4), "unit": "blocks/s", "cores": best_c, "kind": "port",
            "cpu": _cpu_model(), "host_threads": hw,
            "steps": {"warmup": warm_steps, "timed": len(times), "median_s": round(med, 4),
                      "min_s": round(min(times), 4), "max_s": round(max(times), 4),
                      "probe_s_per_thread_count": {str(k): round(v, 4) for k, v in probe.items()}},
            "one_thread": {"value": round(1.0 / med1, 4), "unit": "blocks/s", "blocks_per_step": 1,
                           "timed_steps": len(t1), "median_s": round(med1, 4)},
            "threads_vs_one_thread": round((sample_blocks / med) * med1, 2),
            "per_op_thread_scaling_4_blocks": scaling,
            "note": "per-cloud parallel parts of the oracle (FPS: one thread per cloud) bound the multi-thread figure: with 16 clouds "
                    "per step at most 16 threads work during the sampling chain",
            "sample": "%d S3DIS-like 8192-pt blocks per step (full SPH3D_s3dis graph build + fwd + bwd + Adam on oracle/ "
                      "C99+OpenMP, torch-CPU GEMM/BN): median of %d steps after %d warm-up step(s) at %d threads"
                      % (sample_blocks, len(times), warm_steps, best_c)}


def isolated_call_seconds(name, ints, dev, reps=20):
    """mean device time of ONE C-ABI call of the given op and dims, alone on an idle GPU (synthetic operands)"""
    from sph3d_gcn_amd import tf_gemm, tf_conv3d, tf_nnquery, tf_buildkernel
    try:
        if "gemm" in name:
            R_, Ci_, Co_ = ints[:3]
            x = torch.randn(R_, Ci_, device=dev)
            w = torch.randn(Ci_, Co_, device=dev)
            dy = torch.randn(R_, Co_, device=dev)
            if name.endswith("_tn"):
                fn = lambda: tf_gemm._pointwise_gemm_tn_impl(x, dy)
            else:
                fn = lambda: tf_gemm._pointwise_gemm_impl(x, w, False)
        elif name in ("sph3d_depthwise_conv3d", "sph3d_depthwise_conv3d_grad_t"):
            if name == "sph3d_depthwise_conv3d":
                B, N, M, F, C, r, K = ints[:7]
            else:
                B, N, M, F, C, r = ints[:6]
                K = 64
            if N != M:
                return None
            radius = {8192: 0.1, 2048: 0.2, 768: 0.4, 384: 0.8, 128: 1.6}.get(N, 0.1)
            xyz = torch.from_numpy(synth.s3dis_batch(1000, B, NUM_POINT)[0]).to(dev)
            from sph3d_gcn_amd import tf_sample
            while xyz.shape[1] > N:
                nxt = {8192: 2048, 2048: 768, 768: 384, 384: 128}[xyz.shape[1]]
                idx = tf_sample.farthest_point_sample(nxt, xyz)
                xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
            nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, radius, None, K)
            filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, radius, [8, 2, 2])
            x = torch.randn(B, N, C, device=dev)
            w = torch.randn(F, C, r, device=dev)
            go = torch.randn(B, M, C * r, device=dev)
            if name == "sph3d_depthwise_conv3d":
                fn = lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
            else:
                fn = lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
        else:
            return None
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps / 1e3
    except Exception as e:                      # a measurement aid, never a reason to lose the bench line
        sys.stderr.write("isolated timing of %s failed: %s\n" % (name, e))
        return None


def gemm_accuracy_check(dev):
    """the pointwise product of one mid-size layer shape in both modes of the library against float64 ON THIS BOX, before the timed
    region: max |error| / (sum of the magnitudes of the element's terms) of the forward product, the input gradient and the weight
    gradient — what `config.gemm` claims (fp32-sized error of the split products), measured in the same run"""
    from sph3d_gcn_amd import tf_gemm
    l = _lib.lib()
    R, Ci, Co = 12288, 512, 256
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(R, Ci, device=dev, generator=g) * torch.exp2(torch.randint(-4, 4, (R, Ci), device=dev, generator=g).float())
    w = torch.randn(Ci, Co, device=dev, generator=g) / Ci ** 0.5
    dy = torch.randn(R, Co, device=dev, generator=g)
    xd, wd, dyd = x.double(), w.double(), dy.double()
    want = (xd @ wd, dyd @ wd.t(), xd.t() @ dyd)
    mags = (xd.abs() @ wd.abs(), dyd.abs() @ wd.abs().t(), xd.abs().t() @ dyd.abs())
    prev = l.sph3d_pointwise_gemm_mode(-1)
    out = {"shape": [R, Ci, Co], "metric": "max |err| / sum |terms| over NN, NT, TN vs float64"}
    try:
        for mode, name in ((1, "split_bf16x6"), (0, "fp32_mfma")):
            l.sph3d_pointwise_gemm_mode(mode)
            got = (tf_gemm._pointwise_gemm_impl(x, w, False), tf_gemm._pointwise_gemm_impl(dy, w, True), tf_gemm._pointwise_gemm_tn_impl(x, dy))
            out[name] = float("%.3g" % max(float(((a.double() - c).abs() / m).max()) for a, c, m in zip(got, want, mags)))
    finally:
        l.sph3d_pointwise_gemm_mode(prev)
    return out


def _rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return None


def _self_launch(args):
    """`python bench.py --gpus N` with no launcher: re-run this script under torch.distributed.run, one rank per GPU"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # every rank is a launch-issuing Python process: keep their math-library thread pools small (torch.distributed.run's own
    # default of OMP_NUM_THREADS=1 also works; hdist.pin_rank() sets the final value from the rank's CPU slice)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // max(1, args.gpus)))))
    return subprocess.call(cmd, env=env)


def event_families(events, ev_steps):
    """per-call device times of the event pass -> (per-call table, family table, top kernels)"""
    per = {}
    for name, ints, e0, e1 in events:
        d = per.setdefault((name, ints), [0.0, 0])
        d[0] += e0.elapsed_time(e1)
        d[1] += 1
    fam = {}
    for (name, ints), (ms, cnt) in per.items():
        f = "sph3d_pointwise_gemm*" if "gemm" in name else name
        fam[f] = fam.get(f, 0.0) + ms
    families = {k: round(v / ev_steps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}
    kernels = []
    for (name, ints), (ms, cnt) in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]:
        kernels.append({"op": name, "dims": list(ints[:7]), "calls_per_step": cnt / ev_steps,
                        "avg_us": round(ms / cnt * 1e3, 1), "ms_per_step": round(ms / ev_steps, 3)})
    return per, families, kernels


def eval_main(args, rank, world, dev, pinned_cpus):
    """forward-only secondary line of the headline workload: graph construction + inference forward of the S3DIS net on 16
    resident blocks per GPU, batch-norm moving statistics, every separable layer as one kernel"""
    from sph3d_gcn_amd import sph3gcn_util as s3g_util
    s3dis_net.SAMPLING_STREAMS = args.sampling_streams or 2
    batches = [make_batch(rank, dev, w) for w in range(NUM_BATCHES)]
    model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(NUM_POINT), device=dev)
    pred, _ = model(batches[0][0], is_training=True)            # creates the variables; one training step moves the statistics
    model.loss(pred, batches[0][1], batches[0][2]).backward()
    step_no = [0]
    torch.cuda.synchronize()
    ready = torch.cuda.Event()          # the batches are resident: a forward's plan waits for this only, not for the previous forward
    ready.record()

    def one_step():
        p_ = batches[step_no[0] % NUM_BATCHES][0]
        step_no[0] += 1
        with torch.no_grad():
            return model(p_, is_training=False, points_ready=ready)[0]

    for _ in range(PRIME_STEPS):
        one_step()
    torch.cuda.synchronize()
    elapsed, pred = run_timed(one_step, args.steps, args.warmup, world, torch.cuda.synchronize)
    ev_steps = min(args.steps, 10)
    _lib.timing_start()
    for _ in range(ev_steps):
        one_step()
    torch.cuda.synchronize()
    events = _lib.timing_stop()
    elapsed = reduce_max_seconds(elapsed, world, dev)
    per, families, kernels = event_families(events, ev_steps)
    # the same forward with the separable layers run kernel by kernel (depthwise -> GEMM -> affine), for the record
    fuse_mode = s3g_util.FUSE_SEPARABLE_INFERENCE
    s3g_util.FUSE_SEPARABLE_INFERENCE = False
    try:
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        t_unfused, _ = run_timed(one_step, min(args.steps, 10), 1, 1, torch.cuda.synchronize)
        t_unfused /= min(args.steps, 10)
    finally:
        s3g_util.FUSE_SEPARABLE_INFERENCE = fuse_mode
    if rank == 0:
        out = {"metric": "point-cloud blocks/sec (inference) SPH3D_s3dis 8192-pt", "value": round(world * BLOCKS_PER_GPU * args.steps / elapsed, 3),
               "unit": "blocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "secondary": True,
               "config": {"workload": "SPH3D_s3dis seg net, forward only (is_training=False, no_grad), S3DIS-like 8192-pt blocks, %d "
                                      "blocks/GPU, graph build + forward; separable layers as one kernel where that is the faster "
                                      "form (FUSE_SEPARABLE_INFERENCE = %r)" % (BLOCKS_PER_GPU, fuse_mode),
                          "global_batch": world * BLOCKS_PER_GPU, "points_per_block": NUM_POINT, "atan2": args.atan2,
                          "sampling_streams": s3dis_net.SAMPLING_STREAMS, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                          "parallelism": "dp%d (replicas, no collective)" % world, "cpus_per_rank": pinned_cpus},
               "ms_per_step_layer_by_layer": round(t_unfused * 1e3, 3),
               "families_ms_per_step": families, "kernels": kernels, "roofline": None, "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def secondary_main(args, rank, world, dev, pinned_cpus):
    """BASELINE configs 2, 3 and 5 on the same harness and contract as the headline (SECONDARY lines: `metric` names the
    config; BASELINE.json's own metric is the s3dis line).  A step = graph construction + forward + loss + backward +
    gradient all-reduce (N > 1) + Adam on one resident synthetic batch; two batches alternate."""
    from sph3d_gcn_amd.harness import modelnet_net, shapenet_net
    name = args.config
    rng = np.random.RandomState(17 + rank)
    s3dis_net.SAMPLING_STREAMS = args.sampling_streams or (6 if name == "scannet" else 1)
    if name == "modelnet":
        per_gpu, npts = 32, 10000
        cfg = modelnet_net.modelnet_config(npts)
        model = modelnet_net.SPH3DModelNet(cfg, device=dev)
        batches = [(torch.from_numpy(synth.modelnet_batch(1000 + (w * 64 + rank) * per_gpu, per_gpu, npts)).to(dev),
                    torch.from_numpy(rng.randint(0, 40, (per_gpu,))).to(dev)) for w in range(NUM_BATCHES)]
        fwd = lambda b: model.loss(model(b[0], is_training=True, points_ready=ready)[0], b[1])
        metric = "point clouds/sec (fwd+bwd) SPH3D_modelnet 10000-pt"
        workload = ("SPH3D_modelnet cls net (modelnet_config.py plan, 788 396 parameters), ModelNet-like 10000-pt clouds, "
                    "%d clouds/GPU, graph build + fwd + bwd + Adam" % per_gpu)
    elif name == "shapenet":
        per_gpu, npts = 64, 2048
        cfg = shapenet_net.shapenet_config(npts)
        model = shapenet_net.SPH3DShapeNet(3, cfg, device=dev)
        batches = [(torch.from_numpy(synth.modelnet_batch(5000 + (w * 64 + rank) * per_gpu, per_gpu, npts)).to(dev),
                    torch.from_numpy(rng.randint(0, 3, (per_gpu, npts))).to(dev)) for w in range(NUM_BATCHES)]
        fwd = lambda b: model.loss(model(b[0], is_training=True, points_ready=ready)[0], b[1])
        metric = "point clouds/sec (fwd+bwd) SPH3D_shapenet 2048-pt"
        workload = ("SPH3D_shapenet part-seg net (shapenet_config.py plan, category Table: 3 parts), 2048-pt objects, "
                    "%d objects/GPU, graph build + fwd + bwd + Adam" % per_gpu)
    else:
        per_gpu, npts = 1, 65536
        cfg = s3dis_net.scannet_config(npts)
        model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
        batches = []
        for w in range(NUM_BATCHES):
            xyz, label, inner = synth.s3dis_batch(7000 + w * 64 + rank, per_gpu, npts, extent=(6.0, 6.0, 3.0))
            pts = np.concatenate([xyz, rng.rand(per_gpu, npts, 6).astype(np.float32)], axis=2)
            batches.append((torch.from_numpy(pts).to(dev), torch.from_numpy(rng.randint(0, cfg.num_cls, (per_gpu, npts))).to(dev),
                            torch.from_numpy(inner).to(dev)))
        fwd = lambda b: model.loss(model(b[0], is_training=True, points_ready=ready)[0], b[1], b[2])
        metric = "point-cloud blocks/sec (fwd+bwd) SPH3D seg net 65536-pt"
        workload = ("SPH3D_s3dis plan with ScanNet's 21 classes on 65536-pt blocks (sample counts x8: 16384/6144/3072/1024), "
                    "K = 64, reference radius semantics, %d block/GPU, graph build + fwd + bwd + Adam" % per_gpu)
    torch.cuda.synchronize()
    # the batches are resident: the plans' side streams wait for this event only, not for the previous step (as in the headline)
    ready = torch.cuda.Event()
    ready.record()
    fwd(batches[0]).backward()                                   # creates the variables
    flat = hdist.FlatGradAllReduce(model.parameters())
    flat.broadcast_params(0)
    opt = hoptim.FlatAdam(flat.flat_param, lr=1e-3, eps=1e-4)
    step_no = [0]

    def one_step():
        b = batches[step_no[0] % NUM_BATCHES]
        step_no[0] += 1
        loss = fwd(b)
        flat.backward(loss)
        flat.all_reduce()
        opt.step()
        return loss

    for _ in range(PRIME_STEPS):
        one_step()
    torch.cuda.synchronize()
    elapsed, loss = run_timed(one_step, args.steps, args.warmup, world, torch.cuda.synchronize)
    ev_steps = min(args.steps, 10)
    _lib.timing_start()
    for _ in range(ev_steps):
        one_step()
    torch.cuda.synchronize()
    events = _lib.timing_stop()
    per_rank_s = hdist.gather_floats(elapsed, world, dev)
    elapsed = reduce_max_seconds(elapsed, world, dev)
    per, families, kernels = event_families(events, ev_steps)
    # roofline object of the dominant family's largest call (same rules as the headline line)
    roofline = None
    if families:
        # (the sampling chain runs on a side stream and is latency-bound: it is in `families`, not the roofline kernel)
        fname = next(k for k in families if k != "sph3d_farthest_point_sample")
        best = None
        for (cname, ints), (ms, cnt) in per.items():
            if ("sph3d_pointwise_gemm*" if "gemm" in cname else cname) != fname:
                continue
            work = 2.0 * ints[0] * ints[1] * ints[2] if "gemm" in cname else float(algorithmic_bytes(cname, ints))
            if best is None or work > best[0]:
                best = (work, cname, ints, ms / cnt / 1e3)
        if best is not None and best[0] > 0:
            work, cname, ints, avg_s = best
            gemm = "gemm" in cname
            peak = ((round(BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS, 1) if _lib.lib().sph3d_pointwise_gemm_mode(-1) else FP32_MFMA_PEAK_TFLOPS)
                    if gemm else HBM_PEAK_GBS)
            ach = work / (1e12 if gemm else 1e9) / avg_s
            roofline = {"kernel": cname, "family": fname, "dims": list(ints[:7]), "bound": "mfma" if gemm else "hbm",
                        "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s" if gemm else "GB/s", "frac": round(ach / peak, 4),
                        "avg_us": round(avg_s * 1e6, 1), "traffic": None,
                        "family_ms_per_step": families[fname],
                        "note": "avg_us: HIP events around the C-ABI call inside the running step; no PMC pass for this config"}
    if rank == 0:
        units = world * per_gpu * args.steps
        out = {"metric": metric, "value": round(units / elapsed, 3), "unit": "clouds/s" if name != "scannet" else "blocks/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "secondary": True,
               "config": {"workload": workload, "global_batch": world * per_gpu, "points_per_cloud": npts,
                          "parallelism": "dp%d (one cloud shard per GPU; flat gradient all-reduced over RCCL in %d buckets)"
                                         % (world, len(flat.buckets)),
                          "resident_batches": NUM_BATCHES, "params": flat.num_parameters, "launch_mode": "eager",
                          "sampling_streams": s3dis_net.SAMPLING_STREAMS, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                          "atan2": args.atan2,
                          "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                          "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                          "cpus_per_rank": pinned_cpus},
               "dist": {"per_rank_ms_per_step": [round(x / args.steps * 1e3, 3) for x in per_rank_s]},
               "loss": round(float(loss.detach()), 5), "families_ms_per_step": families, "roofline": roofline, "kernels": kernels,
               "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true", help="skip the one-block host-issue and sampling-chain probes of `dist`")
    ap.add_argument("--atan2", choices=("ocml", "shared"), default="ocml",
                    help="angle function of the spherical-kernel binning: 'ocml' = ROCm's device-library atan2f, the function the "
                         "reference's own kernel calls when built for this GPU (bins bit-identical to the reference build); "
                         "'shared' = the correctly rounded atan2f shared with the CPU oracle")
    ap.add_argument("--config", choices=("s3dis", "modelnet", "shapenet", "scannet"), default="s3dis",
                    help="workload: 's3dis' = the headline (BASELINE.json's metric); the others are SECONDARY lines for BASELINE "
                         "configs 2, 3 and 5 with the same JSON contract (per-GPU batch 32 / 64 / 1, weak scaling)")
    ap.add_argument("--sampling-streams", type=int, default=0,
                    help="HIP streams the plans' sampling chains rotate over (0 = the line's default: 1 for the training lines "
                         "whose step outweighs its sampling chain, 2 for --eval, 6 for scannet (with GPU_MAX_HW_QUEUES=8: six sampling streams + graph + main; round 6: 4 -> 6 streams "
                         "59 -> 63 blocks/s, profiles/r06_ab_scannet_streams.log), where the chain is what a step waits for)")
    ap.add_argument("--eval", action="store_true",
                    help="SECONDARY line: forward only (is_training=False under no_grad: every separable layer is ONE kernel, "
                         "csrc/sepconv.hip) on the headline's batch; metric 'point-cloud blocks/sec (inference)'")
    args = ap.parse_args()

    if args.config == "scannet" or args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # The runtime multiplexes a process's streams onto 4 hardware queues by default, and two streams on one queue run one
        # behind the other (tools/exp_scannet_timeline.py: a third sampling stream's chain waited for another's).  scannet: six
        # sampling streams + the graph and the main stream.  Multi-rank runs: RCCL's streams come on top of the step's three,
        # and a gradient bucket's all-reduce queued behind a 1.7-ms sampling kernel would be waited for before Adam.  (One
        # GPU, three streams: 1814 blocks/s with 8 queues against 1820-1840 — left at the default there.)  The variable is read
        # when the HIP runtime starts, so it is set before the first device call (the self-launched ranks inherit it).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args))
    rank, world, local_rank = hdist.init_from_env()
    is_rank0 = rank == 0          # (kept apart: the JSON line must not depend on a name that later code could reuse)
    assert world == args.gpus, "launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback in the product path)"
    assert torch.cuda.device_count() >= (local_rank + 1), "rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pinned_cpus = hdist.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    _lib.lib()
    from sph3d_gcn_amd import tf_buildkernel
    tf_buildkernel.set_atan2(args.atan2)          # reaches the fused graph kernel too (tf_nnquery.build_sphere_graph)
    if args.config != "s3dis":
        return secondary_main(args, rank, world, dev, pinned_cpus)
    if args.eval:
        return eval_main(args, rank, world, dev, pinned_cpus)

    gemm_check = gemm_accuracy_check(dev) if is_rank0 else None
    batches = [make_batch(rank, dev, w) for w in range(NUM_BATCHES)]
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    ev.record()
    for bt in batches:
        _PTS_READY[bt[0].data_ptr()] = ev
    pts, label, inner = batches[0]
    step_no = [0]
    s3dis_net.SAMPLING_STREAMS = args.sampling_streams or 1
    model = s3dis_net.SPH3DS3DIS(s3dis_net.s3dis_config(NUM_POINT), device=dev)
    # variables are created by the first forward (TF-style scopes): one untimed pass, then flat buffers + Adam
    graphs = s3dis_net.build_graphs(pts, model.config)
    pred, _ = model(pts, is_training=True, graphs=graphs)
    model.loss(pred, label, inner).backward()
    flat = hdist.FlatGradAllReduce(model.parameters())
    flat.broadcast_params(0)
    # train_s3dis.py:224 (epsilon=1e-4); one fused kernel over the flat parameter buffer instead of the foreach chain
    opt = hoptim.FlatAdam(flat.flat_param, lr=1e-3, eps=1e-4)        # one streaming kernel, torch.optim.Adam's arithmetic
    nparams = flat.num_parameters

    # (launch mode: eager on three HIP streams.  A HIP-graph replay of the whole step does capture once every autograd node lives
    #  on a non-default stream (tools/exp_capture.py, round 3) and buys nothing: the feature path replays in 7.46 ms against 7.46
    #  eager — the GPU, not the host, is the bound — and the three-branch graph replays serialised, 12.1 against 10.7 ms)
    mode = "eager"

    def one_step():
        p_, l_, i_ = batches[step_no[0] % NUM_BATCHES]        # a different resident batch every step
        step_no[0] += 1
        return train_step(model, flat, opt, p_, l_, i_)

    # Priming (setup, not measurement): the first ~14 steps of a fresh process contain one-off host stalls — the caching
    # allocator still growing its pools (hipMalloc is synchronous) and one 80-90 ms pause at the 14th step (first
    # generation-2 Python GC of the autograd / graph-cache objects).  With a short --warmup that pause would land inside
    # the K timed steps (measured: 17.1 vs 14.2 ms/step for the same code).  PRIME_STEPS untimed steps put it behind us;
    # the W warm-up steps and the K timed steps that follow are exactly as requested.
    for _ in range(PRIME_STEPS):
        one_step()
    torch.cuda.synchronize()
    elapsed, loss = run_timed(one_step, args.steps, args.warmup, world, torch.cuda.synchronize)

    # Per-kernel device times: the same K steps are run once more, right after the timed region (same process, same
    # data, same kernels), with every C-ABI launch bracketed by two HIP events on its launching stream.  They are
    # not recorded inside the timed region because the eager step is host-bound and ~460 extra event records per
    # step would lengthen the very interval being measured.
    ev_steps = min(args.steps, 20)
    _lib.timing_start()
    flat.time_wait_events = True            # + an event pair around the wait for the gradient all-reduce (main stream)
    for _ in range(ev_steps):
        one_step()
    torch.cuda.synchronize()
    events = _lib.timing_stop()
    flat.time_wait_events = False
    ar_wait_ms = [a.elapsed_time(b) for a, b in flat.wait_events]

    # what this rank's Python needs to ISSUE a step, whatever the device does: the same step on ONE block (the same ~450 launches,
    # a sixteenth of the device work) and the per-rank time of the sampling chain alone —
    # a multi-GPU run whose ranks sit well above the 1-GPU ms/step is launch-bound where host_issue approaches that figure, and
    # device-bound where it does not
    if args.no_probes:            # (the rocprofv3 run of tools/gpu_profile_round.sh: its kernel statistics must hold whole 16-block steps only)
        host_issue_ms, fps_chain_ms = 0.0, 0.0
    else:
        host_issue_ms, fps_chain_ms = probe_host_issue(model, flat, opt, batches[0], dev), probe_fps_chain(batches[0][0], model.config)
    per_rank_issue = hdist.gather_floats(host_issue_ms, world, dev)
    per_rank_fps = hdist.gather_floats(fps_chain_ms, world, dev)

    # every rank's own time for the K timed steps (diagnosis of a slow rank), then the contract's MAX over ranks
    per_rank_s = hdist.gather_floats(elapsed, world, dev)
    elapsed = reduce_max_seconds(elapsed, world, dev)

    # ---- per-kernel device time from the HIP events recorded during the timed steps ----
    per = {}
    for name, ints, e0, e1 in events:
        ms = e0.elapsed_time(e1)
        key = (name, ints)
        d = per.setdefault(key, [0.0, 0])
        d[0] += ms
        d[1] += 1
    ranked = sorted(per.items(), key=lambda kv: -kv[1][0])
    kernels = []
    for (name, ints), (ms, cnt) in ranked[:8]:
        ab = algorithmic_bytes(name, ints)
        avg_ms = ms / cnt
        kernels.append({"op": name, "dims": list(ints[:7]), "calls_per_step": cnt / ev_steps,
                        "avg_us": round(avg_ms * 1e3, 1), "ms_per_step": round(ms / ev_steps, 3),
                        "alg_GB": round(ab / 1e9, 4), "GBps": round(ab / 1e9 / (avg_ms / 1e3), 1) if avg_ms > 0 else None})
    # dominant kernel = the op family with the largest summed device time on the main stream (the FPS chain runs
    # on a side stream, overlapped, and is latency-bound: it is reported in `kernels`, not as the roofline kernel);
    # the roofline numbers are those of that family's largest call
    def family(name):
        return "sph3d_pointwise_gemm*" if "gemm" in name else name

    fam = {}
    for (name, ints), (ms, cnt) in per.items():
        if name == "sph3d_farthest_point_sample":
            continue
        f = fam.setdefault(family(name), [0.0, None, (0.0, 0, 0, 0.0)])
        f[0] += ms
        # the family's representative call: the one with the most work (FLOPs for the GEMMs, algorithmic bytes otherwise);
        # among equals the weight-gradient product (split-K + slab sum), then the slowest — NOT simply the slowest, which
        # flips between runs when several level-0 products take within a few percent of each other
        work = 2.0 * ints[0] * ints[1] * ints[2] if "gemm" in name else float(algorithmic_bytes(name, ints))
        call_rank = (work, 1 if name.endswith("_tn") else 0, ints[0], ms / cnt)      # ... then the most rows (level 0), then the slowest
        if call_rank > f[2]:
            f[1], f[2] = (name, ints, ms, cnt), call_rank
    roofline = None
    families = {k: round(v[0] / ev_steps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
    if fam:
        fname = max(fam, key=lambda k: fam[k][0])
        name, ints, ms, cnt = fam[fname][1]
        ab = algorithmic_bytes(name, ints)
        avg_s = ms / cnt / 1e3
        is_gemm = "gemm" in name
        iso_s = isolated_call_seconds(name, ints, dev)
        gemm_split = bool(_lib.lib().sph3d_pointwise_gemm_mode(-1))
        if is_gemm:
            R_, Ci_, Co_ = ints[:3]
            # split mode (the default): the whole-tile products run on the BF16 matrix pipe, six piece products per fp32 product
            # (include/sph3d.h: sph3d_pointwise_gemm_mode): the fp32-equivalent ceiling is the dense bf16 peak / 6
            mfma_peak = BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS if gemm_split else FP32_MFMA_PEAK_TFLOPS
            flops = 2.0 * R_ * Ci_ * Co_
            # the roof that binds this call: its FLOPs at the matrix peak or its algorithmic bytes at the HBM peak, whichever takes longer
            # (on the bf16 pipe the level-0 products — K = 128 / 256 — are bound by their bytes)
            if flops / 1e12 / mfma_peak >= ab / 1e9 / HBM_PEAK_GBS:
                work, peak, unit, bound = flops / 1e12, round(mfma_peak, 1), "TFLOP/s", "mfma"
            else:
                work, peak, unit, bound = ab / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
        else:
            work, peak, unit, bound = ab / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
        achieved = work / avg_s
        traffic, trace_us, mfma_busy = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                table = json.load(open(tpath))
                key = "%s%s" % (name, list(ints[:7]))
                traffic = table.get(key, table.get("%s%s" % (name, list(ints[:6]))))
                trace_us = table.get("trace_us", {}).get(key)
                mfma_busy = table.get("mfma_pipe_busy", {}).get(key)      # committed PMC pass of the same call (profiles/)
            except Exception:
                traffic = None
        roofline = {"kernel": name, "family": fname, "dims": list(ints[:7]), "bound": bound, "achieved": round(achieved, 1),
                    "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
                    "alg_bytes": ab, "avg_us": round(avg_s * 1e6, 1),
                    "isolated_us": round(iso_s * 1e6, 1) if iso_s else None,
                    "isolated_frac": round(work / iso_s / peak, 4) if iso_s else None,
                    "trace_us": trace_us, "traffic": traffic, "mfma_pipe_busy_pmc": mfma_busy,
                    "family_ms_per_step": round(fam[fname][0] / ev_steps, 3),
                    "note": "avg_us: HIP events around the C-ABI call inside the running step (three streams share the "
                            "CUs); isolated_us: the same call alone on an idle GPU; trace_us: rocprofv3 kernel trace"}
        if is_gemm:
            roofline["mfma"] = ("v_mfma_f32_32x32x16_bf16, %d piece products per fp32 product (operands cut exactly into three bf16 pieces, "
                                "fp32 accumulate): matrix roof = %.0f TF dense bf16 / %d per ALGORITHMIC fp32 FLOP; executed on the pipe: "
                                "%.0f TFLOP/s" % (SPLIT_PRODUCTS, BF16_MFMA_PEAK_TFLOPS, SPLIT_PRODUCTS,
                                                  2.0 * ints[0] * ints[1] * ints[2] / 1e12 / avg_s * SPLIT_PRODUCTS)
                                if gemm_split else "v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain): peak = the fp32 MFMA peak")
            roofline["hbm_bound_us"] = round(ab / 1e9 / HBM_PEAK_GBS * 1e6, 1)      # what the call's algorithmic bytes cost at the HBM peak
            roofline["mfma_bound_us"] = round(2.0 * ints[0] * ints[1] * ints[2] / 1e12 / (BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS if gemm_split else FP32_MFMA_PEAK_TFLOPS) * 1e6, 1)
            roofline["alg_TFLOPs"] = round(2.0 * ints[0] * ints[1] * ints[2] / 1e12 / avg_s, 1)
    # the north-star "conv gather" line: depthwise forward at (B=16, N=M=8192, C=128, r=2, K=64)
    conv_gather = None
    for (name, ints), (ms, cnt) in per.items():
        if name == "sph3d_depthwise_conv3d" and ints[:7] == (BLOCKS_PER_GPU, NUM_POINT, NUM_POINT, 33, 128, 2, 64):
            ab = algorithmic_bytes(name, ints)
            avg_s = ms / cnt / 1e3
            conv_gather = {"avg_us": round(avg_s * 1e6, 1), "achieved": round(ab / 1e9 / avg_s, 1), "unit": "GB/s",
                           "frac": round(ab / 1e9 / avg_s / HBM_PEAK_GBS, 4), "alg_bytes": ab}
            iso = isolated_call_seconds(name, ints, dev)
            if iso:
                conv_gather["isolated_us"] = round(iso * 1e6, 1)
                conv_gather["isolated_frac"] = round(ab / 1e9 / iso / HBM_PEAK_GBS, 4)
    sph3d_ms = sum(v[0] for v in per.values()) / ev_steps

    if is_rank0:
        blocks = world * BLOCKS_PER_GPU * args.steps
        out = {
            "metric": "point-cloud blocks/sec (fwd+bwd) SPH3D_s3dis 8192-pt",
            "value": round(blocks / elapsed, 3),
            "unit": "blocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "SPH3D_s3dis seg net (s3dis_config.py plan), S3DIS-like 8192-pt blocks, "
                                   "%d blocks/GPU, graph build + fwd + bwd + Adam" % BLOCKS_PER_GPU,
                       "global_batch": world * BLOCKS_PER_GPU, "points_per_block": NUM_POINT,
                       "parallelism": "dp%d (one cloud shard per GPU; flat gradient all-reduced over RCCL in %d buckets, "
                                      "overlapped with backward)" % (world, len(flat.buckets)),
                       "resident_batches": NUM_BATCHES, "event_pass_steps": ev_steps,
                       "params": nparams, "launch_mode": mode, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                       "atan2": args.atan2,
                       "gemm": ("split: fp32 operands as three exact bf16 pieces, six bf16-MFMA piece products accumulated in fp32 "
                                "(error <= 2^-23 per product, the size of one fp32 rounding; tests/test_gpu_gemm_split.py); "
                                "SPH3D_GEMM_SPLIT=0 = v_mfma_f32_32x32x2_f32" if _lib.lib().sph3d_pointwise_gemm_mode(-1)
                                else "v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain)"),
                       "gemm_check": gemm_check,
                       "bin_ids": ("bit-identical to the reference build (same ocml atan2f; tests/test_gpu_round3.py)" if args.atan2 == "ocml"
                                   else "shared correctly-rounded atan2f: == CPU oracle, differs from the reference build within an "
                                        "ulp of a bin boundary (0.07 % of level-0 slots)"),
                       "sampling_streams": s3dis_net.SAMPLING_STREAMS,
                       "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "rccl_version": _rccl_version(), "cpus_per_rank": pinned_cpus},
            # multi-GPU diagnosis: each rank's ms/step over the same K steps, the gradient buckets, where
            # their all-reduces were started, and how long the main stream waited for them before the optimiser step
            "dist": {"per_rank_ms_per_step": [round(x / args.steps * 1e3, 3) for x in per_rank_s],
                     "rank_ms_per_step_min_max": [round(min(per_rank_s) / args.steps * 1e3, 3),
                                                  round(max(per_rank_s) / args.steps * 1e3, 3)],
                     "host_issue_ms_per_step_one_block": [round(x, 3) for x in per_rank_issue],
                     "fps_chain_ms_per_step": [round(x, 3) for x in per_rank_fps],
                     "cpus_per_rank": pinned_cpus,
                     "bucket_bytes": [4 * (f1 - f0) for (_i0, _i1, f0, f1) in flat.buckets],
                     "buckets_started_in_backward": flat.stats["buckets_started_in_backward"],
                     "buckets_started_after_backward": flat.stats["buckets_started_after_backward"],
                     "allreduce_wait_stream_ms_per_step": (round(sum(ar_wait_ms) / max(1, len(ar_wait_ms)), 4) if ar_wait_ms else 0.0),
                     "allreduce_wait_host_ms_per_step": round(flat.stats["allreduce_wait_host_s"] / max(1, flat.stats["allreduce_calls"]) * 1e3, 4),
                     "note": "rank 0's counters; allreduce_wait_stream = HIP events around the wait in FlatGradAllReduce.all_reduce() "
                             "on the main stream during the %d event-pass steps (0 at one GPU: no collective is issued)" % ev_steps},
            "loss": round(float(loss.detach()), 5),
            "sph3d_calls_ms_per_step_summed_over_streams": round(sph3d_ms, 3),
            "families_ms_per_step": families,
            "roofline": roofline,
            "conv_gather": conv_gather,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
