"""N > 1 on real GPUs: `python bench.py --gpus 2` with no launcher must spawn one rank per GPU over RCCL and print one JSON
line.  Skipped on single-GPU boxes (the driver's scaling run and the world-2 gloo test on CPU cover the rest)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_self_launch_two_ranks():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["value"] > 0


def test_bench_under_torchrun_runs_the_rccl_path_on_one_gpu():
    """One rank under torch.distributed.run with SPH3D_FORCE_COLLECTIVES=1: the process group is RCCL, every gradient bucket's
    all-reduce is issued during the backward pass on RCCL's stream and waited for before Adam (at world size 1 the sum is the
    identity, so the loss must equal the plain run's) — what a one-GPU box can check of the multi-rank path"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SPH3D_FORCE_COLLECTIVES"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-probes"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["collective_backend"] == "nccl" and out["config"]["world_size"] == 1
    assert out["dist"]["buckets_started_in_backward"] > 0 and out["dist"]["buckets_started_after_backward"] == 0
    assert out["value"] > 0 and out["loss"] == out["loss"]
    env.pop("SPH3D_FORCE_COLLECTIVES")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-probes"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert q.returncode == 0, q.stdout[-2000:] + q.stderr[-2000:]
    plain = json.loads([l for l in q.stdout.splitlines() if l.startswith("{")][-1])
    assert abs(plain["loss"] - out["loss"]) <= 1e-3 * abs(plain["loss"])


def test_bench_single_gpu_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and r["isolated_us"] and "family" in r
    assert out["steps"] == 3 and out["n_gpus"] == 1 and out["config"]["resident_batches"] == 2


def test_bucket_collective_is_ordered_after_the_concatenation_on_a_side_stream(dev):
    """the backward pass of the bench step does not run on the default stream.  ProcessGroupNCCL (= RCCL)
    orders its communication stream after an event recorded on the CURRENT stream at the all_reduce call; FlatGradAllReduce
    relies on that by issuing the bucket's concatenation on the current stream and the collective right behind it.  A
    stand-in collective with exactly that contract (event on the current stream -> wait on a private stream -> work there ->
    .wait() orders the current stream after it) must therefore see complete buckets when forward and backward run on a
    non-default stream with long-running kernels queued in front."""
    import torch
    from sph3d_gcn_amd.harness import dist as hdist
    comm = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)

    class _Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def collective(view):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        comm.wait_event(ready)
        with torch.cuda.stream(comm):
            view.mul_(2.0)                           # "sum over two identical replicas"
            done = torch.cuda.Event()
            done.record(comm)
        return _Work(done)

    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (1 << 20, 3, 1 << 21, 17, 1 << 20)]
    flat = hdist.FlatGradAllReduce(ps, bucket_bytes=4 << 20, collective=collective)
    assert len(flat.buckets) >= 2
    x = torch.randn(1 << 21, device=dev)
    big = torch.randn(4096, 4096, device=dev)
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(6):
                big = (big @ big).clamp_(-1, 1)      # keeps the side stream busy in front of the step
            loss = sum((p * x[:p.numel()]).sum() * (i + 1) for i, p in enumerate(ps))
            flat.backward(loss)
            flat.all_reduce()
            got = flat.flat.clone()
        side.synchronize()
        want = 2.0 * torch.cat([x[:p.numel()] * (i + 1) for i, p in enumerate(ps)])
        torch.testing.assert_close(flat.unpadded(got), want)
    assert flat.stats["buckets_started_in_backward"] == 3 * len(flat.buckets)
