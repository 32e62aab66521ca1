"""world_size-2 (or -8: GLOO_BENCH_WORLD) gloo worker: bench.py's rank code path (train_step, run_timed, reduce_max_seconds, the flat bucketed
all-reduce, Adam on the flat buffer) end to end on CPU.  The HIP ops are replaced by the oracle ops (test infrastructure) so
that the step runs without a GPU; shapes are a reduced S3DIS plan."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import torch_ops  # noqa: E402
from sph3d_gcn_amd.harness import dist as hdist  # noqa: E402
from sph3d_gcn_amd.harness import s3dis_net, synth  # noqa: E402


def main():
    rank, world, local_rank = hdist.init_from_env(backend="gloo")
    want = int(os.environ.get("GLOO_BENCH_WORLD", "2"))
    assert world == want and dist.get_world_size() == want
    ncpu = hdist.pin_rank(local_rank, world, want_numa=False)             # affinity slice per rank (CPU-count permitting)
    assert ncpu >= 0
    if ncpu > 0:                                                          # the slices of the ranks are disjoint and cover no CPU twice
        mine = sorted(os.sched_getaffinity(0))
        all_lo = hdist.gather_floats(float(mine[0]), world, torch.device("cpu"))
        all_hi = hdist.gather_floats(float(mine[-1]), world, torch.device("cpu"))
        spans = sorted(zip(all_lo, all_hi))
        assert all(a[1] < b[0] for a, b in zip(spans, spans[1:])), spans
    dev = torch.device("cpu")
    npts = 512 if world == 2 else 256
    cfg = s3dis_net.small_config(npts)
    cfg.num_sample = [128, 32] if world == 2 else [64, 16]
    blocks = 2 if world == 2 else 1                                       # per rank: weak scaling, like BLOCKS_PER_GPU
    total = blocks * world
    b0, b1 = hdist.shard_range(total, rank, world)                        # this rank's clouds of the global batch
    assert b1 - b0 == blocks
    first = 100 + b0
    xyz, label, inner = synth.s3dis_batch(first, blocks, npts, extent=(0.8, 0.8, 1.0))
    pts, label, inner = torch.from_numpy(xyz), torch.from_numpy(label), torch.from_numpy(inner)
    with torch_ops.patched_util():
        model = s3dis_net.SPH3DS3DIS(cfg, device=dev, seed=7)
        pred, _ = model(pts, is_training=True)
        model.loss(pred, label, inner).backward()
        flat = hdist.FlatGradAllReduce(model.parameters(), bucket_bytes=64 << 10)
        flat.broadcast_params(0)
        opt = torch.optim.Adam([flat.flat_param], lr=1e-3, eps=1e-4)
        elapsed, loss = bench.run_timed(lambda: bench.train_step(model, flat, opt, pts, label, inner), steps=3, warmup=1,
                                        world=world, sync=lambda: None)
    per_rank = hdist.gather_floats(elapsed, world, dev)
    worst = bench.reduce_max_seconds(elapsed, world, dev)
    assert worst >= elapsed > 0 and torch.isfinite(loss)
    assert len(per_rank) == world and abs(max(per_rank) - worst) < 1e-9 and per_rank[rank] == elapsed
    # the buckets' all-reduces were started from inside the backward pass, one per bucket and step
    st = flat.stats
    assert st["buckets_started_in_backward"] + st["buckets_started_after_backward"] == len(flat.buckets) * 4
    assert st["buckets_started_in_backward"] >= (len(flat.buckets) - 1) * 4 and st["allreduce_calls"] == 4
    # the replicas stayed identical: same parameters after four optimiser steps on different shards
    mine = flat.flat_param.data.clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(other, mine)
    # and the all-reduced gradient of the last step is the same on both ranks
    g = flat.flat.clone()
    dist.broadcast(g, src=0)
    assert torch.equal(g, flat.flat)
    dist.barrier()
    if rank == 0:
        print("GLOO_BENCH_OK %.3f" % worst)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
