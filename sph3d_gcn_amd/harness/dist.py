"""Data-parallel plumbing: one process per GPU, batch sharded by cloud, the flat gradient all-reduced in a few
multi-MiB buckets overlapped with the backward pass (the reference is single-GPU).  Every op of the path is independent per cloud, so the
forward/backward data path has no collective; only the parameter gradient is summed (RCCL over xGMI:
backend "nccl" on ROCm; "gloo" in the CPU tests)."""
import os

import torch
import torch.distributed as dist


FORCE_COLLECTIVES = os.environ.get("SPH3D_FORCE_COLLECTIVES", "0") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SPH3D_FORCE_COLLECTIVES=1: create the group and issue the gradient buckets' all-reduces at world size 1 too — the only way
    # to run the RCCL path (communicator, its stream, the event ordering of FlatGradAllReduce) on a one-GPU box
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def _gpu_numa_cpus(local_rank):
    """host CPUs of the NUMA node the rank's GPU hangs off (sysfs), or None"""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        return _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
    except Exception:
        return None


def pin_rank(local_rank, local_world, want_numa=True):
    """Give every rank of the node its own host cores: eight Python ranks each issue ~400 launches per step, and unpinned
    they migrate across both sockets and share cores with one another's OpenMP / autograd threads.  Rank r takes the r-th
    slice of the CPUs of its GPU's NUMA node (sysfs) when that is known, else the r-th contiguous slice of the process's
    affinity mask; OMP / MKL thread counts follow the slice.  -> number of CPUs the rank may run on (0 = left alone)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return 0
    try:
        allowed = sorted(os.sched_getaffinity(0))
        mine = None
        if want_numa and torch.cuda.is_available():
            node = _gpu_numa_cpus(local_rank)
            if node:
                node = [c for c in node if c in set(allowed)]
                # ranks sharing a node: assume GPUs are spread evenly over the nodes in index order
                nodes = max(1, len(allowed) // max(1, len(node)))
                per_node = max(1, (local_world + nodes - 1) // nodes)
                k = local_rank % per_node
                w = max(1, len(node) // per_node)
                mine = node[k * w:(k + 1) * w]
        if not mine:
            w = max(1, len(allowed) // local_world)
            mine = allowed[local_rank * w:(local_rank + 1) * w]
        if not mine:
            return 0
        os.sched_setaffinity(0, mine)
        n = max(1, min(len(mine), 8))
        os.environ["OMP_NUM_THREADS"] = str(n)
        os.environ["MKL_NUM_THREADS"] = str(n)
        torch.set_num_threads(n)
        return len(mine)
    except Exception:
        return 0


def gather_floats(value, world, dev):
    """every rank's `value` (a float) on every rank, in rank order"""
    if world <= 1 or not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def shard_range(total, rank, world):
    """Contiguous [begin, end) of `total` clouds owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class FlatGradAllReduce:
    """All parameters and all gradients live in two flat fp32 buffers (the nn.Parameters are views), the optimiser
    update is one elementwise kernel over ``flat_param`` (pass ``[self.flat_param]`` to the optimiser), and the
    gradient sum over the replicas is bucketed and overlapped with the backward pass:

      * the parameters are cut into buckets of >= ``bucket_bytes`` in creation order (= forward order);
      * ``backward(loss)`` takes the gradients with torch.autograd.grad; tensor hooks on the parameters collect them, and
        as soon as the LAST gradient of a bucket exists (the backward pass reaches the buckets in reverse order) the
        bucket is written into the flat buffer with one concatenation and its all-reduce is started (RCCL over xGMI: backend "nccl"
        on ROCm; "gloo" in the CPU tests) asynchronously on the process group's stream;
      * ``all_reduce()`` waits for the outstanding buckets before the optimiser step.
    xGMI rings are per-link bound, so a bucket is several MiB (default 4 MiB: 4 buckets for the 15-MiB S3DIS net); with
    one replica no collective is issued and the hooks only do the concatenations."""

    def __init__(self, params, bucket_bytes=4 << 20, collective=None):
        """collective: optional replacement of dist.all_reduce for tests — called as collective(view) right after the
        bucket's concatenation, on the stream the concatenation was issued on; returns an object with .wait() (or None)"""
        self._collective = collective
        self.stats = {"allreduce_wait_host_s": 0.0, "allreduce_calls": 0, "buckets_started_in_backward": 0,
                      "buckets_started_after_backward": 0}
        self.time_wait_events = False          # record a HIP-event pair around the wait in all_reduce() (bench.py's event pass)
        self.wait_events = []
        self.params = [p for p in params if p.requires_grad]
        # every parameter starts on a 16-byte boundary of the flat buffers (the HIP kernels take 16-byte vector loads of their
        # operands: the ModelNet plan's odd channel counts would otherwise leave later weights 8-byte aligned); the padding
        # elements stay zero in both buffers
        n = sum((p.numel() + 3) // 4 * 4 for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_param = torch.nn.Parameter(torch.empty(n, dtype=torch.float32, device=dev))
        self.flat_param.grad = self.flat
        off = 0
        self.buckets = []                      # (first param index, one past last, flat begin, flat end)
        b_first, b_off = 0, 0
        with torch.no_grad():
            self.flat_param.data.zero_()
            self._slots = []                   # (flat begin, numel) of every parameter
            for i, p in enumerate(self.params):
                k = p.numel()
                self.flat_param.data[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param.data[off:off + k].view_as(p)
                p.grad = None
                self._slots.append((off, k))
                off += (k + 3) // 4 * 4
                if (off - b_off) * 4 >= bucket_bytes or i == len(self.params) - 1:
                    self.buckets.append((b_first, i + 1, b_off, off))
                    b_first, b_off = i + 1, off
        self._zero_pad = torch.zeros(4, dtype=torch.float32, device=dev)
        self._pending = []
        self._done = set()
        self._got = [dict() for _ in self.buckets]        # per bucket: parameter index -> gradient of this backward pass
        self._armed = False
        for bi, (i0, i1, f0, f1) in enumerate(self.buckets):
            for i in range(i0, i1):
                self.params[i].register_hook(self._make_hook(bi, i))

    def _world(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    @property
    def num_parameters(self):
        """parameter elements (the flat buffers also hold the alignment padding)"""
        return sum(k for _off, k in self._slots)

    def unpadded(self, flat):
        """the parameters' elements of a flat buffer, in order, without the alignment padding"""
        return torch.cat([flat[o:o + k] for o, k in self._slots])

    def _finish_bucket(self, bi, grads):
        i0, i1, f0, f1 = self.buckets[bi]
        parts = []
        for j, (g, p) in enumerate(zip(grads, self.params[i0:i1])):
            parts.append((g if g is not None else torch.zeros_like(p)).reshape(-1))
            pad = (-p.numel()) % 4
            if pad:
                parts.append(self._zero_pad[:pad])
        view = self.flat[f0:f1]
        # ORDER: the concatenation is issued on the CURRENT stream — inside a tensor hook that is the stream of the backward
        # node that produced the bucket's last gradient — and the collective is started right behind it from the same
        # thread: ProcessGroupNCCL (= RCCL) orders its communication stream after an event it records on the current stream
        # at this call, so the all-reduce cannot read the bucket before the concatenation has written it, whichever stream
        # the backward pass runs on (tests/test_gpu_dist.py checks exactly this with a stand-in collective).
        torch.cat(parts, out=view)
        if self._collective is not None:
            w = self._collective(view)
            if w is not None:
                self._pending.append(w)
        elif self._world() > 1 or (FORCE_COLLECTIVES and dist.is_initialized()):
            self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        self.stats["buckets_started_in_backward" if self._armed else "buckets_started_after_backward"] += 1
        self._done.add(bi)

    def _make_hook(self, bi, i):
        def hook(grad):
            if not self._armed:                 # a backward pass that is not ours (e.g. the variable-creating first step)
                return
            got = self._got[bi]
            got[i] = grad
            i0, i1, _f0, _f1 = self.buckets[bi]
            if len(got) == i1 - i0:
                self._finish_bucket(bi, [got[j] for j in range(i0, i1)])
                got.clear()
        return hook

    def broadcast_params(self, src=0):
        if self._world() > 1 or (FORCE_COLLECTIVES and dist.is_initialized()):
            dist.broadcast(self.flat_param.data, src=src)

    def zero(self):
        self.flat.zero_()

    def backward(self, loss):
        """loss.backward() for the flat layout: gradients by torch.autograd.grad, written into the flat buffer one bucket
        at a time as the backward pass produces them (one concatenation per bucket — through .grad views every parameter
        would cost its own in-place `add` launch: 69 of them, 0.35 ms of 5-us kernels per S3DIS step), each bucket's
        all-reduce started at once."""
        self._done.clear()
        for got in self._got:
            got.clear()
        self._armed = True
        try:
            # backward on the ISSUING thread: handing the pass to the engine's device thread and waking up again when it is done
            # costs 0.1-1.3 ms per step of host time depending on the box (tools/exp_host_floor.py: 7.85 -> 6.56 ms to issue a
            # one-block step), and the ~400 launches of a step are what a rank's host has to keep ahead of its GPU
            with torch.autograd.set_multithreading_enabled(os.environ.get("SPH3D_AUTOGRAD_THREAD", "0") == "1"):
                grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        finally:
            self._armed = False
        for bi, (i0, i1, f0, f1) in enumerate(self.buckets):       # buckets with parameters the loss does not depend on
            if bi not in self._done:
                self._finish_bucket(bi, grads[i0:i1])
            self._got[bi].clear()
        return self.flat

    def all_reduce(self, average=False):
        """wait for the bucket all-reduces started during backward() (with RCCL, wait() orders the CURRENT stream after
        the communication stream; the host does not block)"""
        import time
        t0 = time.perf_counter()
        ev = None
        if self.time_wait_events and self._pending and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._pending:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.wait_events.append(ev)
        self._pending = []
        self.stats["allreduce_wait_host_s"] += time.perf_counter() - t0
        self.stats["allreduce_calls"] += 1
        if average and self._world() > 1:
            self.flat.div_(self._world())
        return self.flat
