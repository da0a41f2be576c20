"""Multi-GPU plumbing of the rollout path: one process per GPU, replicas sharded across
ranks with NO data-path collective (replicas never interact; reference
training/utils/device_child_process/process_group_torch.py:6-20 + trainer_base.py:249-252).

backend "nccl" is RCCL on ROCm (8 x MI355X over xGMI); "gloo" runs the same code on CPU and
is what the world_size-2 tests use.  The only collectives are a barrier and the reduction of
per-rank wall times -- the gradient all-reduce belongs to the trainer (DDP)."""
import os

import torch
import torch.distributed as dist


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def device_index(local_rank=None):
    """HIP device a rank drives: its local rank (one process per GPU).  WD_FORCE_DEVICE pins every
    rank to one device instead -- used by the tests that run the N > 1 path on a 1-GPU box."""
    forced = os.environ.get("WD_FORCE_DEVICE")
    if forced not in (None, ""):
        return int(forced)
    return int(rank_info()[1] if local_rank is None else local_rank)


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_numa_cpus(device):
    """CPUs of the NUMA node the GPU `device` hangs off (sysfs, through the PCI address torch reports), or None"""
    try:
        props = torch.cuda.get_device_properties(device)
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        return _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except Exception:
        return None


def rank_cpu_slice(local_rank, local_world, allowed, numa_cpus=None, ranks_on_node=None):
    """The CPUs rank `local_rank` of `local_world` ranks on this host should run on: a contiguous share of `allowed`
    (the process' affinity mask) -- of the part of it on the GPU's own NUMA node when that is known (`numa_cpus`;
    `ranks_on_node` = (index of this rank among the ranks of that node, their number)).  Never empty."""
    allowed = sorted(allowed)
    pool, idx, n = allowed, int(local_rank), max(1, int(local_world))
    if numa_cpus:
        near = [c for c in allowed if c in numa_cpus]
        if near and ranks_on_node is not None:
            pool, (idx, n) = near, ranks_on_node
    per = max(1, len(pool) // n)
    share = pool[idx * per:(idx + 1) * per] if idx < n - 1 else pool[idx * per:]
    return share or pool or allowed


def pin_rank_to_cpus(local_rank=None, local_world=None, device=None):
    """Give every rank of a node its own CPUs, next to its GPU (sched_setaffinity): with 8 ranks started by one
    launcher the host threads otherwise float over both sockets, and a rank whose Python thread sits on the far
    socket issues its launches later than the others -- host-side skew that reads as poor scaling in a weak-scaling
    run whose time is the SLOWEST rank's.  WD_PIN_CPUS=0 leaves the affinity alone.  Returns the CPU list (or None)."""
    if os.environ.get("WD_PIN_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    _, lr, world = rank_info()
    local_rank = lr if local_rank is None else int(local_rank)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world)) if local_world is None else int(local_world)
    if local_world <= 1:
        return None
    allowed = os.sched_getaffinity(0)
    numa, on_node = None, None
    if device is not None and torch.cuda.is_available() and os.environ.get("WD_FORCE_DEVICE") in (None, ""):
        numa = _gpu_numa_cpus(device)
        if numa:  # ranks drive device = local rank: the ranks whose GPU sits on the same node share its CPUs
            same = [r for r in range(local_world)
                    if r < torch.cuda.device_count() and (_gpu_numa_cpus(r) or set()) == numa]
            if local_rank in same:
                on_node = (same.index(local_rank), len(same))
    share = rank_cpu_slice(local_rank, local_world, allowed, numa, on_node)
    try:
        os.sched_setaffinity(0, share)
    except OSError:
        return None
    return sorted(share)


def init_process_group(backend=None, device_id=None):
    rank, local_rank, world = rank_info()
    if world == 1 or (dist.is_available() and dist.is_initialized()):  # (bench.py builds a trainer inside its own group)
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # RCCL refuses two ranks on one GPU; the 1-GPU tests of the N > 1 path ask for gloo here
    backend = os.environ.get("WD_DIST_BACKEND") or backend
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl" and device_id is not None:
        kwargs["device_id"] = torch.device("cuda", device_id)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_replicas(total_envs, world, rank):
    """Contiguous block of replicas owned by `rank` (16000 -> 8 x 2000): (first, count)."""
    base, extra = divmod(int(total_envs), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def rank_seed(base_seed, rank):
    """seed + device id, trainer_base.py:249-252"""
    return int(base_seed) + int(rank)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(seconds):
    """MAX of a per-rank wall time (the slowest rank defines the job's time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_ranks_ok(ok):
    """True iff `ok` on EVERY rank (one small collective): ranks agree on whether to go on before a step that contains
    collectives -- a rank that failed alone would leave the others waiting in theirs"""
    return min(gather_ints(1 if ok else 0)) == 1


def gather_ints(value):
    """[value of rank 0, value of rank 1, ...] on every rank"""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(value)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, int(value))
    return [int(v) for v in out]


def gather_floats(value):
    """[value of rank 0, value of rank 1, ...] on every rank (a skewed rank shows in one line)"""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, float(value))
    return [float(v) for v in out]


def time_allreduce(n_floats, repeats=50):
    """Median time (us) of one all-reduce of `n_floats` float32 -- the size of the trainer's flattened
    gradient bucket -- over the job's process group (RCCL over xGMI on a GPU node); None without a group."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    on_gpu = dist.get_backend() == "nccl"
    buf = torch.zeros(int(n_floats), dtype=torch.float32, device="cuda" if on_gpu else "cpu")
    for _ in range(5):
        dist.all_reduce(buf)
    times = []
    if on_gpu:
        torch.cuda.synchronize()
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(buf)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
    else:
        import time

        for _ in range(repeats):
            t0 = time.perf_counter()
            dist.all_reduce(buf)
            times.append((time.perf_counter() - t0) * 1e6)
    times.sort()
    return times[len(times) // 2]


def aggregate_throughput(units_this_rank, seconds_this_rank):
    """Whole-job throughput: all ranks' units / the slowest rank's time."""
    return sum_over_ranks(units_this_rank) / max_over_ranks(seconds_this_rank)


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
