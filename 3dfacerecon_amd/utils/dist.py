"""One process per GPU under torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" for the CPU tests).

The decode + render path is embarrassingly parallel over the batch (reference batch loop render_depth_op.cc:180;
per-row decode network.py:153-169): faces are sharded across ranks and NO collective is on the data path.  The
only collectives here are the bench's barrier and the max-over-ranks of the step time.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1.
    Returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return world, rank, local


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` faces for `rank`; sizes differ by at most one, earlier ranks get
    the remainder, every face is owned exactly once."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(int(total), world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _collective_device(device):
    """gloo reduces host tensors (the CPU tests, and `bench.py --dist-backend gloo`); RCCL reduces device tensors."""
    return "cpu" if dist.get_backend() == "gloo" else device


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (the bench's step time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device="cpu"):
    """The python float of every rank, in rank order (the bench's per-rank step times)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def describe(device="cpu"):
    """What the process group itself reports -- so that a bench line can prove what RCCL saw: backend, the world size as
    dist.get_world_size() returns it, and how many ranks answered a SUM all-reduce of ones."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "ranks_reporting": 1}
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
            "ranks_reporting": int(round(sum_over_ranks(1.0, device)))}


def bench_partition(batch, rank, world, scaling="weak"):
    """Faces of one bench step owned by `rank`: -> (first_face, local_batch, global_batch).
    weak: every rank runs its own `batch` faces (global = world * batch); strong: ONE `batch`-face job is cut into
    contiguous shards with shard_range (8 faces per GPU for 64 faces on 8 GPUs), global = batch."""
    if scaling == "weak":
        return rank * int(batch), int(batch), world * int(batch)
    if scaling == "strong":
        lo, hi = shard_range(batch, rank, world)
        return lo, hi - lo, int(batch)
    raise ValueError("scaling must be 'weak' or 'strong'")


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
