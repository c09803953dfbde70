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


def init_from_env(backend=None, force=False):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1 (force=True: also for a
    single rank -- a one-rank RCCL group is how the one-GPU box can exercise the collective library at all).
    Returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    if (world > 1 or force) and not dist.is_initialized():
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


def device_identity(device="cpu"):
    """What THIS rank computes on: host, device name and a device id that is unique per physical GPU on the node (the PCI
    bus id; the uuid when torch exposes it).  A CPU rank (the gloo tests) reports its process id."""
    import socket
    ident = {"host": socket.gethostname(), "pid": os.getpid()}
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.type == "cuda" and torch.cuda.is_available():
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(idx)
        ident.update(device="cuda:%d" % idx, name=pr.name, cus=getattr(pr, "multi_processor_count", None),
                     hbm_gib=round(pr.total_memory / 2.0 ** 30, 1))
        bus = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        ident["pci"] = "%04x:%02x:%02x" % tuple(bus) if all(b is not None for b in bus) else None
        uuid = getattr(pr, "uuid", None)
        ident["uuid"] = str(uuid) if uuid is not None else None
        # (bus id AND uuid: a runtime that reports the same placeholder uuid for every GPU must not make distinct devices look
        # shared -- under nccl that raises -- and the device ordinal stands in when neither is exposed)
        ident["id"] = "%s/pci=%s/uuid=%s" % (ident["host"], ident["pci"] or ident["device"], ident["uuid"] or "-")
    else:
        ident.update(device="cpu", name="cpu", pci=None, uuid=None, id="%s/cpu-pid%d" % (ident["host"], ident["pid"]))
    return ident


def collective_library():
    """The version of the collective library torch would use for backend "nccl" (= RCCL on ROCm), or None."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:  # noqa: BLE001 -- a CPU-only torch build has no nccl module to ask
        return None


def describe(device="cpu"):
    """What the process group itself reports -- so that a bench line can prove what RCCL saw: backend, the world size as
    dist.get_world_size() returns it, how many ranks answered a SUM all-reduce of ones, and WHICH device every rank ran
    on (`devices`, in rank order; `distinct_devices` = no two ranks share one).  Under backend "nccl" two ranks on one GPU
    is a broken launch: it raises."""
    me = device_identity(device)
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "ranks_reporting": 1, "devices": [me], "distinct_devices": True,
                "rccl_version": collective_library()}
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, me)
    ids = [d["id"] for d in everyone]
    distinct = len(set(ids)) == len(ids)
    backend = dist.get_backend()
    if backend == "nccl" and not distinct:
        raise RuntimeError("ranks share a GPU under RCCL: %s" % ids)
    return {"backend": backend, "world_size": dist.get_world_size(),
            "ranks_reporting": int(round(sum_over_ranks(1.0, device))), "devices": everyone,
            "distinct_devices": distinct, "rccl_version": collective_library() if backend == "nccl" else None}


def allreduce_preflight(megabytes, device="cpu", iters=5):
    """Times a SUM all-reduce of `megabytes` MB of fp32 (config 4's ResNet-101 + FC gradient is ~302 MB: SURVEY.md 5)
    on the default group: -> {bytes, ms (median of `iters`, MAX over ranks), algbw_GBs = bytes / t,
    busbw_GBs = algbw * 2 (n - 1) / n: the per-link figure a ring all-reduce is bound by}.  None without a group."""
    import time
    if not (dist.is_available() and dist.is_initialized()) or megabytes <= 0:
        return None
    n = dist.get_world_size()
    cdev = _collective_device(device)
    t = torch.ones((int(megabytes * 1e6) // 4,), dtype=torch.float32, device=cdev)
    on_gpu = torch.device(cdev).type == "cuda"

    def sync():
        if on_gpu:
            torch.cuda.synchronize(cdev)

    dist.all_reduce(t)   # warm-up: communicator set-up, buffer registration
    sync()
    times = []
    for _ in range(iters):
        t.fill_(1.0)
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        dist.all_reduce(t)
        sync()
        times.append(max_over_ranks(time.perf_counter() - t0, device))
    ok = bool((t == float(n)).all().item())
    times.sort()
    sec = times[len(times) // 2]
    nbytes = t.numel() * 4
    return {"bytes": nbytes, "ms": 1e3 * sec, "algbw_GBs": nbytes / sec / 1e9,
            "busbw_GBs": nbytes / sec / 1e9 * 2.0 * (n - 1) / n, "sum_correct": ok, "iters": iters,
            "what": "one SUM all-reduce of config 4's gradient size on the bench's own process group"}


def rccl_selftest(device, megabytes=302.0):
    """ONE-rank RCCL group on `device` (only when no group exists yet): librccl is loaded, a communicator is created on this GPU
    and a SUM all-reduce of config 4's gradient size runs through it; then the group is destroyed.  What a one-GPU box can
    show of the collective library: that it initialises and that the call path torch DDP uses works here -- NOT bandwidth
    (one rank moves nothing over xGMI).  Returns a record; never raises (a failure is reported in the record)."""
    import socket
    if dist.is_available() and dist.is_initialized():
        return {"skipped": "a process group already exists"}
    rec = {"backend": "nccl", "world_size": 1, "rccl_version": collective_library()}
    try:
        with socket.socket() as sck:
            sck.bind(("127.0.0.1", 0))
            port = sck.getsockname()[1]
        dev = torch.device(device)
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port, device_id=dev)
        try:
            rec["allreduce"] = allreduce_preflight(megabytes, device=dev, iters=3)
            rec["describe"] = {k: v for k, v in describe(dev).items() if k != "devices"}
            rec["ok"] = bool(rec["allreduce"] and rec["allreduce"]["sum_correct"] and rec["describe"]["backend"] == "nccl")
        finally:
            dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- the self-test must not take the bench line down
        rec["ok"] = False
        rec["error"] = "%s: %s" % (type(e).__name__, e)
    return rec


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
