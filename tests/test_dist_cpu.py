"""CPU, world_size 2, gloo: the N>1 path of bench.py -- batch sharding with no data-path collective, barrier,
max-over-ranks timing, whole-job aggregate."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = importlib.import_module("3dfacerecon_amd.utils.dist")
    synth = importlib.import_module("3dfacerecon_amd.utils.synth")
    w, r, _ = d.init_from_env("gloo")
    assert (w, r) == (world, rank)
    # strong-scaling style shard of a 7-face job, and the per-rank seeds of the weak-scaling bench
    lo, hi = d.shard_range(7, rank, world)
    P = synth.sample_params_batch(4, n_shape=5, n_exp=3, seed=3456 + rank)
    d.barrier()
    t = d.max_over_ranks(0.25 * (rank + 1))
    faces = d.sum_over_ranks(hi - lo)
    d.barrier()
    # the bench's strong-scaling partition of ONE 64-face batch, and what the process group reports about itself
    first, nloc, nglob = d.bench_partition(64, rank, world, "strong")
    wfirst, wloc, wglob = d.bench_partition(64, rank, world, "weak")
    info = d.describe()
    pre = d.allreduce_preflight(0.5, iters=2)
    times = d.gather_over_ranks(0.1 * (rank + 1))
    owned = d.sum_over_ranks(nloc)
    q.put((rank, lo, hi, t, faces, float(P.sum()), (first, nloc, nglob, wfirst, wloc, wglob, info, times, owned, pre)))
    d.finalize()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, f0, s0, x0), (r1, lo1, hi1, t1, f1, s1, x1) = res
    # strong scaling: rank 0 owns faces [0, 32), rank 1 [32, 64) of the one 64-face batch; weak: 64 each, 128 in all
    assert x0[:6] == (0, 32, 64, 0, 64, 128) and x1[:6] == (32, 32, 64, 64, 64, 128)
    for info in (x0[6], x1[6]):
        assert (info["backend"], info["world_size"], info["ranks_reporting"]) == ("gloo", 2, 2)
        # the line proves WHICH device every rank ran on: one identity per rank, in rank order, all different
        assert len(info["devices"]) == 2 and info["distinct_devices"] and info["rccl_version"] is None
        assert len({dv["id"] for dv in info["devices"]}) == 2 and all(dv["device"] == "cpu" for dv in info["devices"])
    assert x0[6]["devices"] == x1[6]["devices"]            # every rank holds the same, complete table
    for pre in (x0[9], x1[9]):                             # the all-reduce preflight: right sum, a bandwidth, the same on both
        assert pre["sum_correct"] and pre["bytes"] == 500000 and pre["ms"] > 0 and pre["busbw_GBs"] == pre["algbw_GBs"]
    assert x0[9]["ms"] == x1[9]["ms"]
    assert x0[7] == x1[7] == [0.1, 0.2] and x0[8] == x1[8] == 64
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)          # disjoint, covering, no overlap
    assert t0 == t1 == 0.5                                 # MAX over ranks
    assert f0 == f1 == 7                                   # every face counted once
    assert s0 != s1                                        # ranks draw different parameter batches


def test_single_process_helpers_are_noops():
    import importlib
    d = importlib.import_module("3dfacerecon_amd.utils.dist")
    assert d.max_over_ranks(1.5) == 1.5 and d.sum_over_ranks(3) == 3.0
    assert d.gather_over_ranks(2.5) == [2.5] and d.describe()["world_size"] == 1
    assert d.describe()["distinct_devices"] and d.describe()["devices"][0]["device"] == "cpu"
    assert d.allreduce_preflight(1.0) is None
    assert d.bench_partition(64, 0, 1, "strong") == (0, 64, 64) and d.bench_partition(64, 3, 8, "strong") == (24, 8, 64)
    assert d.bench_partition(10, 7, 8, "strong") == (9, 1, 10) and d.bench_partition(3, 7, 8, "strong")[1] == 0
    d.barrier()
