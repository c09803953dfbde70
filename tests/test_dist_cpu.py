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
    q.put((rank, lo, hi, t, faces, float(P.sum())))
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
    (r0, lo0, hi0, t0, f0, s0), (r1, lo1, hi1, t1, f1, s1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)          # disjoint, covering, no overlap
    assert t0 == t1 == 0.5                                 # MAX over ranks
    assert f0 == f1 == 7                                   # every face counted once
    assert s0 != s1                                        # ranks draw different parameter batches


def test_single_process_helpers_are_noops():
    import importlib
    d = importlib.import_module("3dfacerecon_amd.utils.dist")
    assert d.max_over_ranks(1.5) == 1.5 and d.sum_over_ranks(3) == 3.0
    d.barrier()
