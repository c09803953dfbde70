#!/usr/bin/env python3
"""Headline bench: faces/sec of 3DMM decode + depth render, batch 64 @ 200x200, fp32 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one 64-face batch of synthetic 235-d parameters already resident in
HBM: fr_decode_render_forward = fr_decode_3dmm -> fr_render_depth_forward (all four output planes; the vertices handed over
in the library's pitched rows), through the C ABI, on the BFM-scale
synthetic assets of SURVEY.md 8(d) (N = 53,215, T = 105,840, 199 + 29 components).  Every rank runs its own 64
faces (weak scaling; the path has no collective), the K steps are bracketed by barrier + synchronize, the MAX
over ranks is taken and rank 0 prints ONE JSON line.

The oracle (oracle/) is used only for the `cpu_baseline` leg: its single-thread decode + render of a bounded
sample of the same workload, timed on the host cores of the same box.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SURVEY.md 8(d) algorithmic bytes / flops (N=53,215, T=105,840, K=228, H=W=200)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: the measured copy rate (SURVEY.md 8d asks for the fraction of both)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: exact-f32 MFMA == fp32 vector peak


def algorithmic_bytes(N, T, K, H, W, B):
    basis = 4.0 * 3 * N * (K + 1) / B
    params = 4.0 * (7 + K)
    vproj = 4.0 * 3 * N
    shared = (4.0 * 3 * T + 4.0 * 3 * N) / B
    planes = 4.0 * H * W * 8
    return {"pipeline": basis + params + 2 * vproj + shared + planes,
            "render": vproj + shared + planes,
            "decode": basis + params + vproj}


def pkg(name):
    return importlib.import_module("3dfacerecon_amd." + name)


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (nothing
    here has touched the GPU yet) and exit with its code."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(assets, params_np, n_faces, H, W, synth):
    """Single-thread CPU restatement of the reference path on a bounded sample of the same workload (the bench batch
    first, then further draws of the same sampler).  `value` = faces/s of [matmul-style fp32 decode through single-thread
    BLAS (oracle.decode_3dmm_blas: network.py:140-171 as the reference's graph evaluates it) + the op's CPU functor
    restated (render_depth_op.cc:132-322)].  Reported beside it: the render-only rates of the op functor and of the MEX
    z-buffer (prepare_data/ZBuffer, the north_star's 'reference CPU z-buffer'), and the decode time of the parity oracle
    (a deliberately serial fmaf chain, the bit-exact spec -- slow by construction, not part of `value`)."""
    import numpy as np
    from oracle import oracle as O
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except (AttributeError, OSError):
        pass
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:  # noqa: BLE001
        limiter = None
    B = params_np.shape[0]
    O.decode_3dmm_blas(params_np[:2], assets["mu"], assets["pc_shape"], assets["pc_exp"], float(H))  # warm-up (page-in)
    t_dec = t_ren = 0.0
    done, chunk = 0, 0
    V = None
    while done < n_faces:
        P = params_np if chunk == 0 else synth.sample_params_batch(B, im_size=H, beta=0.7, seed=3456 + 1000 * chunk)
        P = P[:min(B, n_faces - done)]
        t0 = time.perf_counter()
        Vc = O.decode_3dmm_blas(P, assets["mu"], assets["pc_shape"], assets["pc_exp"], float(H))
        t1 = time.perf_counter()
        O.render_depth(Vc, assets["tri"], assets["vertex"][None], H, W)
        t2 = time.perf_counter()
        t_dec += t1 - t0
        t_ren += t2 - t1
        V = Vc if V is None else V
        done += P.shape[0]
        chunk += 1
    # the MEX z-buffer of prepare_data/ZBuffer (all double, column-major), one face per call
    nz = min(n_faces, V.shape[0], 32)
    src = np.zeros((H, W, 3))
    tz0 = time.perf_counter()
    for b in range(nz):
        O.zbuffer_mex(V[b].astype(np.float64), assets["tri"].astype(np.float64), assets["vertex"].astype(np.float64), src)
    tz1 = time.perf_counter()
    # the parity oracle's decode (serial fmaf chain == the MFMA's summation order), on a few faces only
    ns = min(16, B)
    ts0 = time.perf_counter()
    O.decode_3dmm(params_np[:ns], assets["mu"], assets["pc_shape"], assets["pc_exp"], float(H))
    ts1 = time.perf_counter()
    if limiter is not None:
        limiter.restore_original_limits()
    total = t_dec + t_ren
    zb = (tz1 - tz0) / nz
    return {
        "value": n_faces / total, "unit": "faces/s", "cores": 1, "kind": "port",
        "sample": "%d faces (the bench batch + %d further draws of the same sampler): single-thread BLAS decode "
                  "(network.py:140-171 as two fp32 matmuls) + fr_oracle_render_depth_forward (render_depth_op.cc:132-322 "
                  "restated), one thread pinned to one core, %.1f s" % (n_faces, chunk - 1, total),
        "value_uses": "decode_blas_ms_per_face + render_op_ms_per_face",
        "value_is": "the WHOLE timed path on one CPU thread (decode + the op's CPU functor, render_depth_op.cc:132-322) -- the "
                    "like-for-like baseline of `value`.  north_star's 'reference CPU z-buffer (prepare_data/ZBuffer)' is the "
                    "RENDER-ONLY figure north_star_reference_cpu_zbuffer below (it has no decode stage); both speed-ups are "
                    "reported in speedup_vs_cpu",
        "north_star_reference_cpu_zbuffer": {
            "faces_per_s": 1.0 / zb, "ms_per_face": 1e3 * zb, "faces_timed": nz, "cores": 1,
            "what": "MM3D::ZBuffer + PointInTri (prepare_data/ZBuffer/ModalAndRef.cpp:3-142) restated in oracle/fr_oracle.c, "
                    "all-double column-major as the MEX computes, render only (no 3DMM decode), one face per call"},
        "decode_blas_ms_per_face": 1e3 * t_dec / n_faces,
        "render_op_ms_per_face": 1e3 * t_ren / n_faces,
        "render_only_faces_per_s": {"op_cpu_functor": n_faces / t_ren, "zbuffer_mex": 1.0 / zb},
        "zbuffer_mex_ms_per_face": 1e3 * zb,
        "decode_spec_oracle_ms_per_face": 1e3 * (ts1 - ts0) / ns,
        "host_cpu_model": host_cpu_model(), "host_cpus": os.cpu_count(),
    }


def parity_gate(plan, net, assets, params_np, H, W, n_faces, run=True):
    """The timed route's own buffers against the CPU oracle, on the first `n_faces` faces of this rank's batch.
      * render: oracle rasteriser (render_depth_op.cc:132-322 restated) on the PLAN'S vertex_proj -> all four planes
        bit for bit (a NaN equals a NaN);
      * decode with the host rotation (what the reference's tf.py_func hands over, network.py:150) through the same
        kernel: bit for bit against the spec oracle (nets/network.py:140-171);
      * the plan's decode itself evaluates the rotation in-kernel in float64 (device sincos vs glibc: last-bit
        differences in fp64 that survive the fp32 rounding only rarely): held to <= 2 ulp and >= 99 % equal, the bar
        tests/test_decode_gpu.py states."""
    import numpy as np
    import torch
    from oracle import oracle as O
    n = min(int(n_faces), plan.B)
    if run:
        plan.step()
    torch.cuda.synchronize(plan.device)
    V = plan.vertex_proj[:n].contiguous().cpu().numpy()   # (the plan hands the vertices over in pitched rows: a strided view)
    got = [t[:n].cpu().numpy() for t in plan.outputs()]
    want = O.render_depth(V, assets["tri"], assets["vertex"][None], H, W)
    bad_planes = 0
    for g, w in zip(got, want):
        for b in range(n):
            if not np.array_equal(g[b], w[b], equal_nan=True):
                bad_planes += 1
    R = O.rotation_matrix_batch(params_np[:n, :3])
    # (the oracle of the arithmetic the plan was built with: the f32 chain, or the opt-in Q30 specification)
    Vo = O.decode_3dmm(params_np[:n], assets["mu"], assets["pc_shape"], assets["pc_exp"], float(H), R=R, q30=int(plan.q30))
    Vr = net.vertices_transform(plan.params[:n], R=torch.as_tensor(R, device=plan.device))
    torch.cuda.synchronize(plan.device)
    bad_decode = int(sum(not np.array_equal(Vr[b].cpu().numpy(), Vo[b]) for b in range(n)))
    # in-kernel rotation leg: distance in units in the last place of the oracle's value
    a = V.view(np.int32).astype(np.int64)
    o = Vo.view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    o = np.where(o < 0, -(o & 0x7FFFFFFF), o)
    ulp = np.abs(a - o)
    frac_equal = float((ulp == 0).mean())
    max_ulp = int(ulp.max())
    ok = bad_planes == 0 and bad_decode == 0 and max_ulp <= 2 and frac_equal >= 0.99
    return {"faces": n, "route": "DecodeRenderPlan (fr_decode_render_forward, phases 8|1|2 on a triangle table packed once)",
            "mismatching_planes": bad_planes, "planes_checked": 4 * n,
            "decode_host_rotation_mismatching_faces": bad_decode,
            "decode_inkernel_rotation": {"max_ulp": max_ulp, "frac_bit_equal": frac_equal, "bar": "<= 2 ulp, >= 0.99 equal"},
            "ok": bool(ok)}


def clock_probe(L, dev, ms=2.0):
    """The clock the chip holds in an MFMA-dense loop right now: fr_debug_clock_probe (one 16-wave workgroup per CU issuing the
    decode's v_mfma_f32_16x16x4_f32 back to back for ~`ms` milliseconds; shader-clock ticks over 100 MHz ticks, per
    workgroup).  Returns the median over the workgroups in GHz."""
    import torch
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    iters = max(1, int(ms * 1e-3 * 2.1e9 / (6 * 32 * 4)))
    ticks = torch.zeros((cus, 2), dtype=torch.int64, device=dev)
    rc = L.fr_debug_clock_probe(ctypes_ptr(ticks), cus, iters, ctypes_stream(dev))
    if rc:
        raise RuntimeError("fr_debug_clock_probe rc=%d" % rc)
    torch.cuda.synchronize(dev)
    t = ticks.cpu().double()
    ghz = (0.1 * t[:, 0] / t[:, 1].clamp(min=1)).sort().values
    return float(ghz[len(ghz) // 2]), float(ghz[0]), float(ghz[-1])


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def ctypes_stream(dev):
    import ctypes
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def ops_surface_leg(net, ops, plan, B, H, W, K, Wm, R, dist_u, dev):
    """K steps of the reference's own call chain (nets/network.py:153-182): net.vertices_transform(params) ->
    rendering_layer.ops.render_depth(ver, tri, texture, image), output allocation per call included (as
    render_depth_op.cc:442-445 allocates per call).  Same brackets as the plan's timed region; median of R blocks."""
    import torch
    image = torch.zeros((B, H, W, 3), dtype=torch.float32, device=dev)
    params = plan.params

    def one():
        ver = net.vertices_transform(params)
        return ops.render_depth(ver, net.tri, net.vertex_code, image)

    with torch.no_grad():
        for _ in range(Wm):
            outs = one()
        blocks = []
        for _ in range(R):
            dist_u.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(K):
                outs = one()
            torch.cuda.synchronize(dev)
            blocks.append(dist_u.max_over_ranks(time.perf_counter() - t0, device=dev))
            dist_u.barrier()
    same = all(torch.equal(a, b) for a, b in zip(outs, plan.outputs()))
    blocks.sort()
    return blocks[(len(blocks) - 1) // 2], same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="faces per GPU per step")
    ap.add_argument("--im-size", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed K-step blocks; the median block is reported.  0 = auto: max(10, 1600 // K) -- the first "
                         "~30 ms of GPU work after the host-side set-up run on a chip that is still ramping its clock "
                         "(blocks_ms_per_step in the line shows it), so the blocks have to span well beyond that for "
                         "the median to be a steady-state block")
    ap.add_argument("--cpu-faces", type=int, default=2048, help="faces in the cpu_baseline sample (0 = skip); 2048 ~ 10 s")
    ap.add_argument("--graph", action="store_true", help="also report hipGraph-replay throughput")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch faces per GPU; strong: ONE --batch-face job cut into per-rank shards")
    ap.add_argument("--parity-faces", type=int, default=-1,
                    help="faces per rank the oracle checks before the line is printed (-1 = all of them; 0 = skip, the "
                         "line then carries parity = null and must not be quoted)")
    ap.add_argument("--no-ops-surface", action="store_true", help="skip the operator-surface leg")
    ap.add_argument("--route", choices=("auto", "inflight", "serial"), default="auto",
                    help="inflight: BatchesInFlight (--in-flight independent batches, each on its own stream, no edge between "
                         "them: consecutive steps go to alternating slots); serial: DecodeRenderPlan (one batch at a time, three "
                         "launches per batch on one stream).  auto = inflight (measured fastest, DESIGN.md 4.7)")
    ap.add_argument("--in-flight", type=int, default=2, help="slots of the inflight route (2 measured best; 3 is slower again)")
    ap.add_argument("--strip-rows", type=int, default=-1,
                    help="inflight route: rows per screen strip of the resolver in the slots' plans (FR_PHASES_STRIP_ROWS).  -1 = "
                         "pipeline.BatchesInFlight's own choice (8 where the library alone picks 10: smaller resolver workgroups fill "
                         "the other batch's gaps), 0 = the library's choice, n = n rows.  No result bit depends on it")
    ap.add_argument("--no-serial-leg", action="store_true", help="skip the serial plan's comparison leg (inflight route)")
    ap.add_argument("--q30-levels", type=int, default=4, choices=(0, 4, 5, 7),
                    help="also time the in-flight route with the Q30 decode (int8 matrix cores) at this many digit-product "
                         "levels, parity-gated against ITS oracle, and report it beside `value` as `q30_inflight` (0 = skip; "
                         "skipped anyway when FR_DECODE_ARITH already selects Q30 for the whole run, or off the inflight route)")
    ap.add_argument("--q30-parity-faces", type=int, default=8,
                    help="faces per slot the Q30 leg's oracle gate checks (its level-by-level oracle takes ~1 s per face)")
    ap.add_argument("--allreduce-mb", type=float, default=-1.0,
                    help="N > 1: time one SUM all-reduce of this many MB on the bench's process group before the timed "
                         "region (config 4's gradient is ~302 MB) and report bus GB/s.  -1 = 302 under nccl, 4 under gloo; 0 = skip")
    ap.add_argument("--no-rccl-selftest", action="store_true",
                    help="N = 1: skip the one-rank RCCL self-test (utils.dist.rccl_selftest: communicator on this GPU + one "
                         "302 MB all-reduce through it, reported as dist.rccl_selftest)")
    ap.add_argument("--config", type=int, default=0, choices=(0, 3, 4, 5),
                    help="3 / 4 / 5: run the caller config of BASELINE.json (examples/coarse_loop.py --config N --steps K) "
                         "instead of the headline hot-path bench, and print ITS line")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for N > 1.  nccl (= RCCL) is the product setting.  gloo exists to walk the "
                         "N > 1 control flow on a box with fewer GPUs than ranks (ranks then share devices, LOCAL_RANK "
                         "modulo the device count); the line's dist.backend says gloo, so it cannot pass for an RCCL run")
    args = ap.parse_args()

    if args.config:   # BASELINE.json configs[2..4]: the caller loops, as a child process (nothing here has touched the GPU)
        cmd = [sys.executable, os.path.join(ROOT, "examples", "coarse_loop.py"), "--config", str(args.config),
               "--steps", str(args.steps if "--steps" in sys.argv else 5)]
        sys.exit(subprocess.call(cmd))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1 and "RANK" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    # ONE line on stdout, whatever the libraries print: RCCL writes a version banner to file descriptor 1 when its first
    # communicator comes up (MIOpen and the runtime have their own moods).  From here on fd 1 is stderr; the JSON line goes to
    # the saved real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    dist_u = pkg("utils.dist")
    world, rank, local = dist_u.init_from_env(args.dist_backend if args.gpus > 1 else None)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; the hot path has no CPU fallback")
    if args.dist_backend == "gloo":
        local %= torch.cuda.device_count()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    synth = pkg("utils.synth")
    netm = pkg("nets.network")
    pipe = pkg("pipeline")
    H, W = args.im_size, args.im_size
    first, B, B_global = dist_u.bench_partition(args.batch, rank, world, args.scaling)
    if B == 0:
        raise SystemExit("bench.py --scaling strong: %d faces cannot be cut into %d non-empty shards" % (args.batch, world))
    assets = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=assets, batch_size=B, im_size=H, device=dev)
    if args.scaling == "strong":   # every rank draws the ONE global batch and keeps its own shard
        params_np = synth.sample_params_batch(args.batch, im_size=H, beta=0.7, seed=3456)[first:first + B]
    else:
        params_np = synth.sample_params_batch(B, im_size=H, beta=0.7, seed=3456 + rank)
    L = pkg("_lib").lib()
    inflight = args.route in ("auto", "inflight")
    serial_plan = pipe.DecodeRenderPlan(net, B, H, W)
    serial_plan.params.copy_(torch.as_tensor(params_np, device=dev))
    plan = serial_plan
    slot_params = [params_np]
    if inflight:
        # every slot its own batch: slot 0 the rank's batch above, slot i another draw of the same sampler
        plan = pipe.BatchesInFlight(net, B, H, W, slots=max(1, args.in_flight), strip_rows=(None if args.strip_rows < 0 else args.strip_rows))
        for i in range(1, len(plan.slots)):
            if args.scaling == "strong":
                slot_params.append(synth.sample_params_batch(args.batch, im_size=H, beta=0.7, seed=3456 + 1000 * i)[first:first + B])
            else:
                slot_params.append(synth.sample_params_batch(B, im_size=H, beta=0.7, seed=3456 + rank + 1000 * i))
        for sl, pn in zip(plan.slots, slot_params):
            sl.params.copy_(torch.as_tensor(pn, device=dev))
    torch.cuda.synchronize(dev)

    K, Wm = args.steps, args.warmup
    R = args.repeats if args.repeats > 0 else max(10, 1600 // max(1, K))
    # N > 1: one all-reduce of config 4's gradient size on the bench's own process group, before anything is timed
    ar_mb = args.allreduce_mb
    if ar_mb < 0:
        ar_mb = 0.0 if world == 1 else (302.0 if args.dist_backend == "nccl" else 4.0)
    allreduce = dist_u.allreduce_preflight(ar_mb, device=dev) if world > 1 and ar_mb > 0 else None

    # The timed region -- EXACTLY K steps between barrier + synchronize brackets -- is run R times back to back and the
    # MEDIAN block is reported (min / max alongside): at ~0.11 ms per step a single K = 20 block is 2.2 ms of wall clock,
    # short enough for clock ramps and host jitter to move it by several per cent.
    # Per-kernel durations are taken live, inside the timed regions, with HIP events on the launch stream -- on every
    # EV_EVERY-th step only: an event pair costs a few microseconds of stream bubble, which at ~110 us per step would
    # otherwise tax every step by ~7 %.
    # (round 6, r6k: with the driver's K = 20 two bracketed steps per block cost the serial leg 0.5-1 us per step against one, and
    # four cost the in-flight route 0.6: one bracketed step per twenty -- 80 samples per kernel at K = 20, R = 80)
    EV_EVERY, EV_FIRST = 20, 5   # steps 5, 25, 45, ...: never the block's first step (it starts on an idle chip)

    def timed_blocks(pl, with_events):
        is_fl = hasattr(pl, "slots")
        for _ in range(Wm):
            pl.submit() if is_fl else pl.step()
        # (events for every block are created up front and the per-block max over ranks is taken after the last block, so
        # that the host does nothing but barrier + synchronize + clock reads between two timed regions)
        evs = [({k: [torch.cuda.Event(enable_timing=True) for _ in range(4)] for k in range(min(EV_FIRST, K - 1), K, EV_EVERY)}
                if with_events else {}) for _ in range(R)]
        loc, ev_all = [], []
        for r in range(R):
            ev = evs[r]
            dist_u.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k in range(K):
                e = ev.get(k)
                if is_fl:        # step k goes to slot k mod S, on that slot's stream; the events are recorded on that stream
                    pl.submit(marks=e)
                elif e is None:
                    pl.step()
                else:            # the same three kernels, each bracketed by events (the render op launched phase by phase)
                    e[0].record()
                    pl.decode()
                    e[1].record()
                    pl.render_phase(1)
                    e[2].record()
                    pl.render_phase(2)
                    e[3].record()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            dist_u.barrier()
            loc.append(t1 - t0)
            ev_all.extend(ev.values())
        blk = [dist_u.max_over_ranks(t, device=dev) for t in loc]
        order = sorted(range(R), key=lambda q: blk[q])
        med = order[(R - 1) // 2]              # the median block (lower median for an even R)
        return blk, blk[med], loc[med], ev_all

    blocks, elapsed, local_med, ev_all = timed_blocks(plan, True)
    clocks = {"after_timed_blocks": clock_probe(L, dev)}   # (straight behind the last timed block: the chip is as warm as it gets)
    per_rank_ms = [1e3 * t / K for t in dist_u.gather_over_ranks(local_med, device=dev)]
    dist_info = dist_u.describe(device=dev)
    faces_per_step = int(round(dist_u.sum_over_ranks(B, device=dev)))   # what the ranks actually ran, counted by the group

    # the serial plan (three launches per batch, nothing in flight across batches) in the same process, same K / W / R
    serial_elapsed = None
    serial_ev = []
    if inflight and not args.no_serial_leg:
        _, serial_elapsed, _, serial_ev = timed_blocks(serial_plan, True)
        clocks["after_serial_leg"] = clock_probe(L, dev)

    # the same in-flight route with the Q30 decode (int8 matrix cores, `--q30-levels` digit-product levels): its own plans,
    # its own oracle gate -- reported BESIDE `value`, never as it
    q30_leg = None
    host = pkg("_lib")
    if inflight and args.q30_levels and not serial_plan.q30 and net._basis.q30_ws_bytes > 0:
        prev_arith, prev_lv = host.decode_arith(), host.q30_levels()
        host.set_decode_arith(host.DECODE_ARITH_Q30, args.q30_levels)
        # An AUXILIARY leg: `value` (the f32 route) is already measured at this point and gated below.  Whatever goes wrong
        # here -- the second 154 MB image, the extra plans, a mismatch against the Q30 oracle -- is reported inside
        # q30_inflight (error / parity.ok = false, no figure quoted) and never takes the line down.
        try:
            if os.environ.get("FR_BENCH_Q30_FAULT"):   # (tests/test_bench_gpu.py: the leg's failure path)
                raise RuntimeError("FR_BENCH_Q30_FAULT is set: the Q30 leg fails on purpose")
            qplan = pipe.BatchesInFlight(net, B, H, W, slots=max(1, args.in_flight), strip_rows=(None if args.strip_rows < 0 else args.strip_rows))
            for sl, pn in zip(qplan.slots, slot_params):
                sl.params.copy_(torch.as_tensor(pn, device=dev))
            qserial = pipe.DecodeRenderPlan(net, B, H, W)
            qserial.params.copy_(torch.as_tensor(params_np, device=dev))
            torch.cuda.synchronize(dev)
            qblocks, qelapsed, _, _ = timed_blocks(qplan, True)    # (event-bracketed steps like the f32 leg: equal treatment)
            _, qs_elapsed, _, qs_ev = timed_blocks(qserial, True)
            qpar = None
            if args.parity_faces != 0:
                for _ in range(3 * len(qplan.slots)):
                    qplan.submit()
                qplan.synchronize()
                nfq = min(B, max(1, args.q30_parity_faces))
                per = [parity_gate(sl, net, assets, pn, H, W, nfq, run=False) for sl, pn in zip(qplan.slots, slot_params)]
                qpar = {"ok": bool(all(q["ok"] for q in per)), "faces_checked": sum(q["faces"] for q in per),
                        "planes_checked": sum(q["planes_checked"] for q in per),
                        "mismatching_planes": sum(q["mismatching_planes"] for q in per),
                        "decode_host_rotation_mismatching_faces": sum(q["decode_host_rotation_mismatching_faces"] for q in per),
                        "decode_inkernel_rotation_max_ulp": max(q["decode_inkernel_rotation"]["max_ulp"] for q in per),
                        "oracle": "oracle.decode_3dmm(q30=%d) -- the Q30 specification with %d levels, bit for bit -- and the oracle "
                                  "rasteriser on the plan's own vertices" % (args.q30_levels, args.q30_levels)}
                if not qpar["ok"]:
                    print("bench.py: Q30 parity gate FAILED on rank %d (the Q30 figures are withheld; `value` is unaffected): %s"
                          % (rank, json.dumps(qpar)), file=sys.stderr)
            q30_leg = {"levels": args.q30_levels, "elapsed": qelapsed, "blocks": qblocks, "serial_elapsed": qs_elapsed, "parity": qpar,
                       "serial_kernels_ms": {"decode (q_stage_kernel + decode_q_ring_kernel)": sum(e[0].elapsed_time(e[1]) for e in qs_ev) / len(qs_ev),
                                             "raster_emit": sum(e[1].elapsed_time(e[2]) for e in qs_ev) / len(qs_ev),
                                             "resolve_write": sum(e[2].elapsed_time(e[3]) for e in qs_ev) / len(qs_ev)}}
            del qplan, qserial
        except Exception as e:  # noqa: BLE001
            if world > 1:   # (the leg is full of collectives: a rank that leaves it alone would hang the others -- better to die)
                raise
            import traceback
            traceback.print_exc()
            q30_leg = {"levels": args.q30_levels, "error": "%s: %s" % (type(e).__name__, e)}
        finally:
            host.set_decode_arith(prev_arith, prev_lv)
        # (a gate that failed on ONE rank is a failed gate for the line: the ranks agree before anything is reported)
        q30_bad = q30_leg.get("parity") is not None and not q30_leg["parity"]["ok"]
        if "error" not in q30_leg and dist_u.sum_over_ranks(1.0 if q30_bad else 0.0, device=dev) > 0 and not q30_bad:
            q30_leg["parity"] = dict(q30_leg["parity"] or {}, ok=False, failed_on="another rank")

    # the operator-surface route (allocations + pack_tri every call): same K / W / R, reported beside `value`
    ops_elapsed = ops_same = None
    if not args.no_ops_surface:
        serial_plan.step()
        ops_elapsed, ops_same = ops_surface_leg(net, pkg("rendering_layer.ops"), serial_plan, B, H, W, K, Wm, R, dist_u, dev)

    # parity gate: the oracle on every face of the timed route's buffers, on every rank
    parity = None
    if args.parity_faces != 0:
        nf = B if args.parity_faces < 0 else args.parity_faces
        if inflight:
            # several batches deep on every slot, all slots running beside each other; then every slot's vertices and planes
            # against the oracle on that slot's own parameters
            for _ in range(3 * len(plan.slots)):
                plan.submit()
            plan.synchronize()
            per_slot = [parity_gate(sl, net, assets, pn, H, W, nf, run=False) for sl, pn in zip(plan.slots, slot_params)]
            parity = dict(per_slot[0])
            parity["route"] = ("BatchesInFlight: %d slots (DecodeRenderPlans, fr_decode_render_forward, phases 8|1|2), each on its own "
                               "stream; checked after %d submits that ran beside each other" % (len(plan.slots), 3 * len(plan.slots)))
            parity["batches_checked"] = len(per_slot)
            parity["faces_checked"] = sum(q["faces"] for q in per_slot)
            parity["planes_checked"] = sum(q["planes_checked"] for q in per_slot)
            parity["mismatching_planes"] = sum(q["mismatching_planes"] for q in per_slot)
            parity["decode_host_rotation_mismatching_faces"] = sum(q["decode_host_rotation_mismatching_faces"] for q in per_slot)
            parity["decode_inkernel_rotation"] = {
                "max_ulp": max(q["decode_inkernel_rotation"]["max_ulp"] for q in per_slot),
                "frac_bit_equal": min(q["decode_inkernel_rotation"]["frac_bit_equal"] for q in per_slot),
                "bar": per_slot[0]["decode_inkernel_rotation"]["bar"]}
            parity["ok"] = bool(all(q["ok"] for q in per_slot))
        else:
            parity = parity_gate(plan, net, assets, params_np, H, W, nf)
        all_ok = dist_u.sum_over_ranks(0.0 if parity["ok"] else 1.0, device=dev) == 0.0
        parity["faces_all_ranks"] = int(round(dist_u.sum_over_ranks(parity["faces"], device=dev)))
        parity["mismatching_planes_all_ranks"] = int(round(dist_u.sum_over_ranks(parity["mismatching_planes"], device=dev)))
        if not all_ok:
            if rank == 0 or not parity["ok"]:
                print("bench.py: PARITY GATE FAILED on rank %d: %s" % (rank, json.dumps(parity)), file=sys.stderr)
            dist_u.barrier()
            dist_u.finalize()
            sys.exit(3)
    decode_ms = sum(e[0].elapsed_time(e[1]) for e in ev_all) / len(ev_all)
    in_region = None
    emit_ms = sum(e[1].elapsed_time(e[2]) for e in ev_all) / len(ev_all)
    resolve_ms = sum(e[2].elapsed_time(e[3]) for e in ev_all) / len(ev_all)
    # one batch's parameters-to-planes time inside the timed region (event before its decode -> event behind its resolve)
    latency_ms = sum(e[0].elapsed_time(e[3]) for e in ev_all) / len(ev_all)
    serial_latency_ms = (sum(e[0].elapsed_time(e[3]) for e in serial_ev) / len(serial_ev)) if serial_ev else None
    cov = float(((plan.slots[0] if inflight else plan).tri_ind >= 0).float().mean().item())
    if inflight and serial_ev:
        # Two batches in flight: a kernel's event-bracketed duration in THAT region measures how the two streams share the
        # chip (a kernel of the low-priority stream waits for the other batch's workgroups), not the kernel.  The per-kernel
        # figures and the roofline object therefore come from the serial leg of this same process -- the same kernels, same
        # K / W / R and brackets, one batch in flight -- and the in-region durations ride along as `in_region_avg_ms`.
        in_region = {"decode": decode_ms, "raster_emit": emit_ms, "resolve_write": resolve_ms}
        decode_ms = sum(e[0].elapsed_time(e[1]) for e in serial_ev) / len(serial_ev)
        emit_ms = sum(e[1].elapsed_time(e[2]) for e in serial_ev) / len(serial_ev)
        resolve_ms = sum(e[2].elapsed_time(e[3]) for e in serial_ev) / len(serial_ev)

    graph_fps = graph_multi = None
    if args.graph:   # (the serial plan: three kernel nodes per replay)
        serial_plan.capture()
        for _ in range(Wm):
            serial_plan.replay()
        torch.cuda.synchronize(dev)
        g0 = time.perf_counter()
        for _ in range(K):
            serial_plan.replay()
        torch.cuda.synchronize(dev)
        graph_fps = B * K / (time.perf_counter() - g0)
        # ... and R batches per graph (pipeline.GraphedSteps): the replay-to-replay bubble paid once per R batches
        graph_multi = {}
        for Rg in (4, 8):
            gs = pipe.GraphedSteps(net, B, Rg, H, W)
            for i, p in enumerate(gs.plans):
                p.params.copy_(torch.as_tensor(slot_params[i % len(slot_params)], device=dev))
            gs.capture()
            nrep = max(1, K // Rg)
            for _ in range(max(1, Wm // Rg)):
                gs.replay()
            torch.cuda.synchronize(dev)
            g0 = time.perf_counter()
            for _ in range(nrep):
                gs.replay()
            torch.cuda.synchronize(dev)
            dtg = time.perf_counter() - g0
            same = all(torch.equal(a, b) for a, b in zip(gs.plans[0].outputs(), serial_plan.outputs()))
            # the same R plans stepped EAGERLY in turn (no graph): separates what the graph costs from what R sets of buffers
            # cycling through the 256 MiB Infinity Cache cost (the serial plan re-uses ONE set)
            for _ in range(max(1, Wm // Rg)):
                for p in gs.plans:
                    p.step()
            torch.cuda.synchronize(dev)
            g0 = time.perf_counter()
            for _ in range(nrep):
                for p in gs.plans:
                    p.step()
            torch.cuda.synchronize(dev)
            dte = time.perf_counter() - g0
            graph_multi["steps_per_graph_%d" % Rg] = {"faces_per_s": B * Rg * nrep / dtg, "us_per_batch": 1e6 * dtg / (Rg * nrep),
                                                      "same_plans_stepped_eagerly_us_per_batch": 1e6 * dte / (Rg * nrep),
                                                      "planes_identical_to_the_serial_plan": bool(same)}
            del gs

    rccl_st = None
    if world == 1 and not args.no_rccl_selftest:   # (after everything timed: the one-GPU box's only contact with RCCL)
        rccl_st = dist_u.rccl_selftest(dev)
    if rank == 0:
        N, T, Kc = net.nvert, int(net.tri.shape[1]), net.ndim_shape + net.ndim_exp
        ab = algorithmic_bytes(N, T, Kc, H, W, B)
        assert faces_per_step == B_global, (faces_per_step, B_global)
        value = faces_per_step * K / elapsed
        flops = 2.0 * 3 * N * Kc * B
        # algorithmic bytes of the render op (1,948,403 B/face) split over its two kernels: the emit kernel
        # owns the vertex / triangle / texture reads, the resolve kernel the four output planes
        emit_bytes = (4.0 * 3 * N + (4.0 * 3 * T + 4.0 * 3 * N) / B) * B
        resolve_bytes = 4.0 * H * W * 8 * B
        q30 = serial_plan.q30   # FR_DECODE_ARITH=q30 (opt-in, frozen experiment; the default is the f32 chain)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            pmc = {}
        ev_note = ("HIP events on the launch stream around each launch of every %dth step of the timed blocks (the step then goes "
                   "out launch by launch: separate C calls + event bubbles): an UPPER bound of the kernel's duration in the "
                   "un-bracketed steps; the rocprofv3 average of the same kernel is `rocprofv3_avg_ms` "
                   "(%s, a PROFILED run: slower clock)" % (EV_EVERY, pmc.get("kernel_stats_csv", "profiles/, the round's kernel_stats.csv")))
        if in_region is not None:
            ev_note += ("; taken in the SERIAL leg of this process (one plan, one stream, one batch in flight; its throughput is "
                        "`serial_plan_faces_per_s`): with two batches in flight a kernel's bracketed duration measures how the two "
                        "streams share the chip, not the kernel -- those durations are `in_region_avg_ms`")
        elif inflight:
            ev_note += "; TWO batches in flight and no serial leg (--no-serial-leg): the durations include the time the kernel waits for or shares the chip with the other batch's kernels"
        if q30:   # int8-MFMA blend: the matrix pipe is no longer the bound, the 153 MB basis + 41 MB output stream is
            roof_decode = {"bound": "hbm", "kernel": "q_stage_kernel + decode_q_ring_kernel<16,4,8,2,nt,%d,1> (fr_decode_render_forward_q30, phase 8)" % int(q30),
                           "achieved": ab["decode"] * B / (decode_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "traffic": None, "avg_ms": decode_ms, "algorithmic_bytes_per_launch": ab["decode"] * B}
        else:
            roof_decode = {"bound": "mfma", "kernel": "decode_ring_kernel<13,2,8,2,16,64,4,nt,prio> (phase 8)",
                           "achieved": flops / (decode_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "traffic": None, "avg_ms": decode_ms, "algorithmic_flop_per_launch": flops,
                           "hbm_GBs": ab["decode"] * B / (decode_ms * 1e-3) / 1e9,
                           "hbm_frac_of_8TBs": ab["decode"] * B / (decode_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_bytes_per_launch": ab["decode"] * B,
                           "which_side_binds": "THIS KERNEL ALONE: priced against the fp32 MFMA peak because the flops put it at the "
                               "ridge (29.6 us of MFMA vs 23.4 us of HBM at the spec peaks), but the ablation says its MEMORY side "
                               "binds: with its MFMAs removed it takes 47-50 us, with its stores removed 44-47 us, MFMA-only 36 us at "
                               "the clock it holds (DESIGN.md 4.1, profiles/round5_decode_breakdown.json): read `frac` as "
                               "matrix-pipe utilisation and hbm_frac_of_8TBs as the HBM roofline fraction.  THE STEP is priced "
                               "under `roofline.step`"}
        kernels = {"decode": roof_decode}
        kernels["raster_emit"] = {"bound": "hbm", "kernel": "raster_emit_kernel (fr_decode_render_forward, phase 1)",
                                  "achieved": emit_bytes / (emit_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "traffic": None, "avg_ms": emit_ms, "algorithmic_bytes_per_launch": emit_bytes}
        kernels["resolve_write"] = {"bound": "hbm", "kernel": "resolve_write_kernel<256> (fr_decode_render_forward, phase 2)",
                                    "achieved": resolve_bytes / (resolve_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "traffic": None, "avg_ms": resolve_ms, "algorithmic_bytes_per_launch": resolve_bytes}
        # HBM traffic per launch + the rocprofv3 kernel averages, from the committed profile passes of this same command
        # (profiles/pmc_traffic.json: FETCH_SIZE / WRITE_SIZE collected in separate --pmc passes; ONE stated correction for all
        # kernels -- see its `correction` field -- so the figures of one line are comparable)
        if pmc.get("batch") == B and (H, W) == (200, 200) and not q30 and isinstance(pmc.get("kernels"), dict):
            for name, r in kernels.items():
                rec = pmc["kernels"].get(name)
                if rec:
                    r["traffic"] = rec.get("traffic_bytes_per_launch")
                    r["traffic_source"] = "%s (%s)" % (pmc.get("source_file", "profiles/pmc_traffic.json"), pmc.get("correction", ""))
                    if rec.get("rocprofv3_avg_ms") is not None:
                        r["rocprofv3_avg_ms"] = rec["rocprofv3_avg_ms"]
        for name, r in kernels.items():
            r["frac"] = r["achieved"] / r["peak"]
            r["avg_ms_is"] = ev_note
            if in_region is not None and name in in_region:
                r["in_region_avg_ms"] = in_region[name]
        dominant = max(kernels.values(), key=lambda r: r["avg_ms"])  # the kernel with the longest average launch
        render_ms = emit_ms + resolve_ms
        kernels["render_op"] = {"bound": "hbm", "kernel": "fr_render_depth_forward (both kernels)", "avg_ms": render_ms,
                                "achieved": ab["render"] * B / (render_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": ab["render"] * B / (render_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_launch": ab["render"] * B}
        route = ("inflight: BatchesInFlight -- %d independent batches in flight, each slot a DecodeRenderPlan "
                 "(fr_decode_render_forward, three launches per batch) on its own stream with no edge between the streams; step k "
                 "goes to slot k mod %d; a timed block = K such steps, i.e. K batches from parameters to planes between the "
                 "brackets (both streams drained at either bracket)" % (len(plan.slots), len(plan.slots))) if inflight else \
                "serial: DecodeRenderPlan / fr_decode_render_forward -- three launches per batch, one batch at a time"
        out = {
            "metric": "faces/sec (3DMM decode+depth render), batch 64 @200x200",
            "value": value, "unit": "faces/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": args.scaling,
            "repeats": R, "value_is": "median over %d timed blocks of K steps each" % R, "route": route,
            "value_route": "inflight" if inflight else "serial",
            "value_meaning": ("THROUGHPUT of independent batches: K batches go from parameters to planes between the brackets with "
                              "%d in flight at any time; ms_per_step is the interval between finished batches, NOT one batch's "
                              "latency (per_batch_latency_ms).  A dependent loop (CoarseNet -> render -> CoarseNet, "
                              "nets/network.py:113-116) gets config.value_one_batch_at_a_time, the serial plan's figure, which is "
                              "also the one comparable with rounds 1-3" % len(plan.slots)) if inflight else
                             "one batch at a time: ms_per_step is also one batch's parameters-to-planes time",
            "per_batch_latency_ms": latency_ms,
            "per_batch_latency_is": "mean over the event-bracketed steps of the timed blocks: HIP event before the batch's decode launch "
                                    "-> event behind its resolve, on the batch's own stream" + (
                                        "; the same for the serial leg: %.4f ms" % serial_latency_ms if serial_latency_ms is not None else ""),
            "clock_GHz_held": {k: {"median": v[0], "min": v[1], "max": v[2]} for k, v in clocks.items()},
            "clock_GHz_held_is": "fr_debug_clock_probe: ~2 ms of back-to-back v_mfma_f32_16x16x4_f32 on 16 waves per CU (the decode's "
                                 "instruction and occupancy), shader-clock ticks / 100 MHz ticks per workgroup, launched straight behind "
                                 "the timed blocks (and behind the serial leg): the clock THIS part holds under matrix load today -- "
                                 "parts differ by several per cent, which is what moves the in-flight figure from box to box "
                                 "(DESIGN.md 4.7)",
            "value_min": faces_per_step * K / max(blocks), "value_max": faces_per_step * K / min(blocks),
            "ms_per_step_min": 1e3 * min(blocks) / K, "ms_per_step_max": 1e3 * max(blocks) / K,
            "blocks_ms_per_step": [round(1e3 * t / K, 5) for t in blocks],   # every timed block, in the order they ran
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: batch %d random 235-d params -> 3DMM decode -> depth render, "
                                   "%dx%d, fp32, all four output planes" % (args.batch, H, W),
                       "faces_per_gpu": B if args.scaling == "weak" else None, "faces_per_step_all_gpus": faces_per_step,
                       "value_one_batch_at_a_time": (faces_per_step * K / serial_elapsed) if serial_elapsed is not None else
                                                    (value if not inflight else None),
                       "faces_per_gpu_rank0": B, "batches_in_flight": len(plan.slots) if inflight else 1,
                       "resolver_strip_rows": ({"in_flight_slots": plan.strip_rows or "library default", "serial_plan": "library default"}
                                               if inflight else "library default"), "nver": N, "ntri": T, "n_shape": net.ndim_shape, "n_exp": net.ndim_exp,
                       "sampler": "sample_test.py:23-38 beta=0.7 " + ("seed=3456+rank" if args.scaling == "weak" else "seed=3456, one batch for all ranks"), "coverage": cov,
                       "sharding": ("weak: every rank runs its own %d faces" % B if args.scaling == "weak" else
                                    "strong: ONE %d-face batch cut into contiguous shards (utils.dist.shard_range)" % args.batch)
                                   + ", no data-path collective",
                       "decode_arith": ("q30, %d digit-product levels (fixed point on the int8 MFMA, include/fr_hotpath.h)" % int(q30)) if q30 else "f32 fmaf chain (f32-input MFMA)",
                       "constants": "the packed basis (fr_decode_pack_basis) and the pre-validated triangle table "
                                    "(fr_decode_render_forward, phase 4) are built once per plan: both are "
                                    "tf.constants of the reference model (network.py:41-43, 178); a caller that repacks the "
                                    "triangle list every call (fr_render_depth_forward) pays one more 5 us kernel per step"},
            "roofline": dominant,
            "kernels": kernels,
            "parity": parity,
            "dist": dict(dist_info, per_rank_ms_per_step=per_rank_ms, launcher_world_size=world),
        }
        # BASELINE.json configs[2..4] (the caller loops around this path): NOT run here -- `bench.py --config 3|4|5` runs them -- but the
        # round's recorded lines ride along inside `config`, marked as what they are
        try:
            cc = json.load(open(os.path.join(ROOT, "profiles", "round6_caller_configs.json")))
            out["config"]["caller_configs_recorded"] = {
                "source": "profiles/round6_caller_configs.json (bench.py --config 3|4|5 on one MI355X = the per-GPU shard of configs 4-5; "
                          "recorded in the round's profile session, not measured by this run)",
                **{k: {"faces_per_s_per_gpu": v["line"]["value"], "ms_per_step": v["line"]["ms_per_step"], "faces_per_gpu": v["line"]["faces_per_gpu"],
                       "hot_path_share_of_gpu_time": v.get("rocprofv3_kernel_stats", {}).get("hot_path_share")}
                   for k, v in cc.items() if "line" in v}}
        except (OSError, ValueError, KeyError):
            pass
        if ops_elapsed is not None:
            out["ops_surface_faces_per_s"] = faces_per_step * K / ops_elapsed
            out["ops_surface"] = {"ms_per_step": 1e3 * ops_elapsed / K, "vs_plan": elapsed / ops_elapsed,
                                  "outputs_identical_to_plan": bool(ops_same),
                                  "route": "FaceRecNet.vertices_transform -> rendering_layer.ops.render_depth (outputs "
                                           "allocated per call; the pre-validated triangle table is reused while the "
                                           "same `tri` tensor is passed: ops._render_phases)"}
        if serial_elapsed is not None:
            out["serial_plan_faces_per_s"] = faces_per_step * K / serial_elapsed
            out["serial_plan"] = {"ms_per_step": 1e3 * serial_elapsed / K, "timed_route_vs_serial": serial_elapsed / elapsed,
                                  "route": "DecodeRenderPlan.step(): decode -> emit -> resolve, three launches, same process, "
                                           "same K / W / R (median block)"}
        need = 0.4 * HBM_PEAK_GBS * 1e9 / ab["pipeline"]   # north_star: ">= 40 % of the HBM roofline", per GPU
        q30_ok = q30_leg is not None and "error" not in q30_leg and (q30_leg["parity"] is None or q30_leg["parity"]["ok"])
        if q30_leg is not None and not q30_ok:
            # the leg failed or its gate did: no Q30 figure is quoted anywhere in the line
            out["q30_inflight"] = {"levels": q30_leg["levels"], "error": q30_leg.get("error", "Q30 parity gate failed"),
                                   "parity": q30_leg.get("parity")}
        elif q30_leg is not None:
            qv = faces_per_step * K / q30_leg["elapsed"]
            out["q30_inflight_faces_per_s"] = qv
            out["q30_inflight"] = {
                "decode_arith": "Q30 fixed point on the int8 matrix cores, %d digit-product levels (include/fr_hotpath.h: "
                                "fr_decode_render_forward_q30)" % q30_leg["levels"],
                "ms_per_step": 1e3 * q30_leg["elapsed"] / K, "vs_value": qv / value,
                "ms_per_step_min": 1e3 * min(q30_leg["blocks"]) / K, "ms_per_step_max": 1e3 * max(q30_leg["blocks"]) / K,
                "frac_of_8TBs": ab["pipeline"] * qv / world / 1e9 / HBM_PEAK_GBS,
                "serial_plan_ms_per_step": 1e3 * q30_leg["serial_elapsed"] / K,
                "serial_plan_faces_per_s": faces_per_step * K / q30_leg["serial_elapsed"],
                "serial_leg_kernels_avg_ms": q30_leg["serial_kernels_ms"],
                "parity": q30_leg["parity"],
                "why_not_value": "the two arithmetics are two written definitions of the same blend, each held to its own CPU "
                                 "restatement bit for bit; `value` stays on the f32 chain (the reference's arithmetic type, the "
                                 "figure rounds 1-4 reported).  Same route, same K / W / R, same slots' parameters"}
        # ---- everything a reader needs to recompute the line's fractions, INSIDE `roofline` (the object the driver's record keeps):
        #      the dominant kernel's own fields (above), the other kernels in short, the whole step, the Q30 step, the clock ----
        per_gpu = value / world
        serial_v = (faces_per_step * K / serial_elapsed) if serial_elapsed is not None else (value if not inflight else None)
        step = {"what": "the WHOLE step (decode -> emit -> resolve) against the HBM roofline: algorithmic bytes per face (SURVEY.md 8d: "
                        "basis / B + params + 2 x vertices + (triangles + texture) / B + four planes) x faces/s per GPU / 8 TB/s",
                "algorithmic_bytes_per_face": ab["pipeline"], "faces_per_step_per_gpu": B, "route": out["value_route"],
                "batches_in_flight": out["config"]["batches_in_flight"],
                "ms_per_step": 1e3 * elapsed / K, "faces_per_s_per_gpu": per_gpu,
                "achieved_GBs": ab["pipeline"] * per_gpu / 1e9, "peak_GBs": HBM_PEAK_GBS,
                "frac_of_8TBs": ab["pipeline"] * per_gpu / 1e9 / HBM_PEAK_GBS,
                "frac_of_measured_copy_6.29TBs": ab["pipeline"] * per_gpu / 1e9 / HBM_COPY_GBS,
                "one_batch_at_a_time": None if serial_v is None else {
                    "faces_per_s_per_gpu": serial_v / world, "ms_per_step": 1e3 * B * world / serial_v,
                    "frac_of_8TBs": ab["pipeline"] * serial_v / world / 1e9 / HBM_PEAK_GBS},
                "north_star_40pct": {"needs_faces_per_s_per_gpu": need, "needs_ms_per_step": 1e3 * B / need,
                                     "value_passes": bool(per_gpu >= need),
                                     "one_batch_at_a_time_passes": None if serial_v is None else bool(serial_v / world >= need),
                                     "q30_passes": bool(out["q30_inflight_faces_per_s"] / world >= need) if q30_ok else None}}
        roof = dict(dominant)
        roof["step"] = step
        roof["q30"] = None if q30_leg is None else (
            {"levels": q30_leg["levels"], "error": out["q30_inflight"]["error"], "parity_ok": False if q30_leg.get("parity") else None}
            if not q30_ok else
            {"levels": q30_leg["levels"], "faces_per_s_per_gpu": out["q30_inflight_faces_per_s"] / world,
             "ms_per_step": out["q30_inflight"]["ms_per_step"], "frac_of_8TBs": out["q30_inflight"]["frac_of_8TBs"],
             "parity_ok": None if q30_leg["parity"] is None else bool(q30_leg["parity"]["ok"]),
             "faces_checked": None if q30_leg["parity"] is None else q30_leg["parity"]["faces_checked"],
             "one_batch_at_a_time_ms_per_step": out["q30_inflight"]["serial_plan_ms_per_step"],
             "one_batch_at_a_time_frac_of_8TBs": ab["pipeline"] * out["q30_inflight"]["serial_plan_faces_per_s"] / world / 1e9 / HBM_PEAK_GBS,
             "is": "the same route with the Q30 decode (a second written definition of the blend, gated against ITS oracle): "
                   "reported beside `value`, never as it"})
        roof["clock_GHz_held"] = {k: v[0] for k, v in clocks.items()}
        roof["kernels"] = {name: {k: r.get(k) for k in ("bound", "avg_ms", "achieved", "peak", "unit", "frac", "traffic",
                                                           "algorithmic_bytes_per_launch", "algorithmic_flop_per_launch",
                                                           "rocprofv3_avg_ms", "in_region_avg_ms") if r.get(k) is not None}
                           for name, r in kernels.items() if name != "render_op"}
        out["roofline"] = roof
        if rccl_st is not None:
            out["dist"]["rccl_selftest"] = rccl_st
        if allreduce is not None:
            out["dist"]["allreduce_preflight"] = allreduce
        if args.scaling == "strong":   # what one GPU's shard was measured to take (profiles/round3_strong_scaling_shards.json)
            try:
                sh = json.load(open(os.path.join(ROOT, "profiles", "round3_strong_scaling_shards.json")))["shards"]
                us = {k.split("(")[1].split()[0]: v["us_per_step"] for k, v in sh.items()}
                sp = {k.split("=")[1].split()[0]: v["predicted_speedup_vs_1_gpu"] for k, v in sh.items()}
            except (OSError, ValueError, KeyError, IndexError):
                us = sp = None
            out["dist"]["strong_scaling_prediction"] = {
                "us_per_step_at_faces_per_gpu": us,
                "predicted_speedup_vs_1_gpu": sp,
                "why": "every rank streams the whole 153 MB basis whatever its share of the batch, and each render kernel "
                       "keeps ~10 us of latency chain: strong scaling of ONE 64-face batch is structurally poor; the "
                       "path is meant to scale weakly (64 faces per GPU, no collective)",
                "source": "profiles/round3_strong_scaling_shards.json (serial plan, one MI355X running a rank's shard)"}
        if graph_fps is not None:
            out["graph_replay_faces_per_s"] = graph_fps
            out["graph_replay"] = {"one_step_per_graph_faces_per_s": graph_fps, "R_steps_per_graph": graph_multi,
                                   "what": "hipGraph replays of the serial plan's three launches; R steps per graph = "
                                           "pipeline.GraphedSteps (R plans, one hipGraphLaunch per R batches): the ~9 us bubble "
                                           "between two replays (profiles/round4_probes/r4h) is paid once per R batches"}
        if args.cpu_faces > 0:   # rank 0 only, at every N (the other ranks wait at the closing barrier)
            cb = cpu_baseline(assets, params_np, args.cpu_faces, H, W, synth)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu"] = {
                "n_gpus": world,
                "vs_cpu_baseline_value (decode + op functor, 1 thread)": value / cb["value"],
                "vs_north_star_reference_cpu_zbuffer (render only, 1 thread; the GPU figure includes the decode)":
                    value / cb["north_star_reference_cpu_zbuffer"]["faces_per_s"]}
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    dist_u.barrier()
    dist_u.finalize()


if __name__ == "__main__":
    main()
