#!/usr/bin/env python3
"""Caller harness for BASELINE.json configs 3-5: CoarseNet (x nIter) around the decode -> render hot path, the depth
rendering layer, FineNet, forward only or the reference's training step, one process per GPU (RCCL all-reduce of the
network gradients through torch DDP).  Shaped like the reference's train loop (trainval.py:79-125: train step, then a
full forward on a validation batch every iteration) with synthetic images / labels.

    python examples/coarse_loop.py --config 3      # presets: 3 = CoarseNet + render forward, batch 32, 1 GPU;
                                                   #          4 = train loop, 256 faces over the ranks (32 per GPU at N = 8);
                                                   #          5 = Coarse + Fine joint forward at 448 x 448, 128 faces over the ranks
    python examples/coarse_loop.py --batch 32 --steps 5                       # config 3 spelled out
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29501 \
        examples/coarse_loop.py --batch 32 --steps 5 --train                  # config 4: 256 faces over 8 GPUs
    ... --im-size 448 --batch 16 --fine                                       # config 5: Coarse + Fine joint forward
    python examples/coarse_loop.py --phase test --batch 64 --steps 20         # the reference's eval loop (trainval.py:192-210):
                                                                              # independent test batches, the depth rendering of
                                                                              # batch k in flight beside the network of batch k+1

The render / decode path needs no collective (the batch is sharded); the only collectives are DDP's gradient all-reduce
and -- with --gather-sfs -- the all-gather that restores the reference's whole-batch lighting estimate of the
shape-from-shading loss (nets/losses.py).  The objective is the reference's (nets/network.py:336-378): pose MSE, geometry
MSE through the basis, SfS, fidelity, Laplacian smoothness; per training forward with nIter = 4 that is 5 decodes + 1
basis product and 7 render_depth calls, as in the reference graph.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


_emit = lambda rec: print(json.dumps(rec))   # noqa: E731  (main() points it at the real stdout)


def build_harness(args, dev, rank, world, local):
    """model (DDP-wrapped when training on > 1 rank), optimiser, step() closure."""
    synth, netm, cn, losses = pkg("utils.synth"), pkg("nets.network"), pkg("nets.coarse_net"), pkg("nets.losses")
    cn.apply_miopen_workaround()   # a process setting, applied by the entry point before the first convolution
    torch.manual_seed(1234)  # same initial weights on every rank
    A = synth.make_small_assets() if args.small else synth.make_assets()
    B, S = args.batch, args.im_size
    face = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    if args.small:  # the tiny mesh is ~30 px wide: centre it and scale it up a little
        face.init_pred_params[..., 6] = 1e-3 * S / 200.0
    # the reference's graph always holds FineNet (build(), network.py:69-101); the forward-only config 3 is CoarseNet + render
    model = cn.FaceReconModel(face, nIter=args.nIter, fine=args.fine or args.train).to(dev)
    net = model
    if args.train and world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local] if dev.type == "cuda" else None)
    opt = torch.optim.Adam(model.parameters(), lr=1e-5) if args.train else None  # run_experiment.sh:13

    g = torch.Generator(device="cpu").manual_seed(100 + rank)  # every rank has its own shard of the data
    def batch(seed):
        im = torch.rand((B, S, S, 1), generator=g).to(dev)
        lab = torch.as_tensor(synth.sample_params_batch(B, im_size=S, n_shape=face.ndim_shape, n_exp=face.ndim_exp,
                                                        beta=0.7, seed=seed + rank), device=dev)
        return im, lab
    train_b, val_b = batch(200), batch(300)

    def forward_loss(im, lab):
        out = net(im)
        return out, losses.get_loss(face, out["pred_params"], lab, im, out["vertices_proj"], out["coarse_depth_map"],
                                    out["pred_depth_map"], gather_sfs=args.gather_sfs)

    def step():
        if not args.train:
            with torch.no_grad():
                return net(*train_b[:1]), None
        model.train()
        out, L = forward_loss(*train_b)
        opt.zero_grad(set_to_none=True)
        L["total_loss"].backward()
        opt.step()
        if args.val:  # trainval.py:96-99: a second full forward on a validation batch every iteration
            model.eval()
            with torch.no_grad():
                forward_loss(*val_b)
        return out, L

    return model, net, opt, step


def run_test_phase(args, dev, rank, world, local, dist_u):
    """The reference's evaluation loop (trainval.py:192-210: `while True: images = next(generator); pred_depth = sess.run(...);
    write depth * 255`) over INDEPENDENT test batches.  CoarseNet of a batch is a dependent chain (network -> decode -> render ->
    network, nets/network.py:113-116) and runs as such on torch's current stream; the batch's final depth rendering
    (depth_rendering_layer, network.py:300-309: decode + render of the predicted parameters) does not feed anything of the NEXT
    batch, so it goes down `pipeline.BatchesInFlight.submit(pred_params)` -- its own stream, two batches in flight -- and its
    consumer (the reference writes depth * 255 to a jpg; here: the same product, reduced to a checksum) is ordered behind the slot
    with `make_current_stream_wait()` one iteration later, i.e. the render of batch k runs beside the network of batch k + 1."""
    synth, netm, cn, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("nets.coarse_net"), pkg("pipeline")
    cn.apply_miopen_workaround()
    torch.manual_seed(1234)
    A = synth.make_small_assets() if args.small else synth.make_assets()
    B, S = args.batch, args.im_size
    face = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    if args.small:
        face.init_pred_params[..., 6] = 1e-3 * S / 200.0
    coarse = cn.CoarseNet(face, nIter=args.nIter).to(dev).eval()
    flights = pipe.BatchesInFlight(face, B, S, S, slots=2)
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    images = [torch.rand((B, S, S, 1), generator=g).to(dev) for _ in range(4)]   # the "test generator": four resident batches

    dumped = []   # --dump-batches: what every timed batch left in its slot (copied out by the CONSUMER, on the current stream)

    def consume(slot, keep=False):
        """trainval.py:196: depth_images = pred_depth * 255.0, one image per face (written to disk there)."""
        slot.make_current_stream_wait()
        if keep:
            dumped.append([t.clone() for t in (slot.params, slot.vertex_proj.contiguous()) + tuple(slot.outputs())])
        return (slot.depth.clamp_min(1e-6) * 255.0).sum(dim=(1, 2, 3))

    def loop(n, keep=False):
        sums, pending = [], None
        with torch.no_grad():
            for k in range(n):
                params = coarse(images[k % len(images)])           # the dependent chain of THIS batch (current stream)
                slot = flights.submit(params)                       # its depth rendering: the slot's stream, nothing waits for it yet
                if pending is not None:
                    sums.append(consume(pending, keep))             # batch k-1's planes, behind ITS slot
                pending = slot
            sums.append(consume(pending, keep))
        return sums

    loop(max(args.warmup, 1))
    torch.cuda.synchronize(dev)
    dist_u.barrier()
    t0 = time.perf_counter()
    sums = loop(args.steps, keep=bool(args.dump_batches))
    torch.cuda.synchronize(dev)
    dt = dist_u.max_over_ranks(time.perf_counter() - t0, device=dev)
    if args.dump_batches and rank == 0:
        # every timed batch's predicted parameters, vertices and four planes as the consumer saw them: tests/test_programs_gpu.py
        # holds them to the CPU oracle (this program never loads it)
        import numpy as np
        names = ("params", "vertex_proj", "depth", "texture_image", "normal", "tri_ind")
        np.savez(args.dump_batches, steps=len(dumped), mu=A["mu"], pc_shape=A["pc_shape"], pc_exp=A["pc_exp"], tri=A["tri"], vertex=A["vertex"],
                 im_size=S, **{"%s_%d" % (n, k): t.cpu().numpy() for k, rec in enumerate(dumped) for n, t in zip(names, rec)})
    # the same batches one at a time through the plain surface: the in-flight loop must reproduce them bit for bit
    with torch.no_grad():
        ref = []
        for k in range(min(args.steps, len(images))):
            p = coarse(images[k])
            d = face.coarse_net_input(face.vertices_transform(p), im_gray=images[k])[1]
            ref.append((d * 255.0).sum(dim=(1, 2, 3)))
    same = all(torch.equal(a, b) for a, b in zip(sums, ref))
    dist_info = dist_u.describe(device=dev)
    if rank == 0:
        _emit(dict({"metric": "faces/sec, evaluation loop (trainval.py:192-210): CoarseNet x %d + depth rendering" % args.nIter,
                          "value": world * B * args.steps / dt, "unit": "faces/s", "higher_is_better": True, "data": "synthetic",
                          "dtype": "f32", "phase": "test", "n_gpus": world, "faces_per_gpu": B, "im_size": S, "steps": args.steps,
                          "ms_per_step": 1e3 * dt / args.steps, "batches_in_flight": len(flights.slots),
                          "route": "CoarseNet on torch's current stream; depth_rendering_layer of batch k through "
                                   "pipeline.BatchesInFlight.submit() beside the network of batch k + 1; consumer ordered with "
                                   "make_current_stream_wait()",
                          "depth_identical_to_one_batch_at_a_time": bool(same), "dist": dist_info}))
    dist_u.finalize()
    return 0 if same else 1


# BASELINE.json configs[2..4] as presets; the global batch is cut over the ranks that are present (one process per GPU)
CONFIG_PRESETS = {
    3: dict(label="configs[2]: CoarseNet (ResNet-101) + render_depth end-to-end forward, batch 32, 1 GPU",
            global_batch=32, im_size=200, train=False, val=False, fine=False),
    4: dict(label="configs[3]: CoarseNet train loop, batch 256 sharded over the ranks, RCCL all-reduce",
            global_batch=256, im_size=200, train=True, val=True, fine=True),
    5: dict(label="configs[4]: CoarseNet + FineNet joint forward, 448x448 input, batch 128 over the ranks",
            global_batch=128, im_size=448, train=False, val=False, fine=True),
}


def ddp_bucket_bytes(model, net):
    """Gradient bytes one training step all-reduces: every parameter that requires grad, fp32, once per step (DDP's
    buckets partition exactly this set); 0 when the model is not wrapped (single rank / forward only)."""
    if net is model:
        return 0
    return int(sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=sorted(CONFIG_PRESETS), default=None,
                    help="BASELINE.json config preset (3, 4 or 5): sets batch / im-size / train / val / fine; on fewer GPUs "
                         "than the config names each rank runs global_batch / max(world, named GPUs) faces (its shard)")
    ap.add_argument("--shard-gpus", type=int, default=8, help="GPUs configs 4 and 5 are quoted on (their per-GPU shard)")
    ap.add_argument("--batch", type=int, default=32, help="faces per GPU")
    ap.add_argument("--im-size", type=int, default=200)
    ap.add_argument("--nIter", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--val", action="store_true", help="with --train: the reference's per-iteration validation forward")
    ap.add_argument("--fine", action="store_true")
    ap.add_argument("--gather-sfs", action="store_true", help="whole-batch SfS lighting estimate across ranks")
    ap.add_argument("--small", action="store_true", help="tiny synthetic assets (smoke runs)")
    ap.add_argument("--dump-batches", default=None, metavar="FILE.npz",
                    help="--phase test: save every timed batch's parameters, vertices and planes (as its consumer saw them) for an "
                         "external checker")
    ap.add_argument("--phase", choices=("train", "test"), default="train",
                    help="trainval.py's phase switch (:223-225).  test = the forward-only evaluation loop over independent batches "
                         "with the depth rendering in flight (run_test_phase); train = everything else this script does")
    args = ap.parse_args()
    # (ONE line on stdout: RCCL prints a version banner to fd 1 when its first communicator comes up -- from here on fd 1 is
    # stderr and the JSON line goes to the saved real stdout)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    global _emit
    _emit = lambda rec: os.write(real_stdout, (json.dumps(rec) + "\n").encode())   # noqa: E731
    dist_u = pkg("utils.dist")
    world, rank, local = dist_u.init_from_env()
    if args.phase == "test":
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        sys.exit(run_test_phase(args, dev, rank, world, local, dist_u))
    preset = None
    if args.config is not None:
        preset = CONFIG_PRESETS[args.config]
        named = 1 if args.config == 3 else args.shard_gpus
        args.batch = preset["global_batch"] // max(world, named)
        args.im_size, args.train, args.val, args.fine = preset["im_size"], preset["train"], preset["val"], preset["fine"]
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    model, net, opt, step = build_harness(args, dev, rank, world, local)
    if not args.train:
        model.eval()
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize(dev)
    dist_u.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, L = step()
    torch.cuda.synchronize(dev)
    dt = dist_u.max_over_ranks(time.perf_counter() - t0, device=dev)
    dist_info = dist_u.describe(device=dev)   # (a collective: every rank calls it)
    if rank == 0:
        name = "configs[4]" if args.fine and not args.train else ("configs[3]" if args.train else "configs[2]")
        rec = {"metric": "faces/sec, caller config (%s)" % (preset["label"] if preset else name + " coarse_loop"),
               "value": world * args.batch * args.steps / dt, "unit": "faces/s", "higher_is_better": True,
               "data": "synthetic", "dtype": "f32", "scaling": "weak (per-GPU shard fixed)",
               "config": "%s coarse_loop" % name, "n_gpus": world, "faces_per_gpu": args.batch, "im_size": args.im_size,
               "global_batch_this_run": world * args.batch,
               "dist": dist_info,
               "ddp_allreduce_bytes_per_step": ddp_bucket_bytes(model, net),
               "nIter": args.nIter, "train": args.train, "val_forward": args.val, "fine": args.fine or args.train,
               "steps": args.steps, "faces_per_s": world * args.batch * args.steps / dt,
               "ms_per_step": 1e3 * dt / args.steps,
               "params_finite": bool(torch.isfinite(out["pred_params"]).all())}
        if L is not None:
            rec["losses"] = {k: float(v) for k, v in L.items()}
        _emit(rec)
    dist_u.finalize()


if __name__ == "__main__":
    main()
