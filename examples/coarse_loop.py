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


def build_harness(args, dev, rank, world, local):
    """model (DDP-wrapped when training on > 1 rank), optimiser, step() closure."""
    synth, netm, cn, losses = pkg("utils.synth"), pkg("nets.network"), pkg("nets.coarse_net"), pkg("nets.losses")
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
    args = ap.parse_args()
    dist_u = pkg("utils.dist")
    world, rank, local = dist_u.init_from_env()
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
        print(json.dumps(rec))
    dist_u.finalize()


if __name__ == "__main__":
    main()
