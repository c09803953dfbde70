#!/usr/bin/env python3
"""Caller harness for BASELINE.json configs 3-5: CoarseNet (x nIter) around the decode -> render hot path, optionally
FineNet, forward only or a training step, one process per GPU (RCCL all-reduce of the network gradients through
torch DDP).  Shaped like the reference's train loop (trainval.py:79-125) but with synthetic images / labels.

    python examples/coarse_loop.py --batch 32 --steps 5                       # config 3: forward, 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29501 \
        examples/coarse_loop.py --batch 32 --steps 5 --train                  # config 4: 256 faces over 8 GPUs
    ... --im-size 448 --batch 16 --fine                                       # config 5

The render / decode path needs no collective (the batch is sharded); the only collective is DDP's gradient all-reduce.
Objective of the harness (the reference's five losses are out of scope): lambda_pose * MSE(pose) + lambda_geo *
MSE(shape, exp) against synthetic labels (network.py:27-28 weights) + a depth-fidelity term that sends gradient through
render backward and decode backward into every iteration's weights.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32, help="faces per GPU")
    ap.add_argument("--im-size", type=int, default=200)
    ap.add_argument("--nIter", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--fine", action="store_true")
    ap.add_argument("--small", action="store_true", help="tiny synthetic assets (smoke runs)")
    args = ap.parse_args()
    dist_u, synth, netm, cn = pkg("utils.dist"), pkg("utils.synth"), pkg("nets.network"), pkg("nets.coarse_net")
    world, rank, local = dist_u.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(1234)  # same initial weights on every rank

    A = synth.make_small_assets() if args.small else synth.make_assets()
    B, S = args.batch, args.im_size
    face = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    if args.small:  # the tiny mesh is ~30 px wide: centre it and scale it up a little
        face.init_pred_params[..., 6] = 1e-3 * S / 200.0
    coarse = cn.CoarseNet(face, nIter=args.nIter).to(dev)
    fine = cn.FineNet().to(dev) if args.fine else None
    model = torch.nn.ModuleList([coarse] + ([fine] if fine else []))
    if args.train and world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local])
    opt = torch.optim.Adam(model.parameters(), lr=1e-5) if args.train else None

    g = torch.Generator(device="cpu").manual_seed(100 + rank)  # every rank has its own shard of the data
    im_gray = torch.rand((B, S, S, 1), generator=g).to(dev)
    labels = torch.as_tensor(synth.sample_params_batch(B, im_size=S, n_shape=face.ndim_shape, n_exp=face.ndim_exp,
                                                       beta=0.7, seed=200 + rank), device=dev)
    label_depth = torch.rand((B, S, S, 1), generator=g).to(dev)

    def step():
        params = coarse(im_gray)
        out = {"params": params}
        if args.train or fine is not None:
            depth = coarse.depth(im_gray, params)
            out["depth"] = depth
            if fine is not None:
                out["fine"] = fine(im_gray, depth)
        if args.train:
            loss = 1e-3 * torch.nn.functional.mse_loss(params[:, :7], labels[:, :7]) \
                + 1e-6 * torch.nn.functional.mse_loss(params[:, 7:], labels[:, 7:]) \
                + 1e-3 * torch.nn.functional.mse_loss(out.get("fine", out["depth"]).clamp(0, 1), label_depth)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            out["loss"] = loss.detach()
        return out

    ctx = torch.enable_grad() if args.train else torch.no_grad()
    with ctx:
        if not args.train:
            model.eval()
        step()
        torch.cuda.synchronize(dev)
        dist_u.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize(dev)
        dt = dist_u.max_over_ranks(time.perf_counter() - t0, device=dev)
    if rank == 0:
        print(json.dumps({"config": "coarse_loop", "n_gpus": world, "faces_per_gpu": B, "im_size": S, "nIter": args.nIter,
                          "train": args.train, "fine": args.fine, "steps": args.steps,
                          "faces_per_s": world * B * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                          "loss": float(out["loss"]) if "loss" in out else None,
                          "params_finite": bool(torch.isfinite(out["params"]).all())}))
    dist_u.finalize()


if __name__ == "__main__":
    main()
