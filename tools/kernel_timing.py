#!/usr/bin/env python3
"""Per-kernel timing helper (development tool): times decode alone, render alone and the pipeline with HIP events.
Usage: python tools/kernel_timing.py [--batch 64] [--iters 50]"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--im-size", type=int, default=200)
    args = ap.parse_args()
    S = args.im_size
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=args.batch, im_size=S, device="cuda:0")
    plan = pipe.DecodeRenderPlan(net, args.batch, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(args.batch, im_size=S, beta=0.7), device="cuda:0"))
    plan.step()
    print("decode only   %.1f us" % timeit(plan.decode, args.iters))
    print("render only   %.1f us" % timeit(plan.render, args.iters))
    print("decode+render %.1f us" % timeit(plan.step, args.iters))
    plan.capture()
    print("graph replay  %.1f us" % timeit(plan.replay, args.iters))
    # backward legs (autograd surface): depth -> vertex z -> params
    ops = pkg("rendering_layer.ops")
    P = plan.params.clone().requires_grad_(True)
    V = net.vertices_transform(P)
    outs = ops.render_depth(V, net.tri, net.vertex_code, torch.zeros((args.batch, S, S, 3), device="cuda:0"))
    g = torch.ones_like(outs[0])

    def bwd():
        P.grad = None
        outs[0].backward(g, retain_graph=True)
    print("render+decode backward %.1f us" % timeit(bwd, args.iters))


if __name__ == "__main__":
    main()
