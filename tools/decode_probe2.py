#!/usr/bin/env python3
"""Development probe: decode variants inside the pipeline (FR_DECODE_NBW item width, FR_DECODE_NT basis-load policy):
wall time per step and event-bracketed kernel times; outputs compared bit for bit."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    B, S, K = 64, 200, 200
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plan = pipe.DecodeRenderPlan(net, B, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7), device=dev))
    plan.step()
    torch.cuda.synchronize()
    ref = plan.vertex_proj.clone()

    def wall(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    def phases():
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(40)]
        for e in ev:
            e[0].record(); plan.decode(); e[1].record(); plan.render_phase(1); e[2].record(); plan.render_phase(2); e[3].record()
        torch.cuda.synchronize()
        med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
        return [round(1e3 * med([e[i].elapsed_time(e[i + 1]) for e in ev[3:]]), 1) for i in range(3)]

    host = pkg("_lib")   # (the launcher knobs are set through fr_set_option: the environment is read once per process)
    for env in ({}, {"FR_DECODE_NBW": 1}, {"FR_DECODE_NT": 0}, {}):
        with host.options(**env):
            plan.decode()
            torch.cuda.synchronize()
            same = torch.equal(plan.vertex_proj, ref)
            print("knobs %s identical=%s: step us %s decode-alone us %s, decode/emit/resolve (events) %s" % (
                env, same, [round(wall(plan.step), 1) for _ in range(3)], round(wall(plan.decode), 1), phases()), flush=True)


if __name__ == "__main__":
    main()
