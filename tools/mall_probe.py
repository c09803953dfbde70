#!/usr/bin/env python3
"""Development probe: how much of the decode's basis stream (153 MB, re-read every step) does the 256 MiB Infinity Cache
keep between two steps?  decode back to back runs ~10 us faster than decode inside the pipeline; this measures the decode
time with Y MB of unrelated write traffic between two launches, and the pipeline with the non-temporal knobs
(FR_RESOLVE_NT: plane stores, FR_DECODE_NT: basis loads)."""
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
    B, S, K = 64, 200, 100
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plan = pipe.DecodeRenderPlan(net, B, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7), device=dev))
    plan.step()
    torch.cuda.synchronize()
    junk = torch.empty((320 << 20,), dtype=torch.uint8, device=dev)

    def ev_time(fn_between):
        es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for e0, e1 in es:
            fn_between()
            e0.record()
            plan.decode()
            e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) for e0, e1 in es[5:])
        return 1e3 * t[len(t) // 2]

    for y in (0, 16, 32, 64, 96, 128, 160, 192, 256, 320):
        n = y << 20
        print("decode after %3d MB of fill traffic: %.1f us" % (y, ev_time((lambda: junk[:n].fill_(1)) if n else (lambda: None))),
              flush=True)
    # read traffic instead of write traffic
    for y in (64, 128, 192):
        n = y << 20
        print("decode after %3d MB of read traffic: %.1f us" % (y, ev_time(lambda: junk[:n].sum())), flush=True)

    def wall():
        for _ in range(10):
            plan.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            plan.decode()
            plan.render()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    def phases():
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(40)]
        for e in ev:
            e[0].record(); plan.decode(); e[1].record(); plan.render_phase(1); e[2].record(); plan.render_phase(2); e[3].record()
        torch.cuda.synchronize()
        med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
        return [round(1e3 * med([e[i].elapsed_time(e[i + 1]) for e in ev[3:]]), 1) for i in range(3)]

    host = pkg("_lib")   # (fr_set_option; the non-temporal plane stores of round 2's FR_RESOLVE_NT experiment are gone)
    for env in ({}, {"FR_DECODE_NT": 0}, {}):
        with host.options(**env):
            w = [wall() for _ in range(3)]
            print("knobs %s: step wall us %s, decode/emit/resolve (events) %s" % (env, [round(x, 1) for x in w], phases()), flush=True)


if __name__ == "__main__":
    main()
