#!/usr/bin/env python3
"""Development probe: emit-kernel A/B knobs in one process, interleaved rounds (FR_EMIT_FILTER: certified fp32 inside test)."""
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
    os.environ["FR_EMIT_FILTER"] = "0"
    ref = [o.clone() for o in plan.step()]
    torch.cuda.synchronize()

    def wall(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    res = {"1": [], "3": []}
    for rnd in range(5):
        for v in ("1", "3"):
            os.environ["FR_EMIT_FILTER"] = v
            outs = plan.step()
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(outs, ref))
            res[v].append((round(wall(lambda: plan.render_phase(1)), 1), round(wall(plan.step), 1)))
    for v in ("1", "3"):
        print("FR_EMIT_FILTER=%s (emit alone us, step us): %s" % (v, res[v]), flush=True)


if __name__ == "__main__":
    main()
