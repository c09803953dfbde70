#!/usr/bin/env python3
"""Development probe: emit-kernel variants (FR_EMIT_PERSIST = persistent item loop with next-item triangle prefetch,
FR_EMIT_CAP = register budget) -- wall time of the kernel alone (back to back) and of the whole step; outputs compared."""
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
    os.environ["FR_EMIT_PERSIST"] = "0"
    os.environ["FR_EMIT_CAP"] = "1"
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

    for persist in ("1", "0"):
        for cap in ("8", "1"):
            os.environ["FR_EMIT_PERSIST"], os.environ["FR_EMIT_CAP"] = persist, cap
            outs = plan.step()
            torch.cuda.synchronize()
            same = all(torch.equal(a, b) for a, b in zip(outs, ref))
            r = [(round(wall(lambda: plan.render_phase(1)), 1), round(wall(plan.step), 1)) for _ in range(3)]
            print("persist=%s cap=%s identical=%s  (emit alone us, step us): %s" % (persist, cap, same, r), flush=True)


if __name__ == "__main__":
    main()
