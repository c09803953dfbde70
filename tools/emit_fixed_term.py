#!/usr/bin/env python3
"""Development probe: raster_emit_kernel at 1 ... 64 faces, to be run under `rocprofv3 --kernel-trace`:
tools/emit_fixed_term_report.py groups the trace's dispatches by grid size and fits the kernel's FIXED term.  Each batch size
gets its own plan (bench assets, bench sampler); the decode runs once, then the emit phase alone, 40 times."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 10, 12, 16, 20, 24, 32, 40, 48, 56, 64]
    for B in sizes:
        net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200, device=dev)
        plan = pipe.DecodeRenderPlan(net, B, 200, 200)
        plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=200, beta=0.7, seed=3456), device=dev))
        plan.step()
        torch.cuda.synchronize()
        for _ in range(40):
            plan.render_phase(1)
            torch.cuda.synchronize()     # one launch at a time: every dispatch starts on an idle chip, like the serial step's
        del plan, net
    print("done", sizes)


if __name__ == "__main__":
    main()
