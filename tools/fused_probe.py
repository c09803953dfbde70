#!/usr/bin/env python3
"""Development probe for the fused emit || resolve launch (render_fused_kernel): back-to-back launches, K each, in ONE process:
the two roles alone (stand-alone kernels and through the fused kernel), one after the other, and fused with the three block
orders (FR_FUSED_ORDER).  All on the pipelined route's 8-row strips; the serial plan's 10-row kernels beside them."""
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
    B, S, K = 64, 200, 100
    synth, netm, pipe, host = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline"), pkg("_lib")
    L = host.lib()
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    sp = pipe.DecodeRenderPlan(net, B, S, S)
    pp = pipe.PipelinedPlan(net, B, S, S)
    P = torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456), device=dev)
    sp.step(P)
    pp.params.copy_(P)
    pp._run(8 | 1, 0, 1)
    pp._run(8 | 1, 1, 0)      # both workspaces hold records, both vertex buffers vertices
    torch.cuda.synchronize()

    def setopt(name, v):
        assert L.fr_set_option(name.encode(), v) == 0

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / K * 1e6, 2)

    out = []
    for rnd in range(3):
        r = {}
        r["serial_emit_10row"] = timed(lambda: sp.render_phase(1))
        r["serial_resolve_10row"] = timed(lambda: sp.render_phase(2))
        r["serial_emit_then_resolve"] = timed(lambda: sp.render_phase(3))
        setopt("FR_FUSED_ALONE", 0)
        r["emit_8row"] = timed(lambda: pp._run(1, 0, 1))
        r["resolve_8row_full_kernel"] = timed(lambda: pp._run(2, 0, 1))
        r["emit_then_resolve_8row"] = timed(lambda: (pp._run(1, 0, 1), pp._run(2, 0, 1)))
        setopt("FR_FUSED_ALONE", 1)
        r["emit_role_alone_in_fused_kernel"] = timed(lambda: pp._run(1, 0, 1))
        r["lean_resolve_role_alone_in_fused_kernel"] = timed(lambda: pp._run(2, 0, 1))
        setopt("FR_FUSED_ALONE", 0)
        for order in (0, 1, 2):
            setopt("FR_FUSED_ORDER", order)
            r["fused_order%d" % order] = timed(lambda: pp._run(3, 0, 1))
        setopt("FR_FUSED_ORDER", 0)
        r["decode"] = timed(lambda: pp._run(8, 0, 1))
        r["decode_then_fused"] = timed(lambda: pp._run(8 | 3, 0, 1))
        r["serial_step"] = timed(lambda: sp.step())
        print(json.dumps(r), flush=True)
        out.append(r)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/fused_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
