#!/usr/bin/env python3
"""Probe: what the resolve role costs the fused launch -- resolve role a no-op (FR_RESOLVE_OPT bit 8), resolve role without its
plane writer (bit 16), full.  Needs a library built with -DFR_FUSED_PROBE (add it to _lib.HIPCC_FLAGS for the session):
the product build carries no probe hooks.  Record: profiles/round4_probes/r4b_overlap_probes_summary.json."""
import importlib, json, os, sys, time
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
    pp = pipe.PipelinedPlan(net, B, S, S)
    P = torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456), device=dev)
    pp.params.copy_(P)
    pp._run(8 | 1, 0, 1)
    pp._run(8 | 1, 1, 0)
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

    for rnd in range(2):
        r = {}
        for order in (0, 1):
            setopt("FR_FUSED_ORDER", order)
            for mode, name in ((2, "full"), (2 | 16, "no_plane_writes"), (2 | 8, "resolve_noop")):
                setopt("FR_RESOLVE_OPT", mode)
                r["order%d_%s" % (order, name)] = timed(lambda: pp._run(3, 0, 1))
        setopt("FR_FUSED_ALONE", 1)
        for mode, name in ((2, "full"), (2 | 16, "no_plane_writes"), (2 | 8, "resolve_noop")):
            setopt("FR_RESOLVE_OPT", mode)
            r["lean_resolve_alone_%s" % name] = timed(lambda: pp._run(2, 0, 1))
        r["emit_role_alone"] = timed(lambda: pp._run(1, 0, 1))
        setopt("FR_FUSED_ALONE", 0)
        setopt("FR_RESOLVE_OPT", 2)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
