#!/usr/bin/env python3
"""Development probe: resolve(batch k) on a side stream while emit(batch k+1) runs -- do the two kernels share CUs
productively?  Run with FR_RESOLVE_BLOCK=256 FR_RENDER_ROWS=13 so that a resolve workgroup is small enough to sit next
to emit workgroups."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    B, K = 64, 200
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200, device="cuda:0")
    plans = [pipe.DecodeRenderPlan(net, B, 200, 200) for _ in range(2)]
    P = torch.as_tensor(synth.sample_params_batch(B, beta=0.7), device="cuda:0")
    for p in plans:
        p.params.copy_(P)
        p.step()
    torch.cuda.synchronize()
    ref = [t.clone() for t in plans[0].outputs()]

    def timed(fn):
        fn(10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(K)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / K * 1e3

    def serial(n):
        for k in range(n):
            p = plans[k & 1]
            p.render_phase(1)
            p.render_phase(2)
    print("serial emit+resolve      %.1f us/batch" % timed(serial))
    s2 = torch.cuda.Stream()
    ev_e = [torch.cuda.Event() for _ in range(2)]
    ev_r = [torch.cuda.Event() for _ in range(2)]

    def piped(n):
        s1 = torch.cuda.current_stream()
        for k in range(n):
            p = plans[k & 1]
            s1.wait_event(ev_r[k & 1])      # the resolve that last read this workspace has finished
            p.render_phase(1)
            ev_e[k & 1].record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(ev_e[k & 1])
                p.render_phase(2)
                ev_r[k & 1].record(s2)
        s1.wait_stream(s2)
    print("resolve(k) || emit(k+1)  %.1f us/batch" % timed(piped))
    torch.cuda.synchronize()
    print("outputs identical:", all(torch.equal(a, b) for a, b in zip(ref, plans[0].outputs())))


if __name__ == "__main__":
    main()
