#!/usr/bin/env python3
"""Development probe: can the decode of batch k+1 run BESIDE the render (emit + resolve) of batch k on one MI355X?
decode is MFMA + HBM-stream bound, emit is VALU / L2-gather bound, resolve is LDS + HBM-write bound.  The 16-wave decode
workgroup owns the whole register file of its CU; the 8-wave form (FR_DECODE_WAVES=8, <= 128 VGPRs) leaves half of it.
Times: each kernel alone, then decode on one stream and render on another (no dependencies), then the dependent pipeline
(render k waits for decode k; decode k+1 runs beside render k; vertex buffers double-buffered)."""
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
    P = torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7), device=dev)
    plans = [pipe.DecodeRenderPlan(net, B, S, S) for _ in range(2)]
    for p in plans:
        p.params.copy_(P)
        p.step()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def wall(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    def dec_alone():
        with torch.cuda.stream(s1):
            for _ in range(K):
                plans[0].decode()

    def ren_alone():
        with torch.cuda.stream(s2):
            for _ in range(K):
                plans[1].render()

    def both_independent():
        for _ in range(K):
            with torch.cuda.stream(s1):
                plans[0].decode()
            with torch.cuda.stream(s2):
                plans[1].render()

    def sequential():
        with torch.cuda.stream(s1):
            for _ in range(K):
                plans[0].decode()
                plans[0].render()

    ev_dec = [torch.cuda.Event() for _ in range(2)]
    ev_emit = [torch.cuda.Event() for _ in range(2)]

    def pipelined():
        # vertex buffers: plans[k % 2].vertex_proj; one shared render workspace would also do, each plan has its own
        for k in range(K):
            p = plans[k % 2]
            with torch.cuda.stream(s1):
                if k >= 2:
                    s1.wait_event(ev_emit[k % 2])      # emit(k-2) has finished reading this vertex buffer
                p.decode()
                ev_dec[k % 2].record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(ev_dec[k % 2])
                p.render_phase(1)
                ev_emit[k % 2].record(s2)
                p.render_phase(2)

    for waves in (16, 8):
        pkg("_lib").set_option("FR_DECODE_WAVES", waves)
        for _ in range(2):
            r = {"decode_alone": wall(dec_alone), "render_alone": wall(ren_alone), "sequential": wall(sequential),
                 "two_streams_independent": wall(both_independent), "pipelined_dependent": wall(pipelined)}
        print("FR_DECODE_WAVES=%s  us per step: %s" % (waves, {k: round(v, 1) for k, v in r.items()}), flush=True)
    # correctness of the pipelined schedule: outputs of both plans equal a plain step
    ref = [o.clone() for o in plans[0].step()]
    pipelined()
    torch.cuda.synchronize()
    for p in plans:
        assert all(torch.equal(a, b) for a, b in zip(p.outputs(), ref))
    print("pipelined outputs identical to the sequential step")


if __name__ == "__main__":
    main()
