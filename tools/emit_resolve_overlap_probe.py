#!/usr/bin/env python3
"""Development probe: how much would emit (VALU-bound) and resolve (store-bound) gain from running CONCURRENTLY?
Upper bound, no dependency between them: plan A's emit kernel on one stream, plan B's resolve kernel (B's hit records were
produced once, before) on another, K launches each; compared with the same launches one after the other on one stream.
If the concurrent time is near max(emit, resolve) a fused / dependency-tracked schedule could pay; if it is near the sum,
the two kernels do not overlap on this hardware and no schedule will make them."""
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
    plans = [pipe.DecodeRenderPlan(net, B, S, S) for _ in range(2)]
    for i, p in enumerate(plans):
        p.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456 + i), device=dev))
        p.step()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    def emit_only():
        with torch.cuda.stream(s1):
            for _ in range(K):
                plans[0].render_phase(1)

    def resolve_only():
        with torch.cuda.stream(s2):
            for _ in range(K):
                plans[1].render_phase(2)

    def sequential():
        with torch.cuda.stream(s1):
            for _ in range(K):
                plans[0].render_phase(1)
                plans[1].render_phase(2)

    def concurrent():
        for _ in range(K):
            with torch.cuda.stream(s1):
                plans[0].render_phase(1)
            with torch.cuda.stream(s2):
                plans[1].render_phase(2)

    def decode_only():
        with torch.cuda.stream(s1):
            for _ in range(K):
                plans[0].decode()

    def decode_resolve_concurrent():
        for _ in range(K):
            with torch.cuda.stream(s1):
                plans[0].decode()
            with torch.cuda.stream(s2):
                plans[1].render_phase(2)

    def decode_emit_concurrent():
        for _ in range(K):
            with torch.cuda.stream(s1):
                plans[0].decode()
            with torch.cuda.stream(s2):
                plans[1].render_phase(1)

    for rnd in range(3):
        print({"emit_alone": round(timed(emit_only), 1), "resolve_alone": round(timed(resolve_only), 1),
               "emit_then_resolve_one_stream": round(timed(sequential), 1),
               "emit_and_resolve_two_streams": round(timed(concurrent), 1),
               "decode_alone": round(timed(decode_only), 1),
               "decode_and_resolve_two_streams": round(timed(decode_resolve_concurrent), 1),
               "decode_and_emit_two_streams": round(timed(decode_emit_concurrent), 1)}, flush=True)


if __name__ == "__main__":
    main()
