#!/usr/bin/env python3
"""Probe: upper bound of running decode, emit and resolve of THREE different batches concurrently (three streams, no
dependencies between them, K launches each) against the same launches one after the other."""
import importlib, json, os, sys, time
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
    plans = [pipe.DecodeRenderPlan(net, B, S, S) for _ in range(3)]
    for i, p in enumerate(plans):
        p.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456 + i), device=dev))
        p.step()
    torch.cuda.synchronize()
    ss = [torch.cuda.Stream() for _ in range(3)]

    def timed(fn):
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(K)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / K * 1e6, 1)

    def seq(n):
        with torch.cuda.stream(ss[0]):
            for _ in range(n):
                plans[0].decode(); plans[1].render_phase(1); plans[2].render_phase(2)

    def three(n):
        for _ in range(n):
            with torch.cuda.stream(ss[0]):
                plans[0].decode()
            with torch.cuda.stream(ss[1]):
                plans[1].render_phase(1)
            with torch.cuda.stream(ss[2]):
                plans[2].render_phase(2)

    def de_then_r(n):   # decode || emit, resolve on the emit stream behind the emit
        for _ in range(n):
            with torch.cuda.stream(ss[0]):
                plans[0].decode()
            with torch.cuda.stream(ss[1]):
                plans[1].render_phase(1); plans[2].render_phase(2)

    for rnd in range(3):
        print(json.dumps({"one_stream_decode_emit_resolve": timed(seq), "three_streams": timed(three),
                          "decode_stream_and_render_stream": timed(de_then_r)}), flush=True)


if __name__ == "__main__":
    main()
