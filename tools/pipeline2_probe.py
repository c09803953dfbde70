#!/usr/bin/env python3
"""Development probe: across-batch software pipeline  decode(k) ; [emit(k) || resolve(k-1)]  on two streams with two
plans (double-buffered vertices / workspace / planes), against the plain one-stream step.  Outputs compared bit for bit."""
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
    refs = [[o.clone() for o in p.step()] for p in plans]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def plain():
        for k in range(K):
            plans[k % 2].step()

    def pipelined(variant):
        ev_dec = [torch.cuda.Event() for _ in range(2)]
        ev_res = [torch.cuda.Event() for _ in range(2)]
        ev_emit = [torch.cuda.Event() for _ in range(2)]
        for k in range(K):
            p = plans[k % 2]
            with torch.cuda.stream(sa):
                if k >= 2:
                    sa.wait_event(ev_res[k % 2])          # this plan's planes / workspace are free again
                p.decode()
                ev_dec[k % 2].record(sa)
            if k >= 1 and variant == 0:
                with torch.cuda.stream(sb):                # resolve(k-1) beside emit(k): both start when decode(k) is done
                    sb.wait_event(ev_dec[k % 2])
                    plans[(k - 1) % 2].render_phase(2)
                    ev_res[(k - 1) % 2].record(sb)
            with torch.cuda.stream(sa):
                p.render_phase(1)
                ev_emit[k % 2].record(sa)
            if k >= 1 and variant == 1:
                with torch.cuda.stream(sb):
                    sb.wait_event(ev_dec[k % 2])
                    plans[(k - 1) % 2].render_phase(2)
                    ev_res[(k - 1) % 2].record(sb)
        with torch.cuda.stream(sb):
            sb.wait_event(ev_emit[(K - 1) % 2])
            plans[(K - 1) % 2].render_phase(2)
        torch.cuda.current_stream().wait_stream(sa)
        torch.cuda.current_stream().wait_stream(sb)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    for rnd in range(3):
        r = {"plain_one_stream": round(timed(plain), 1), "pipelined_v0": round(timed(lambda: pipelined(0)), 1),
             "pipelined_v1": round(timed(lambda: pipelined(1)), 1)}
        ok = all(torch.equal(a, b) for p, ref in zip(plans, refs) for a, b in zip(p.outputs(), ref))
        print(r, "outputs identical:", ok, flush=True)


if __name__ == "__main__":
    main()
