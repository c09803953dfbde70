#!/usr/bin/env python3
"""Development probe: throughput of decode(k+1) || render(k) on two HIP streams with double buffering."""
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
    # serial
    t0 = time.perf_counter()
    for k in range(K):
        plans[k & 1].step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("serial      %.1f us/step" % ((t1 - t0) / K * 1e6))
    sd, sr = torch.cuda.Stream(), torch.cuda.Stream()
    ev_dec = [torch.cuda.Event() for _ in range(2)]
    ev_ren = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        p = plans[k & 1]
        with torch.cuda.stream(sd):
            sd.wait_event(ev_ren[k & 1])      # the render that last read this vertex buffer has finished
            p.decode()
            ev_dec[k & 1].record(sd)
        with torch.cuda.stream(sr):
            sr.wait_event(ev_dec[k & 1])
            p.render()
            ev_ren[k & 1].record(sr)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("2 streams   %.1f us/step" % ((t1 - t0) / K * 1e6))


if __name__ == "__main__":
    main()
