#!/usr/bin/env python3
"""Development probe: time of fr_decode_3dmm_backward (three kernels) through the autograd surface, full-size basis."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


synth, netm = pkg("utils.synth"), pkg("nets.network")
A = synth.make_assets()
for B in [int(x) for x in os.environ.get("BWD_B", "16,32,64").split(",")]:
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200, device="cuda:0")
    P = torch.as_tensor(synth.sample_params_batch(B, im_size=200, beta=0.7), device="cuda:0").requires_grad_(True)
    V = net.vertices_transform(P)
    G = torch.randn_like(V)
    def bwd():
        P.grad = None
        V.backward(G, retain_graph=True)
    for _ in range(5):
        bwd()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    # host side of one backward through the autograd surface: wall clock per call with the GPU never waited for (the queue runs
    # full: what the host spends issuing it), beside the device figure above
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        bwd()
    host_us = (time.perf_counter() - t0) / 30 * 1e6
    torch.cuda.synchronize()
    print("B=%d decode backward: %s us   (host issue time per call %.1f us)" % (B, " ".join("%.1f" % t for t in ts), host_us), flush=True)
