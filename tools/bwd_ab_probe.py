#!/usr/bin/env python3
"""Round-5 development probe: the two forms of the fused decode backward's d f in ONE process, interleaved rounds, raw C ABI
(no autograd): 'vproj' = fr_decode_3dmm_backward_packed (reads the forward's output), 'mu' = fr_decode_3dmm_backward_packed_mu
(d f from mu and the coefficient gradients: no forward output).  (profiles/round5_probes/r5d also holds the LDS-staged
variant of the mu form, which was measured with this script and removed.)"""
import ctypes
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

h = importlib.import_module("3dfacerecon_amd._lib")
synth = importlib.import_module("3dfacerecon_amd.utils.synth")
netm = importlib.import_module("3dfacerecon_amd.nets.network")
L = h.lib()
A = synth.make_assets()
dev = torch.device("cuda:0")
out = {}
for B in (64, 32, 16):
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200, device=dev)
    P = torch.as_tensor(synth.sample_params_batch(B, im_size=200, beta=0.7), device=dev)
    V = net.vertices_transform(P).detach()
    G = torch.randn_like(V)
    nws = L.fr_decode_backward_workspace_bytes(B, net.nvert, 199, 29)
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    gp = torch.empty_like(P)
    img = net._basis.image_t()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(kind):
        if kind == "vproj":
            return L.fr_decode_3dmm_backward_packed(h.ptr(G), h.ptr(P), h.ptr(V), h.ptr(img), None, B, net.nvert, 199, 29, 200.0,
                                                    h.ptr(gp), h.ptr(ws), nws, st)
        return L.fr_decode_3dmm_backward_packed_mu(h.ptr(G), h.ptr(P), h.ptr(net.mu), h.ptr(img), None, B, net.nvert, 199, 29, 200.0,
                                                   h.ptr(gp), h.ptr(ws), nws, st)
    res = {k: [] for k in ("vproj", "mu")}
    for k in res:
        for _ in range(5):
            assert run(k) == 0
    torch.cuda.synchronize()
    for rnd in range(4):
        for k in res:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run(k)
            e1.record()
            torch.cuda.synchronize()
            res[k].append(round(e0.elapsed_time(e1) / 50 * 1e3, 2))
    out["B=%d" % B] = res
    print("B=%d" % B, res, flush=True)
print(json.dumps({"what": "us per backward (fused kernel + reduce, both launches), 50 calls per figure, four interleaved rounds, one process", "results": out}))
