"""Development probe: Q30 (int8-MFMA) decode vs its CPU spec, bit for bit, on several basis shapes; then timing of both
arithmetic modes at the bench shape."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

L = importlib.import_module("3dfacerecon_amd._lib")
netm = importlib.import_module("3dfacerecon_amd.nets.network")
synth = importlib.import_module("3dfacerecon_amd.utils.synth")
lib = L.lib()


def rand_params(rs, B, ns, ne, im):
    P = np.zeros((B, 7 + ns + ne), np.float32)
    P[:, 0:3] = rs.uniform(-1.5, 1.5, (B, 3))
    P[:, 3:5] = rs.uniform(0, im, (B, 2))
    P[:, 5] = rs.uniform(-1, 1, B)
    P[:, 6] = rs.uniform(0, 1e-3, B)
    P[:, 7:7 + ns] = rs.uniform(0, 1e4, (B, ns))
    P[:, 7 + ns:] = rs.uniform(-1.5, 1.5, (B, ne))
    return P


def dec(net, P, R):
    out = net.vertices_transform(torch.as_tensor(P, device="cuda:0"), R=torch.as_tensor(R, device="cuda:0"))
    torch.cuda.synchronize()
    return out.cpu().numpy()


ok = True
for gu, gv, ns, ne, B in [(20, 24, 9, 5, 3), (7, 9, 1, 1, 1), (13, 17, 199, 29, 17), (12, 31, 199, 29, 5), (10, 23, 199, 29, 133),
                          (15, 16, 200, 17, 40), (11, 19, 33, 16, 64), (9, 10, 40, 7, 65), (6, 8, 256, 0, 20), (5, 7, 300, 100, 33),
                          (6, 8, 0, 0, 4), (6, 9, 64, 0, 16)]:
    A = synth.make_assets(gu, gv, ns, ne, patch=None, seed_basis=gu * gv)
    P = rand_params(np.random.RandomState(B), B, ns, ne, 200)
    R = orc.rotation_matrix_batch(P[:, :3])
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
    for mode, q30 in ((0, True), (1, False)):
        assert lib.fr_decode_set_arith(mode) == 0
        got = dec(net, P, R)
        want = orc.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R, q30=q30)
        eq = np.array_equal(got, want)
        ok &= eq
        print("shape", (gu * gv, ns, ne, B), "mode", "q30" if q30 else "f32", "bit-exact" if eq else
              "MISMATCH frac=%.4f maxabs=%.3e" % ((got != want).mean(), np.nanmax(np.abs(got - want))), flush=True)
print("ALL OK" if ok else "FAILURES")

A = synth.make_assets()
B = 64
P = synth.sample_params_batch(B, im_size=200, beta=0.7, seed=3456)
R = orc.rotation_matrix_batch(P[:, :3])
net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200, device=torch.device("cuda:0"))
for mode in (0, 1):
    lib.fr_decode_set_arith(mode)
    got = dec(net, P, R)
    want = orc.decode_3dmm(P[:1], A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R[:1], q30=(mode == 0))
    print("full size mode", mode, "face 0 bit-exact:", np.array_equal(got[:1], want))
    p = torch.as_tensor(P, device="cuda:0")
    for _ in range(5):
        net.vertices_transform(p)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            net.vertices_transform(p)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    print("mode", mode, "decode back-to-back us:", ["%.1f" % t for t in ts], flush=True)
