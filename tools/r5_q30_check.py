"""Round-5 development check: the Q30 decode with 7 / 5 / 4 levels under every schedule against the oracle (bit for bit)."""
import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
h = importlib.import_module("3dfacerecon_amd._lib")
synth = importlib.import_module("3dfacerecon_amd.utils.synth")
netm = importlib.import_module("3dfacerecon_amd.nets.network")
ok = True
full = synth.make_assets()
for (assets, name, Bs) in ((full, "full", (8, 64)), (synth.make_assets(13, 17, 199, 29, patch=None, seed_basis=5), "small199", (17, 5, 133, 48)),
                           (synth.make_assets(15, 16, 200, 17, patch=None, seed_basis=6), "generic217", (40,))):
    for B in Bs:
        ns, ne = assets["pc_shape"].shape[1], assets["pc_exp"].shape[1]
        rs = np.random.RandomState(B)
        P = np.zeros((B, 7 + ns + ne), np.float32)
        P[:, 0:3] = rs.uniform(-1.5, 1.5, (B, 3)); P[:, 3:5] = rs.uniform(0, 200, (B, 2)); P[:, 6] = rs.uniform(0, 1e-3, B)
        P[:, 7:7 + ns] = rs.uniform(0, 1e4, (B, ns)); P[:, 7 + ns:] = rs.uniform(-1.5, 1.5, (B, ne))
        if B > 3:   # a face of zeros, one with a non-finite parameter, one with huge / tiny ones
            P[1, 7:] = 0
            P[2, 9] = np.inf
            P[3, 7:12] = (3e38, -1e-40, 1e-30, 5e-39, 0)
        net = netm.FaceRecNet(mesh_data=assets, batch_size=B, im_size=200, device=torch.device("cuda:0"))
        R = O.rotation_matrix_batch(P[:, :3])
        nchk = min(B, 8) if name == "full" else B
        for lv in (7, 5, 4):
            want = O.decode_3dmm_q30(P[:nchk], assets["mu"], assets["pc_shape"], assets["pc_exp"], 200.0, R=R[:nchk], levels=lv)
            h.set_decode_arith(h.DECODE_ARITH_Q30, lv)
            for sched in (0, 1):
                h.set_option("FR_Q30_SCHED", sched)
                got = net.vertices_transform(torch.as_tensor(P, device="cuda:0"), R=torch.as_tensor(R, device="cuda:0"))
                torch.cuda.synchronize()
                g = got[:nchk].cpu().numpy()
                same = np.array_equal(g, want, equal_nan=True)
                ok &= same
                print(name, "B", B, "levels", lv, "sched", sched, "bit-exact" if same else "MISMATCH %d of %d, max |d| %g" % ((g != want).sum(), g.size, np.nanmax(np.abs(g - want))), flush=True)
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
