#!/usr/bin/env python3
"""Development probe: same-process, interleaved A/B of the decode backward's launcher knobs (fr_set_option: FR_BWD_CHUNKS) through the autograd surface; K backward calls per figure, four rounds; results identical across settings."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    synth, netm, host = pkg("utils.synth"), pkg("nets.network"), pkg("_lib")
    L = host.lib()
    A = synth.make_assets()
    K = 40
    out = {}
    for B in (32, 64):
        net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200, device="cuda:0")
        P = torch.as_tensor(synth.sample_params_batch(B, im_size=200, beta=0.7), device="cuda:0").requires_grad_(True)
        V = net.vertices_transform(P)
        G = torch.randn_like(V)

        def bwd():
            P.grad = None
            V.backward(G, retain_graph=True)
        ref = None
        res = {}
        for rnd in range(4):
            for chunks, ring in ((512, 3), (256, 3), (384, 3), (128, 3)):
                assert L.fr_set_option(b"FR_BWD_CHUNKS", chunks) == 0
                for _ in range(5):
                    bwd()
                torch.cuda.synchronize()
                g = P.grad.clone()
                ref = g if ref is None else ref
                same = bool(torch.equal(g, ref))
                t0 = time.perf_counter()
                for _ in range(K):
                    bwd()
                torch.cuda.synchronize()
                res.setdefault("chunks=%s ring=%d" % (chunks or "cus", ring), []).append((round((time.perf_counter() - t0) / K * 1e6, 1), same))
        out["B=%d" % B] = res
        print(json.dumps({"B": B, **res}), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/decode_bwd_ab.json", "w"), indent=1)


if __name__ == "__main__":
    main()
