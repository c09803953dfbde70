#!/usr/bin/env python3
"""Development probe: fr_render_depth_backward alone (B = 64 and 32, 200x200, full-size mesh), wall per launch."""
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
    S, K = 200, 100
    synth, netm, pipe, ops = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline"), pkg("rendering_layer.ops")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    for B, dbg in ((64, 0), (32, 0)):
        net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
        plan = pipe.DecodeRenderPlan(net, B, S, S)
        plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7), device=dev))
        plan.step()
        g = torch.randn((B, S, S, 1), device=dev)
        img = torch.zeros((B, S, S, 3), device=dev)

        def run():
            return ops.render_depth_grad(g, plan.vertex_proj, net.tri, plan.depth, plan.tri_ind, img)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            run()
        torch.cuda.synchronize()
        print("B=%d dbg=%d render backward: %.1f us per launch (incl. pack + output allocation)" % (B, dbg, (time.perf_counter() - t0) / K * 1e6),
              flush=True)


if __name__ == "__main__":
    main()
