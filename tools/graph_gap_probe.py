#!/usr/bin/env python3
"""Probe for DESIGN 4.5: why is hipGraph replay of the three-kernel step slower than three direct launches?
Runs 40 eager steps, then 40 graph replays (serial plan), under `rocprofv3 --kernel-trace` (tools/graph_gap_report.py reads the
trace): gaps between the kernels INSIDE a step and BETWEEN steps, eager vs replay."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    B, S, K = 64, 200, 40
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    dev = torch.device("cuda:0")
    net = netm.FaceRecNet(mesh_data=synth.make_assets(), batch_size=B, im_size=S, device=dev)
    plan = pipe.DecodeRenderPlan(net, B, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456), device=dev))
    for _ in range(10):
        plan.step()
    plan.capture()
    for _ in range(10):
        plan.replay()
    torch.cuda.synchronize()
    time.sleep(0.05)
    t0 = time.perf_counter()
    for _ in range(K):
        plan.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    time.sleep(0.05)
    t2 = time.perf_counter()
    for _ in range(K):
        plan.replay()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print({"eager_us_per_step": round((t1 - t0) / K * 1e6, 1), "replay_us_per_step": round((t3 - t2) / K * 1e6, 1)})


if __name__ == "__main__":
    main()
