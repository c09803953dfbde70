#!/usr/bin/env python3
"""Development probe: ablated builds of raster_emit_kernel (tools/emit_probe.hip, fr_probe_emit_ablate) on the bench workload --
the kernel returns at cut point N, so the differences are what each stage ADDS to the kernel's duration (throughput, not a
wave's latency): 1 = empty workgroups, 2 = table loads, 3 = + eighteen gathers, 5 = + bbox / pre-cull + compaction,
6 = + phase B, 7 = + bucket scan / barrier 3, 0 = whole kernel.  Back to back, K launches each, three interleaved rounds."""
import ctypes, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main(B=64):
    S, K = 200, 100
    synth, netm, pipe, host = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline"), pkg("_lib")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plan = pipe.DecodeRenderPlan(net, B, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456), device=dev))
    plan.step()
    torch.cuda.synchronize()
    P = ctypes.CDLL(os.path.join(ROOT, "tools", "libemit_probe.so"))
    vp, i = ctypes.c_void_p, ctypes.c_int
    P.fr_probe_emit_ablate.argtypes = [vp, vp, vp, i, i, i, i, i, i, vp, vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_longlong, i, vp]
    P.fr_probe_emit_ablate.restype = i
    p = host.ptr

    def run(level):
        rc = P.fr_probe_emit_ablate(p(plan._vertex), p(net.tri), p(plan.texture), B, plan.N, plan.T, S, S, plan.tex_batch,
                                    p(plan.depth), p(plan.texture_image), p(plan.normal), p(plan.tri_ind), p(plan._ws),
                                    plan._ws_bytes, plan.pitch, level, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc

    def wall(level):
        for _ in range(5):
            run(level)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            run(level)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / K * 1e6, 2)

    names = {1: "empty workgroups", 2: "+ table loads", 3: "+ eighteen gathers",
             5: "+ bbox / pre-cull / compaction (phase A complete)", 6: "+ phase B", 7: "+ bucket scan, barrier 3", 0: "+ record stores (whole kernel)"}
    res = {v: [] for v in names.values()}
    for rnd in range(3):
        for lv in (1, 2, 3, 5, 6, 7, 0):
            res[names[lv]].append(wall(lv))
    return res


if __name__ == "__main__":
    # emit_ablate.py [B ...]: one table per batch size (default 64); several sizes itemise the kernel's FIXED term
    sizes = [int(x) for x in sys.argv[1:]] or [64]
    out = {("%d faces" % b): main(b) for b in sizes}
    if len(sizes) == 1:
        out = out["%d faces" % sizes[0]]
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/emit_ablate.json", "w"), indent=1)
