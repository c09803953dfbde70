#!/usr/bin/env python3
"""Probe: can the resolve kernel (store-path-bound) keep its rate on a SUBSET of the CUs, and do emit and resolve overlap
when they run on DISJOINT CU sets?  Streams with CU masks (hipExtStreamCreateWithCUMask), existing stand-alone kernels."""
import ctypes, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    B, S, K = 64, 200, 60
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plans = [pipe.DecodeRenderPlan(net, B, S, S) for _ in range(2)]
    for i, p in enumerate(plans):
        p.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456 + i), device=dev))
        p.step()
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    FULL = (1 << 256) - 1

    def timed(fn):
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(K)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / K * 1e6, 1)

    def strided(n_of, every):   # n_of CUs out of every `every`
        m = 0
        for i in range(256):
            if i % every < n_of:
                m |= 1 << i
        return m

    masks = {"all": FULL, "first128": (1 << 128) - 1, "first64": (1 << 64) - 1, "1of2": strided(1, 2), "1of4": strided(1, 4),
             "1of8": strided(1, 8), "2of8": strided(2, 8), "3of8": strided(3, 8)}
    res = {}
    for name, m in masks.items():
        st = masked_stream(hip, m)

        def run(n, st=st):
            with torch.cuda.stream(st):
                for _ in range(n):
                    plans[1].render_phase(2)
        res["resolve_on_" + name] = timed(run)

        def run_e(n, st=st):
            with torch.cuda.stream(st):
                for _ in range(n):
                    plans[0].render_phase(1)
        res["emit_on_" + name] = timed(run_e)
    print(json.dumps(res), flush=True)
    # disjoint sets: emit on the complement of the resolve set
    out = {}
    for name in ("first64", "first128", "1of4", "2of8", "3of8", "1of2"):
        m = masks[name]
        s_res, s_emit = masked_stream(hip, m), masked_stream(hip, FULL & ~m)

        def both(n):
            for _ in range(n):
                with torch.cuda.stream(s_emit):
                    plans[0].render_phase(1)
                with torch.cuda.stream(s_res):
                    plans[1].render_phase(2)
        out["emit_on_rest_and_resolve_on_" + name] = timed(both)
    s1, s2 = masked_stream(hip, FULL), masked_stream(hip, FULL)

    def both_all(n):
        for _ in range(n):
            with torch.cuda.stream(s1):
                plans[0].render_phase(1)
            with torch.cuda.stream(s2):
                plans[1].render_phase(2)
    out["both_on_all"] = timed(both_all)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
