#!/usr/bin/env python3
"""Development probe: does write traffic into memory allocated with hipExtMallocWithFlags(hipDeviceMallocUncached /
hipDeviceMallocFinegrained) stay out of the 256 MiB Infinity Cache?  If it does, a plan could keep its write-only output
planes (82 MB per step) in such memory and leave the cache to the 153 MB basis the decode re-reads every step.
Method: the decode with DEFAULT-policy basis loads (FR_DECODE_NT=0: cacheable), timed after Y MB of hipMemsetAsync traffic
into (a) ordinary device memory, (b) uncached memory, (c) fine-grained memory."""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    B, S, K = 64, 200, 60
    synth, netm, pipe, host = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline"), pkg("_lib")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plan = pipe.DecodeRenderPlan(net, B, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7), device=dev))
    plan.step()
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    size = 320 << 20
    bufs = {}
    for name, flags in (("ordinary", None), ("uncached", 0x3), ("finegrained", 0x1)):
        p = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(p), size) if flags is None else hip.hipExtMallocWithFlags(ctypes.byref(p), size, flags)
        print("alloc %s: rc=%d ptr=%s" % (name, rc, hex(p.value or 0)), flush=True)
        if rc == 0:
            bufs[name] = p
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def ev_time(fn_between):
        es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for e0, e1 in es:
            fn_between()
            e0.record()
            plan.decode()
            e1.record()
        torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) for e0, e1 in es[5:])
        return 1e3 * t[len(t) // 2]

    with host.options(FR_DECODE_NT=0):
        print("decode, cacheable basis, back to back: %.1f us" % ev_time(lambda: None), flush=True)
        for name, p in bufs.items():
            for y in (64, 128, 192, 256, 320):
                n = y << 20
                t = ev_time(lambda: hip.hipMemsetAsync(p, 1, n, stream))
                print("decode after %3d MB of memset into %-11s memory: %.1f us" % (y, name, t), flush=True)
    with host.options(FR_DECODE_NT=1):
        print("decode, non-temporal basis (product), back to back: %.1f us" % ev_time(lambda: None), flush=True)


if __name__ == "__main__":
    main()
