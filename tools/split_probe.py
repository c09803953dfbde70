#!/usr/bin/env python3
"""Development probe: does resolve(chunk k) overlap emit(chunk k+1)?  Splits the batch into face chunks and runs
emit / resolve of the chunks (a) serially, (b) on two streams inside a captured hipGraph with fork/join edges."""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def timeit(fn, iters=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B = 64
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=200, device="cuda:0")
    plan = pipe.DecodeRenderPlan(net, B, 200, 200)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, beta=0.7), device="cuda:0"))
    plan.step()
    torch.cuda.synchronize()
    ref = [t.clone() for t in plan.outputs()]
    L, h = plan._L, plan._h
    N, T, H, W = plan.N, plan.T, plan.H, plan.W

    def chunk_calls(nch):
        Bc = B // nch
        wsb = L.fr_render_depth_workspace_bytes(Bc, N, T, H, W)
        ws = torch.empty((nch * wsb,), dtype=torch.uint8, device="cuda:0")
        calls = []
        for c in range(nch):
            b0 = c * Bc
            args = (h.ptr(plan.vertex_proj[b0:]), h.ptr(net.tri), h.ptr(plan.texture), Bc, N, T, H, W, 3, 1,
                    h.ptr(plan.depth[b0:]), h.ptr(plan.texture_image[b0:]), h.ptr(plan.normal[b0:]), h.ptr(plan.tri_ind[b0:]),
                    ctypes.c_void_p(ws.data_ptr() + c * wsb), wsb)
            calls.append(args)
        return calls, ws

    def launch(args, phase):
        rc = L.fr_render_depth_forward_phases(*args, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), phase)
        assert rc == 0, rc

    print("current: decode+render      %.1f us" % timeit(plan.step))
    for nch in (2, 4):
        calls, ws = chunk_calls(nch)

        def serial():
            plan.decode()
            for c in calls:
                launch(c, 1)
            for c in calls:
                launch(c, 2)
        print("%d chunks serial            %.1f us" % (nch, timeit(serial)))
        s2 = torch.cuda.Stream()

        def forked():
            plan.decode()
            s1 = torch.cuda.current_stream()
            evs = []
            for k, c in enumerate(calls):
                launch(c, 1)
                ev = torch.cuda.Event()
                ev.record(s1)
                evs.append(ev)
                if k >= 1:  # resolve of the previous chunk on the side stream, overlapping this chunk's emit
                    with torch.cuda.stream(s2):
                        s2.wait_event(evs[k - 1])
                        launch(calls[k - 1], 2)
            launch(calls[-1], 2)
            s1.wait_stream(s2)
        print("%d chunks 2 streams eager   %.1f us" % (nch, timeit(forked)))
        g = torch.cuda.CUDAGraph()
        forked()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            forked()
        print("%d chunks 2 streams graph   %.1f us" % (nch, timeit(g.replay)))
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) for a, b in zip(ref, plan.outputs()))
        print("   outputs identical:", ok)
    plan.capture()
    print("current graph replay        %.1f us" % timeit(plan.replay))


if __name__ == "__main__":
    main()
