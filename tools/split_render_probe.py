#!/usr/bin/env python3
"""Development probe: the render of a 64-face batch split into two 32-face halves on two streams, so that the resolver of
the first half (HBM-bound) runs beside the emit kernel of the second (VALU-bound).  Outputs are compared with the
one-launch plan; step times are taken over interleaved rounds in one process."""
import ctypes
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
    B, S, K = 64, 200, 200
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plan = pipe.DecodeRenderPlan(net, B, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7), device=dev))
    ref = [o.clone() for o in plan.step()]
    torch.cuda.synchronize()
    h, L = plan._h, plan._L
    p = h.ptr
    vdense = plan.vertex_proj.contiguous()   # (the plan hands the vertices over in pitched rows; this probe calls the dense op)
    halves = []
    nh = int(os.environ.get("SPLIT", "2"))
    per = B // nh
    outs = [torch.empty_like(o) for o in ref]
    for i in range(nh):
        b0 = i * per
        ws_bytes = L.fr_render_depth_workspace_bytes(per, plan.N, plan.T, S, S)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        args = (p(vdense[b0:b0 + per]), p(net.tri), p(plan.texture), per, plan.N, plan.T, S, S, 3, 1,
                p(outs[0][b0:b0 + per]), p(outs[1][b0:b0 + per]), p(outs[2][b0:b0 + per]), p(outs[3][b0:b0 + per]), p(ws), ws_bytes)
        rc = L.fr_render_depth_forward_phases(*args, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream), 4)
        assert rc == 0
        halves.append((args, ws, torch.cuda.Stream(dev)))
    torch.cuda.synchronize()
    s0 = torch.cuda.current_stream(dev)

    def ph(args, st, phases):
        rc = L.fr_render_depth_forward_phases(*args, ctypes.c_void_p(st.cuda_stream), phases)
        assert rc == 0

    evD = torch.cuda.Event()
    evE = [torch.cuda.Event() for _ in range(nh)]
    evR = [torch.cuda.Event() for _ in range(nh)]
    mode = {"v": 0}

    def split_step():
        # decode of this step must not overwrite vertices an emit of the previous step still reads
        for e in evE:
            s0.wait_event(e)
        plan.decode()
        evD.record(s0)
        for i, (args, ws, st) in enumerate(halves):
            st.wait_event(evD)
            if i > 0:
                st.wait_event(evE[i - 1])    # emits run one after the other; resolve(i-1) overlaps emit(i)
            ph(args, st, 1)
            evE[i].record(st)
            ph(args, st, 2)
            evR[i].record(st)

    def finish():
        for e in evR:
            s0.wait_event(e)

    split_step(); finish()
    torch.cuda.synchronize()
    print("split outputs identical:", all(torch.equal(a, b) for a, b in zip(outs, ref)), flush=True)

    # both schedules as hipGraphs of NS consecutive steps (the eager split is bound by the host's launch rate)
    NS = 4

    def graph_of(body, fin=None):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cs = torch.cuda.current_stream(dev)
            nonlocal s0
            s0_saved, s0 = s0, cs
            for _ in range(NS):
                body()
            if fin:
                fin()
            s0 = s0_saved
        return g

    g_one = graph_of(plan.step)
    g_split = graph_of(split_step, finish)

    def wall_g(g):
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K // NS):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (K // NS * NS) * 1e6

    for rnd in range(4):
        print("graphs of %d steps: one launch %.1f us/step   split in %d: %.1f us/step" % (NS, wall_g(g_one), nh, wall_g(g_split)), flush=True)
    for o in outs:
        o.zero_()
    g_split.replay()
    torch.cuda.synchronize()
    print("split graph outputs identical:", all(torch.equal(a, b) for a, b in zip(outs, ref)), flush=True)

    def wall(fn, fin=None):
        for _ in range(10):
            fn()
        if fin:
            fin()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        if fin:
            fin()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    for rnd in range(4):
        print("one launch: %.1f us   split in %d on %d streams: %.1f us" % (wall(plan.step), nh, nh, wall(split_step, finish)), flush=True)


if __name__ == "__main__":
    main()
