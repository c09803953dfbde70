#!/usr/bin/env python3
"""Probe: S INDEPENDENT batches in flight -- S DecodeRenderPlans (each its own workspace, vertex buffer and planes), each
stepping on its own stream with NO cross-stream edge -- against one plan stepping on one stream.  Microseconds per
(64-face) batch step = wall time / (K * S).  Eager launches and hipGraph replays."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    B, S, K = 64, 200, 200
    SMAX = int(os.environ.get("SMAX", "4"))
    synth, netm, pipe = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plans = [pipe.DecodeRenderPlan(net, B, S, S) for _ in range(SMAX)]
    PRIO = os.environ.get("PRIO", "")
    if "," in PRIO:   # explicit list: PRIO=-1,1
        pl = [int(x) for x in PRIO.split(",")]
        ss = [torch.cuda.Stream(priority=pl[i % len(pl)]) for i in range(SMAX)]
    else:
        ss = [torch.cuda.Stream(priority=(-1 if (PRIO == "first" and i == 0) or PRIO == "all" else 0)) for i in range(SMAX)]
    print(json.dumps({"priority_range": list(torch.cuda.Stream.priority_range()) if hasattr(torch.cuda.Stream, "priority_range") else None,
                      "priorities": [s.priority for s in ss]}))
    for i, p in enumerate(plans):
        p.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456 + i), device=dev))
        p.step()
    torch.cuda.synchronize()
    for i, p in enumerate(plans):
        with torch.cuda.stream(ss[i]):
            p.capture()
    torch.cuda.synchronize()

    def timed(fn, n):
        fn(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(K)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / (K * n) * 1e6, 2)

    def eager(n):
        def f(k):
            for _ in range(k):
                for i in range(n):
                    with torch.cuda.stream(ss[i]):
                        plans[i].step()
        return f

    def graph(n):
        def f(k):
            for _ in range(k):
                for i in range(n):
                    with torch.cuda.stream(ss[i]):
                        plans[i]._graph.replay()
        return f

    for rnd in range(3):
        row = {}
        for n in range(1, SMAX + 1):
            row["eager_%d" % n] = timed(eager(n), n)
            row["graph_%d" % n] = timed(graph(n), n)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
