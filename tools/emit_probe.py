#!/usr/bin/env python3
"""Development probe of raster_emit_kernel on the bench workload (64 faces, 200x200, BFM-scale mesh):
  (1) A/B of the launcher knobs in one process, interleaved rounds (fr_set_option: FR_EMIT_FILTER), outputs compared;
  (2) the stamped build (tools/libemit_probe.so from tools/emit_probe.hip): every wave's s_memtime at the phase
      boundaries -> shares of a workgroup's life per phase (median over all 13,440 workgroups), printed as JSON.
Build the probe library first (command in tools/emit_probe.hip)."""
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def pkg(n):
    return importlib.import_module("3dfacerecon_amd." + n)


def main():
    B, S, K = 64, 200, 100
    synth, netm, pipe, host = pkg("utils.synth"), pkg("nets.network"), pkg("pipeline"), pkg("_lib")
    dev = torch.device("cuda:0")
    A = synth.make_assets()
    net = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    plan = pipe.DecodeRenderPlan(net, B, S, S)
    plan.params.copy_(torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=3456), device=dev))
    with host.options(FR_EMIT_FILTER=0):
        ref = [o.clone() for o in plan.step()]
    torch.cuda.synchronize()

    def wall(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    out = {"ab_us": {}}
    res = {1: [], 3: []}
    for rnd in range(3):
        for v in (1, 3):
            with host.options(FR_EMIT_FILTER=v):
                outs = plan.step()
                torch.cuda.synchronize()
                assert all(torch.equal(a, b) for a, b in zip(outs, ref))
                res[v].append((round(wall(lambda: plan.render_phase(1)), 1), round(wall(plan.step), 1)))
    for v in (1, 3):
        out["ab_us"]["FR_EMIT_FILTER=%d (emit alone, step)" % v] = res[v]

    # ---- stamps -------------------------------------------------------------------------------------------------------
    so = os.path.join(ROOT, "tools", "libemit_probe.so")
    if os.path.exists(so):
        P = ctypes.CDLL(so)
        vp, i = ctypes.c_void_p, ctypes.c_int
        P.fr_probe_emit_stamps.argtypes = [vp, vp, vp, i, i, i, i, i, i, vp, vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_longlong,
                                           vp, vp]
        P.fr_probe_emit_stamps.restype = i
        nwg_max = B * 256
        stamps = torch.zeros((nwg_max * 4 * 10,), dtype=torch.int64, device=dev)
        p = host.ptr
        for _ in range(50):
            plan.step()
        torch.cuda.synchronize()
        nwg = P.fr_probe_emit_stamps(p(plan._vertex), p(net.tri), p(plan.texture), B, plan.N, plan.T, S, S, plan.tex_batch,
                                     p(plan.depth), p(plan.texture_image), p(plan.normal), p(plan.tri_ind), p(plan._ws),
                                     plan._ws_bytes, plan.pitch, p(stamps),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert nwg > 0, nwg
        plan.render_phase(2)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(plan.outputs(), ref)), "stamped build changed the records"
        st = stamps.cpu().numpy().reshape(-1, 4, 10)[:nwg].astype(np.float64)
        t = st[:, :, :9]
        nq = st[:, 0, 9]
        seg = np.diff(t, axis=2)                      # [wg, wave, 8 segments]
        names = ["entry -> counters zeroed + barrier", "A: table + 18 gathers + bbox + single-pixel pre-cull",
                 "A: compaction into the LDS queue", "barrier 1", "B: depth / inside test / normal / LDS record",
                 "barrier 2", "C: bucket scan (wave 0) + barrier 3", "C: records out"]
        life = t[:, :, 8] - t[:, :, 0]
        rep = {"workgroups": int(nwg), "survivors_per_workgroup_median": float(np.median(nq)),
               "wave_life_cycles_median": float(np.median(life)), "segments_cycles_median_over_waves": {},
               "segments_share_of_wave_life": {}}
        for k, nm in enumerate(names):
            rep["segments_cycles_median_over_waves"][nm] = float(np.median(seg[:, :, k]))
            rep["segments_share_of_wave_life"][nm] = float(np.mean(seg[:, :, k]) / np.mean(life))
        # phase B by wave index (the queue is dense: waves 2-3 usually have nothing to do)
        rep["phase_B_cycles_median_by_wave"] = [float(np.median(seg[:, w, 4])) for w in range(4)]
        out["stamps"] = rep
        # ---- emit at reduced residency (extra, unused dynamic LDS): 8 / 7 / 6 / 5 / 4 workgroups per CU ---------------------
        P.fr_probe_emit_with_extra_lds.argtypes = [vp, vp, vp, i, i, i, i, i, i, vp, vp, vp, vp, vp, ctypes.c_size_t,
                                                   ctypes.c_longlong, i, vp]
        P.fr_probe_emit_with_extra_lds.restype = i
        occ = {}
        for wgs, extra in ((8, 0), (7, 2400), (6, 6000), (5, 11000), (4, 19000)):
            def run():
                rc = P.fr_probe_emit_with_extra_lds(p(plan._vertex), p(net.tri), p(plan.texture), B, plan.N, plan.T, S, S,
                                                    plan.tex_batch, p(plan.depth), p(plan.texture_image), p(plan.normal),
                                                    p(plan.tri_ind), p(plan._ws), plan._ws_bytes, plan.pitch, extra,
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            occ["%d workgroups per CU" % wgs] = round(wall(run), 1)
        out["emit_alone_us_by_residency"] = occ
        # ---- the resolver, same method --------------------------------------------------------------------------------
        P.fr_probe_resolve_stamps.argtypes = P.fr_probe_emit_stamps.argtypes
        P.fr_probe_resolve_stamps.restype = i
        for _ in range(20):
            plan.step()
        plan.render_phase(1)
        torch.cuda.synchronize()
        stamps.zero_()
        nb = P.fr_probe_resolve_stamps(p(plan._vertex), p(net.tri), p(plan.texture), B, plan.N, plan.T, S, S, plan.tex_batch,
                                       p(plan.depth), p(plan.texture_image), p(plan.normal), p(plan.tri_ind), p(plan._ws),
                                       plan._ws_bytes, plan.pitch, p(stamps),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert nb > 0, nb
        assert all(torch.equal(a, b) for a, b in zip(plan.outputs(), ref)), "stamped resolver changed the planes"
        st = stamps.cpu().numpy().reshape(-1, 4, 10)[:nb].astype(np.float64)
        t = st[:, :, :9]
        # stamps: 0 entry, 1 offsets back, 2 wave scan + wave list, 3 record loads issued, 4 LDS-only barrier + records back +
        # LDS max, 5 barrier, 6 winners' normals stored, 7 unused (= 0 on the fast path), 8 end (plane stores issued)
        fast = t[:, 0, 4] > 0          # bins that took the wave-local register path
        names = [("entry -> offsets back (keys initialised meanwhile)", 0, 1), ("wave scan (DPP) + the wave's slot list", 1, 2),
                 ("slot list read, record + normal loads issued", 2, 3),
                 ("LDS-only barrier, records back, LDS max (+ the busy waves' overflow loop)", 3, 4), ("barrier", 4, 5),
                 ("winners' normals stored", 5, 6), ("plane writer: keys -> texture-mean gathers -> stores issued", 6, 8)]
        rr = {"bins": int(nb), "bins_on_the_single_trip_path": int(fast.sum()), "segments_cycles_median_over_waves": {},
              "wave_life_cycles_median": float(np.median(t[fast][:, :, 8] - t[fast][:, :, 0]))}
        for nm, i0, i1 in names:
            rr["segments_cycles_median_over_waves"][nm] = float(np.median(t[fast][:, :, i1] - t[fast][:, :, i0]))
        out["resolve_stamps"] = rr
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
