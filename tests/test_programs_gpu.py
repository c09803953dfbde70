"""GPU: the callers run AS PROGRAMS, the way a user or the driver runs them.

* examples/coarse_loop.py --phase test -- the reference's evaluation loop (trainval.py:192-210) with the depth rendering of batch k
  in flight (pipeline.BatchesInFlight, its own stream) beside the network of batch k + 1 and a consumer on torch's current stream,
  ordered with make_current_stream_wait(): exactly where round 4's stream-ordering bug lived.  The program dumps what every timed
  batch's consumer saw; HERE the CPU oracle (the program never loads it) re-renders every batch's vertices and re-decodes its
  parameters: planes bit for bit, vertices to the in-kernel-rotation bar of tests/test_decode_gpu.py (<= 2 ulp, >= 99 % equal).
* bench.py --config 3 -- BASELINE.json configs[2] (CoarseNet + render_depth forward, batch 32) through the driver-facing entry:
  ONE parsable line, finite parameters."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]     # ONE line on stdout, nothing else
    return json.loads(lines[0])


def test_eval_loop_program_against_the_oracle(oracle, tmp_path):
    dump = str(tmp_path / "batches.npz")
    B, S, K = 8, 120, 4
    d = _run([sys.executable, "examples/coarse_loop.py", "--phase", "test", "--small", "--batch", str(B), "--im-size", str(S),
              "--steps", str(K), "--warmup", "1", "--dump-batches", dump])
    assert d["phase"] == "test" and d["batches_in_flight"] == 2 and d["faces_per_gpu"] == B and d["steps"] == K
    assert d["depth_identical_to_one_batch_at_a_time"] is True and d["value"] > 0
    z = np.load(dump)
    assert int(z["steps"]) == K and int(z["im_size"]) == S
    tri, tex = z["tri"], z["vertex"][None]
    seen = []
    for k in range(K):
        P, V = z["params_%d" % k], z["vertex_proj_%d" % k]
        assert P.shape == (B, 7 + z["pc_shape"].shape[-1] + z["pc_exp"].shape[-1]) and np.isfinite(P).all()
        # the planes the consumer read are the rasterisation of the vertices it read: bit for bit, all four planes
        want = oracle.render_depth(V, tri, tex, S, S)
        for name, w in zip(("depth", "texture_image", "normal", "tri_ind"), want):
            g = z["%s_%d" % (name, k)]
            assert np.array_equal(g, w, equal_nan=True), "batch %d: %s differs from the oracle" % (k, name)
        assert (z["tri_ind_%d" % k] >= 0).mean() > 0.005          # (a face is on screen: the checks above are not of empty planes)
        # ... and those vertices are the decode of the parameters CoarseNet predicted for THIS batch
        Vo = oracle.decode_3dmm(P, z["mu"], z["pc_shape"], z["pc_exp"], float(S), R=oracle.rotation_matrix_batch(P[:, :3]))
        a, o = V.view(np.int32).astype(np.int64), Vo.view(np.int32).astype(np.int64)
        a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
        o = np.where(o < 0, -(o & 0x7FFFFFFF), o)
        ulp = np.abs(a - o)
        assert ulp.max() <= 2 and (ulp == 0).mean() >= 0.99, (k, int(ulp.max()), float((ulp == 0).mean()))
        seen.append(P)
    # the four test batches are four different images: a slot handing its consumer the PREVIOUS batch's planes would have passed
    # the per-batch checks above only if the parameters had been the previous batch's too
    assert all(not np.array_equal(seen[0], s) for s in seen[1:])


def test_bench_config3_program():
    d = _run([sys.executable, "bench.py", "--config", "3", "--steps", "2"], timeout=1500)
    assert d["config"].startswith("configs[2]") and d["faces_per_gpu"] == 32 and d["im_size"] == 200 and d["n_gpus"] == 1
    assert d["params_finite"] is True and d["value"] > 0 and d["steps"] == 2 and d["train"] is False
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
