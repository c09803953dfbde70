"""The pipelined step (fr_decode_render_pipelined / pipeline.PipelinedPlan): emit of batch k beside resolve of batch k-1 in
ONE launch.  Bars: the planes are bit-identical to the serial plan's on the same parameters (all 64 full-size faces) and to
the CPU oracle on arbitrary scenes -- sub-pixel grids, random soup, oversized triangles (the resolver rasterises those from
the PREVIOUS batch's vertices), integer ties, NaN / Inf / bad ids -- through both roles of the fused launch."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from gpu_util import assert_render_equal, net_mod, render_pipelined_gpu
from test_fuzz_gpu import _scene

pytestmark = pytest.mark.gpu


def _second_batch(ver, seed):
    """Another batch of the same shape: the faces in another order, every face moved a little."""
    rs = np.random.RandomState(seed)
    v = ver[::-1].copy()
    v[:, :2] += rs.uniform(-0.7, 0.7, (v.shape[0], 2, 1)).astype(np.float32)
    v[:, 2] *= np.float32(0.5)
    return v


N_SCENES = max(48, int(os.environ.get("FR_FUZZ_CASES", "0")) * 3 // 10)   # FR_FUZZ_CASES widens this sweep too


@pytest.mark.parametrize("seed", range(N_SCENES))
def test_fused_launch_both_roles_vs_oracle(oracle, seed):
    ver, tri, tex, H, W = _scene(7000 + seed)
    ver_b = _second_batch(ver, seed)
    got = render_pipelined_gpu(ver, ver_b, tri, tex, H, W)
    if got is None:
        pytest.skip("shape not served by the pipelined entry point (W = %d)" % W)
    assert_render_equal(got[0], oracle.render_depth(ver, tri, tex, H, W), "pipelined, batch a (fused resolve role), seed %d" % seed)
    assert_render_equal(got[1], oracle.render_depth(ver_b, tri, tex, H, W), "pipelined, batch b (drain), seed %d" % seed)


def test_support_matrix_and_error_codes():
    L = pkg("_lib").lib()
    assert L.fr_decode_render_pipelined_supported(64, 53215, 105840, 200, 200) == 1
    assert L.fr_decode_render_pipelined_supported(64, 53215, 105840, 448, 448) == 0      # a 4-row strip does not fit the keys
    assert L.fr_decode_render_pipelined_supported(1, 100, 50, 3, 3) == 1
    assert L.fr_decode_render_pipelined_supported(0, 100, 50, 8, 8) == 0
    assert L.fr_decode_render_pipelined_supported(2, 100, 0, 8, 8) == 0
    assert L.fr_decode_render_pipelined_supported(2, 100, 50, 3, 401) == 1               # H <= rows: one strip
    assert L.fr_decode_render_pipelined_supported(2, 100, 50, 40, 401) == 0


def test_plan_stream_of_batches_bit_identical_to_the_serial_plan(full_assets, synth):
    """Three 64-face batches through PipelinedPlan: every plane of every face equals DecodeRenderPlan.step() on the same
    parameters (which tests/test_pipeline_gpu.py holds to the oracle); step() = submit + flush agrees too."""
    dev = torch.device("cuda:0")
    B = 64
    net = net_mod().FaceRecNet(mesh_data=full_assets, batch_size=B, im_size=200, device=dev)
    pipe = pkg("pipeline")
    serial = pipe.DecodeRenderPlan(net, B, 200, 200)
    piped = pipe.PipelinedPlan(net, B, 200, 200)
    P = [torch.as_tensor(synth.sample_params_batch(B, im_size=200, beta=0.7, seed=s), device=dev) for s in (3456, 11, 12)]
    want, wantv = [], []
    for p in P:
        want.append([t.clone() for t in serial.step(p)])
        wantv.append(serial.vertex_proj.clone())
    assert piped.submit(P[0]) is None
    assert torch.equal(piped.vertex_proj, wantv[0])
    got = []
    for k in (1, 2):
        outs = piped.submit(P[k])
        got.append([t.clone() for t in outs])
        assert torch.equal(piped.vertex_proj, wantv[k])
    got.append([t.clone() for t in piped.flush()])
    assert piped.flush() is None
    for k in range(3):
        for g, w, n in zip(got[k], want[k], ("depth", "texture_image", "normal", "tri_ind")):
            assert torch.equal(g, w), "batch %d, %s: %d elements differ" % (k, n, int((g != w).sum()))
    assert float((want[0][3] >= 0).float().mean()) > 0.2
    for g, w in zip(piped.step(P[1]), want[1]):
        assert torch.equal(g, w)
    # launch by launch (what bench.py brackets with events) is the same stream of launches
    piped.submit_phases(8); piped.submit_phases(3)        # batch P[1] again (the plan's params buffer still holds it)
    piped.params.copy_(P[2])
    piped.submit_phases(8); piped.submit_phases(3)
    for g, w in zip(piped.outputs(), want[1]):
        assert torch.equal(g, w)
    for g, w in zip(piped.flush(), want[2]):
        assert torch.equal(g, w)
