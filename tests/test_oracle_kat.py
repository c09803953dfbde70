"""CPU: the oracle against the known answers the reference itself produced."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, grid_from_rows, kat_inputs

KAT = json.load(open(os.path.join(GOLDEN, "kat_survey.json")))
W, H = KAT["W"], KAT["H"]


@pytest.mark.parametrize("case", [c for c in KAT["cases"] if c["flavour"] == "op"], ids=lambda c: c["name"])
def test_kat_op(oracle, case):
    ver, tri, tex = kat_inputs(case, W, H)
    depth, timg, nrm, tind = oracle.render_depth(ver, tri, tex, H, W)
    if "tri_ind" in case:
        np.testing.assert_array_equal(tind[0, :, :, 0], grid_from_rows(case["tri_ind"]))
    if "tri_ind_row0" in case:
        np.testing.assert_array_equal(tind[0, 0, :, 0], grid_from_rows([case["tri_ind_row0"]])[0])
    for x, y, d in case.get("depth_at", []):
        assert depth[0, y, x, 0] == np.float32(d)
    for x, y, t in case.get("tex_at", []):
        np.testing.assert_array_equal(timg[0, y, x], (np.array([0.1, 0.2, 0.3], np.float32) * 1 +
                                                       np.array([0.1, 0.2, 0.3], np.float32) * 2 +
                                                       np.array([0.1, 0.2, 0.3], np.float32) * 3) / np.float32(3))
        np.testing.assert_allclose(timg[0, y, x], t, rtol=1e-6)
    for x, y, n in case.get("normal_at", []):
        np.testing.assert_array_equal(nrm[0, y, x], np.array(n, np.float32))
    bg = tind[0, :, :, 0] < 0
    assert np.all(depth[0, :, :, 0][bg] == np.float32(-100000000376832.0))
    assert np.all(timg[0][bg] == 0) and np.all(nrm[0][bg] == 0)
    if "background" in case:
        assert bg.all()
        assert float(depth[0, 0, 0, 0]) == case["background"]["depth"]


@pytest.mark.parametrize("case", [c for c in KAT["cases"] if c["flavour"] == "mex"], ids=lambda c: c["name"])
def test_kat_mex(oracle, case):
    ver, tri, tex = kat_inputs(case, W, H)
    img, tind = oracle.zbuffer_mex(ver[0].astype(np.float64), tri.astype(np.float64), tex[0].astype(np.float64),
                                   np.zeros((H, W, 3)))
    np.testing.assert_array_equal(tind, grid_from_rows(case["tri_ind"]))


def test_kat_grad(oracle):
    k = KAT["K6_grad"]
    ver, tri, tex = kat_inputs(k, W, H)
    _, _, _, tind = oracle.render_depth(ver, tri, tex, H, W)
    g = np.where(tind >= 0, np.float32(k["depth_grad_on_covered"]), np.float32(0)).astype(np.float32)
    vg = oracle.render_depth_grad(g, tri, tind, ver.shape[2])
    np.testing.assert_array_equal(vg[0, 2], np.array(k["vertex_grad_z"], np.float32))
    assert np.all(vg[0, :2] == k["vertex_grad_xy"])


def test_rotation_probe(oracle):
    p = KAT["rotation_probe"]
    R = oracle.rotation_matrix(np.array(p["angles"], np.float32))
    np.testing.assert_array_equal(R, np.array(p["R"], np.float32))


def test_background_constant(oracle):
    assert float(oracle.BG_DEPTH) == -100000000376832.0


def test_blas_decode_baseline_tracks_the_float64_formula(oracle, small_assets):
    """decode_3dmm_blas (the cpu_baseline's matmul-style decode) is the same formula as the parity oracle: both sit
    within a few fp32 ulp of the float64 evaluation (different summation orders, so not bit-equal to each other)."""
    A = small_assets
    rs = np.random.RandomState(0)
    B = 5
    P = np.zeros((B, 7 + A["ndim_shape"] + A["ndim_exp"]), np.float32)
    P[:, 0:3] = rs.uniform(-1, 1, (B, 3))
    P[:, 3:5] = rs.uniform(15, 25, (B, 2))
    P[:, 6] = rs.uniform(1.5e-4, 2.5e-4, B)
    P[:, 7:] = np.concatenate([rs.uniform(0, 1e4, (B, A["ndim_shape"])), rs.uniform(-1.5, 1.5, (B, A["ndim_exp"]))], 1)
    v64 = oracle.decode_3dmm_f64(P, A["mu"], A["pc_shape"], A["pc_exp"], 40.0)
    for v in (oracle.decode_3dmm_blas(P, A["mu"], A["pc_shape"], A["pc_exp"], 40.0),
              oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 40.0)):
        assert v.shape == v64.shape and v.dtype == np.float32
        assert np.abs(v - v64).max() <= 8 * np.spacing(np.float32(np.abs(v64).max()))
