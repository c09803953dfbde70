"""CPU: the oracle's whole render function against a brute-force restatement of render_depth_op.cc:201-316 whose inside
test is the REFERENCE's PointInTri binary (oracle/_ref/libref_pit.so, built from /root/reference by oracle/Makefile).
VERDICT round 1, item 2(i): the whole-function pin no longer rests on the hand-transcribed K1-K6 alone."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, kat_inputs
from ref_bruteforce import render_depth_bruteforce, render_depth_grad_bruteforce


@pytest.fixture(scope="module")
def pit(oracle):
    if not oracle.ref_point_in_tri_available():
        pytest.skip("oracle/_ref/libref_pit.so not built (reference tree absent)")
    return oracle.ref_point_in_tri_batch


def _eq(got, want, what):
    for g, w, n in zip(got, want, ("depth", "texture_image", "normal", "tri_ind")):
        assert g.shape == w.shape and g.dtype == w.dtype, (what, n)
        assert np.array_equal(g, w, equal_nan=True), (what, n, int((g != w).sum()))


def _random_scene(rs, B, nver, ntri, H, W, scale):
    ver = np.empty((B, 3, nver), np.float32)
    ver[:, 0] = rs.uniform(-0.15 * W, 1.15 * W, (B, nver))
    ver[:, 1] = rs.uniform(-0.15 * H, 1.15 * H, (B, nver))
    ver[:, 2] = rs.uniform(-50, 50, (B, nver))
    tri = rs.randint(0, nver, (3, ntri))
    k = ntri // 2          # half the triangles small: vertices near a base vertex
    base = rs.randint(0, nver, k)
    for b in range(B):
        for j in (1, 2):
            idx = tri[j, :k]
            ver[b, :2, idx] = ver[b, :2, base] + rs.uniform(-scale, scale, (k, 2)).astype(np.float32)
    tri[0, :k] = base
    tex = rs.uniform(0, 1, (B, 3, nver)).astype(np.float32)
    return ver, tri.astype(np.float32), tex


@pytest.mark.parametrize("B,nver,ntri,H,W,scale", [(2, 60, 150, 16, 18, 2.0), (1, 300, 800, 40, 33, 1.2),
                                                   (3, 40, 60, 9, 7, 6.0)])
def test_random_scenes(oracle, pit, B, nver, ntri, H, W, scale):
    rs = np.random.RandomState(ntri)
    ver, tri, tex = _random_scene(rs, B, nver, ntri, H, W, scale)
    want = render_depth_bruteforce(ver, tri, tex, H, W, pit)
    assert (want[3] >= 0).mean() > 0.05
    _eq(oracle.render_depth(ver, tri, tex, H, W), want, "random")


def test_subpixel_mesh_and_duplicate_triangles(oracle, pit):
    """jittered ~1 px grid (the BFM regime) with a duplicated patch (equal-h ties -> first triangle wins, :295 strict <)"""
    rs = np.random.RandomState(3)
    H, W, nu, nv = 24, 26, 34, 36
    gx, gy = np.meshgrid(np.linspace(-2.0, W + 1.0, nv), np.linspace(-2.0, H + 1.0, nu))
    ver = np.empty((2, 3, nu * nv), np.float32)
    for b in range(2):
        ver[b, 0] = (gx + rs.uniform(-0.45, 0.45, gx.shape)).reshape(-1)
        ver[b, 1] = (gy + rs.uniform(-0.45, 0.45, gy.shape)).reshape(-1)
        ver[b, 2] = rs.uniform(-5, 5, nu * nv)
    iu, iv = np.meshgrid(np.arange(nu - 1), np.arange(nv - 1), indexing="ij")
    v00 = (iu * nv + iv).reshape(-1)
    tri = np.concatenate([np.stack([v00, v00 + nv, v00 + 1]), np.stack([v00 + 1, v00 + nv, v00 + nv + 1])], 1)
    tri = np.concatenate([tri, tri[:, 100:160]], 1)        # exact duplicates later in the list
    tex = rs.uniform(0, 1, (1, 3, nu * nv)).astype(np.float32)
    want = render_depth_bruteforce(ver, tri.astype(np.float32), tex, H, W, pit)
    assert (want[3] >= 0).mean() > 0.5
    _eq(oracle.render_depth(ver, tri.astype(np.float32), tex, H, W), want, "subpixel")


def test_integer_edges_degenerate_and_border_rules(oracle, pit):
    """pixel centres exactly on edges / vertices, zero-area triangles (den == 0 paints the bbox), bboxes touching and
    crossing the image border (whole-triangle reject, no clipping), negative-sub-pixel overhang (ceil(-0.5) = 0)."""
    H, W = 8, 9
    ver = np.zeros((1, 3, 16), np.float32)
    pts = [(1, 1), (4, 1), (1, 4), (4, 4), (2.5, 2.5), (-0.5, 1), (3, 1), (1, 3), (8, 0), (8, 7), (5, 7), (8.01, 3),
           (9, 3), (6, 5), (0, 0), (7.5, 6.5)]
    for k, (x, y) in enumerate(pts):
        ver[0, 0, k], ver[0, 1, k], ver[0, 2, k] = x, y, 5 + 0.25 * k
    tri = np.array([[0, 1, 2], [0, 3, 4], [5, 6, 7], [8, 9, 10], [8, 11, 10], [8, 12, 10], [13, 13, 13], [14, 1, 2],
                    [15, 9, 13], [2, 1, 0]], np.float32).T.copy()
    tex = np.linspace(0, 1, 48, dtype=np.float32).reshape(1, 3, 16)
    want = render_depth_bruteforce(ver, tri, tex, H, W, pit)
    _eq(oracle.render_depth(ver, tri, tex, H, W), want, "edges")
    assert (want[3] >= 0).sum() > 10


def test_nan_inf_negative_zero_and_bad_ids(oracle, pit):
    rs = np.random.RandomState(9)
    ver, tri, tex = _random_scene(rs, 1, 50, 120, 12, 12, 3.0)
    ver[0, 0, 3] = np.nan
    ver[0, 1, 7] = np.inf
    ver[0, 2, 11] = np.nan
    ver[0, 2, 12] = -0.0
    ver[0, 2, 13] = np.inf
    tri[0, 5] = -1.0
    tri[1, 6] = 50.0
    tri[2, 7] = np.nan
    tri[0, 8] = 3.99          # truncates to 3
    tri[1, 9] = -0.5          # truncates to 0
    want = render_depth_bruteforce(ver, tri, tex, 12, 12, pit)
    _eq(oracle.render_depth(ver, tri, tex, 12, 12), want, "nan/inf")


def test_survey_known_answers_through_the_bruteforce(pit):
    """the hand-transcribed K cases (tests/golden/kat_survey.json) agree with the brute force too: two independent
    witnesses of the reference's behaviour say the same thing"""
    KAT = json.load(open(os.path.join(GOLDEN, "kat_survey.json")))
    W, H = KAT["W"], KAT["H"]
    n = 0
    for case in KAT["cases"]:
        if case["flavour"] != "op" or "tri_ind" not in case:
            continue
        ver, tri, tex = kat_inputs(case, W, H)
        got = render_depth_bruteforce(ver, tri, tex, H, W, pit)
        want = np.array([[-1 if ch == "." else int(ch) for ch in r] for r in case["tri_ind"]], np.float32)
        np.testing.assert_array_equal(got[3][0, :, :, 0], want, err_msg=case["name"])
        n += 1
    assert n >= 5


def test_one_full_size_face(oracle, pit, full_assets, synth):
    """BFM-scale mesh (53,215 vertices, 105,840 triangles, duplicated patch), 200x200, config-1 parameters."""
    A = full_assets
    P = synth.sample_params_batch(1, beta=0.7, seed=3456)
    V = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    want = render_depth_bruteforce(V, A["tri"], A["vertex"][None], 200, 200, pit)
    assert (want[3] >= 0).mean() > 0.2
    _eq(oracle.render_depth(V, A["tri"], A["vertex"][None], 200, 200), want, "full-size")


# ---- the backward as a WHOLE function -----------------------------------------------------
# render_depth_grad_bruteforce (tests/ref_bruteforce.py) is a plain-Python restatement of render_depth_op.cc:344-366 --
# row-major pixel order, fp32 `g * 1.0f / 3.0f`, sequential fp32 `+=` -- that shares no code with oracle/fr_oracle.c; the
# oracle's backward is held to it BIT FOR BIT (one fixed summation order on both sides).  No reference binary is involved
# (the loop has no helper function to compile), so these tests run on any box.

def _bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _grad_case(rs, B, H, W, kind):
    if kind == "unit":
        return np.ones((B, H, W, 1), np.float32)
    if kind == "wide":       # 12 decades of magnitude, both signs: the order of the fp32 adds matters
        return (rs.standard_normal((B, H, W, 1)) * np.exp(rs.uniform(-14, 14, (B, H, W, 1)))).astype(np.float32)
    g = rs.standard_normal((B, H, W, 1)).astype(np.float32)
    g[rs.rand(B, H, W, 1) < 0.1] = 0.0
    return g


@pytest.mark.parametrize("kind", ["unit", "normal", "wide"])
def test_backward_whole_function_random_scenes(oracle, kind):
    rs = np.random.RandomState(77)
    for B, nver, ntri, H, W, scale in ((2, 60, 150, 16, 18, 2.0), (1, 300, 800, 40, 33, 1.2), (3, 40, 60, 9, 7, 6.0)):
        ver, tri, tex = _random_scene(rs, B, nver, ntri, H, W, scale)
        tri_ind = oracle.render_depth(ver, tri, tex, H, W)[3]
        assert (tri_ind >= 0).mean() > 0.05
        g = _grad_case(rs, B, H, W, kind)
        want = render_depth_grad_bruteforce(g, tri, tri_ind, nver)
        got = oracle.render_depth_grad(g, tri, tri_ind, nver)
        assert _bits_equal(got, want), (kind, B, nver, int((got != want).sum()))
        assert not got[:, :2].any()                       # x and y rows carry no gradient (:359-363)


def test_backward_whole_function_ties_bad_ids_and_hand_made_tri_ind(oracle):
    """Many pixels on few triangles (long per-vertex sums: the order of the adds is what is being pinned), shared
    vertices, out-of-range / negative / fractional / NaN tri_ind values, triangles with ids outside [0, nver)."""
    rs = np.random.RandomState(5)
    B, H, W, nver = 2, 23, 19, 12
    tri = np.array([[0, 1, 2, 3, 0, 11, 12, -1, 5],
                    [1, 2, 3, 4, 5, 10, 3, 2, 5.9],
                    [2, 3, 4, 0, 6, 9, 1, 1, 6.2]], np.float32)      # triangle 6: id 12 = nver (bad), 7: -1 (bad), 8: fractional
    ntri = tri.shape[1]
    tri_ind = rs.randint(-1, ntri, (B, H, W, 1)).astype(np.float32)
    tri_ind[0, 0, 0, 0] = np.nan
    tri_ind[0, 0, 1, 0] = ntri            # one past the end
    tri_ind[0, 0, 2, 0] = 3.9             # (int) truncates to 3
    tri_ind[0, 0, 3, 0] = -0.5            # truncates to 0 -> triangle 0
    tri_ind[0, 0, 4, 0] = 1e20            # out of int range
    tri_ind[1, :, :, 0] = 4               # a whole face on ONE triangle: 437 sequential adds per vertex
    for kind in ("normal", "wide"):
        g = _grad_case(rs, B, H, W, kind)
        want = render_depth_grad_bruteforce(g, tri, tri_ind, nver)
        got = oracle.render_depth_grad(g, tri, tri_ind, nver)
        assert _bits_equal(got, want), kind
    g = _grad_case(rs, B, H, W, "normal")
    g[1, 3, 3, 0] = np.inf
    g[0, 5, 5, 0] = np.nan
    want = render_depth_grad_bruteforce(g, tri, tri_ind, nver)
    got = oracle.render_depth_grad(g, tri, tri_ind, nver)
    assert np.array_equal(got, want, equal_nan=True) and np.isnan(got).any()


def test_backward_whole_function_one_full_size_face(oracle, full_assets, synth):
    """BFM-scale mesh, 200x200, the bench's parameters: 40,000 pixels through the Python loop (a few seconds)."""
    A = full_assets
    P = synth.sample_params_batch(1, beta=0.7, seed=3456)
    V = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    tri_ind = oracle.render_depth(V, A["tri"], A["vertex"][None], 200, 200)[3]
    g = _grad_case(np.random.RandomState(9), 1, 200, 200, "wide")
    want = render_depth_grad_bruteforce(g, A["tri"], tri_ind, V.shape[2])
    got = oracle.render_depth_grad(g, A["tri"], tri_ind, V.shape[2])
    assert _bits_equal(got, want)
    assert np.count_nonzero(got[0, 2]) > 5000
