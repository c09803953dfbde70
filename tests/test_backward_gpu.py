"""GPU parity: fr_render_depth_backward + autograd wiring vs the CPU oracle (render_depth_op.cc:325-368 with
zero-init and the tri_ind<0 guard).  The HIP kernel scatter-adds with f32 atomics, so the per-vertex sum order
is not the oracle's row-major order: tolerance = count * ulp of the partial sums (documented in DESIGN.md)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, kat_inputs
from gpu_util import ops

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(GOLDEN, "kat_survey.json")))


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a, np.float32), device="cuda:0")


def test_kat_k6():
    k = KAT["K6_grad"]
    W, H = KAT["W"], KAT["H"]
    ver, tri, tex = kat_inputs(k, W, H)
    v = _t(ver).requires_grad_(True)
    depth, timg, nrm, tind = ops().render_depth(v, _t(tri), _t(tex), torch.zeros((1, H, W, 3), device="cuda:0"))
    g = torch.where(tind >= 0, torch.full_like(depth, k["depth_grad_on_covered"]), torch.zeros_like(depth))
    depth.backward(g)
    vg = v.grad.cpu().numpy()
    np.testing.assert_array_equal(vg[0, 2], np.array(k["vertex_grad_z"], np.float32))
    assert np.all(vg[0, :2] == 0)


def test_golden_small_backward(small_assets):
    z = np.load(os.path.join(GOLDEN, "render_small_oracle.npz"))
    H, W = int(z["H"]), int(z["W"])
    B = z["vertex"].shape[0]
    vg = ops().render_depth_grad(_t(z["depth_grad"]), _t(z["vertex"]), _t(small_assets["tri"]), _t(z["depth"]),
                                 _t(z["tri_ind"]), torch.zeros((B, H, W, 3), device="cuda:0"))
    got = vg.cpu().numpy()
    want = z["vertex_grad"]
    assert np.all(got[:, :2] == 0)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
    assert np.abs(want).max() > 0.1


def test_autograd_full_size(oracle, full_assets, synth):
    A = full_assets
    P = synth.sample_params_batch(2, beta=0.7, seed=1)
    V = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    v = _t(V).requires_grad_(True)
    outs = ops().render_depth(v, _t(A["tri"]), _t(A["vertex"]), torch.zeros((2, 200, 200, 3), device="cuda:0"))
    rs = np.random.RandomState(0)
    g = rs.standard_normal((2, 200, 200, 1)).astype(np.float32)
    (outs[0] * _t(g)).sum().backward()
    got = v.grad.cpu().numpy()
    tind = outs[3].detach().cpu().numpy()
    want = oracle.render_depth_grad(g, A["tri"], tind, V.shape[2])
    assert np.all(got[:, :2] == 0)
    np.testing.assert_allclose(got[:, 2], want[:, 2], rtol=0, atol=2e-5)
    # conservation: every covered pixel hands out exactly g (3 x g/3), up to rounding
    tot = g[tind >= 0].astype(np.float64).sum()
    assert abs(got.astype(np.float64).sum() - tot) < 1e-2


def test_background_and_bad_ids_are_skipped(oracle):
    H, W = 8, 8
    tri = np.array([[0, 1, 2]], np.float32).T.copy()
    tind = -np.ones((1, H, W, 1), np.float32)
    tind[0, 2, 2, 0] = 0
    tind[0, 3, 3, 0] = 5      # >= ntri -> skipped
    tind[0, 4, 4, 0] = np.nan
    g = np.ones((1, H, W, 1), np.float32)
    vg = ops().render_depth_grad(_t(g), torch.zeros((1, 3, 4), device="cuda:0"), _t(tri), _t(g), _t(tind),
                                 torch.zeros((1, H, W, 3), device="cuda:0")).cpu().numpy()
    np.testing.assert_array_equal(vg, oracle.render_depth_grad(g, tri, tind, 4))
    np.testing.assert_array_equal(vg[0, 2], np.array([1 / 3, 1 / 3, 1 / 3, 0], np.float32))


def test_backward_is_bit_reproducible_and_the_rounded_exact_sum(oracle, full_assets, synth):
    """fr_render_depth_backward adds the contributions as exact fixed-point integers (order independent) and rounds once:
    two launches are bit-equal, and the result is the float64 sum of the fp32 terms (g*1.0f)/3.0f rounded to fp32, up to
    the stated quantisation n * 2^-39 * max|term| -- closer to the real sum than the reference's sequential fp32 order."""
    A = full_assets
    B = 4
    P = synth.sample_params_batch(B, beta=0.7, seed=21)
    V = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    dev = "cuda:0"
    img = torch.zeros((B, 200, 200, 3), device=dev)
    depth, _, _, tind = ops().render_depth(_t(V), _t(A["tri"]), _t(A["vertex"]), img)
    rs = np.random.RandomState(2)
    g = (rs.standard_normal((B, 200, 200, 1)) * np.exp(rs.uniform(-6, 6, (B, 200, 200, 1)))).astype(np.float32)
    runs = [ops().render_depth_grad(_t(g), _t(V), _t(A["tri"]), depth, tind, img).cpu().numpy() for _ in range(3)]
    np.testing.assert_array_equal(runs[0], runs[1])
    np.testing.assert_array_equal(runs[0], runs[2])
    got = runs[0]
    ti = tind.cpu().numpy().reshape(B, -1).astype(np.int64)
    tri = A["tri"].astype(np.int64)
    c = (g.reshape(B, -1) * np.float32(1.0)) / np.float32(3.0)          # fp32 terms, as render_depth_op.cc:361 forms them
    want = np.zeros((B, V.shape[2]), np.float64)
    nterm = np.zeros((B, V.shape[2]), np.int64)
    for b in range(B):
        cov = ti[b] >= 0
        for k in range(3):
            np.add.at(want[b], tri[k, ti[b][cov]], c[b][cov].astype(np.float64))
            np.add.at(nterm[b], tri[k, ti[b][cov]], 1)
    assert np.all(got[:, :2] == 0)
    cmax = np.abs(np.where(ti >= 0, c, 0)).max(axis=1)                  # per-face scale of the fixed-point grid
    bound = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) * 0.5 + \
        (nterm + 1) * 2.0 ** -38 * cmax[:, None]
    err = np.abs(got[:, 2].astype(np.float64) - want)
    assert np.all(err <= bound), float((err / bound).max())
    # and it agrees with the oracle's sequential fp32 order to that order's own rounding error
    seq = oracle.render_depth_grad(g, A["tri"], tind.cpu().numpy(), V.shape[2])
    np.testing.assert_allclose(got[:, 2], seq[:, 2], rtol=0, atol=float(1e-5 * max(1.0, cmax.max())))


def test_backward_inf_nan_gradients_and_small_batch_owners():
    """Inf / NaN depth gradients cannot be scaled to the fixed-point grid: that face takes fp32 LDS atomics and the class
    of every result (NaN / +-Inf / finite) matches the sequential sum; B = 1 exercises the many-owners geometry."""
    H, W = 6, 5
    tri = np.array([[0, 1, 2], [2, 3, 4], [1, 3, 5]], np.float32).T.copy()
    tind = np.array([[0, 0, 1, 1, -1], [2, 2, 2, 0, 1], [1, 1, -1, -1, 2], [0, 2, 1, 0, 0], [-1, -1, -1, -1, -1],
                     [2, 1, 0, 2, 1]], np.float32).reshape(1, H, W, 1)
    g = np.arange(H * W, dtype=np.float32).reshape(1, H, W, 1) - 7
    want = np.zeros(6, np.float64)
    for i in range(H * W):
        t = int(tind.reshape(-1)[i])
        if t >= 0:
            for k in range(3):
                want[int(tri[k, t])] += np.float32(g.reshape(-1)[i] / np.float32(3.0))
    z = torch.zeros((1, H, W, 3), device="cuda:0")
    vg = ops().render_depth_grad(_t(g), torch.zeros((1, 3, 6), device="cuda:0"), _t(tri), _t(g), _t(tind), z).cpu().numpy()
    np.testing.assert_allclose(vg[0, 2], want, rtol=1e-6, atol=1e-6)
    g2 = g.copy()
    g2[0, 0, 0, 0] = np.inf       # triangle 0 -> vertices 0, 1, 2
    g2[0, 1, 0, 0] = np.nan       # triangle 2 -> vertices 1, 3, 5
    vg2 = ops().render_depth_grad(_t(g2), torch.zeros((1, 3, 6), device="cuda:0"), _t(tri), _t(g2), _t(tind), z).cpu().numpy()
    assert np.isposinf(vg2[0, 2, 0]) and np.isposinf(vg2[0, 2, 2])
    assert np.isnan(vg2[0, 2, 1]) and np.isnan(vg2[0, 2, 3]) and np.isnan(vg2[0, 2, 5])
    assert np.isfinite(vg2[0, 2, 4]) and abs(vg2[0, 2, 4] - want[4]) < 1e-5


def test_workspace_and_plain_entry_points_agree(oracle, full_assets, synth):
    """fr_render_depth_backward_ws (packed triangle table, one id gather per pixel) == fr_render_depth_backward (float
    ids, three gathers), bit for bit; bad ids / background handled identically."""
    import ctypes
    from conftest import pkg
    h = pkg("_lib")
    L = h.lib()
    A = full_assets
    B = 3
    P = synth.sample_params_batch(B, beta=0.7, seed=8)
    V = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    img = torch.zeros((B, 200, 200, 3), device="cuda:0")
    tri = A["tri"].copy()
    tri[1, 100:140] = 60000.0          # out-of-range ids: those triangles are skipped everywhere
    depth, _, _, tind = ops().render_depth(_t(V), _t(tri), _t(A["vertex"]), img)
    g = _t(np.random.RandomState(4).standard_normal((B, 200, 200, 1)))
    got_ws = ops().render_depth_grad(g, _t(V), _t(tri), depth, tind, img)
    plain = torch.empty_like(got_ws)
    rc = L.fr_render_depth_backward(h.ptr(g), h.ptr(_t(tri)), h.ptr(tind), h.ptr(plain), B, V.shape[2], tri.shape[1], 200,
                                    200, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(got_ws, plain)
    assert float(got_ws[:, 2].abs().max()) > 0


def test_backward_of_an_image_above_2_pow_20_pixels(oracle):
    """the forward renders a 1,100 x 1,000 image (scan fallback: the strip does not fit the binned path's
    geometry) -- its backward must not refuse it.  Above 2^20 pixels the fixed-point scale gives up one bit per doubling
    (header): still bit-reproducible, within the stated bound of the float64 sum; both entry points agree."""
    import ctypes
    from conftest import pkg
    h = pkg("_lib")
    L = h.lib()
    rs = np.random.RandomState(12)
    H, W, nver = 1100, 1000, 400
    assert H * W > (1 << 20)
    ver = np.empty((1, 3, nver), np.float32)
    ver[0, 0] = rs.uniform(0, W, nver)
    ver[0, 1] = rs.uniform(0, H, nver)
    ver[0, 2] = rs.uniform(-5, 5, nver)
    tri = rs.randint(0, nver, (3, 300)).astype(np.float32)
    tex = rs.uniform(0, 1, (1, 3, nver)).astype(np.float32)
    img = torch.zeros((1, H, W, 3), device="cuda:0")
    depth, _, _, tind = ops().render_depth(_t(ver), _t(tri), _t(tex), img)
    want_f = oracle.render_depth(ver, tri, tex, H, W)
    np.testing.assert_array_equal(tind.cpu().numpy(), want_f[3])
    assert (want_f[3] >= 0).sum() > 50000
    g = rs.standard_normal((1, H, W, 1)).astype(np.float32)
    a = ops().render_depth_grad(_t(g), _t(ver), _t(tri), depth, tind, img)
    b = ops().render_depth_grad(_t(g), _t(ver), _t(tri), depth, tind, img)
    plain = torch.empty_like(a)
    rc = L.fr_render_depth_backward(h.ptr(_t(g)), h.ptr(_t(tri)), h.ptr(tind), h.ptr(plain), 1, nver, tri.shape[1], H, W,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, plain)
    # float64 sum of the fp32 terms c = g / 3
    ti = want_f[3].reshape(-1).astype(np.int64)
    c = (g.reshape(-1) * np.float32(1.0) / np.float32(3.0)).astype(np.float64)
    want = np.zeros(nver, np.float64)
    cnt = np.zeros(nver, np.int64)
    for k in range(3):
        p = tri[k].astype(np.int64)[ti[ti >= 0]]
        np.add.at(want, p, c[ti >= 0])
        np.add.at(cnt, p, 1)
    got = a.cpu().numpy()[0, 2].astype(np.float64)
    cmax = np.abs(c).max()
    bound = 0.5 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) + cnt * cmax * 2.0 ** -37
    assert np.all(np.abs(got - want) <= bound + 1e-300)
    assert not a[:, :2].any()
