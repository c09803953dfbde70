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
