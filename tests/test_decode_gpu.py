"""GPU parity: fr_decode_3dmm (HIP MFMA kernel, through nets/network.py and the C ABI) vs the CPU oracle.
Bar: bit-exact against the written fp32 spec (oracle/fr_oracle.c, fmaf chains == gfx950 f32 MFMA) when the
rotation is supplied by the host (as the reference does via tf.py_func); with the in-kernel float64 rotation,
R may differ from glibc's by the last fp64 bit of sin/cos, so that leg is held to 2 ulp of the output; and
within 4 fp32 ulp of the magnitude against a float64 evaluation of the same formula."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gpu_util import net_mod

pytestmark = pytest.mark.gpu


def _decode_gpu(net, P, R=None):
    p = torch.as_tensor(P, device="cuda:0")
    out = net.vertices_transform(p, R=None if R is None else torch.as_tensor(R, device="cuda:0"))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _rand_params(rs, B, ns, ne, im):
    P = np.zeros((B, 7 + ns + ne), np.float32)
    P[:, 0:3] = rs.uniform(-1.5, 1.5, (B, 3))
    P[:, 3:5] = rs.uniform(0, im, (B, 2))
    P[:, 5] = rs.uniform(-1, 1, B)
    P[:, 6] = rs.uniform(0, 1e-3, B)
    P[:, 7:7 + ns] = rs.uniform(0, 1e4, (B, ns))
    P[:, 7 + ns:] = rs.uniform(-1.5, 1.5, (B, ne))
    return P


def test_golden_small(small_assets):
    z = np.load(os.path.join(GOLDEN, "decode_small_oracle.npz"))
    net = net_mod().FaceRecNet(mesh_data=small_assets, batch_size=3, im_size=float(z["im_size"]))
    got = _decode_gpu(net, z["params"], z["R"])
    np.testing.assert_array_equal(got, z["vertex_proj"])


@pytest.mark.parametrize("gu,gv,ns,ne,B", [
    (20, 24, 9, 5, 3),       # N=480
    (7, 9, 1, 1, 1),         # N=63 (< 4 tiles, ragged last tile), single components
    (13, 17, 199, 29, 17),   # real component counts (the ring-scheduled kernel), B not a multiple of 16
    (12, 31, 199, 29, 5),    # ring kernel, one column block (16-column items)
    (10, 23, 199, 29, 133),  # ring kernel: one 128-column pass + 5 columns
    (8, 21, 199, 29, 128),   # exactly one 128-column pass
    (7, 15, 199, 29, 70),    # 128-column pass with 70 live columns (second half nearly empty)
    (6, 11, 199, 29, 300),   # two wide passes + 44 columns
    (15, 16, 200, 17, 40),   # 13 + 2 groups again (200 -> 13, 17 -> 2) with different paddings
    (11, 19, 33, 16, 64),
    (9, 10, 40, 7, 65),      # > 64 columns: second pass
    (6, 8, 20, 3, 130),
])
def test_vs_oracle_bit_exact(oracle, synth, gu, gv, ns, ne, B):
    A = synth.make_assets(gu, gv, ns, ne, patch=None, seed_basis=gu * gv)
    rs = np.random.RandomState(B)
    P = _rand_params(rs, B, ns, ne, 200)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
    R = oracle.rotation_matrix_batch(P[:, :3])
    want = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R)
    np.testing.assert_array_equal(_decode_gpu(net, P, R), want)
    # in-kernel float64 rotation
    got = _decode_gpu(net, P)
    ulp = np.spacing(np.maximum(np.abs(want), np.float32(1.0)))
    assert np.all(np.abs(got - want) <= 2 * ulp)
    assert (got == want).mean() > 0.99
    # fp64 truth
    truth = oracle.decode_3dmm_f64(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    # absolute error against the largest magnitude in play (the projection sums terms of that size)
    scale = max(float(np.abs(truth).max()), float(np.abs(P[:, 3:6]).max()), 1.0)
    assert np.max(np.abs(got - truth)) / scale < 4 * 2.0 ** -23


def test_zero_batch(small_assets):
    net = net_mod().FaceRecNet(mesh_data=small_assets, batch_size=1, im_size=200)
    out = net.vertices_transform(torch.zeros((0, net.ndim), device="cuda:0"))
    assert tuple(out.shape) == (0, 3, net.nvert)


def test_nan_basis_row_does_not_leak(oracle, synth):
    # padded k-steps must not let a NaN/Inf basis entry of one vertex reach another vertex
    A = synth.make_assets(6, 8, 5, 3, patch=None, seed_basis=1)
    A = dict(A)
    A["pc_shape"] = A["pc_shape"].copy()
    A["pc_shape"][7, 0] = np.nan
    A["pc_exp"] = A["pc_exp"].copy()
    A["pc_exp"][9, 2] = np.inf
    P = _rand_params(np.random.RandomState(2), 4, 5, 3, 200)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=4, im_size=200)
    R = oracle.rotation_matrix_batch(P[:, :3])
    got = _decode_gpu(net, P, R)
    want = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R)
    np.testing.assert_array_equal(np.isfinite(got), np.isfinite(want))
    np.testing.assert_array_equal(got[np.isfinite(want)], want[np.isfinite(want)])
    assert np.isfinite(want).mean() > 0.9


def test_full_size_batch64(oracle, full_assets, synth):
    """BASELINE config 2 shape: N=53,215, 199+29 components, B=64.  Oracle check on ALL 64 faces (~36 ms per face),
    plus size-independent properties."""
    A = full_assets
    P = synth.sample_params_batch(64, beta=0.7, seed=3456)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=64, im_size=200)
    R = oracle.rotation_matrix_batch(P[:, :3])
    got = _decode_gpu(net, P, R)
    want = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R)
    for b in range(64):
        np.testing.assert_array_equal(got[b], want[b], err_msg="face %d" % b)
    # determinism + independence of the batch composition (each column is its own fmaf chain)
    np.testing.assert_array_equal(_decode_gpu(net, P, R), got)
    perm = np.random.RandomState(1).permutation(64)
    np.testing.assert_array_equal(_decode_gpu(net, P[perm], R[perm]), got[perm])
    np.testing.assert_array_equal(_decode_gpu(net, P[5:22], R[5:22]), got[5:22])
    # y-flip / translation structure: shifting tx, ty moves x, y by the same amount up to fp32 rounding
    P2 = P.copy()
    P2[:, 3] += 8.0
    P2[:, 4] -= 4.0
    got2 = _decode_gpu(net, P2, R)
    assert np.max(np.abs((got2[:, 0] - got[:, 0]) - 8.0)) < 1e-4
    assert np.max(np.abs((got2[:, 1] - got[:, 1]) - 4.0)) < 1e-4
    np.testing.assert_array_equal(got2[:, 2], got[:, 2])


@pytest.mark.parametrize("knobs", [{"FR_DECODE_NT": 0}, {"FR_DECODE_NT": 1}, {"FR_DECODE_NBW": 1}, {"FR_DECODE_WAVES": 8}, {"FR_DECODE_IMPL": 1},
                                   {"FR_DECODE_WIDE": 0}])
def test_launcher_knobs_do_not_change_a_bit(oracle, synth, knobs):
    """The decode launcher's A/B knobs (fr_set_option; the environment is read once per process) select other schedules of
    the same arithmetic: the model's basis shape at B = 70 (one 128-column pass, or 64 + 6 with FR_DECODE_WIDE=0) stays
    bit-identical to the default and to the oracle."""
    from conftest import pkg
    A = synth.make_assets(9, 11, 199, 29, patch=None, seed_basis=3)
    B = 70
    P = _rand_params(np.random.RandomState(6), B, 199, 29, 200)
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
    base = _decode_gpu(net, P, R)
    np.testing.assert_array_equal(base, oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R))
    with pkg("_lib").options(**knobs):
        np.testing.assert_array_equal(_decode_gpu(net, P, R), base)
        np.testing.assert_array_equal(_decode_gpu(net, P[:33], R[:33]), base[:33])
