"""CPU: the oracle against pieces of the reference that run in the build container.

  * PointInTri (render_depth_op.cc:76-122) compiled from /root/reference by oracle/Makefile into oracle/_ref
    (binary only; travels to the GPU box, so this test also runs there);
  * rotation_matrix / get_random_params: fixtures made by executing the reference's numpy functions
    (tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _pit_cases():
    rs = np.random.RandomState(11)
    n = 20000
    P = np.empty((0, 8))
    # sub-pixel triangles around integer pixel centres (the BFM regime), fp32-valued vertices
    c = rs.randint(0, 200, (n, 2)).astype(np.float64)
    tri = (c[:, None, :] + rs.uniform(-1.2, 1.2, (n, 3, 2))).astype(np.float32).astype(np.float64)
    P = np.concatenate([P, np.concatenate([c, tri.reshape(n, 6)], 1)])
    # integer vertices: pixel centres exactly on edges / hypotenuse / vertices
    v = rs.randint(0, 8, (n, 3, 2)).astype(np.float64)
    q = rs.randint(0, 8, (n, 2)).astype(np.float64)
    P = np.concatenate([P, np.concatenate([q, v.reshape(n, 6)], 1)])
    # degenerate (collinear / repeated) triangles
    a = rs.uniform(0, 10, (2000, 2))
    d = rs.uniform(-3, 3, (2000, 2))
    deg = np.stack([a, a + d, a + 2.5 * d], 1).astype(np.float32).astype(np.float64)
    q = rs.randint(0, 12, (2000, 2)).astype(np.float64)
    P = np.concatenate([P, np.concatenate([q, deg.reshape(2000, 6)], 1)])
    # huge / tiny magnitudes, NaN, inf
    big = rs.uniform(-1, 1, (2000, 8)) * 10.0 ** rs.randint(-30, 30, (2000, 1))
    P = np.concatenate([P, big])
    sp = rs.uniform(0, 5, (64, 8))
    sp[np.arange(64), rs.randint(0, 8, 64)] = np.where(np.arange(64) % 2 == 0, np.nan, np.inf)
    P = np.concatenate([P, sp])
    return P


def test_point_in_tri_vs_reference_binary(oracle):
    if not oracle.ref_point_in_tri_available():
        pytest.skip("oracle/_ref/libref_pit.so not built (reference tree absent)")
    P = _pit_cases()
    ref = oracle.ref_point_in_tri_batch(P)
    ours = oracle.point_in_tri_op_batch(P)
    assert ref.sum() > 1000 and (~ref).sum() > 1000
    np.testing.assert_array_equal(ours, ref)


def test_point_in_tri_edge_rules(oracle):
    # op: u+v<1 excludes the hypotenuse; mex: u+v<=1 includes it (SURVEY 8a deviation 5)
    p1, p2, p3 = (1, 1), (4, 1), (1, 4)
    assert oracle.point_in_tri_op((1, 1), p1, p2, p3) and oracle.point_in_tri_mex((1, 1), p1, p2, p3)
    assert not oracle.point_in_tri_op((2, 3), p1, p2, p3)
    assert oracle.point_in_tri_mex((2, 3), p1, p2, p3)
    # degenerate: den == 0 -> inside
    assert oracle.point_in_tri_op((0, 7), (1, 1), (4, 4), (2.5, 2.5))


def test_rotation_vs_reference_fixture(oracle):
    z = np.load(os.path.join(GOLDEN, "rotation_ref.npz"))
    R = oracle.rotation_matrix_batch(z["angles"])
    np.testing.assert_array_equal(R, z["R"])


def test_sampler_vs_reference_fixture(synth):
    z = np.load(os.path.join(GOLDEN, "sampler_ref.npz"))
    for tag in "abc":
        seed, beta = z["cfg_" + tag]
        rs = np.random.RandomState(int(seed))
        pose, shp, exp = synth.get_random_params(200, 199, 29, float(beta), rand=rs.rand)
        for name, ours in (("pose_", pose), ("shape_", shp), ("exp_", exp)):
            ref = z[name + tag]
            assert ours.dtype == ref.dtype and ours.shape == ref.shape
            np.testing.assert_array_equal(ours, ref)


def test_set_constraints_vs_reference_structure_fixture(oracle):
    """oracle.set_constraints == the reference's own set_constraints body (nets/network.py:204-218) executed with numpy
    standing in for its three tf calls (tests/golden/make_golden.py::gen_set_constraints), bit for bit."""
    z = np.load(os.path.join(GOLDEN, "set_constraints_ref.npz"))
    for tag in "abc":
        n_shape, n_exp, im = (int(v) for v in z["cfg_" + tag])
        got = oracle.set_constraints(z["raw_" + tag], im, 7, n_shape)
        want = z["out_" + tag]
        assert got.dtype == want.dtype and got.shape == want.shape
        np.testing.assert_array_equal(got, want)
        q = want.reshape(-1, 7 + n_shape + n_exp)
        assert np.all(np.abs(q[:, 0:3]) <= 1.5) and np.all((q[:, 3:5] >= 0) & (q[:, 3:5] <= im))
        assert np.all(q[:, 5] == 0) and np.all((q[:, 6] >= 0) & (q[:, 6] <= np.float32(1e-3)))
        assert np.all((q[:, 7:7 + n_shape] >= 0) & (q[:, 7:7 + n_shape] <= 1e4)) and np.all(np.abs(q[:, 7 + n_shape:]) <= 1.5)
    a = z["out_a"][0, 0, 0]
    np.testing.assert_array_equal(a[:4], np.array([0.0, -1.5, 1.5, 100.0], np.float32))   # sigmoid(0, -120, 120, -0)
