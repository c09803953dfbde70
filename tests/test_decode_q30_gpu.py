"""GPU parity of the opt-in Q30 decode arithmetic (csrc/fr_decode_q.hip: exact fixed-point basis blend on the int8
matrix cores) against its written CPU specification (oracle/fr_oracle.c "Q30 decode").  Bar: BIT-EXACT -- the arithmetic
is integer plus exactly specified float64 steps, so there is no tolerance to state -- and, against the reference's own
fp32 arithmetic (the f32 chain oracle) and the float64 evaluation, the tolerances written in the tests."""
import numpy as np
import pytest
import torch

from conftest import pkg
from gpu_util import net_mod

pytestmark = pytest.mark.gpu

Q30, F32 = 0, 1


@pytest.fixture(params=[7, 5, 4], ids=["levels7", "levels5", "levels4"])
def q30_mode(request):
    """The Q30 arithmetic with 7 (all sixteen digit products), 5 or 4 levels kept; `q30_mode.levels` names the spec."""
    h = pkg("_lib")
    prev, prev_lv = h.decode_arith(), h.q30_levels()
    h.set_decode_arith(Q30, request.param)
    assert h.decode_arith() == Q30 and h.q30_levels() == request.param
    h.levels = request.param
    yield h
    h.set_decode_arith(prev, prev_lv)


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


def test_setter_rejects_unknown_mode():
    h = pkg("_lib")
    prev = h.decode_arith()
    with pytest.raises(ValueError):
        h.set_decode_arith(7)
    with pytest.raises(ValueError):
        h.set_decode_arith(Q30, 6)
    assert h.decode_arith() == prev


@pytest.mark.parametrize("gu,gv,ns,ne,B", [
    (20, 24, 9, 5, 3),       # one k-step, 1 live 16-k group
    (7, 9, 1, 1, 1),         # N=63: ragged last tile, single coefficients
    (13, 17, 199, 29, 17),   # the model's shape: streaming kernel, two column blocks
    (12, 31, 199, 29, 5),    # streaming kernel, one column block
    (10, 23, 199, 29, 133),  # streaming kernel: two full 64-face passes + 5 faces
    (9, 14, 199, 29, 48),    # three live column blocks (the fourth is zero digits)
    (15, 16, 200, 17, 40),   # 217 coefficients: 14 groups, generic kernel
    (11, 19, 33, 16, 64),
    (9, 10, 40, 7, 65),      # second pass with a single face
    (6, 8, 256, 0, 20),      # exactly four full k-steps, no expression basis
    (5, 7, 300, 100, 33),    # 400 coefficients: seven k-steps
    (6, 8, 0, 0, 4),         # no basis at all: v = mu
    (6, 9, 64, 0, 16),       # exactly one full k-step (payload cannot ride in a short fragment)
])
def test_vs_q30_spec_bit_exact(q30_mode, oracle, synth, gu, gv, ns, ne, B):
    A = synth.make_assets(gu, gv, ns, ne, patch=None, seed_basis=gu * gv)
    P = _rand_params(np.random.RandomState(B), B, ns, ne, 200)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
    R = oracle.rotation_matrix_batch(P[:, :3])
    want = oracle.decode_3dmm_q30(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R, levels=q30_mode.levels)
    got = _decode_gpu(net, P, R)
    np.testing.assert_array_equal(got, want)
    for sched in (1,):   # the two-halves schedule: the same bits
        with q30_mode.options(FR_Q30_SCHED=sched):
            np.testing.assert_array_equal(_decode_gpu(net, P, R), want, err_msg="FR_Q30_SCHED=%d" % sched)
    # against the reference's arithmetic type (f32 chain spec): both are within a few ulp of the float64 evaluation
    chain = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R)
    truth = oracle.decode_3dmm_f64(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    scale = max(float(np.abs(truth).max()), float(np.abs(P[:, 3:6]).max()), 1.0)
    got_r = _decode_gpu(net, P)          # in-kernel float64 rotation for the comparison with the f64 formula
    assert np.max(np.abs(got_r - truth)) / scale < 4 * 2.0 ** -23
    assert np.max(np.abs(got - chain)) / scale < 4 * 2.0 ** -23


def test_special_values(q30_mode, oracle, synth):
    """Inf / NaN parameters make that face NaN and nothing else; subnormal, zero and huge coefficients, an all-zero face,
    and columns of wildly different scale follow the spec bit for bit."""
    A = synth.make_assets(9, 13, 199, 29, patch=None, seed_basis=5)
    rs = np.random.RandomState(3)
    B = 12
    P = _rand_params(rs, B, 199, 29, 200)
    P[1, 7 + 5] = np.inf
    P[2, 7 + 200] = np.nan
    P[3, 7:] = 0.0
    P[4, 7:] = 0.0
    P[4, 7 + 3] = 1e-41          # subnormal fp32 parameter
    P[5, 7 + 17] = 3e38          # near the top of the fp32 range
    P[6, 7:] *= 1e-30
    P[7, 7:7 + 199:2] = 0.0
    P[8, 7 + 198] = -1e7
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
    got = _decode_gpu(net, P, R)
    want = oracle.decode_3dmm_q30(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R, levels=q30_mode.levels)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(got[~np.isnan(want)], want[~np.isnan(want)])
    with q30_mode.options(FR_Q30_SCHED=1):
        np.testing.assert_array_equal(_decode_gpu(net, P, R), got)
    assert np.isnan(got[1]).all() and np.isnan(got[2]).all() and not np.isnan(got[[0, 3, 4, 5, 6, 7, 8]]).any()


def test_special_basis(q30_mode, oracle, synth):
    """A NaN / Inf basis entry poisons its own vertex coordinate only; an all-zero column, an all-zero row and a column
    1e30 times larger than the rest (the per-column exponent's job) follow the spec."""
    A = dict(synth.make_assets(6, 8, 37, 11, patch=None, seed_basis=1))
    A["pc_shape"] = A["pc_shape"].copy()
    A["pc_exp"] = A["pc_exp"].copy()
    A["pc_shape"][7, 0] = np.nan
    A["pc_exp"][9, 2] = np.inf
    A["pc_shape"][:, 4] = 0.0
    A["pc_shape"][20, :] = 0.0
    A["pc_exp"][20, :] = 0.0
    A["pc_shape"][:, 9] *= np.float32(1e30)
    A["pc_exp"][:, 1] *= np.float32(1e-30)
    P = _rand_params(np.random.RandomState(2), 5, 37, 11, 200)
    P[:, 7 + 9] *= np.float32(1e-30)
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=5, im_size=200)
    got = _decode_gpu(net, P, R)
    want = oracle.decode_3dmm_q30(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R, levels=q30_mode.levels)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(got[~np.isnan(want)], want[~np.isnan(want)])
    N = 48
    assert np.isnan(got[:, :, 7 % N]).all() and np.isnan(got[:, :, 9 % N]).all() and np.isnan(want).mean() < 0.1


def test_full_size_batch64(q30_mode, oracle, full_assets, synth):
    """BASELINE configs[1] shape (N = 53,215, 199 + 29 coefficients, B = 64): spec check on 2 faces, size-independent
    properties on the rest, accuracy against the float64 blend compared with the f32 chain's."""
    A = full_assets
    P = synth.sample_params_batch(64, beta=0.7, seed=3456)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=64, im_size=200)
    R = oracle.rotation_matrix_batch(P[:, :3])
    got = _decode_gpu(net, P, R)
    for b in (0, 63):
        want = oracle.decode_3dmm_q30(P[b:b + 1], A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R[b:b + 1], levels=q30_mode.levels)
        np.testing.assert_array_equal(got[b:b + 1], want)
    for sched in (1,):
        with q30_mode.options(FR_Q30_SCHED=sched):
            np.testing.assert_array_equal(_decode_gpu(net, P, R), got, err_msg="FR_Q30_SCHED=%d" % sched)
    # deterministic; a face's result does not depend on what else is in the batch (its scale is its own)
    np.testing.assert_array_equal(_decode_gpu(net, P, R), got)
    perm = np.random.RandomState(1).permutation(64)
    np.testing.assert_array_equal(_decode_gpu(net, P[perm], R[perm]), got[perm])
    np.testing.assert_array_equal(_decode_gpu(net, P[5:22], R[5:22]), got[5:22])
    # accuracy of the blend itself: identity pose, so x and z ARE v = mu + S + E
    P2 = P[:4].copy()
    P2[:, 3:6] = 0
    P2[:, 6] = 1.0
    I = np.tile(np.eye(3, dtype=np.float32).reshape(1, 3, 3), (4, 1, 1))
    vq = _decode_gpu(net, P2, I)
    q30_mode.set_decode_arith(F32)
    vc = _decode_gpu(net, P2, I)
    q30_mode.set_decode_arith(Q30, q30_mode.levels)
    Ab = np.concatenate([A["pc_shape"], A["pc_exp"]], 1).astype(np.float64)
    vt = (A["mu"].reshape(-1).astype(np.float64)[None] + P2[:, 7:].astype(np.float64) @ Ab.T).reshape(4, 3, -1)
    cr = vt.astype(np.float32)
    for c in (0, 2):
        eq, ec = np.abs(vq[:, c] - vt[:, c]), np.abs(vc[:, c] - vt[:, c])
        # levels 7 / 5: the correctly rounded fp32 value almost everywhere; levels 4: still ahead of the f32 chain
        assert eq.mean() < (0.8 if q30_mode.levels >= 5 else 0.95) * ec.mean() and eq.max() <= ec.max()
        assert (vq[:, c] == cr[:, c]).mean() > (0.99 if q30_mode.levels >= 5 else 0.95)
        assert (vc[:, c] == cr[:, c]).mean() < 0.9           # (the f32 chain is not)
