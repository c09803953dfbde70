"""CPU checks of the Q30 decode specification (oracle/fr_oracle.c "Q30 decode"): the restatement against an independent
big-integer evaluation in Python, and its accuracy against the float64 blend next to the reference-type f32 chain."""
import math

import numpy as np


def _q30_python(P, mu, ps, pe):
    """v = mu + S + E per the written spec, with Python integers (no digits, no level chain).  Returns [B, 3N] float32 and
    the mask of entries whose integer sum stays below 2^52 (there the spec's level chain is exact, so the two must agree
    to the bit; above, the chain's intermediate float64 roundings may move the last float64 bit)."""
    A = np.concatenate([ps, pe], 1).astype(np.float64)
    rows, K = A.shape
    ce = np.zeros(K, int)
    for k in range(K):
        m = np.abs(A[:, k]).max()
        ce[k] = math.frexp(float(m))[1] if m > 0 else 0
    out = np.zeros((P.shape[0], rows), np.float32)
    small = np.zeros((P.shape[0], rows), bool)
    for b in range(P.shape[0]):
        x = P[b, 7:].astype(np.float64)
        es = [math.frexp(float(x[k]))[1] + int(ce[k]) for k in range(K) if x[k] != 0]
        be = max(es) if es else 0
        qB = [int(np.rint(math.ldexp(float(x[k]), int(ce[k]) + 30 - be))) for k in range(K)]
        for r in range(rows):
            ex = [math.frexp(float(A[r, k]))[1] - int(ce[k]) for k in range(K) if A[r, k] != 0]
            re = max(ex) if ex else 0
            I = sum(int(np.rint(math.ldexp(float(A[r, k]), 30 - re - int(ce[k])))) * qB[k] for k in range(K))
            small[b, r] = abs(I) < 2 ** 52
            out[b, r] = np.float32(float(mu[r]) + math.ldexp(float(I), re + be - 60))
    return out, small


def test_spec_vs_python_integers(oracle, synth):
    A = synth.make_assets(5, 6, 21, 9, patch=None, seed_basis=4)
    rs = np.random.RandomState(0)
    P = np.zeros((3, 7 + 30), np.float32)
    P[:, 6] = 1.0
    P[:, 7:28] = rs.uniform(-1e4, 1e4, (3, 21))
    P[:, 28:] = rs.uniform(-1.5, 1.5, (3, 9))
    P[2, 7:] = 0
    P[2, 9] = 123.5
    I = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (3, 1))
    got = oracle.decode_3dmm_q30(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=I)
    want, small = _q30_python(P, A["mu"].reshape(-1), A["pc_shape"], A["pc_exp"])
    want, small = want.reshape(3, 3, -1), small.reshape(3, 3, -1)
    assert small.any() and not small.all()                   # both branches of the spec are exercised
    for c in (0, 2):                                         # identity pose: x and z rows are v itself
        np.testing.assert_array_equal(got[:, c][small[:, c]], want[:, c][small[:, c]])
        assert np.all(np.abs(got[:, c] - want[:, c]) <= np.spacing(np.abs(want[:, c])))
        assert (got[:, c] == want[:, c]).mean() > 0.999


def test_level_chain_equals_wide_integer_path(oracle):
    """One coefficient with both operands at full scale: |I| ~ 2^60 takes the digit / level-sum branch of the spec; the
    product of two fp32 numbers is exact in float64, so the result must be the correctly rounded mu + a*x."""
    mu = np.array([1.0, -2.0, 3.0] * 1, np.float32)                      # N = 1
    ps = np.array([[0.999999], [-0.75], [0.5000001]], np.float32)
    pe = np.zeros((3, 0), np.float32)
    P = np.zeros((2, 8), np.float32)
    P[:, 6] = 1.0
    P[0, 7] = 12345.678
    P[1, 7] = -0.99999994
    I = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (2, 1))
    got = oracle.decode_3dmm_q30(P, mu, ps, pe, 200.0, R=I)
    for b in range(2):
        for c in (0, 2):
            want = np.float32(float(mu[c]) + float(ps[c, 0]) * float(P[b, 7]))
            assert got[b, c, 0] == want


def test_accuracy_next_to_the_f32_chain(oracle, synth):
    """On the model's data (every 23rd vertex of the bench mesh, the bench's parameter sampler) the Q30 blend is the
    correctly rounded fp32 value of the float64 blend almost everywhere; the f32 chain carries about twice its error."""
    full = synth.make_assets()
    N0 = full["mu"].shape[0] // 3
    sel = np.arange(0, N0, 23)
    rows = np.concatenate([sel, N0 + sel, 2 * N0 + sel])
    mu, ps, pe = full["mu"].reshape(-1)[rows], full["pc_shape"][rows], full["pc_exp"][rows]
    B = 8
    P = synth.sample_params_batch(B, im_size=200, beta=0.7, seed=11).astype(np.float32)
    P[:, 3:6] = 0
    P[:, 6] = 1.0
    I = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (B, 1))
    vq = oracle.decode_3dmm_q30(P, mu, ps, pe, 200.0, R=I)
    vc = oracle.decode_3dmm(P, mu, ps, pe, 200.0, R=I)
    A = np.concatenate([ps, pe], 1).astype(np.float64)
    vt = (mu.astype(np.float64)[None] + P[:, 7:].astype(np.float64) @ A.T).reshape(B, 3, -1)
    cr = vt.astype(np.float32)
    for c in (0, 2):
        eq, ec = np.abs(vq[:, c] - vt[:, c]), np.abs(vc[:, c] - vt[:, c])
        assert eq.mean() < 0.8 * ec.mean() and eq.max() <= ec.max()
        assert (vq[:, c] == cr[:, c]).mean() > 0.99
