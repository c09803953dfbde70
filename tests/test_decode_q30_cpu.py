"""CPU checks of the Q30 decode specification (oracle/fr_oracle.c "Q30 decode"): the restatement against an independent
big-integer evaluation in Python, and its accuracy against the float64 blend next to the reference-type f32 chain."""
import math

import numpy as np
import pytest


def _digits(q):
    """balanced base-256 digits of a Python int, most significant first (the spec's fr_q30_digits)"""
    d = []
    for _ in range(3):
        l = ((q + 128) & 255) - 128
        d.append(l)
        q = (q - l) >> 8
    d.append(q)
    return d[::-1]


def _kept(qa, qb, levels):
    """sum of the digit products a_i b_j 256^(6-i-j) with i + j < levels, as a Python int"""
    if levels == 7:
        return qa * qb
    da, db = _digits(qa), _digits(qb)
    return sum(da[i] * db[j] * 256 ** (6 - i - j) for i in range(4) for j in range(4) if i + j < levels)


def _q30_python(P, mu, ps, pe, levels=7):
    """v = mu + S + E per the written spec, with Python integers (digit by digit only where levels < 7; no level chain).
    Returns [B, 3N] float32 and the mask of entries whose integer sum stays below 2^52 (there the spec's level chain is
    exact, so the two must agree to the bit; above, the chain's intermediate float64 roundings may move the last float64
    bit)."""
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
            I = sum(_kept(int(np.rint(math.ldexp(float(A[r, k]), 30 - re - int(ce[k])))), qB[k], levels) for k in range(K))
            small[b, r] = abs(I) < 2 ** (52 + 8 * (7 - levels))
            out[b, r] = np.float32(float(mu[r]) + math.ldexp(float(I), re + be - 60))
    return out, small


@pytest.mark.parametrize("levels", [7, 5, 4])
def test_spec_vs_python_integers(oracle, synth, levels):
    A = synth.make_assets(5, 6, 21, 9, patch=None, seed_basis=4)
    rs = np.random.RandomState(0)
    P = np.zeros((3, 7 + 30), np.float32)
    P[:, 6] = 1.0
    P[:, 7:28] = rs.uniform(-1e4, 1e4, (3, 21))
    P[:, 28:] = rs.uniform(-1.5, 1.5, (3, 9))
    P[2, 7:] = 0
    P[2, 9] = 123.5
    I = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (3, 1))
    got = oracle.decode_3dmm_q30(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=I, levels=levels)
    want, small = _q30_python(P, A["mu"].reshape(-1), A["pc_shape"], A["pc_exp"], levels)
    want, small = want.reshape(3, 3, -1), small.reshape(3, 3, -1)
    assert small.any() and (levels < 7 or not small.all())   # both branches of the spec are exercised (levels = 7)
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


def test_levels_5_and_4_next_to_the_exact_product(oracle, synth):
    """Dropping the digit products below 2^-32 (levels 5) or 2^-24 (levels 4) of a term's full scale: levels 5 gives the same
    fp32 value as the exact product almost everywhere, levels 4 stays ahead of the f32 chain (what include/fr_hotpath.h
    states about the `levels` argument)."""
    full = synth.make_assets()
    N0 = full["mu"].shape[0] // 3
    sel = np.arange(0, N0, 37)
    rows = np.concatenate([sel, N0 + sel, 2 * N0 + sel])
    mu, ps, pe = full["mu"].reshape(-1)[rows], full["pc_shape"][rows], full["pc_exp"][rows]
    B = 6
    P = synth.sample_params_batch(B, im_size=200, beta=0.7, seed=12).astype(np.float32)
    P[:, 3:6] = 0
    P[:, 6] = 1.0
    I = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (B, 1))
    v7 = oracle.decode_3dmm_q30(P, mu, ps, pe, 200.0, R=I)
    v5 = oracle.decode_3dmm_q30(P, mu, ps, pe, 200.0, R=I, levels=5)
    v4 = oracle.decode_3dmm_q30(P, mu, ps, pe, 200.0, R=I, levels=4)
    vc = oracle.decode_3dmm(P, mu, ps, pe, 200.0, R=I)
    A = np.concatenate([ps, pe], 1).astype(np.float64)
    vt = (mu.astype(np.float64)[None] + P[:, 7:].astype(np.float64) @ A.T).reshape(B, 3, -1)
    for c in (0, 2):
        assert (v5[:, c] == v7[:, c]).mean() > 0.9995
        assert (v4[:, c] == v7[:, c]).mean() > 0.95
        e7, e4, ec = (np.abs(v[:, c] - vt[:, c]).mean() for v in (v7, v4, vc))
        assert e7 <= e4 < 0.95 * ec
    with pytest.raises(ValueError):
        oracle.decode_3dmm_q30(P, mu, ps, pe, 200.0, R=I, levels=0)
