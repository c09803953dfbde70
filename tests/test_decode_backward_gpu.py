"""GPU parity: fr_decode_3dmm_backward (HIP, MFMA reduction over the vertices) vs a float64 evaluation of the gradient
TF autodiff derives from nets/network.py:140-171.  The kernels sum in fixed-order fp32 partials, so the comparison is a
tolerance relative to the magnitude of the summed terms; the result itself is bit-reproducible run to run."""
import numpy as np
import pytest
import torch

from gpu_util import net_mod, ops

pytestmark = pytest.mark.gpu


def _params(rs, B, ns, ne):
    P = np.zeros((B, 7 + ns + ne), np.float32)
    P[:, 0:3] = rs.uniform(-1.0, 1.0, (B, 3))
    P[:, 3:5] = rs.uniform(60, 140, (B, 2))
    P[:, 5] = rs.uniform(-1, 1, B)
    P[:, 6] = rs.uniform(2e-4, 1e-3, B)
    P[:, 7:7 + ns] = rs.uniform(0, 1e4, (B, ns))
    P[:, 7 + ns:] = rs.uniform(-1.5, 1.5, (B, ne))
    return P


def _check(oracle, A, P, G, tol=2e-5, from_mu=False):
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=P.shape[0], im_size=200)
    net._basis.backward_from_mu = from_mu      # the autograd node's form of d f (nets/network.py::_Decode3DMM)
    p = torch.as_tensor(P, device="cuda:0").requires_grad_(True)
    V = net.vertices_transform(p)
    V.backward(torch.as_tensor(G, device="cuda:0"))
    got = p.grad.cpu().numpy().astype(np.float64)
    want = oracle.decode_3dmm_backward_f64(G, P, A["mu"], A["pc_shape"], A["pc_exp"])
    ns = A["ndim_shape"]
    assert np.all(got[:, 0:3] == 0) and np.all(want[:, 0:3] == 0)       # angles: no gradient (tf.py_func)
    # scale of the sums behind each block of outputs
    for sl in (slice(3, 6), slice(6, 7), slice(7, 7 + ns), slice(7 + ns, None)):
        scale = np.abs(want[:, sl]).max() + 1e-30
        err = np.abs(got[:, sl] - want[:, sl]).max() / scale
        assert err < tol, (sl, err)
    return got


@pytest.mark.parametrize("gu,gv,ns,ne,B", [(7, 9, 3, 2, 2), (20, 24, 9, 5, 5), (13, 17, 199, 29, 17), (9, 10, 40, 7, 65)])
def test_vs_f64_small(oracle, synth, gu, gv, ns, ne, B):
    A = synth.make_assets(gu, gv, ns, ne, patch=None, seed_basis=gu + gv)
    rs = np.random.RandomState(B)
    P = _params(rs, B, ns, ne)
    G = rs.standard_normal((B, 3, gu * gv)).astype(np.float32)
    a = _check(oracle, A, P, G)
    b = _check(oracle, A, P, G, from_mu=True)      # no forward output kept for the backward: the same gradient, d f to rounding
    keep = np.ones(a.shape[1], bool)
    keep[6] = False
    np.testing.assert_array_equal(a[:, keep], b[:, keep])


@pytest.mark.parametrize("gu,gv", [(2, 3), (3, 5), (4, 4)])
def test_tiny_mesh_takes_the_reference_layout_path(oracle, synth, gu, gv):
    """the fused (packed) backward loads sixteen floats per tile row, so a mesh of fewer than 16 vertices must
    not reach it -- fr_decode_backward_basis_bytes answers 0, the pack / packed entry points FR_ERR_UNSUPPORTED, and the autograd
    node falls back to the reference-layout kernels; N = 16 is the first mesh the packed path serves."""
    from conftest import pkg
    h = pkg("_lib")
    L = h.lib()
    N = gu * gv
    A = synth.make_assets(gu, gv, 9, 5, patch=None, seed_basis=N)
    assert (L.fr_decode_backward_basis_bytes(N, 9, 5) == 0) == (N < 16)
    rs = np.random.RandomState(N)
    P = _params(rs, 4, 9, 5)
    G = rs.standard_normal((4, 3, N)).astype(np.float32)
    _check(oracle, A, P, G)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=4, im_size=200)
    assert (net._basis.image_t() is None) == (N < 16)
    if N < 16:
        one = torch.zeros((1 << 16,), dtype=torch.uint8, device="cuda:0")
        assert L.fr_decode_backward_pack_basis(h.ptr(net._basis.pc_shape), h.ptr(net._basis.pc_exp), N, 9, 5, h.ptr(one), 1 << 16,
                                               None) == -4


def test_full_size_deterministic(oracle, full_assets):
    A = full_assets
    rs = np.random.RandomState(3)
    B = 3
    P = _params(rs, B, 199, 29)
    G = (rs.standard_normal((B, 3, 53215)) * rs.uniform(0, 1, (B, 3, 53215))).astype(np.float32)
    got1 = _check(oracle, A, P, G, tol=5e-5)
    got2 = _check(oracle, A, P, G, tol=5e-5)
    np.testing.assert_array_equal(got1, got2)      # no float atomics: bit-reproducible


def test_gradient_flows_from_depth_to_params(full_assets, synth):
    """CoarseNet -> render loop: d(depth loss)/d params through render backward (z only) and decode backward."""
    A = full_assets
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=2, im_size=200)
    P = torch.as_tensor(synth.sample_params_batch(2, beta=0.7, seed=4), device="cuda:0").requires_grad_(True)
    V = net.vertices_transform(P)
    depth = ops().render_depth(V, net.tri, net.vertex_code, torch.zeros((2, 200, 200, 3), device="cuda:0"))[0]
    loss = depth.clamp_min(1e-6).sum()
    loss.backward()
    g = P.grad
    assert bool(torch.isfinite(g).all())
    assert float(g[:, 0:3].abs().max()) == 0.0          # angles
    assert float(g[:, 3:5].abs().max()) == 0.0          # tx, ty: depth only feeds the z row
    assert float(g[:, 5].abs().min()) > 0.0             # tz: every covered pixel contributes g/3 * 3
    assert float(g[:, 7:].abs().max()) > 0.0
    # d loss / d tz == number of covered pixels with depth > 1e-6 (each hands out 3 * 1/3)
    ncov = (depth > 1e-6).sum(dim=(1, 2, 3)).float()
    assert torch.allclose(g[:, 5], ncov, rtol=1e-4)


def test_zero_focal_column(oracle, synth):
    """f == 0 in one batch column (include/fr_hotpath.h): d f is defined as 0 there, d alpha = d beta = 0 (dv = f R^T dq
    vanishes), d t3d = sum dq stays exact, and the other columns are untouched."""
    A = synth.make_assets(9, 10, 12, 4, patch=None, seed_basis=19)
    rs = np.random.RandomState(0)
    P = _params(rs, 3, 12, 4)
    P[1, 6] = 0.0
    G = rs.standard_normal((3, 3, 90)).astype(np.float32)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=3, im_size=200)
    p = torch.as_tensor(P, device="cuda:0").requires_grad_(True)
    net.vertices_transform(p).backward(torch.as_tensor(G, device="cuda:0"))
    got = p.grad.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(got))
    assert got[1, 6] == 0.0 and np.all(got[1, 7:] == 0.0) and np.all(got[1, 0:3] == 0.0)
    dq = G[1].astype(np.float64) * np.array([1.0, -1.0, 1.0])[:, None]
    np.testing.assert_allclose(got[1, 3:6], dq.sum(1), rtol=1e-5, atol=1e-5)
    keep = [0, 2]
    want = oracle.decode_3dmm_backward_f64(G[keep], P[keep], A["mu"], A["pc_shape"], A["pc_exp"])
    for sl in (slice(3, 6), slice(6, 7), slice(7, None)):
        scale = np.abs(want[:, sl]).max() + 1e-30
        assert np.abs(got[keep][:, sl] - want[:, sl]).max() / scale < 2e-5


def test_packed_and_reference_layout_entry_points(oracle, synth):
    """fr_decode_3dmm_backward (basis in its reference layout, no extra memory: prepass + GEMM + reduce),
    fr_decode_3dmm_backward_packed (packed image, ONE fused kernel + reduce) and fr_decode_3dmm_backward_packed_mu (the same
    kernel with d f formed from mu instead of the forward's output: what the autograd node calls when `basis.backward_from_mu` is set
    -- off by default, it saves 41 MB per decode and no time, nets/network.py::_Decode3DMM) are the same
    gradient with differently ordered -- each fixed -- partial sums: both within tolerance of the float64 gradient, each
    bit-reproducible, at a ragged shape (N = 187: the last vertex group is 11 vertices; 217 coefficients: 13 + 2 blocks, the
    last wave has one live), at the model's 199 + 29 with 70 faces (two passes) and at a one-block basis (staging-only waves)."""
    import ctypes
    from conftest import pkg
    h = pkg("_lib")
    L = h.lib()
    for gu, gv, ns, ne, B in ((11, 17, 200, 17, 5), (13, 17, 199, 29, 70), (7, 9, 3, 0, 2)):
        A = synth.make_assets(gu, gv, ns, ne, patch=None, seed_basis=gu * gv)
        N = gu * gv
        rs = np.random.RandomState(N + B)
        P = _params(rs, B, ns, ne)
        G = rs.standard_normal((B, 3, N)).astype(np.float32)
        net = net_mod().FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
        dev = torch.device("cuda:0")
        p = torch.as_tensor(P, device=dev)
        g = torch.as_tensor(G, device=dev)
        V = net.vertices_transform(p).detach()
        nws = L.fr_decode_backward_workspace_bytes(B, N, ns, ne)
        ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        outs = []
        for packed in (False, True, True, False, 2, 2):
            gp = torch.full_like(p, 7.0)
            if packed == 2:      # round 5: the same fused kernel with d f from mu -- no forward output in the call at all
                rc = L.fr_decode_3dmm_backward_packed_mu(h.ptr(g), h.ptr(p), h.ptr(net.mu), h.ptr(net._basis.image_t()), None, B, N,
                                                         ns, ne, 200.0, h.ptr(gp), h.ptr(ws), nws, st)
            elif packed:
                rc = L.fr_decode_3dmm_backward_packed(h.ptr(g), h.ptr(p), h.ptr(V), h.ptr(net._basis.image_t()), None, B, N, ns,
                                                      ne, 200.0, h.ptr(gp), h.ptr(ws), nws, st)
            else:
                rc = L.fr_decode_3dmm_backward(h.ptr(g), h.ptr(p), h.ptr(V), h.ptr(net.pc_shape), h.ptr(net.pc_exp), None, B, N,
                                               ns, ne, 200.0, h.ptr(gp), h.ptr(ws), nws, st)
            assert rc == 0
            torch.cuda.synchronize()
            outs.append(gp.cpu().numpy().astype(np.float64))
        np.testing.assert_array_equal(outs[0], outs[3])      # each entry point reproduces its own bits
        np.testing.assert_array_equal(outs[1], outs[2])
        np.testing.assert_array_equal(outs[4], outs[5])
        keep = np.ones(outs[1].shape[1], bool)
        keep[6] = False                                       # the mu form differs from the packed one in d f only -- bit for bit elsewhere
        np.testing.assert_array_equal(outs[4][:, keep], outs[1][:, keep])
        want = oracle.decode_3dmm_backward_f64(G, P, A["mu"], A["pc_shape"], A["pc_exp"])
        for got in (outs[0], outs[1], outs[4]):
            assert np.all(got[:, 0:3] == 0)
            for sl in (slice(3, 6), slice(6, 7), slice(7, 7 + ns), slice(7 + ns, None)):
                if sl.start >= got.shape[1]:
                    continue
                scale = np.abs(want[:, sl]).max() + 1e-30
                assert np.abs(got[:, sl] - want[:, sl]).max() / scale < 2e-5, (gu, gv, ns, ne, sl)
