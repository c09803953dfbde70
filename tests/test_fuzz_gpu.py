"""Seeded differential fuzzing of the HIP path against the CPU oracle: scene type, batch, image size, triangle count,
coordinate scale and special values are all drawn from the seed, so every case is reproducible by its id.  Bars as in
the targeted tests: forward planes bit-exact; backward bit-reproducible and within the stated bound of the oracle;
decode bit-exact against the spec of the selected arithmetic."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from gpu_util import assert_render_equal, net_mod, ops, render_gpu

pytestmark = pytest.mark.gpu

# FR_FUZZ_CASES=3000 widens the sweep (a couple of minutes); the default keeps the suite short
N_RENDER = int(os.environ.get("FR_FUZZ_CASES", "160"))
N_DECODE = max(1, N_RENDER * 3 // 10)


def _scene(seed):
    rs = np.random.RandomState(seed)
    B = int(rs.choice([1, 1, 2, 3, 5, 8, 17, 64, 70]))
    H = int(rs.choice([1, 2, 3, 5, 9, 16, 31, 40, 64, 100, 200, 257]))
    W = int(rs.choice([1, 2, 4, 7, 8, 9, 33, 64, 100, 200, 300]))
    if B * H * W > 3_000_000:
        B = max(1, 3_000_000 // (H * W))
    kind = int(rs.randint(0, 5))
    if kind == 0:      # jittered sub-pixel grid (the 3DMM regime), random triangle order
        nu, nv = max(2, int(H * rs.uniform(0.8, 1.6))), max(2, int(W * rs.uniform(0.8, 1.6)))
        gx, gy = np.meshgrid(np.linspace(-2.0, W + 1.0, nv), np.linspace(-2.0, H + 1.0, nu))
        nver = nu * nv
        ver = np.empty((B, 3, nver), np.float32)
        j = rs.uniform(0.0, 0.6)
        for b in range(B):
            ver[b, 0] = (gx + rs.uniform(-j, j, gx.shape)).reshape(-1)
            ver[b, 1] = (gy + rs.uniform(-j, j, gy.shape)).reshape(-1)
            ver[b, 2] = rs.uniform(-5, 5, nver)
        iu, iv = np.meshgrid(np.arange(nu - 1), np.arange(nv - 1), indexing="ij")
        v00 = (iu * nv + iv).reshape(-1)
        tri = np.concatenate([np.stack([v00, v00 + nv, v00 + 1]), np.stack([v00 + 1, v00 + nv, v00 + nv + 1])], 1)
        tri = tri[:, rs.permutation(tri.shape[1])]
    else:
        nver = int(rs.randint(3, 3000))
        ntri = int(rs.randint(1, 6000))
        ver = np.empty((B, 3, nver), np.float32)
        ver[:, 0] = rs.uniform(-0.2 * W - 1, 1.2 * W + 1, (B, nver))
        ver[:, 1] = rs.uniform(-0.2 * H - 1, 1.2 * H + 1, (B, nver))
        ver[:, 2] = rs.uniform(-50, 50, (B, nver))
        tri = rs.randint(0, nver, (3, ntri))
        if kind >= 2:   # most triangles small: two vertices near the first
            scale = float(rs.choice([0.3, 1.0, 3.0, 12.0]))
            k = min(ntri, nver // 3)
            idx = rs.permutation(nver)[:3 * k].reshape(-1, 3)
            for b in range(B):
                c = ver[b, :2][:, idx[:, 0]]
                ver[b, :2][:, idx[:, 1]] = c + rs.uniform(-scale, scale, c.shape).astype(np.float32)
                ver[b, :2][:, idx[:, 2]] = c + rs.uniform(-scale, scale, c.shape).astype(np.float32)
            tri[:, :k] = idx.T
        if kind == 3:   # integer coordinates: pixel centres on edges and vertices, equal depths (ties)
            ver[:, :2] = np.round(ver[:, :2])
            ver[:, 2] = np.round(ver[:, 2] / 10)
        if kind == 4:   # duplicates, degenerate triangles and special values
            tri[:, ::7] = tri[:, ::7][:, ::-1] if tri[:, ::7].shape[1] > 1 else tri[:, ::7]
            tri[1, ::11] = tri[0, ::11]
            flat = ver.reshape(-1)
            for val in (np.nan, np.inf, -np.inf, 3e9, -3e9, 1e-40, -0.0):
                flat[rs.randint(0, flat.size, 3)] = val
    tri = tri.astype(np.float32)
    if kind == 4:
        tri[rs.randint(0, 3), rs.randint(0, tri.shape[1])] = -1.0
        tri[rs.randint(0, 3), rs.randint(0, tri.shape[1])] = float(nver)
        tri[rs.randint(0, 3), rs.randint(0, tri.shape[1])] = np.nan
        tri[rs.randint(0, 3), rs.randint(0, tri.shape[1])] = 0.75
    tex_b = B if rs.rand() < 0.5 else 1
    tex = rs.uniform(0, 1, (tex_b, 3, nver)).astype(np.float32)
    return ver, tri, tex, H, W


@pytest.mark.parametrize("seed", range(N_RENDER))
def test_render_forward_and_backward(oracle, seed):
    ver, tri, tex, H, W = _scene(1000 + seed)
    B, nver = ver.shape[0], ver.shape[2]
    want = oracle.render_depth(ver, tri, tex, H, W)
    got = render_gpu(ver, tri, tex, H, W)
    assert_render_equal(got, want, "fuzz seed %d" % seed)
    # backward through the operator surface: two launches bit-equal, and equal to the oracle's sequential sum up to that
    # order's own rounding (n terms of size <= max|g|/3)
    rs = np.random.RandomState(seed)
    g = (rs.uniform(-2, 2, (B, H, W, 1)) * (rs.rand(B, H, W, 1) < 0.7)).astype(np.float32)
    dev = torch.device("cuda:0")
    outs = []
    for _ in range(2):
        v = torch.as_tensor(ver, device=dev).requires_grad_(True)
        d = ops().render_depth(v, torch.as_tensor(tri, device=dev), torch.as_tensor(tex, device=dev),
                               torch.zeros((B, H, W, 3), device=dev))[0]
        d.backward(torch.as_tensor(g, device=dev))
        outs.append(v.grad.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    wantg = oracle.render_depth_grad(g, tri, want[3], nver)
    n_terms = 3 * H * W
    tol = n_terms * np.float32(2.0 / 3.0) * np.float32(2.0 ** -23) + 1e-30
    assert np.max(np.abs(outs[0] - wantg)) <= tol
    np.testing.assert_array_equal(outs[0][:, :2], 0.0)


@pytest.mark.parametrize("seed", range(N_DECODE))
def test_decode_both_arithmetics(oracle, synth, seed):
    rs = np.random.RandomState(2000 + seed)
    gu, gv = int(rs.randint(2, 20)), int(rs.randint(2, 24))
    ns = int(rs.choice([0, 1, 5, 16, 17, 63, 64, 65, 100, 199, 200, 255, 256, 300]))
    ne = int(rs.choice([0, 1, 3, 15, 16, 29, 40, 64]))
    B = int(rs.choice([1, 2, 15, 16, 17, 31, 33, 48, 64, 65, 129]))
    A = synth.make_assets(gu, gv, ns, ne, patch=None, seed_basis=seed)
    P = np.zeros((B, 7 + ns + ne), np.float32)
    P[:, 0:3] = rs.uniform(-1.5, 1.5, (B, 3))
    P[:, 3:5] = rs.uniform(0, 200, (B, 2))
    P[:, 6] = rs.uniform(0, 1e-3, B)
    P[:, 7:7 + ns] = rs.uniform(0, 1e4, (B, ns)) * (rs.rand(B, ns) < 0.9)
    P[:, 7 + ns:] = rs.uniform(-1.5, 1.5, (B, ne))
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
    h = pkg("_lib")
    prev, prev_lv = h.decode_arith(), h.q30_levels()
    lv = int(rs.choice([7, 5, 4]))          # digit-product levels of the Q30 leg
    sched = int(rs.choice([0, 1]))       # and its schedule (matters for the model's shape only)
    try:
        for mode, q30 in ((1, False), (0, lv)):
            h.set_decode_arith(mode, lv)
            with h.options(FR_Q30_SCHED=sched):
                got = net.vertices_transform(torch.as_tensor(P, device="cuda:0"), R=torch.as_tensor(R, device="cuda:0"))
                torch.cuda.synchronize()
            want = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R, q30=q30)
            np.testing.assert_array_equal(got.cpu().numpy(), want,
                                          err_msg="seed %d %s" % (seed, "q30 levels %d sched %d" % (lv, sched) if q30 else "f32"))
    finally:
        h.set_decode_arith(prev, prev_lv)


@pytest.mark.parametrize("seed", range(max(1, N_RENDER // 4)))
def test_fused_rendering_layer(oracle, seed):
    """fr_rendering_layer_forward (render + network.py:185-199 in one pass) on the finite scene kinds: depth / tri_ind and
    the post-processed channels bit-exact against the oracle's planes pushed through the same fp32 formulas in numpy."""
    s = 5000 + seed
    while True:
        ver, tri, tex, H, W = _scene(s)
        if np.isfinite(ver).all() and np.isfinite(tri).all():
            break
        s += 100000
    B = ver.shape[0]
    im = np.random.RandomState(seed).uniform(0, 1, (B, H, W, 1)).astype(np.float32)
    t = lambda a: torch.as_tensor(a, device="cuda:0")  # noqa: E731
    net_in, depth_img, depth, tri_ind = ops().rendering_layer_fused(t(ver), t(tri), t(tex), t(im))
    d, tx, n, ti = oracle.render_depth(ver, tri, tex, H, W)
    np.testing.assert_array_equal(depth.cpu().numpy(), d)
    np.testing.assert_array_equal(tri_ind.cpu().numpy(), ti)
    np.testing.assert_array_equal(net_in[..., 1:4].cpu().numpy(), np.clip(tx, np.float32(1e-6), np.float32(1.0)))
    np.testing.assert_array_equal(net_in[..., 0:1].cpu().numpy(), np.clip(d, np.float32(1e-6), np.float32(1.0)) * im)
    np.testing.assert_array_equal(depth_img.cpu().numpy(), np.maximum(d, np.float32(1e-6)))
    nn = n.copy()
    nn[nn[..., 2] < 0] *= -1.0
    with np.errstate(over="ignore", invalid="ignore"):
        mag = (nn[..., 0] * nn[..., 0] + nn[..., 1] * nn[..., 1]) + nn[..., 2] * nn[..., 2]
        mag = np.where(mag > np.float32(1e-6), mag, np.float32(1.0)).astype(np.float32)
        want = (nn / (np.sqrt(mag) + np.float32(1e-6))[..., None]).astype(np.float32)
    np.testing.assert_array_equal(net_in[..., 4:7].cpu().numpy(), want)


@pytest.mark.parametrize("seed", range(max(1, N_RENDER // 8)))
def test_decode_backward(oracle, synth, seed):
    """fr_decode_3dmm_backward vs the float64 gradient, random basis shapes and batches; bit-reproducible."""
    rs = np.random.RandomState(7000 + seed)
    gu, gv = int(rs.randint(2, 24)), int(rs.randint(2, 24))
    ns = int(rs.choice([1, 5, 16, 17, 64, 100, 199]))
    ne = int(rs.choice([1, 3, 16, 29]))
    B = int(rs.choice([1, 2, 16, 17, 33, 64, 65]))
    A = synth.make_assets(gu, gv, ns, ne, patch=None, seed_basis=seed)
    P = np.zeros((B, 7 + ns + ne), np.float32)
    P[:, 0:3] = rs.uniform(-1.0, 1.0, (B, 3))
    P[:, 3:5] = rs.uniform(60, 140, (B, 2))
    P[:, 5] = rs.uniform(-1, 1, B)
    P[:, 6] = rs.uniform(2e-4, 1e-3, B)
    P[:, 7:7 + ns] = rs.uniform(0, 1e4, (B, ns))
    P[:, 7 + ns:] = rs.uniform(-1.5, 1.5, (B, ne))
    G = rs.standard_normal((B, 3, gu * gv)).astype(np.float32)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=B, im_size=200)
    grads = []
    for _ in range(2):
        p = torch.as_tensor(P, device="cuda:0").requires_grad_(True)
        net.vertices_transform(p).backward(torch.as_tensor(G, device="cuda:0"))
        grads.append(p.grad.cpu().numpy().astype(np.float64))
    np.testing.assert_array_equal(grads[0], grads[1])
    want = oracle.decode_3dmm_backward_f64(G, P, A["mu"], A["pc_shape"], A["pc_exp"])
    assert np.all(grads[0][:, 0:3] == 0)
    # every output is a sum over the vertices; the tolerance is relative to the size of the block's largest output, with a
    # floor of 1 % of the largest sum of ABSOLUTE terms for d f (a random sum of ~N terms cancels to a small value every few
    # hundred seeds, and an fp32 sum cannot be held relative to THAT)
    V64 = oracle.decode_3dmm_f64(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    q = V64.copy()
    q[:, 1] = (200.0 - 1.0) - q[:, 1]
    q -= P[:, 3:6].astype(np.float64)[:, :, None]
    dq = G.astype(np.float64) * np.array([1.0, -1.0, 1.0])[None, :, None]
    f_abs_terms = (np.abs(q) * np.abs(dq)).sum(axis=(1, 2)) / P[:, 6].astype(np.float64)
    for sl in (slice(3, 6), slice(6, 7), slice(7, 7 + ns), slice(7 + ns, None)):
        scale = np.abs(want[:, sl]).max() + 1e-30
        if sl == slice(6, 7):
            scale = max(scale, 1e-2 * float(f_abs_terms.max()))
        assert np.abs(grads[0][:, sl] - want[:, sl]).max() / scale < 2e-5, (seed, sl)
