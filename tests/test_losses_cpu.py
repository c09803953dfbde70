"""CPU: the torch objective (3dfacerecon_amd/nets/losses.py, reference nets/network.py:336-392, 420-462) against its
numpy restatement (oracle/losses_np.py) term by term; the terms that call the hot path (geometry product on the MFMA
decode kernel, the two extra renders of the shading model) are covered on the GPU in tests/test_losses_gpu.py."""
import numpy as np
import torch

from conftest import pkg
from oracle import losses_np as LN


def test_laplace_and_smoothness():
    L = pkg("nets.losses")
    rs = np.random.RandomState(0)
    x = rs.standard_normal((3, 17, 13, 1)).astype(np.float32)
    got = L.laplace_transform(torch.as_tensor(x[..., 0])).numpy()
    want = np.stack([LN.laplace_transform(d[:, :, 0]) for d in x])
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(L.laplace_transform(torch.as_tensor(x[0, :, :, 0])).numpy(), want[0], rtol=0, atol=2e-5)
    # a constant image has a zero Laplacian away from the zero-padded border only ('SAME' padding, as in the reference)
    c = L.laplace_transform(torch.ones((6, 7))).numpy()
    assert np.all(c[1:-1, 1:-1] == 0) and c[0, 0] == -6 + 1 + 1 + 0.5
    assert abs(float(L.laplace_transform(torch.as_tensor(x[..., 0])).abs().sum()) - LN.smoothness_loss(x)) < 1e-2


def test_shading_model_core_vs_numpy_pinv():
    L = pkg("nets.losses")
    rs = np.random.RandomState(1)
    B, H, W = 6, 5, 4
    nrm = rs.standard_normal((B, H, W, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    nrm[:, 0, 0] = 0                                  # a background pixel: Y = 0 -> pinv = 0 -> intensity 0
    nrm[:, 1, 1] = nrm[0, 1, 1]                       # rank-1 pixel: every face has the same normal
    alb = rs.uniform(0.2, 0.8, (B, H, W, 1)).astype(np.float32)
    alb2 = rs.uniform(0.2, 0.8, (B, H, W, 1)).astype(np.float32)
    nrm2 = np.roll(nrm, 1, axis=0)
    im = rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32)
    t = lambda a: torch.as_tensor(a)  # noqa: E731
    got = L.spherical_harmonics_intensity(t(alb), t(nrm), t(im), t(alb2), t(nrm2)).numpy()
    want = LN.spherical_harmonics_intensity(alb, nrm, im, alb2, nrm2)
    assert got.shape == (B, H, W, 1)
    # the rank-1 pixel is ill-posed in the reference itself: with the 1e-15 cutoff np.linalg.pinv inverts the rounding
    # noise in the two null directions (singular values ~1e-7 of 6), so any two correct implementations disagree there;
    # it must stay finite, nothing more
    ok = np.ones((H, W), bool)
    ok[1, 1] = False
    np.testing.assert_allclose(got[:, ok], want[:, ok], rtol=2e-3, atol=2e-4)
    assert np.all(np.isfinite(got[:, 1, 1]))
    assert np.all(got[:, 0, 0] == 0)
    # the pinv carries no gradient (tf.py_func, network.py:431) but the rest of the expression does
    a = t(alb2).requires_grad_(True)
    y = t(nrm).requires_grad_(True)
    L.spherical_harmonics_intensity(t(alb), y, t(im), a, t(nrm2)).sum().backward()
    assert a.grad is not None and y.grad is not None and bool(torch.isfinite(y.grad).all())


def test_pose_fidelity_and_total():
    L = pkg("nets.losses")
    rs = np.random.RandomState(2)
    P, Q = rs.standard_normal((4, 235)).astype(np.float32), rs.standard_normal((4, 235)).astype(np.float32)
    got = float(torch.nn.functional.mse_loss(torch.as_tensor(P[:, :7]), torch.as_tensor(Q[:, :7])))
    assert abs(got - LN.pose_loss(P, Q)) < 1e-6
    d = {k: torch.tensor(v) for k, v in (("pose_loss", 2.0), ("geometry_loss", 3e5), ("spherical_harmonics_loss", 0.5),
                                         ("fidelity_loss", 0.01), ("smoothness_loss", 700.0))}
    assert abs(float(L.combine_losses(d)) - LN.total_loss({k: float(v) for k, v in d.items()})) < 1e-6
    assert (L.LAMBDA_POSE, L.LAMBDA_GEO, L.LAMBDA_SH, L.LAMBDA_F, L.LAMBDA_SM) == (1e-3, 1e-6, 1e-3, 100.0, 1e-5)


def _sfs_gather_worker(rank, world, port, q):
    import importlib
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = importlib.import_module("3dfacerecon_amd.utils.dist")
    L = importlib.import_module("3dfacerecon_amd.nets.losses")
    d.init_from_env("gloo")
    rs = np.random.RandomState(7)
    B, H, W = 8, 4, 5
    nrm = rs.standard_normal((B, H, W, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    alb = rs.uniform(0.2, 0.8, (B, H, W, 1)).astype(np.float32)
    alb2 = rs.uniform(0.2, 0.8, (B, H, W, 1)).astype(np.float32)
    im = rs.uniform(0, 1, (B, H, W, 1)).astype(np.float32)
    t = lambda a: torch.as_tensor(a)  # noqa: E731
    full = L.spherical_harmonics_intensity(t(alb), t(nrm), t(im), t(alb2), t(nrm)).numpy()      # the single-process estimate
    sl = slice(rank * B // world, (rank + 1) * B // world)
    local = L.spherical_harmonics_intensity(t(alb[sl]), t(nrm[sl]), t(im[sl]), t(alb2[sl]), t(nrm[sl])).numpy()
    y = t(nrm[sl]).requires_grad_(True)
    gathered = L.spherical_harmonics_intensity(t(alb[sl]), y, t(im[sl]), t(alb2[sl]), y, gather=True)
    gathered.sum().backward()
    q.put((rank, full[sl], local, gathered.detach().numpy(), bool(torch.isfinite(y.grad).all())))
    d.finalize()


def test_sfs_lighting_under_batch_sharding_gloo():
    """the SfS lighting is one per-pixel least squares over the WHOLE batch (network.py:430-434).  Under batch
    sharding each rank's own estimate differs from the single-process one; with gather=True (all-gather of the per-pixel
    normals / intensities) every rank reproduces the single-process result for its shard."""
    import torch.multiprocessing as mp
    from test_callers import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sfs_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, want, local, gathered, finite in res:
        np.testing.assert_allclose(gathered, want, rtol=2e-4, atol=2e-5)
        assert np.abs(local - want).max() > 1e-3          # the per-shard estimator really is a different one
        assert finite
