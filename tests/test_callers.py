"""The callers of the hot path (BASELINE.json configs 3-5): CoarseNet / FineNet modules, the DDP gradient all-reduce
(world-size-2 gloo on CPU for the network alone; the render loop needs a GPU), and hot-path parity at those configs'
shapes (batch 32 @ 200x200; 448x448 images)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = importlib.import_module("3dfacerecon_amd.utils.dist")
    cn = importlib.import_module("3dfacerecon_amd.nets.coarse_net")
    d.init_from_env("gloo")
    torch.manual_seed(0)
    net = cn.CoarseNetIter(ndim=20)
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    x = torch.randn((2, 32, 32, 7), generator=torch.Generator().manual_seed(10 + rank))
    ddp(x).square().mean().backward()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    # the same computation without DDP, for the averaged-gradient check done by the parent
    torch.manual_seed(0)
    ref = cn.CoarseNetIter(ndim=20)
    ref(x).square().mean().backward()
    gl = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    q.put((rank, g.numpy(), gl.numpy()))
    d.finalize()


def test_miopen_workaround_is_the_entry_points_job(monkeypatch):
    """importing nets.coarse_net must not touch the process environment (the MIOpen solver switch is a setting of
    the whole host application); entry points call apply_miopen_workaround(), and building a trainable net without it warns."""
    import subprocess
    name = "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC"
    code = ("import os, sys, importlib; sys.path.insert(0, %r); os.environ.pop(%r, None); "
            "cn = importlib.import_module('3dfacerecon_amd.nets.coarse_net'); assert %r not in os.environ; "
            "assert cn.apply_miopen_workaround() == '0' and os.environ[%r] == '0'; print('ok')" % (ROOT, name, name, name))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    cn = pkg("nets.coarse_net")
    monkeypatch.delenv(name, raising=False)
    with pytest.warns(RuntimeWarning, match="apply_miopen_workaround"):
        cn.FineNet()
    monkeypatch.setenv(name, "1")      # an explicit '1' is respected by the helper and still warned about
    assert cn.apply_miopen_workaround() == "1"
    with pytest.warns(RuntimeWarning):
        cn.FineNet()
    monkeypatch.setenv(name, "0")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        cn.FineNet()


def test_ddp_gradient_allreduce_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, l0), (_, g1, l1) = res
    np.testing.assert_array_equal(g0, g1)                       # all ranks hold the reduced gradient
    np.testing.assert_allclose(g0, 0.5 * (l0 + l1), rtol=1e-4, atol=1e-7)   # ... which is the mean of the local ones
    assert np.abs(g0).max() > 0


class _FakeFaceNet:
    """CPU stand-in for nets.network.FaceRecNet in the DDP wiring test: the same methods FaceReconModel calls, with a
    small differentiable torch body instead of the HIP decode / render (which need a GPU).  Test double only."""

    def __init__(self, ndim=12, im_size=16):
        self.ndim, self.im_size = ndim, im_size
        self.init_pred_params = torch.zeros((8, 1, 1, ndim))
        g = torch.Generator().manual_seed(5)
        self.basis = torch.randn((ndim, 3 * 10), generator=g)

    def vertices_transform(self, params, R=None):
        return (params.reshape(params.shape[0], -1) @ self.basis).reshape(-1, 3, 10)

    def coarse_net_input(self, v, triangles=None, colors=None, im_gray=None):
        B, S = v.shape[0], self.im_size
        d = v[:, 2].mean(dim=1).reshape(B, 1, 1, 1) + 0.0 * im_gray
        net_in = torch.cat([d * im_gray] + [im_gray] * 6, dim=3)
        return net_in, d.expand(B, S, S, 1).contiguous()

    def set_constraints(self, p):
        return torch.sigmoid(p)


def _ddp_model_worker(rank, world, port, q):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    d = importlib.import_module("3dfacerecon_amd.utils.dist")
    cn = importlib.import_module("3dfacerecon_amd.nets.coarse_net")
    d.init_from_env("gloo")

    def build():
        torch.manual_seed(0)
        return cn.FaceReconModel(_FakeFaceNet(), nIter=2, fine=True)
    x = torch.rand((2, 16, 16, 1), generator=torch.Generator().manual_seed(10 + rank))

    def loss(out):
        return out["pred_params"].square().mean() + out["pred_depth_map"].square().mean() + out["coarse_depth_map"].mean()
    model = build()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    loss(ddp(x)).backward()                      # the way examples/coarse_loop.py calls it: through the DDP wrapper
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    ref = build()
    loss(ref(x)).backward()
    gl = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    q.put((rank, g.numpy(), gl.numpy()))
    d.finalize()


def test_face_recon_model_under_ddp_allreduces_gloo():
    """the harness wrapped a ModuleList (no forward) in DDP and called the inner modules, so no gradient
    was ever all-reduced.  FaceReconModel is one nn.Module; called THROUGH the wrapper, both ranks end with the mean of
    the two local gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, l0), (_, g1, l1) = res
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_allclose(g0, 0.5 * (l0 + l1), rtol=1e-4, atol=1e-7)
    assert np.abs(l0 - l1).max() > 1e-6 and np.abs(g0).max() > 0          # the shards really differ


def test_module_shapes_cpu():
    cn = pkg("nets.coarse_net")
    it = cn.CoarseNetIter(ndim=235)
    n_params = sum(p.numel() for p in it.parameters())
    assert 15e6 < n_params < 25e6                                # ~18.3 M parameters per iteration
    y = it.eval()(torch.zeros((1, 64, 64, 7)))
    assert tuple(y.shape) == (1, 235)
    fine = cn.FineNet().eval()
    d = fine(torch.zeros((1, 32, 32, 1)), torch.zeros((1, 32, 32, 1)))
    assert tuple(d.shape) == (1, 32, 32, 1)


@pytest.mark.gpu
def test_coarse_loop_forward_backward_gpu(small_assets):
    netm, cn = pkg("nets.network"), pkg("nets.coarse_net")
    S, B = 40, 3
    face = netm.FaceRecNet(mesh_data=small_assets, batch_size=B, im_size=S)
    face.init_pred_params[..., 6] = 2e-4
    torch.manual_seed(0)
    coarse = cn.CoarseNet(face, nIter=2).cuda()
    im = torch.rand((B, S, S, 1), device="cuda:0")
    params = coarse(im)
    assert tuple(params.shape) == (B, face.ndim) and bool(torch.isfinite(params).all())
    assert float(params.detach()[:, 0:3].abs().max()) <= 1.5 and float(params.detach()[:, 6].max()) <= 1e-3
    depth = coarse.depth(im, params)
    assert tuple(depth.shape) == (B, S, S, 1)
    # depth-only loss: the gradient reaches the last iteration's weights ONLY through render backward (vertex z) and
    # decode backward (235-d parameters)
    (1e-3 * depth.mean()).backward()
    last = coarse.iters[-1]
    assert max(float(p.grad.abs().max()) for p in last.parameters() if p.grad is not None) > 0
    # earlier iterations see the later ones only through the rendered input, whose sole vertex-dependent channel is
    # mask = clip(depth, 1e-6, 1) (network.py:195): saturated wherever depth > 1, so their gradient may be exactly zero --
    # as in the reference -- but it must be finite
    for it in coarse.iters[:-1]:
        for p in it.parameters():
            assert p.grad is None or bool(torch.isfinite(p.grad).all())


@pytest.mark.gpu
def test_config3_batch32_hot_path(oracle, full_assets, synth):
    """config 3 shape: 32 faces @ 200x200 through decode -> render; oracle check on the first and last face."""
    from gpu_util import assert_render_equal, net_mod, ops
    A = full_assets
    P = synth.sample_params_batch(32, beta=0.7, seed=77)
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=32, im_size=200)
    V = net.vertices_transform(torch.as_tensor(P, device="cuda:0"), R=torch.as_tensor(R, device="cuda:0"))
    outs = ops().render_depth(V, net.tri, net.vertex_code, torch.zeros((32, 200, 200, 3), device="cuda:0"))
    got = tuple(o.cpu().numpy() for o in outs)
    for b in (0, 31):
        Vo = oracle.decode_3dmm(P[b:b + 1], A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R[b:b + 1])
        np.testing.assert_array_equal(V[b:b + 1].cpu().numpy(), Vo)
        assert_render_equal(tuple(g[b:b + 1] for g in got), oracle.render_depth(Vo, A["tri"], A["vertex"][None], 200, 200),
                            "config 3 face %d" % b)


@pytest.mark.gpu
def test_config5_448_hot_path(oracle, full_assets, synth):
    """config 5 shape: 448x448 images, 16 faces per GPU (128 over 8); the face is scaled to fill the larger image."""
    from gpu_util import assert_render_equal, net_mod, ops
    A = full_assets
    P = synth.sample_params_batch(16, im_size=448, beta=0.7, seed=5)
    P[:, 6] *= 2.24                                        # f: same face, 448/200 times larger
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=16, im_size=448)
    V = net.vertices_transform(torch.as_tensor(P, device="cuda:0"), R=torch.as_tensor(R, device="cuda:0"))
    outs = ops().render_depth(V, net.tri, net.vertex_code, torch.zeros((16, 448, 448, 3), device="cuda:0"))
    got = tuple(o.cpu().numpy() for o in outs)
    assert (got[3] >= 0).mean() > 0.2
    for b in (3, 15):
        Vo = oracle.decode_3dmm(P[b:b + 1], A["mu"], A["pc_shape"], A["pc_exp"], 448.0, R=R[b:b + 1])
        np.testing.assert_array_equal(V[b:b + 1].cpu().numpy(), Vo)
        assert_render_equal(tuple(g[b:b + 1] for g in got), oracle.render_depth(Vo, A["tri"], A["vertex"][None], 448, 448),
                            "config 5 face %d" % b)
