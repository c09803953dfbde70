"""GPU end-to-end: 235-d parameters -> decode (HIP) -> render (HIP) vs oracle decode -> oracle render, and the
caller-side rendering_layer wrapper (network.py:174-201)."""
import numpy as np
import pytest
import torch

from gpu_util import assert_render_equal, net_mod, ops

pytestmark = pytest.mark.gpu


def test_params_to_depth_bit_exact(oracle, full_assets, synth):
    A = full_assets
    P = synth.sample_params_batch(2, beta=0.7, seed=3456)
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=2, im_size=200)
    V = net.vertices_transform(torch.as_tensor(P, device="cuda:0")[:, None, None, :], R=torch.as_tensor(R, device="cuda:0"))
    outs = ops().render_depth(V, net.tri, net.vertex_code, torch.zeros((2, 200, 200, 3), device="cuda:0"))
    got = tuple(o.cpu().numpy() for o in outs)
    Vo = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R)
    want = oracle.render_depth(Vo, A["tri"], A["vertex"][None], 200, 200)
    assert_render_equal(got, want, "params->depth")
    # against the float64 decode: depth within 1e-5 (north_star) wherever the winning triangle agrees
    V64 = oracle.decode_3dmm_f64(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    want64 = oracle.render_depth(V64.astype(np.float32), A["tri"], A["vertex"][None], 200, 200)
    same = (want64[3] == got[3]) & (got[3] >= 0)
    assert same.mean() > 0.3
    assert np.max(np.abs(got[0][same] - want64[0][same])) <= 1e-5 * 4   # fp32 ulp at |z|~100 is 7.6e-6
    assert (want64[3] != got[3]).mean() < 2e-3                           # edge-ambiguous pixels only


def test_rendering_layer_wrapper(full_assets, synth):
    A = full_assets
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=2, im_size=200)
    P = torch.as_tensor(synth.sample_params_batch(2, beta=0.7, seed=5), device="cuda:0")
    V = net.vertices_transform(P)
    im = torch.rand((2, 200, 200, 1), device="cuda:0")
    pncc, nrm, mask, dimg = net.rendering_layer(V, net.tri, net.vertex_code, im_gray=im)
    assert tuple(pncc.shape) == (2, 200, 200, 3) and tuple(nrm.shape) == (2, 200, 200, 3)
    assert tuple(mask.shape) == (2, 200, 200, 1) and tuple(dimg.shape) == (2, 200, 200, 1)
    assert float(pncc.min()) >= float(np.float32(1e-6)) and float(pncc.max()) <= 1.0
    assert float(nrm[..., 2].min()) >= 0.0
    n2 = (nrm * nrm).sum(-1)
    fg = n2 > 0.5
    assert fg.float().mean() > 0.2 and float((n2[fg] - 1).abs().max()) < 1e-3
    assert float(dimg.min()) >= float(np.float32(1e-6))
