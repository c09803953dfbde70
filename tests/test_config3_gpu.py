"""BASELINE.json configs[3] on the HIP path at FULL size: 32 faces per GPU (N = 53,215, T = 105,840, 200x200),
CoarseNet (nIter = 4) -> decode -> fused rendering layer -> FineNet -> the reference's objective (network.py:336-378)
-> render backward -> decode backward -> every iteration's weights.

Checked: (i) finite gradients on every parameter of every iteration; (ii) on the SAME graph, the render-backward output
(d loss / d vertices_proj, captured with a hook) against fr_oracle_render_depth_backward fed with the captured depth
gradient, and the decode-backward output (d loss / d params of the depth rendering layer) against the float64 oracle, on
faces 0 and 31; (iii) one optimiser step changes the weights and a second forward is still finite."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def test_config3_full_size_train_step(oracle, full_assets, synth):
    netm, cn, losses = pkg("nets.network"), pkg("nets.coarse_net"), pkg("nets.losses")
    A = full_assets
    B, S = 32, 200
    dev = torch.device("cuda:0")
    face = netm.FaceRecNet(mesh_data=A, batch_size=B, im_size=S, device=dev)
    torch.manual_seed(7)
    model = cn.FaceReconModel(face, nIter=4, fine=True).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    g = torch.Generator().manual_seed(11)
    im = torch.rand((B, S, S, 1), generator=g).to(dev)
    lab = torch.as_tensor(synth.sample_params_batch(B, im_size=S, beta=0.7, seed=200), device=dev)

    # capture the tensors on the depth-rendering-layer edge of the graph (network.py:300-309)
    cap = {}
    orig_vt, orig_in = face.vertices_transform, face.coarse_net_input
    calls = {"n": 0}

    def vt(params, R=None):
        v = orig_vt(params, R)
        calls["n"] += 1
        if calls["n"] == 5:   # 4 CoarseNet iterations, then the depth rendering layer
            cap["params"] = params
            params.retain_grad()
            cap["v"] = v
            v.retain_grad()
        return v

    def cin(v, triangles=None, colors=None, im_gray=None):
        net_in, depth_img = orig_in(v, triangles, colors, im_gray)
        if v is cap.get("v"):
            cap["depth_img"] = depth_img
            depth_img.retain_grad()
        return net_in, depth_img

    face.vertices_transform, face.coarse_net_input = vt, cin
    try:
        out = model(im)
        L = losses.get_loss(face, out["pred_params"], lab, im, out["vertices_proj"], out["coarse_depth_map"],
                            out["pred_depth_map"])
        assert all(bool(torch.isfinite(v)) for v in L.values()), {k: float(v) for k, v in L.items()}
        opt.zero_grad(set_to_none=True)
        L["total_loss"].backward()
    finally:
        face.vertices_transform, face.coarse_net_input = orig_vt, orig_in
    assert calls["n"] == 5

    # (i) every parameter of every iteration (and FineNet) has a finite gradient; the last iteration's is non-zero
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        assert bool(torch.isfinite(p.grad).all()), name
    assert max(float(p.grad.abs().max()) for p in model.coarse.iters[-1].parameters()) > 0
    assert max(float(p.grad.abs().max()) for p in model.fine.parameters()) > 0

    # (ii) hot-path backward parity on faces 0 and 31 of this very graph.  d loss / d v comes from two consumers of v:
    # the depth image of the fused layer (render backward) and the SfS renders (no vertex gradient: texture / normal
    # outputs have none, ops.py:95) -- so v.grad must equal the render backward of the captured depth gradient.
    v, dimg = cap["v"], cap["depth_img"]
    gd = dimg.grad
    assert gd is not None and float(gd.abs().max()) > 0
    # depth_img = max(depth, 1e-6): the gradient reaches depth where depth > 1e-6 (network.py:199)
    fused = pkg("rendering_layer.ops").rendering_layer_fused(v.detach(), face.tri, face.vertex_code, im)
    depth_raw, tind = fused[2], fused[3]
    gdepth = torch.where(depth_raw > 1e-6, gd, torch.zeros_like(gd))
    for b in (0, 31):
        want = oracle.render_depth_grad(gdepth[b:b + 1].cpu().numpy(), A["tri"], tind[b:b + 1].cpu().numpy(), face.nvert)
        got = v.grad[b:b + 1].cpu().numpy()
        assert np.all(got[:, :2] == 0)
        scale = np.abs(want).max()
        assert scale > 0
        np.testing.assert_allclose(got[:, 2], want[:, 2], rtol=0, atol=2e-6 * max(scale, 1.0) + 1e-9)
    # decode backward of the SAME upstream gradient, isolated (the graph's params.grad also holds the pose / geometry
    # loss terms): 32 faces at full size through fr_decode_3dmm_backward, faces 0 and 31 against the float64 oracle
    p2 = cap["params"].detach().clone().requires_grad_(True)
    orig_vt(p2).backward(v.grad)
    P = p2.detach().cpu().numpy()
    G = v.grad.detach().cpu().numpy()
    gp = p2.grad.detach().cpu().numpy().astype(np.float64)
    assert bool(torch.isfinite(cap["params"].grad).all())
    for b in (0, 31):
        want = oracle.decode_3dmm_backward_f64(G[b:b + 1], P[b:b + 1], A["mu"], A["pc_shape"], A["pc_exp"])
        ns = A["ndim_shape"]
        assert np.all(gp[b, 0:3] == 0)
        for sl in (slice(3, 6), slice(6, 7), slice(7, 7 + ns), slice(7 + ns, None)):
            scale = np.abs(want[0, sl]).max() + 1e-30
            assert np.abs(gp[b, sl] - want[0, sl]).max() / scale < 5e-5, (b, sl)

    # (iii) the step moves the weights; the next forward is finite
    w0 = model.coarse.iters[-1].fc.weight.detach().clone()
    opt.step()
    assert float((model.coarse.iters[-1].fc.weight.detach() - w0).abs().max()) > 0
    with torch.no_grad():
        out2 = model(im)
    assert bool(torch.isfinite(out2["pred_params"]).all()) and bool(torch.isfinite(out2["pred_depth_map"]).all())
