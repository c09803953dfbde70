"""The reference's training objective (nets/network.py:336-392, 420-462) on PyTorch-ROCm -- the caller of the hot path in
BASELINE.json config 4.  Stock torch ops, except where the objective itself calls the hot path:

  * the geometry loss's basis product [pc_shape | pc_exp] . coeff (network.py:347-353) runs on the MFMA decode kernel
    (FaceRecNet.geometry_product), and
  * the shape-from-shading model issues two more render_depth calls (network.py:423, 454 through compute_abedo_image).

Names and weights follow the reference: pose MSE (lambda 1e-3), geometry MSE through the basis (1e-6), spherical
harmonics / SfS MSE (1e-3), fidelity MSE between the coarse and the fine depth map (100), Laplacian-L1 smoothness (1e-5)
(network.py:27-31, 373).

Batch sharding: every term but one is a mean / sum over independent faces, so a data-parallel shard
computes its share and DDP averages the gradients.  The exception is the SfS lighting estimate: the reference solves ONE
per-pixel least squares over the whole batch (network.py:430-434: (Y Y^T)^+ Y (I/(albedo+1))^T with Y = [3 x B] normals
of that pixel), so under batch sharding each rank's lighting is estimated from its own B/world faces -- a different
(noisier) estimator, not a bug; `get_spherical_harmonics_model(..., gather=True)` all-gathers the per-pixel normal,
intensity and albedo planes first and reproduces the single-process estimate (one all-gather of 5 floats per pixel per
face, no gradient through the gathered remote shards, like the reference's py_func pinv has none).
"""
import torch
import torch.nn.functional as F

LAMBDA_POSE = 1e-3   # network.py:27
LAMBDA_GEO = 1e-6    # :28
LAMBDA_SH = 1e-3     # :29
LAMBDA_F = 100.0     # :30
LAMBDA_SM = 1e-5     # :31

_LAPLACE_K = ((0.5, 1.0, 0.5), (1.0, -6.0, 1.0), (0.5, 1.0, 0.5))  # network.py:383-385


def laplace_transform(x):
    """2-D Laplacian of (H,W) or (B,H,W) maps with the reference's 3x3 kernel, zero 'SAME' padding (network.py:381-392)."""
    single = x.dim() == 2
    xx = x[None] if single else x
    k = torch.tensor(_LAPLACE_K, dtype=xx.dtype, device=xx.device)[None, None]
    y = F.conv2d(xx[:, None], k, padding=1)[:, 0]
    return y[0] if single else y


def _pinv_sym3(A):
    """Moore-Penrose inverse of a batch of symmetric 3x3 matrices (the reference calls np.linalg.pinv on Y Y^T through
    tf.py_func, network.py:431: cutoff 1e-15 x the largest singular value, no gradient)."""
    return torch.linalg.pinv(A.detach(), rtol=1e-15, hermitian=True)


def _all_gather_batch(t):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t.detach().contiguous())
    parts[dist.get_rank()] = t  # keep the local shard differentiable
    return torch.cat(parts, dim=-1)


def spherical_harmonics_intensity(abedo_image, normal_map, im_gray, abedo_image_new, normal_map_new, gather=False):
    """The linear-algebra core of get_spherical_harmonics_model (network.py:424-460) on already rendered maps:
    per pixel, lighting l = (Y Y^T)^+ Y (I / (albedo + 1))^T over the batch (Y = [3 x B] normals), then the recovered
    intensity albedo_new * (l^T Y_new).  All inputs [B,H,W,c]; returns [B,H,W,1]."""
    abedo = abedo_image.permute(1, 2, 3, 0)            # (H,W,1,B)
    Yz0 = normal_map.permute(1, 2, 3, 0)               # (H,W,3,B)
    I = im_gray.permute(1, 2, 3, 0)                    # (H,W,1,B)
    rhs = I / (abedo + 1.0)
    Yl, rl = (Yz0, rhs) if not gather else (_all_gather_batch(Yz0), _all_gather_batch(rhs))
    Yz0_nec_inv = _pinv_sym3(Yl @ Yl.transpose(-1, -2))                                   # (H,W,3,3)
    lighting_lse = (Yz0_nec_inv @ Yl) @ rl.transpose(-1, -2)                               # (H,W,3,1)
    Yz = normal_map_new.permute(1, 2, 3, 0)
    intensity = abedo_image_new.permute(1, 2, 3, 0) * (lighting_lse.transpose(-1, -2) @ Yz)  # Eqn (8), (H,W,1,B)
    return intensity.permute(3, 0, 1, 2)


def get_spherical_harmonics_model(face_net, vertices_proj, im_gray, gather=False):
    """Recovered intensity (B,H,W,1) of the first-order spherical-harmonics shading model (network.py:420-462): two
    more render_depth calls (mean albedo, then mean + pc_tex . param_tex) feed spherical_harmonics_intensity."""
    fn = face_net
    if fn.mu_tex is None or fn.pc_tex is None or fn.param_tex is None:
        raise ValueError("the asset dict has no texture model (mu_tex / pc_tex / param_tex)")
    abedo_image, normal_map = fn.compute_abedo_image(vertices_proj, fn.tri, fn.mu_tex)   # (B,H,W,1), (B,H,W,3)
    texture_new = fn.mu_tex + (fn.pc_tex @ fn.param_tex).reshape(3, -1)                    # network.py:446-448
    abedo_new, normal_new = fn.compute_abedo_image(vertices_proj, fn.tri, texture_new)
    return spherical_harmonics_intensity(abedo_image, normal_map, im_gray, abedo_new, normal_new, gather=gather)


def combine_losses(losses):
    """total = 1e-3 pose + 1e-6 geometry + 1e-3 SfS + 100 fidelity + 1e-5 smoothness (network.py:27-31, 373)"""
    return (LAMBDA_POSE * losses['pose_loss'] + LAMBDA_GEO * losses['geometry_loss'] +
            LAMBDA_SH * losses['spherical_harmonics_loss'] + LAMBDA_F * losses['fidelity_loss'] +
            LAMBDA_SM * losses['smoothness_loss'])


def get_loss(face_net, pred_params, params_label, im_gray, vertices_proj, coarse_depth_map, pred_depth_map,
             gather_sfs=False):
    """dict of the reference's six scalars (network.py:336-378).  pred_params / params_label: (B,d) or (B,1,1,d)."""
    fn = face_net
    B = pred_params.shape[0]
    pred = pred_params.reshape(B, fn.ndim)
    label = params_label.reshape(B, fn.ndim).to(pred.dtype)
    losses = {}
    losses['pose_loss'] = F.mse_loss(pred[:, :fn.ndim_pose], label[:, :fn.ndim_pose])
    # geometry: MSE(basis . label^T, basis . pred^T) == mean over (3N x B) of (basis . (pred - label)^T)^2 up to fp32
    # rounding of the two products; the difference form needs one pass of the basis instead of two
    g = fn.geometry_product(pred[:, fn.ndim_pose:] - label[:, fn.ndim_pose:])
    losses['geometry_loss'] = (g * g).mean()
    intensity_recover = get_spherical_harmonics_model(fn, vertices_proj, im_gray, gather=gather_sfs)
    losses['spherical_harmonics_loss'] = F.mse_loss(intensity_recover, im_gray)
    losses['fidelity_loss'] = F.mse_loss(pred_depth_map, coarse_depth_map)
    filtered_depth = laplace_transform(pred_depth_map[..., 0])
    losses['smoothness_loss'] = filtered_depth.abs().sum()    # tf.contrib.layers.l1_regularizer(1.0), network.py:367
    losses['total_loss'] = combine_losses(losses)
    return losses
