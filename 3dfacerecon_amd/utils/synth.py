"""Synthetic BFM-scale assets and the reference's 235-d parameter sampler.

The Basel Face Model is licensed and absent (reference 3dmm/.gitignore:1-4), so benches and tests run on a
deterministic synthetic stand-in with the public 3DDFA/BFM trim's sizes (reference README.md:30-46):
N = 53,215 vertices, T = 105,840 triangles, 199 shape + 29 expression components, 235 = 7 + 199 + 29.
The dict returned by `make_assets` has the keys of utils/parser_3dmm.py:50-60 (`read_3dmm_model`).

This module is input generation only (numpy); it computes nothing on the hot path.
"""
import numpy as np

N_SHAPE = 199
N_EXP = 29
N_POSE = 7  # utils/parser_3dmm.py:49
N_TEX = 10  # texture components the reference keeps (network.py:26)

GRID_U = 145  # rows  (145 * 367 = 53,215 vertices)
GRID_V = 367  # cols
PATCH = (60, 170, 12, 18)  # (u0, v0, du, dv): duplicated 12x18-cell patch -> 432 extra triangles


def make_mesh(grid_u=GRID_U, grid_v=GRID_V, patch=PATCH):
    """UV-grid mesh: vertex id = iu*grid_v + iv; two triangles per cell; plus a duplicated patch of cells that
    re-uses existing vertices (exact duplicates -> equal-h ties, mimicking BFM's overlapping inner-mouth region).
    Returns tri as float32 [3, T] with 0-based indices (the op takes float-stored indices,
    reference network.py:178)."""
    iu, iv = np.meshgrid(np.arange(grid_u - 1), np.arange(grid_v - 1), indexing="ij")
    v00 = (iu * grid_v + iv).reshape(-1)
    v01 = v00 + 1
    v10 = v00 + grid_v
    v11 = v10 + 1
    t_a = np.stack([v00, v10, v01], 0)
    t_b = np.stack([v01, v10, v11], 0)
    tri = np.empty((3, 2 * v00.size), np.int64)
    tri[:, 0::2] = t_a
    tri[:, 1::2] = t_b
    if patch is not None:
        u0, v0, du, dv = patch
        du = min(du, max(grid_u - 1 - u0, 0))
        dv = min(dv, max(grid_v - 1 - v0, 0))
        if du > 0 and dv > 0:
            pu, pv = np.meshgrid(np.arange(u0, u0 + du), np.arange(v0, v0 + dv), indexing="ij")
            cell = (pu * (grid_v - 1) + pv).reshape(-1)
            extra = np.concatenate([tri[:, 2 * cell], tri[:, 2 * cell + 1]], 1)
            tri = np.concatenate([tri, extra], 1)
    return tri.astype(np.float32)


def _grid_coords(grid_u, grid_v):
    a = np.linspace(-1.0, 1.0, grid_v)[None, :].repeat(grid_u, 0)  # horizontal
    b = np.linspace(-1.0, 1.0, grid_u)[:, None].repeat(grid_v, 1)  # vertical
    return a, b


def make_assets(grid_u=GRID_U, grid_v=GRID_V, n_shape=N_SHAPE, n_exp=N_EXP, patch=PATCH, seed_basis=2345):
    """Returns the parser_3dmm-style dict: vertex (PNCC code [3,N]), tri [3,T], mu [3N,1] (blocked: element r is
    coordinate r//N of vertex r%N, reference network.py:157), pc_shape [3N,n_shape], pc_exp [3N,n_exp],
    mu_tex [3,N], ndim_*.  Units are BFM micrometres: with f=1e-3, t=(100,100) the dome lands as a ~140x160 px
    face in a 200x200 image."""
    a, b = _grid_coords(grid_u, grid_v)
    # square -> disc, then an oval dome
    xd = a * np.sqrt(1.0 - 0.5 * b * b)
    yd = b * np.sqrt(1.0 - 0.5 * a * a)
    r2 = np.clip(xd * xd + yd * yd, 0.0, 1.0)
    x = 7.0e4 * xd
    y = 8.0e4 * yd
    z = 1.2e5 * (1.0 - r2)
    N = grid_u * grid_v
    mu = np.concatenate([x.reshape(-1), y.reshape(-1), z.reshape(-1)]).astype(np.float32).reshape(3 * N, 1)

    rng = np.random.default_rng(seed_basis)

    def smooth_modes(k, noise):
        out = np.empty((3 * N, k), np.float32)
        for j in range(k):
            for c in range(3):
                fu, fv = rng.integers(0, 7, 2)
                ph_u, ph_v = rng.uniform(0, 2 * np.pi, 2)
                amp = rng.normal()
                m = amp * np.cos(np.pi * fu * b + ph_u) * np.cos(np.pi * fv * a + ph_v)
                m = m + noise * rng.standard_normal(m.shape)
                out[c * N:(c + 1) * N, j] = m.reshape(-1)
        return out

    pc_shape = smooth_modes(n_shape, 0.05)
    nrm = np.linalg.norm(pc_shape.astype(np.float64), axis=0, keepdims=True)
    pc_shape = (pc_shape / np.maximum(nrm, 1e-12)).astype(np.float32)  # unit-norm columns (raw shapePC)
    pc_exp = smooth_modes(n_exp, 0.02)
    rms = np.sqrt(np.mean(pc_exp.astype(np.float64) ** 2, axis=0, keepdims=True))
    pc_exp = (pc_exp / np.maximum(rms, 1e-12) * 300.0).astype(np.float32)

    u01 = (b.reshape(-1) + 1.0) * 0.5
    v01 = (a.reshape(-1) + 1.0) * 0.5
    z01 = (z.reshape(-1) / 1.2e5)
    vertex_code = np.stack([v01, u01, z01], 0).astype(np.float32)  # PNCC-like colour in [0,1]
    mu_tex = (0.2 + 0.6 * vertex_code).astype(np.float32)
    # albedo model of the SfS loss (reference network.py:45-47 uses the first 10 texture components): drawn AFTER the
    # geometry bases so that those keep their values
    pc_tex = (0.05 * smooth_modes(N_TEX, 0.0)).astype(np.float32)
    param_tex = rng.normal(size=(N_TEX, 1)).astype(np.float32)
    return {
        "vertex": vertex_code,
        "tri": make_mesh(grid_u, grid_v, patch),
        "mu": mu,
        "pc_shape": pc_shape,
        "pc_exp": pc_exp,
        "mu_tex": mu_tex,
        "pc_tex": pc_tex,
        "param_tex": param_tex,
        "ndim_shape": n_shape,
        "ndim_exp": n_exp,
        "ndim_pose": N_POSE,
    }


def make_small_assets(grid_u=20, grid_v=24, n_shape=9, n_exp=5, scale_px=None, seed_basis=77):
    """A tiny asset set with the same structure (for CPU-speed parity tests)."""
    return make_assets(grid_u, grid_v, n_shape, n_exp, patch=(3, 4, 2, 3), seed_basis=seed_basis)


def get_random_params(im_size, num_shape_param, num_exp_param, beta=1.0, rand=None):
    """The reference sampler (rendering_layer/sample_test.py:23-38 == prepare_data/get_random_params.m:1-11):
    phi in U[-75,45] deg, gamma in U[-90,90] deg, theta in U[-30,30] deg (radians), f in U[0,1e-3],
    t3d = (U[0,60], U[0,60], 0); pose = beta*[0,0,0,im/2,im/2,0,1e-3] + (1-beta)*rand;
    shape in U[0,1e4]^ns, exp in U[-1.5,1.5]^ne.  `rand` is a numpy-legacy-style callable (default
    numpy.random.rand) and is drawn in the reference's order, so a seeded legacy generator reproduces it."""
    if rand is None:
        rand = np.random.rand
    phi = (-75 + 120 * rand()) * np.pi / 180
    gamma = (-90 + 180 * rand()) * np.pi / 180
    theta = (-30 + 60 * rand()) * np.pi / 180
    focal_factor = rand() * 1e-3
    t3d = np.vstack([rand(2, 1) * 60, np.array([[0.0]], dtype=np.float32)])
    pose_rand = np.vstack([np.array([[phi], [gamma], [theta]], dtype=np.float32), t3d,
                           np.array([[focal_factor]], dtype=np.float32)])
    pose_base = np.reshape(np.array([0, 0, 0, im_size / 2, im_size / 2, 0, 0.001], dtype=np.float32), [7, 1])
    pose_param = beta * pose_base + (1 - beta) * pose_rand
    shape_param = rand(num_shape_param, 1) * 1e04
    exp_param = -1.5 + 3 * rand(num_exp_param, 1)
    return pose_param, shape_param, exp_param


def sample_params_batch(batch, im_size=200, n_shape=N_SHAPE, n_exp=N_EXP, beta=0.7, seed=3456):
    """[batch, 7+n_shape+n_exp] float32 parameter vectors in the 235-d layout
    [phi,gamma,theta,tx,ty,tz,f | shape | exp] (reference README.md:43-46, network.py:253-263)."""
    rs = np.random.RandomState(seed)
    out = np.empty((batch, N_POSE + n_shape + n_exp), np.float32)
    for i in range(batch):
        pose, shp, exp = get_random_params(im_size, n_shape, n_exp, beta, rand=rs.rand)
        out[i] = np.concatenate([pose[:, 0], shp[:, 0], exp[:, 0]]).astype(np.float32)
    return out
