"""GPU parity: fr_render_depth_forward (HIP, through rendering_layer/ops.py and the C ABI) vs the CPU oracle.
Bar: bit-exact on all four planes (tri_ind / coverage are integers; depth, texture, normal are pure functions of
the fp32 inputs under the reference's operation order)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, grid_from_rows, kat_inputs, pkg
from gpu_util import assert_render_equal, ops, render_gpu

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(GOLDEN, "kat_survey.json")))


@pytest.mark.parametrize("case", [c for c in KAT["cases"] if c["flavour"] == "op"], ids=lambda c: c["name"])
def test_kat(oracle, case):
    W, H = KAT["W"], KAT["H"]
    ver, tri, tex = kat_inputs(case, W, H)
    got = render_gpu(ver, tri, tex, H, W)
    if "tri_ind" in case:
        np.testing.assert_array_equal(got[3][0, :, :, 0], grid_from_rows(case["tri_ind"]))
    assert_render_equal(got, oracle.render_depth(ver, tri, tex, H, W), case["name"])


def test_golden_small(oracle, small_assets):
    z = np.load(os.path.join(GOLDEN, "render_small_oracle.npz"))
    got = render_gpu(z["vertex"], small_assets["tri"], small_assets["vertex"][None], int(z["H"]), int(z["W"]))
    assert_render_equal(got, (z["depth"], z["texture_image"], z["normal"], z["tri_ind"]), "golden_small")


def _random_scene(rs, B, nver, ntri, H, W, scale):
    ver = np.empty((B, 3, nver), np.float32)
    ver[:, 0] = rs.uniform(-0.1 * W, 1.1 * W, (B, nver))
    ver[:, 1] = rs.uniform(-0.1 * H, 1.1 * H, (B, nver))
    ver[:, 2] = rs.uniform(-50, 50, (B, nver))
    # triangles: a base vertex plus two near-by ones so that sizes range from sub-pixel to tens of pixels
    base = rs.randint(0, nver, ntri)
    tri = np.stack([base, rs.randint(0, nver, ntri), rs.randint(0, nver, ntri)]).astype(np.float32)
    # make most triangles small: overwrite vertices 2,3 positions relative to base for a subset
    k = ntri // 2
    idx = rs.permutation(nver)[: min(nver, 3 * k) // 3 * 3].reshape(-1, 3)
    for b in range(B):
        c = ver[b, :2, idx[:, 0]]
        ver[b, :2, idx[:, 1]] = c + rs.uniform(-scale, scale, c.shape).astype(np.float32)
        ver[b, :2, idx[:, 2]] = c + rs.uniform(-scale, scale, c.shape).astype(np.float32)
    m = min(k, idx.shape[0])
    tri[:, :m] = idx[:m].T
    tex = rs.uniform(0, 1, (B, 3, nver)).astype(np.float32)
    return ver, tri, tex


@pytest.mark.parametrize("B,nver,ntri,H,W,scale", [
    (1, 50, 80, 16, 16, 3.0),
    (3, 300, 700, 33, 47, 2.0),     # odd sizes: scalar store path, ragged strips
    (2, 1000, 3000, 64, 64, 1.0),
    (5, 400, 900, 200, 200, 8.0),
    (1, 200, 300, 7, 450, 5.0),     # wide image (448+ wide rows)
    (4, 64, 40, 100, 3, 2.0),       # narrow image
    (70, 120, 200, 24, 28, 2.5),    # more faces than one pass of bins per CU
    (2, 50, 100, 1, 1, 1.0),        # one-pixel image
    (2, 50, 100, 2, 3, 1.0),        # fewer rows than the 8x4 hit-mask window
    (1, 60, 150, 3, 64, 2.0),
])
def test_random_scenes(oracle, B, nver, ntri, H, W, scale):
    rs = np.random.RandomState(B * 1000 + ntri)
    ver, tri, tex = _random_scene(rs, B, nver, ntri, H, W, scale)
    assert_render_equal(render_gpu(ver, tri, tex, H, W), oracle.render_depth(ver, tri, tex, H, W), "random")


def _subpixel_mesh(rs, B, H, W, nu, nv, jitter):
    """A jittered nu x nv vertex grid stretched over the image (plus a margin): every cell is ~1 px, so the triangles are
    the sub-pixel kind the 3DMM mesh projects to, and every strip boundary is crossed by a row of them."""
    gx, gy = np.meshgrid(np.linspace(-2.0, W + 1.0, nv), np.linspace(-2.0, H + 1.0, nu))
    nver = nu * nv
    ver = np.empty((B, 3, nver), np.float32)
    for b in range(B):
        ver[b, 0] = (gx + rs.uniform(-jitter, jitter, gx.shape)).reshape(-1)
        ver[b, 1] = (gy + rs.uniform(-jitter, jitter, gy.shape)).reshape(-1)
        ver[b, 2] = rs.uniform(-5, 5, nver)
    iu, iv = np.meshgrid(np.arange(nu - 1), np.arange(nv - 1), indexing="ij")
    v00 = (iu * nv + iv).reshape(-1)
    tri = np.concatenate([np.stack([v00, v00 + nv, v00 + 1]), np.stack([v00 + 1, v00 + nv, v00 + nv + 1])], 1)
    tri = tri[:, rs.permutation(tri.shape[1])].astype(np.float32)   # no helpful ordering
    tex = rs.uniform(0, 1, (1, 3, nver)).astype(np.float32)
    return ver, tri, tex


@pytest.mark.parametrize("B,H,W,nu,nv", [
    (2, 60, 64, 90, 96),      # few faces: many thin strips (the strip count is capped by the offset table)
    (64, 40, 40, 56, 56),     # the bench's batch size: ~512 bins
    (9, 100, 37, 140, 50),    # odd sizes, ragged last strip
])
def test_subpixel_mesh_across_strip_boundaries(oracle, B, H, W, nu, nv):
    rs = np.random.RandomState(B * 7 + H)
    ver, tri, tex = _subpixel_mesh(rs, B, H, W, nu, nv, 0.45)
    want = oracle.render_depth(ver, tri, tex, H, W)
    assert (want[3] >= 0).mean() > 0.5     # the mesh really covers the image
    assert_render_equal(render_gpu(ver, tri, tex, H, W), want, "subpixel")


def test_pixel_centres_within_ulps_of_edges(oracle):
    """Triangles whose edges pass exactly through pixel centres, then with one vertex moved by -2..2 float ulps: the hit
    test must follow the reference's fp64 arithmetic bit for bit (u, v, u+v sit at 0 / 1 up to rounding)."""
    rs = np.random.RandomState(77)
    H = W = 48
    vs, ts = [], []
    for k in range(600):
        # an integer-coordinate pixel centre c, and an edge through it: a = c - d*s, b = c + d*t with a random direction
        c = rs.randint(4, 44, 2).astype(np.float64)
        d = rs.uniform(-1, 1, 2)
        a = c - d * rs.uniform(0.2, 3.0)
        b = c + d * rs.uniform(0.2, 3.0)
        third = c + rs.uniform(-3, 3, 2)
        pts = np.stack([a, b, third]).astype(np.float32)
        pts = pts[rs.permutation(3)]
        j, ax = rs.randint(0, 3), rs.randint(0, 2)
        steps = int(rs.randint(-2, 3))
        for _ in range(abs(steps)):
            pts[j, ax] = np.nextafter(pts[j, ax], np.float32(np.inf if steps > 0 else -np.inf))
        n0 = len(vs)
        for q in pts:
            vs.append([q[0], q[1], rs.uniform(-3, 3)])
        ts.append([n0, n0 + 1, n0 + 2])
    # exact cases too: vertices on half-integer / integer lattices (edges through many centres, zero-area slivers)
    for k in range(300):
        q = rs.randint(2, 2 * 44, (3, 2)).astype(np.float32) / 2.0
        n0 = len(vs)
        for r in q:
            vs.append([r[0], r[1], rs.uniform(-3, 3)])
        ts.append([n0, n0 + 1, n0 + 2])
    ver = np.asarray(vs, np.float32).T[None].copy()
    tri = np.asarray(ts, np.float32).T.copy()
    tex = rs.uniform(0, 1, (1, 3, ver.shape[2])).astype(np.float32)
    assert_render_equal(render_gpu(ver, tri, tex, H, W), oracle.render_depth(ver, tri, tex, H, W), "near-edge")
    # one triangle at a time (no occlusion hiding a wrong hit): coverage of each must match
    for k in range(0, 900, 9):
        t1 = tri[:, k:k + 1]
        assert_render_equal(render_gpu(ver, t1, tex, H, W), oracle.render_depth(ver, t1, tex, H, W), "near-edge single %d" % k)


def test_more_segments_than_resolver_threads(oracle):
    """> 512 * 504 triangles: the resolver walks the segment list in more than one chunk."""
    rs = np.random.RandomState(11)
    B, nver, ntri, H, W = 2, 5000, 300000, 48, 64
    ver, tri, tex = _random_scene(rs, B, nver, ntri, H, W, 1.5)
    assert_render_equal(render_gpu(ver, tri, tex, H, W), oracle.render_depth(ver, tri, tex, H, W), "many segments")


def test_shared_texture(oracle):
    rs = np.random.RandomState(3)
    ver, tri, tex = _random_scene(rs, 3, 100, 200, 20, 20, 3.0)
    want = oracle.render_depth(ver, tri, tex[:1], 20, 20)
    assert_render_equal(render_gpu(ver, tri, tex[:1], 20, 20), want, "tex_batch=1")
    assert_render_equal(render_gpu(ver, tri, tex[0], 20, 20), want, "tex 2-D")


def test_integer_coordinates_edge_rule(oracle):
    # pixel centres exactly on edges / vertices / hypotenuse: pins u+v<1 and the inclusive u,v bounds
    rs = np.random.RandomState(8)
    nver, ntri, H, W = 60, 150, 12, 12
    ver = np.empty((2, 3, nver), np.float32)
    ver[:, 0] = rs.randint(0, W, (2, nver))
    ver[:, 1] = rs.randint(0, H, (2, nver))
    ver[:, 2] = rs.randint(0, 4, (2, nver))          # many equal h -> ties to the lowest index
    tri = rs.randint(0, nver, (3, ntri)).astype(np.float32)
    tex = rs.uniform(0, 1, (2, 3, nver)).astype(np.float32)
    assert_render_equal(render_gpu(ver, tri, tex, H, W), oracle.render_depth(ver, tri, tex, H, W), "integer")


def test_big_triangles_and_full_cover(oracle):
    H, W = 40, 56
    ver = np.array([[[0, W - 1, 0, W - 1, 10.5, 30.2, 20.1], [0, 0, H - 1, H - 1, 5.5, 9.1, 30.7],
                     [1, 2, 3, 4, 9, 9, 9]]], np.float32)
    tri = np.array([[0, 1, 2], [3, 2, 1], [4, 5, 6]], np.float32).T.copy()
    tex = np.random.RandomState(1).uniform(0, 1, (1, 3, 7)).astype(np.float32)
    got = render_gpu(ver, tri, tex, H, W)
    assert (got[3] >= 0).mean() > 0.9
    assert_render_equal(got, oracle.render_depth(ver, tri, tex, H, W), "big")


def test_bad_indices_nan_inf_and_fractional_ids(oracle):
    rs = np.random.RandomState(5)
    ver, tri, tex = _random_scene(rs, 2, 80, 160, 24, 24, 3.0)
    tri[0, 3] = -1            # out of range -> triangle skipped (deviation 3)
    tri[1, 5] = 80            # == nver
    tri[2, 7] = 1e9
    tri[0, 9] = np.nan
    tri[1, 11] += 0.75        # (int) truncation of float-stored ids
    ver[0, 0, 5] = np.nan     # NaN x
    ver[0, 1, 6] = np.inf
    ver[1, 2, 7] = np.nan     # NaN z -> h NaN -> never drawn
    ver[1, 2, 8] = -np.inf
    ver[1, 2, 9] = -3e14      # below the background depth
    ver[0, 0, 10] = 1e20
    assert_render_equal(render_gpu(ver, tri, tex, 24, 24), oracle.render_depth(ver, tri, tex, 24, 24), "bad")


def test_signed_zero_depth_ties(oracle):
    # h = -0.0 and +0.0 compare equal in the serial code: lowest index wins
    ver = np.array([[[1, 6, 1, 1, 6, 1], [1, 1, 6, 1, 1, 6], [-0.0, -0.0, -0.0, 0.0, 0.0, 0.0]]], np.float32)
    tex = np.ones((1, 3, 6), np.float32)
    for tri in ([[0, 1, 2], [3, 4, 5]], [[3, 4, 5], [0, 1, 2]]):
        t = np.array(tri, np.float32).T.copy()
        got = render_gpu(ver, t, tex, 8, 8)
        want = oracle.render_depth(ver, t, tex, 8, 8)
        np.testing.assert_array_equal(got[3], want[3])
        np.testing.assert_array_equal(got[0], want[0])   # value-equal (+-0)


def test_empty_inputs(oracle):
    ver = np.zeros((2, 3, 5), np.float32)
    tex = np.zeros((2, 3, 5), np.float32)
    tri0 = np.zeros((3, 0), np.float32)
    got = render_gpu(ver, tri0, tex, 6, 9)
    assert_render_equal(got, oracle.render_depth(ver, tri0, tex, 6, 9), "ntri=0")
    assert np.all(got[3] == -1) and np.all(got[0] == np.float32(-100000000376832.0))
    # zero batch / zero-size image: no launch, empty outputs
    got = render_gpu(np.zeros((0, 3, 5), np.float32), np.zeros((3, 2), np.float32), np.zeros((0, 3, 5), np.float32), 6, 9)
    assert got[0].shape == (0, 6, 9, 1)
    got = render_gpu(ver, np.zeros((3, 2), np.float32), tex, 0, 9)
    assert got[1].shape == (2, 0, 9, 3)


def test_argument_errors():
    from gpu_util import ops
    o = ops()
    dev = torch.device("cuda:0")
    ver = torch.zeros((2, 3, 5), device=dev)
    tri = torch.zeros((3, 4), device=dev)
    tex = torch.zeros((2, 3, 5), device=dev)
    img = torch.zeros((2, 8, 8, 3), device=dev)
    with pytest.raises(ValueError, match="vertex's batch"):
        o.render_depth(ver[:1], tri, tex, img)
    with pytest.raises(ValueError, match="Batch x 3 x nver"):
        o.render_depth(torch.zeros((2, 2, 5), device=dev), tri, tex, img)
    with pytest.raises(ValueError, match="3 x ntri"):
        o.render_depth(ver, torch.zeros((2, 4), device=dev), tex, img)
    with pytest.raises(ValueError, match="texture channel"):
        o.render_depth(ver, tri, torch.zeros((2, 4, 5), device=dev), img)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        o.render_depth(ver.cpu(), tri, tex, img)
    with pytest.raises(TypeError):
        o.render_depth(ver.double(), tri, tex, img)


def test_full_size_face_bit_exact(oracle, full_assets, synth):
    """BFM-scale mesh (53,215 vertices, 105,840 triangles), 200x200, 2 faces: bit-exact vs the oracle."""
    A = full_assets
    P = synth.sample_params_batch(2, beta=0.7, seed=3456)
    V = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    want = oracle.render_depth(V, A["tri"], A["vertex"][None], 200, 200)
    got = render_gpu(V, A["tri"], A["vertex"], 200, 200)
    assert (want[3] >= 0).mean() > 0.2
    assert_render_equal(got, want, "full-size")


def test_batch64_properties(full_assets, synth, oracle):
    """BASELINE config 2 size (B=64): determinism, batch-permutation equivariance, and agreement of ALL 64 faces with
    the oracle (2.3 ms per face)."""
    A = full_assets
    P = synth.sample_params_batch(4, beta=0.7, seed=99)
    V4 = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    V = np.concatenate([V4] * 16, 0)
    rs = np.random.RandomState(0)
    V = V + rs.uniform(-0.3, 0.3, (64, 1, 1)).astype(np.float32)   # distinct faces
    got1 = render_gpu(V, A["tri"], A["vertex"], 200, 200)
    got2 = render_gpu(V, A["tri"], A["vertex"], 200, 200)
    assert_render_equal(got1, got2, "determinism")
    perm = rs.permutation(64)
    gotp = render_gpu(V[perm], A["tri"], A["vertex"], 200, 200)
    assert_render_equal(gotp, tuple(g[perm] for g in got1), "batch permutation")
    want = oracle.render_depth(V, A["tri"], A["vertex"][None], 200, 200)
    for b in range(64):
        assert_render_equal(tuple(g[b:b + 1] for g in got1), tuple(w[b:b + 1] for w in want), "face %d" % b)
    # triangle-order invariance of depth: reversing the triangle list changes tri_ind but not the depth map,
    # and coverage stays identical
    tri_rev = np.ascontiguousarray(A["tri"][:, ::-1])
    gotr = render_gpu(V[:8], tri_rev, A["vertex"], 200, 200)
    np.testing.assert_array_equal(gotr[0], got1[0][:8])
    np.testing.assert_array_equal(gotr[3] >= 0, got1[3][:8] >= 0)


def test_div3_matches_division_on_every_fp32():
    """The kernels' 3-instruction x / 3.0f (centroid depth render_depth_op.cc:217, texture mean :223) against the IEEE
    division, on all 2^32 fp32 bit patterns (NaN results compare equal as NaN)."""
    import ctypes
    L = pkg("_lib").lib()
    cnt = torch.zeros((1,), dtype=torch.int64, device="cuda:0")
    rc = L.fr_debug_div3_sweep(0, 1 << 32, ctypes.c_void_p(cnt.data_ptr()),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0


@pytest.mark.parametrize("env", [{"FR_RENDER_IMPL": 1}, {"FR_RENDER_ROWS": 1}, {"FR_RENDER_ROWS": 3},
                                 {"FR_RENDER_ROWS": 7}, {"FR_EMIT_FILTER": 0}, {"FR_EMIT_FILTER": 1},
                                 {"FR_EMIT_FILTER": 2}, {"FR_RESOLVE_OPT": 0}, {"FR_RESOLVE_BLOCK": 512},
                                 {"FR_RESOLVE_BLOCK": 1024}, {"FR_EMIT_ORDER": 0}, {"FR_EMIT_ORDER": 1}])
def test_fallback_and_row_override_paths(oracle, env):
    """render_strip_kernel (every bin scans every triangle) is the path for shapes the binned rasteriser rejects; the
    tuning knob FR_RENDER_ROWS below the hit-window height must route there too, larger overrides
    stay binned.  The A/B knobs of the fast paths (FR_EMIT_FILTER: 0 = every pixel through the reference's fp64 sequence,
    1 = certified fp32 test only, 2 = phase-A pre-cull only; FR_RESOLVE_OPT=0 = two-pass resolver; FR_EMIT_ORDER: the lane
    order of a segment's triangles forced to the list order / to even-then-odd instead of scored) must not change a bit
    either.  All bit-exact.  The knobs are set through fr_set_option (the environment is read once per process)."""
    rs = np.random.RandomState(11)
    scenes = [_random_scene(rs, 3, 300, 700, 33, 47, 2.0) + (33, 47), _random_scene(rs, 2, 400, 900, 64, 64, 6.0) + (64, 64)]
    ver, tri, tex = _subpixel_mesh(rs, 4, 40, 40, 56, 56, 0.45)
    scenes.append((ver, tri, tex, 40, 40))
    with pkg("_lib").options(**env):
        for ver, tri, tex, H, W in scenes:
            assert_render_equal(render_gpu(ver, tri, tex, H, W), oracle.render_depth(ver, tri, tex, H, W), str(env))


@pytest.mark.parametrize("order", ["cells", "halves", "shuffled"])
def test_lane_order_of_the_emit_table_follows_the_list_and_changes_nothing(oracle, order):
    """pack_tri_kernel writes every segment's triangles in the lane order that makes the emit kernel's gathers cheapest -- list
    order, or even triangles then odd ones when the list walks the grid cell by cell -- scored per segment.  Whatever the list
    looks like (cell by cell; all upper triangles, then all lower ones; shuffled) and whichever order is scored or forced,
    the planes are the oracle's.  2,090 triangles: four full segments and a ragged fifth."""
    rs = np.random.RandomState(17)
    B, H, W, nu, nv = 3, 48, 52, 20, 56
    gx, gy = np.meshgrid(np.linspace(-1.0, W, nv), np.linspace(-1.0, H, nu))
    nver = nu * nv
    ver = np.empty((B, 3, nver), np.float32)
    for b in range(B):
        ver[b, 0] = (gx + rs.uniform(-0.4, 0.4, gx.shape)).reshape(-1)
        ver[b, 1] = (gy + rs.uniform(-0.4, 0.4, gy.shape)).reshape(-1)
        ver[b, 2] = rs.uniform(-5, 5, nver)
    iu, iv = np.meshgrid(np.arange(nu - 1), np.arange(nv - 1), indexing="ij")
    v00 = (iu * nv + iv).reshape(-1)
    ta, tb = np.stack([v00, v00 + nv, v00 + 1]), np.stack([v00 + 1, v00 + nv, v00 + nv + 1])
    if order == "cells":
        tri = np.empty((3, 2 * v00.size), np.int64)
        tri[:, 0::2], tri[:, 1::2] = ta, tb
    else:
        tri = np.concatenate([ta, tb], 1)
        if order == "shuffled":
            tri = tri[:, rs.permutation(tri.shape[1])]
    tri = tri.astype(np.float32)
    assert tri.shape[1] == 2090
    tex = rs.uniform(0, 1, (1, 3, nver)).astype(np.float32)
    want = oracle.render_depth(ver, tri, tex, H, W)
    assert (want[3] >= 0).mean() > 0.5
    for forced in (-1, 0, 1):
        with pkg("_lib").options(FR_EMIT_ORDER=forced):
            assert_render_equal(render_gpu(ver, tri, tex, H, W), want, "%s list, FR_EMIT_ORDER=%d" % (order, forced))


def test_very_wide_image_takes_the_scan_path(oracle):
    """W = 6000: only three rows of keys fit the CU's LDS, shorter than the 8x4 hit window -> strip-scan path."""
    rs = np.random.RandomState(5)
    H, W = 9, 6000
    ver, tri, tex = _random_scene(rs, 1, 200, 400, H, W, 40.0)
    want = oracle.render_depth(ver, tri, tex, H, W)
    assert (want[3] >= 0).sum() > 100
    assert_render_equal(render_gpu(ver, tri, tex, H, W), want, "wide")


def test_certified_fp32_inside_filter_adversarial(oracle):
    """The emit kernel decides pixel-in-triangle in fp32 with error bounds and falls back to the reference's fp64 operation
    sequence when it cannot certify the decision.  Triangles built so that an edge passes a pixel centre at distances from
    1e-1 down to 1e-9 px on either side, slivers of aspect 10 .. 1e6, sub-ulp triangles, large (multi-pixel) and huge
    coordinates: coverage, depth, normals bit-exact against the oracle."""
    rs = np.random.RandomState(123)
    H = W = 64
    tris, zs = [], []

    def add(p1, p2, p3):
        tris.append((p1, p2, p3))
        zs.append(rs.uniform(1, 50, 3))
    for _ in range(6000):
        c = rs.randint(2, 62, 2).astype(np.float64)                   # the pixel centre under attack
        ang = rs.uniform(0, 2 * np.pi)
        d = 10.0 ** rs.uniform(-9, -1) * rs.choice([-1, 1])           # signed distance of the edge from the centre
        n = np.array([np.cos(ang), np.sin(ang)])
        tdir = np.array([-n[1], n[0]])
        L = 10.0 ** rs.uniform(-0.5, 0.7)                             # edge length 0.3 .. 5 px
        a = c + d * n + tdir * L * rs.uniform(0.2, 0.8)
        b = c + d * n - tdir * L * rs.uniform(0.2, 0.8)
        apex = c + n * (10.0 ** rs.uniform(-6, 0.5)) * rs.choice([-1, 1]) + tdir * rs.uniform(-0.3, 0.3)   # slivers too
        add(a, b, apex)
    for _ in range(500):                                              # sub-ulp / tiny triangles around a centre
        c = rs.randint(1, 63, 2).astype(np.float64)
        s = 10.0 ** rs.uniform(-7, -2)
        add(c + s * rs.uniform(-1, 1, 2), c + s * rs.uniform(-1, 1, 2), c + s * rs.uniform(-1, 1, 2))
    for _ in range(300):                                              # multi-pixel triangles (window path, several pixels)
        c = rs.uniform(4, 60, 2)
        add(c + rs.uniform(-3.4, 3.4, 2), c + rs.uniform(-3.4, 3.4, 2) * [1, 0.5], c + rs.uniform(-3.4, 3.4, 2) * [1, 0.5])
    nt = len(tris)
    ver = np.zeros((2, 3, 3 * nt), np.float32)
    P = np.array(tris, np.float64).reshape(nt * 3, 2)
    ver[0, 0], ver[0, 1], ver[0, 2] = P[:, 0], P[:, 1], np.array(zs).reshape(-1)
    ver[1] = ver[0]
    ver[1, :2] = ver[1, :2] + np.float32(0.5)                         # second face: every centre half a pixel away
    tri = np.arange(3 * nt, dtype=np.float32).reshape(nt, 3).T.copy()
    tex = rs.uniform(0, 1, (1, 3, 3 * nt)).astype(np.float32)
    want = oracle.render_depth(ver, tri, tex, H, W)
    assert (want[3] >= 0).mean() > 0.3
    assert_render_equal(render_gpu(ver, tri, tex, H, W), want, "certified filter")
    # a few hand-made extremes: exact zero-area, vertex on the centre, huge and tiny scales
    v = np.zeros((1, 3, 15), np.float32)
    pts = [(5, 5), (7, 5), (6, 5), (10, 10), (12, 10), (10, 12), (20, 20), (20 + 1e-7, 20), (20, 20 + 1e-7),
           (30.5, 30.5), (3e9, 30), (30, 3e9), (40, 40), (40.00001, 40.00002), (40.00002, 40.00001)]
    for k, (x, y) in enumerate(pts):
        v[0, 0, k], v[0, 1, k], v[0, 2, k] = x, y, 3 + k
    t5 = np.arange(15, dtype=np.float32).reshape(5, 3).T.copy()
    x5 = rs.uniform(0, 1, (1, 3, 15)).astype(np.float32)
    assert_render_equal(render_gpu(v, t5, x5, H, W), oracle.render_depth(v, t5, x5, H, W), "extremes")


def test_emit_phase_refuses_a_table_packed_for_other_arguments(oracle):
    """fr_render_depth_forward_phases(3) consumes whatever triangle table the workspace holds.  The table
    now names the (nver, ntri) it was packed for; an emit phase handed a table that was never packed, or packed for
    another shape, treats every triangle as invalid: pure background planes, no out-of-bounds gather, no error code."""
    import ctypes
    h = pkg("_lib")
    L = h.lib()
    rs = np.random.RandomState(21)
    H = W = 48
    dev = torch.device("cuda:0")

    def scene(nver, ntri):
        ver = np.empty((2, 3, nver), np.float32)
        ver[:, 0] = rs.uniform(0, W, (2, nver))
        ver[:, 1] = rs.uniform(0, H, (2, nver))
        ver[:, 2] = rs.uniform(-5, 5, (2, nver))
        tri = rs.randint(0, nver, (3, ntri)).astype(np.float32)
        tex = rs.uniform(0, 1, (2, 3, nver)).astype(np.float32)
        return ver, tri, tex

    def run(ver, tri, tex, ws, phases):
        t = [torch.as_tensor(a, device=dev) for a in (ver, tri, tex)]
        outs = [torch.full((2, H, W, c), 7.0, device=dev) for c in (1, 3, 3, 1)]
        nws = L.fr_render_depth_workspace_bytes(2, ver.shape[2], tri.shape[1], H, W)
        assert ws.numel() >= nws
        rc = L.fr_render_depth_forward_phases(h.ptr(t[0]), h.ptr(t[1]), h.ptr(t[2]), 2, ver.shape[2], tri.shape[1], H, W, 3,
                                              2, h.ptr(outs[0]), h.ptr(outs[1]), h.ptr(outs[2]), h.ptr(outs[3]), h.ptr(ws),
                                              nws, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), phases)
        assert rc == 0
        torch.cuda.synchronize()
        return tuple(o.cpu().numpy() for o in outs)

    a = scene(500, 900)
    b = scene(500, 1400)     # another ntri: the table sits elsewhere in the workspace and names other sizes
    ws = torch.zeros((L.fr_render_depth_workspace_bytes(2, 500, 1400, H, W) + 64,), dtype=torch.uint8, device=dev)
    empty = oracle.render_depth(a[0], np.zeros((3, 0), np.float32), a[2], H, W)
    assert_render_equal(run(*a, ws, 3), empty, "never packed")                 # zeroed workspace, no pack phase
    want_a = oracle.render_depth(*a, H, W)
    assert (want_a[3] >= 0).mean() > 0.3
    assert_render_equal(run(*a, ws, 7), want_a, "packed + rendered")
    assert_render_equal(run(*a, ws, 3), want_a, "table reused")
    assert_render_equal(run(*b, ws, 7), oracle.render_depth(*b, H, W), "other shape")
    # a workspace whose only table was packed for ANOTHER shape (pack phase alone: bit 4), then emit + resolve for `a`
    ws2 = torch.zeros_like(ws)
    run(*b, ws2, 4)
    assert_render_equal(run(*a, ws2, 3), empty, "table of another shape")


@pytest.mark.parametrize("nsegs,with_big", [(4, False), (8, False), (12, False), (5, True)])
def test_resolver_wave_local_front_overflow_and_fallback(oracle, nsegs, with_big):
    """The resolver's wave-local front (FR_RESOLVE_OPT=2) keeps six records per lane in registers, sends what a busy wave has
    beyond 384 through a reload loop and hands a bin whose wave holds more than 768 records -- or any big record -- to the
    general path.  One-pixel triangles, all of them inside strip 0, 504 per segment: with 4 segments every wave owns 504
    records (overflow loop), with 8 it owns 1,008 (general path), with 12 the first strip holds 6,048 records (longer than
    the LDS slot list: the search path); a half-screen triangle makes bucket 0 non-empty.  Many triangles share a pixel
    (depth ties and near-ties), every plane bit for bit; the same scene through the block-wide front (FR_RESOLVE_OPT=1) and
    the two-pass resolver (0) gives the same bits."""
    host = pkg("_lib")
    rs = np.random.RandomState(100 + nsegs)
    H = W = 64
    ntri = 504 * nsegs
    nver = 3 * ntri + 3
    px = rs.randint(0, W, ntri).astype(np.float32)
    py = rs.randint(0, 4, ntri).astype(np.float32)          # rows 0..3 = strip 0 at this size
    ver = np.zeros((1, 3, nver), np.float32)
    k = np.arange(ntri)
    jit = rs.uniform(-0.05, 0.05, (6, ntri)).astype(np.float32)
    ver[0, 0, 3 * k] = px - 0.3 + jit[0]; ver[0, 1, 3 * k] = py - 0.3 + jit[1]
    ver[0, 0, 3 * k + 1] = px + 0.4 + jit[2]; ver[0, 1, 3 * k + 1] = py - 0.2 + jit[3]
    ver[0, 0, 3 * k + 2] = px + jit[4]; ver[0, 1, 3 * k + 2] = py + 0.4 + jit[5]
    ver[0, 2, :3 * ntri] = np.repeat(rs.randint(0, 40, ntri).astype(np.float32) * 0.25, 3)   # few distinct depths: ties
    tri = np.stack([3 * k, 3 * k + 1, 3 * k + 2]).astype(np.float32)
    if with_big:   # one triangle over half the screen (inside it: the reference drops a triangle whose bbox leaves the image), behind everything
        ver[0, :, 3 * ntri:] = np.array([[0.5, 62.5, 0.5], [0.5, 0.5, 62.5], [-5, -5, -5]], np.float32)
        tri = np.concatenate([tri, np.array([[3 * ntri], [3 * ntri + 1], [3 * ntri + 2]], np.float32)], axis=1)
    tex = rs.uniform(0, 1, (1, 3, nver)).astype(np.float32)
    want = oracle.render_depth(ver, tri, tex, H, W)
    assert (want[3] >= 0).mean() > (0.4 if with_big else 0.05)
    for opt in (2, 1, 0):
        with host.options(FR_RESOLVE_OPT=opt):
            got = render_gpu(ver, tri, tex, H, W)
        assert_render_equal(got, want, "FR_RESOLVE_OPT=%d, %d segments%s" % (opt, nsegs, ", big triangle" if with_big else ""))


def test_render_depth_under_inference_mode(oracle):
    """A forward-only serving set-up: tensors created under torch.inference_mode() have no version counter (reading it
    raises).  The operator surface must render them -- the triangle table is then packed every call instead of being
    cached against a version it cannot see -- and an in-place change of `tri` must show in the very next call."""
    rs = np.random.RandomState(77)
    B, nver, ntri, H, W = 2, 60, 90, 24, 20
    ver = np.stack([rs.uniform(-2, W + 2, (B, nver)), rs.uniform(-2, H + 2, (B, nver)), rs.uniform(-5, 5, (B, nver))], 1).astype(np.float32)
    tri = rs.randint(0, nver, (3, ntri)).astype(np.float32)
    tex = rs.uniform(0, 1, (1, 3, nver)).astype(np.float32)
    dev = torch.device("cuda:0")
    with torch.inference_mode():
        v, t, x = (torch.as_tensor(a_, device=dev) for a_ in (ver, tri, tex))
        assert t.is_inference()
        img = torch.zeros((B, H, W, 3), device=dev)
        for rnd in range(2):
            got = tuple(o.cpu().numpy() for o in ops().render_depth(v, t, x, img))
            assert_render_equal(got, oracle.render_depth(ver, tri, tex, H, W), "inference mode, call %d" % rnd)
        tri2 = tri.copy()
        tri2[:, ::2] = tri2[::-1, ::2]                       # another triangle list, written IN PLACE into the same tensor
        t.copy_(torch.as_tensor(tri2, device=dev))
        got = tuple(o.cpu().numpy() for o in ops().render_depth(v, t, x, img))
        assert_render_equal(got, oracle.render_depth(ver, tri2, tex, H, W), "inference mode, tri rewritten in place")
