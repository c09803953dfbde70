"""ctypes front-end of the CPU oracle (oracle/fr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (3dfacerecon_amd/) never imports this module.

Every function takes / returns numpy arrays in the reference's layouts:
  vertex [B,3,N] f32, tri [3,T] f32 (float-stored indices), texture [B|1,3,N] f32,
  outputs depth [B,H,W,1], texture_image [B,H,W,3], normal [B,H,W,3], tri_ind [B,H,W,1]
  (rendering_layer/ops_src/render_depth_op.cc:437-445).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfr_oracle.so")
_REF_PIT_PATH = os.path.join(_HERE, "_ref", "libref_pit.so")

BG_DEPTH = np.float32(-99999999999999.0)  # render_depth_op.cc:186 -> -100000000376832.0

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present) with oracle/Makefile."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "fr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libfr_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF_PIT_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.fr_oracle_point_in_tri_op.argtypes = [ctypes.c_double] * 8
        L.fr_oracle_point_in_tri_op.restype = ctypes.c_int
        L.fr_oracle_point_in_tri_mex.argtypes = [ctypes.c_double] * 8
        L.fr_oracle_point_in_tri_mex.restype = ctypes.c_int
        L.fr_oracle_render_depth_forward.argtypes = [_f32p, _f32p, _f32p] + [ctypes.c_int] * 7 + [_f32p] * 4
        L.fr_oracle_render_depth_forward.restype = ctypes.c_int
        L.fr_oracle_render_depth_backward.argtypes = [_f32p, _f32p, _f32p] + [ctypes.c_int] * 5 + [_f32p]
        L.fr_oracle_render_depth_backward.restype = ctypes.c_int
        L.fr_oracle_zbuffer.argtypes = [_f64p, _f64p, _f64p, ctypes.c_int, ctypes.c_int, _f64p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, _f64p, _f64p]
        L.fr_oracle_zbuffer.restype = ctypes.c_int
        L.fr_oracle_rotation_matrix.argtypes = [ctypes.c_float] * 3 + [_f32p]
        L.fr_oracle_rotation_matrix.restype = None
        for name in ("fr_oracle_decode_3dmm", "fr_oracle_decode_3dmm_nofma", "fr_oracle_decode_3dmm_q30"):
            fn = getattr(L, name)
            fn.argtypes = [_f32p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_float, _f32p]
            fn.restype = ctypes.c_int
        L.fr_oracle_decode_3dmm_q30_lv.argtypes = [_f32p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, _f32p]
        L.fr_oracle_decode_3dmm_q30_lv.restype = ctypes.c_int
        L.fr_oracle_decode_3dmm_f64.argtypes = [_f32p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_double, _f64p]
        L.fr_oracle_decode_3dmm_f64.restype = ctypes.c_int
        L.fr_oracle_decode_3dmm_backward_f64.argtypes = [_f32p] * 5 + [ctypes.c_int] * 4 + [_f64p]
        L.fr_oracle_decode_3dmm_backward_f64.restype = ctypes.c_int
        _lib = L
    return _lib


def _c32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_f64p)


def point_in_tri_op(p, p1, p2, p3):
    return bool(lib().fr_oracle_point_in_tri_op(p[0], p[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]))


def point_in_tri_mex(p, p1, p2, p3):
    return bool(lib().fr_oracle_point_in_tri_mex(p[0], p[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]))


def render_depth(vertex, tri, texture, H, W):
    """CPU restatement of RenderDepth(CPUDevice) (render_depth_op.cc:132-322)."""
    vertex, vp = _c32(vertex)
    tri, tp = _c32(tri)
    texture, xp = _c32(texture)
    if texture.ndim == 2:
        texture = texture[None]
    B, three, nver = vertex.shape
    assert three == 3 and tri.shape[0] == 3 and texture.shape[1] == 3 and texture.shape[2] == nver
    ntri = tri.shape[1]
    depth = np.empty((B, H, W, 1), np.float32)
    tex = np.empty((B, H, W, 3), np.float32)
    nrm = np.empty((B, H, W, 3), np.float32)
    tind = np.empty((B, H, W, 1), np.float32)
    rc = lib().fr_oracle_render_depth_forward(vp, tp, xp, B, nver, ntri, H, W, 3, texture.shape[0],
                                              depth.ctypes.data_as(_f32p), tex.ctypes.data_as(_f32p),
                                              nrm.ctypes.data_as(_f32p), tind.ctypes.data_as(_f32p))
    if rc != 0:
        raise ValueError("fr_oracle_render_depth_forward rc=%d" % rc)
    return depth, tex, nrm, tind


def render_depth_grad(depth_grad, tri, tri_ind, nver):
    """CPU restatement of RenderDepthGrad(CPUDevice) (render_depth_op.cc:325-368), zero-initialised."""
    depth_grad, gp = _c32(depth_grad)
    tri, tp = _c32(tri)
    tri_ind, ip = _c32(tri_ind)
    B, H, W = depth_grad.shape[:3]
    out = np.empty((B, 3, nver), np.float32)
    rc = lib().fr_oracle_render_depth_backward(gp, tp, ip, B, nver, tri.shape[1], H, W,
                                               out.ctypes.data_as(_f32p))
    if rc != 0:
        raise ValueError("fr_oracle_render_depth_backward rc=%d" % rc)
    return out


def zbuffer_mex(vertex_3xN, tri_3xT, texture_CxN, src_img_HxWxC):
    """CPU restatement of MM3D::ZBuffer (ModalAndRef.cpp:3-94) behind Mex_ZBuffer.m:1-6.

    Arguments are numpy arrays shaped as MATLAB sees them (vertex 3xN, tri 3xT, texture CxN, img HxWxC);
    MATLAB's column-major memory is reproduced with order='F'.  Returns (img HxWxC, tri_ind HxW) float64.
    """
    H, W, C = src_img_HxWxC.shape
    v = np.asfortranarray(vertex_3xN, dtype=np.float64)
    t = np.asfortranarray(tri_3xT, dtype=np.float64)
    x = np.asfortranarray(texture_CxN, dtype=np.float64)
    s = np.asfortranarray(src_img_HxWxC, dtype=np.float64)
    img = np.empty((H, W, C), np.float64, order="F")
    tind = np.empty((H, W), np.float64, order="F")
    rc = lib().fr_oracle_zbuffer(v.ctypes.data_as(_f64p), t.ctypes.data_as(_f64p), x.ctypes.data_as(_f64p),
                                 v.shape[1], t.shape[1], s.ctypes.data_as(_f64p), W, H, C,
                                 img.ctypes.data_as(_f64p), tind.ctypes.data_as(_f64p))
    if rc != 0:
        raise ValueError("fr_oracle_zbuffer rc=%d" % rc)
    return img, tind


def rotation_matrix(angles):
    """CPU restatement of FaceRecNet.rotation_matrix (nets/network.py:266-291): (3,) -> (3,3) fp32."""
    a = np.asarray(angles, dtype=np.float32).reshape(3)
    R = np.empty(9, np.float32)
    lib().fr_oracle_rotation_matrix(float(a[0]), float(a[1]), float(a[2]), R.ctypes.data_as(_f32p))
    return R.reshape(3, 3)


def rotation_matrix_batch(angles_batch):
    """nets/network.py:292-297."""
    a = np.asarray(angles_batch, dtype=np.float32)
    return np.stack([rotation_matrix(a[i]) for i in range(a.shape[0])]) if a.shape[0] else np.zeros((0, 3, 3), np.float32)


def decode_3dmm(params, mu, pc_shape, pc_exp, im_size, R=None, nofma=False, q30=False):
    """CPU restatement of FaceRecNet.vertices_transform (nets/network.py:140-171) under the written fp32 spec
    (q30=True or a level count 4 / 5 / 7: under the written fixed-point spec of the product's int8-MFMA decode, fr_oracle.c
    "Q30 decode"; True = 7 = all sixteen digit products)."""
    params, pp = _c32(params)
    mu, mp = _c32(np.asarray(mu).reshape(-1))
    pc_shape, sp = _c32(pc_shape)
    pc_exp, ep = _c32(pc_exp)
    B = params.shape[0]
    N = mu.shape[0] // 3
    ns, ne = pc_shape.shape[1], pc_exp.shape[1]
    assert params.shape[1] == 7 + ns + ne and pc_shape.shape[0] == 3 * N and pc_exp.shape[0] == 3 * N
    if R is not None:
        R, rp = _c32(np.asarray(R).reshape(B, 9))
    else:
        rp = None
    out = np.empty((B, 3, N), np.float32)
    fn = lib().fr_oracle_decode_3dmm_nofma if nofma else lib().fr_oracle_decode_3dmm
    if q30:
        levels = 7 if q30 is True else int(q30)
        rc = lib().fr_oracle_decode_3dmm_q30_lv(pp, mp, sp, ep, rp, B, N, ns, ne, float(im_size), levels,
                                                out.ctypes.data_as(_f32p))
    else:
        rc = fn(pp, mp, sp, ep, rp, B, N, ns, ne, float(im_size), out.ctypes.data_as(_f32p))
    if rc != 0:
        raise ValueError("fr_oracle_decode_3dmm rc=%d" % rc)
    return out


def decode_3dmm_q30(params, mu, pc_shape, pc_exp, im_size, R=None, levels=7):
    if not 1 <= int(levels) <= 7:
        raise ValueError("Q30 levels must be in 1..7")
    return decode_3dmm(params, mu, pc_shape, pc_exp, im_size, R=R, q30=int(levels))


def decode_3dmm_f64(params, mu, pc_shape, pc_exp, im_size):
    """Same formula evaluated in float64 (tolerance reference for the fp32 spec)."""
    params, pp = _c32(params)
    mu, mp = _c32(np.asarray(mu).reshape(-1))
    pc_shape, sp = _c32(pc_shape)
    pc_exp, ep = _c32(pc_exp)
    B = params.shape[0]
    N = mu.shape[0] // 3
    out = np.empty((B, 3, N), np.float64)
    rc = lib().fr_oracle_decode_3dmm_f64(pp, mp, sp, ep, B, N, pc_shape.shape[1], pc_exp.shape[1],
                                         float(im_size), out.ctypes.data_as(_f64p))
    if rc != 0:
        raise ValueError("fr_oracle_decode_3dmm_f64 rc=%d" % rc)
    return out


def decode_3dmm_backward_f64(grad_vertex_proj, params, mu, pc_shape, pc_exp):
    """float64 gradient of the parameters given dL/d vertex_proj [B,3,N] (what TF autodiff yields for
    nets/network.py:140-171: no gradient to the three angles, which pass through tf.py_func)."""
    g, gp = _c32(grad_vertex_proj)
    params, pp = _c32(params)
    mu, mp = _c32(np.asarray(mu).reshape(-1))
    pc_shape, sp = _c32(pc_shape)
    pc_exp, ep = _c32(pc_exp)
    B, N = params.shape[0], mu.shape[0] // 3
    out = np.empty((B, params.shape[1]), np.float64)
    rc = lib().fr_oracle_decode_3dmm_backward_f64(gp, pp, mp, sp, ep, B, N, pc_shape.shape[1], pc_exp.shape[1],
                                                  out.ctypes.data_as(_f64p))
    if rc != 0:
        raise ValueError("fr_oracle_decode_3dmm_backward_f64 rc=%d" % rc)
    return out


# ---- the reference's own PointInTri, compiled from /root/reference into oracle/_ref (if present) ----
_ref_pit = None


def ref_point_in_tri_available():
    return os.path.exists(_REF_PIT_PATH)


def ref_point_in_tri(p, p1, p2, p3):
    """Calls the reference's PointInTri (render_depth_op.cc:76-122) from oracle/_ref/libref_pit.so."""
    global _ref_pit
    if _ref_pit is None:
        L = ctypes.CDLL(_REF_PIT_PATH)
        fn = getattr(L, "_Z10PointInTriPdS_S_S_")  # bool PointInTri(double*, double*, double*, double*)
        fn.argtypes = [_f64p] * 4
        fn.restype = ctypes.c_bool
        _ref_pit = fn
    arrs = [(ctypes.c_double * 2)(float(q[0]), float(q[1])) for q in (p, p1, p2, p3)]
    return bool(_ref_pit(*arrs))


def ref_point_in_tri_batch(P):
    """P: [n, 8] float64 rows (px,py,x1,y1,x2,y2,x3,y3) -> bool[n] via the reference binary."""
    global _ref_pit
    P = np.ascontiguousarray(P, dtype=np.float64)
    ref_point_in_tri((0, 0), (0, 0), (1, 0), (0, 1))  # load
    out = np.empty(P.shape[0], bool)
    base = P.ctypes.data
    for i in range(P.shape[0]):
        a = base + i * 64
        out[i] = _ref_pit(ctypes.cast(a, _f64p), ctypes.cast(a + 16, _f64p), ctypes.cast(a + 32, _f64p),
                          ctypes.cast(a + 48, _f64p))
    return out


def point_in_tri_op_batch(P):
    P = np.ascontiguousarray(P, dtype=np.float64)
    fn = lib().fr_oracle_point_in_tri_op
    return np.array([fn(*row) for row in P.tolist()], bool)


def set_constraints(pred_params, im_size, ndim_pose=7, ndim_shape=199):
    """numpy fp32 restatement of FaceRecNet.set_constraints (nets/network.py:204-218): sigmoid, then
    angles -> [-1.5,1.5] (s*3-1.5), tx,ty -> [0,im_size] (s*im_size), tz -> 0 (s*0), f -> [0,1e-3] (s*1e-3),
    shape -> [0,1e4] (s*1e4), expression -> [-1.5,1.5] (s*3-1.5).  pred_params: (..., d) float32.
    The sigmoid is 1/(1+exp(-x)) in fp32 (TF 1.2's own kernel bits are not observable here: tests compare the HIP-side
    torch.sigmoid to this within a few ulp, and this function bit-for-bit with tests/golden/set_constraints_ref.npz)."""
    x = np.asarray(pred_params, np.float32)
    with np.errstate(over="ignore"):
        s = (np.float32(1.0) / (np.float32(1.0) + np.exp(-x))).astype(np.float32)
    out = np.empty_like(s)
    out[..., 0:3] = s[..., 0:3] * np.float32(3.0) - np.float32(1.5)
    out[..., 3:5] = s[..., 3:5] * np.float32(im_size)
    out[..., 5] = s[..., 5] * np.float32(0.0)
    out[..., 6] = s[..., 6] * np.float32(1e-3)
    ps = ndim_pose + ndim_shape
    out[..., ndim_pose:ps] = s[..., ndim_pose:ps] * np.float32(1e4)
    out[..., ps:] = s[..., ps:] * np.float32(3.0) - np.float32(1.5)
    return out


def decode_3dmm_blas(params, mu, pc_shape, pc_exp, im_size, R=None):
    """numpy transcription of FaceRecNet.vertices_transform (nets/network.py:140-171) the way the reference's graph
    evaluates it: two dense fp32 matmuls through BLAS ([3N x ns] . [ns x B], [3N x ne] . [ne x B]), then the pose
    product and the y flip.  Its summation order is BLAS's, not the written spec's, so it is NOT the parity oracle
    (decode_3dmm is): it exists as the realistic CPU-baseline timing of the decode (bench.py cpu_baseline) and is held
    to the float64 evaluation by tolerance in tests/test_oracle_kat.py."""
    P = np.asarray(params, np.float32)
    mu = np.asarray(mu, np.float32).reshape(-1)
    pc_shape = np.asarray(pc_shape, np.float32)
    pc_exp = np.asarray(pc_exp, np.float32)
    B, N = P.shape[0], mu.shape[0] // 3
    ns, ne = pc_shape.shape[1], pc_exp.shape[1]
    pose, alpha, beta = P[:, :7], P[:, 7:7 + ns], P[:, 7 + ns:7 + ns + ne]
    Rm = rotation_matrix_batch(pose[:, :3]) if R is None else np.asarray(R, np.float32).reshape(B, 3, 3)
    shapes = (pc_shape @ alpha.T).T.reshape(B, 3, N)          # network.py:153-154 (blocked rows: coordinate r // N)
    exprs = (pc_exp @ beta.T).T.reshape(B, 3, N)              # :155-156
    vertex = (mu.reshape(1, 3, N) + shapes) + exprs           # :157-159
    M = pose[:, 6].reshape(B, 1, 1) * Rm                      # f (.) R, :163-165
    proj = np.matmul(M, vertex) + pose[:, 3:6].reshape(B, 3, 1)   # :165
    proj[:, 1] = (np.float32(im_size) - proj[:, 1]) - np.float32(1.0)   # :167-169
    return proj.astype(np.float32)
