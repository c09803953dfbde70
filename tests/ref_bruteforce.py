"""TEST-SIDE brute-force rasteriser that pins fr_oracle_render_depth_forward as a whole function to the reference.

It restates the two loops of RenderDepth(CPUDevice) (rendering_layer/ops_src/render_depth_op.cc:201-246 per-triangle
setup, :263-316 raster + z-test) in plain numpy / Python, sequentially in triangle order with the reference's
`depth < h && PointInTri(...)` overwrite -- NOT the packed-key max the HIP kernels use -- and its ONLY inside test is the
reference's own PointInTri compiled from /root/reference into oracle/_ref/libref_pit.so (oracle/Makefile).  It shares no
code with oracle/fr_oracle.c.  Slow by design; used on small scenes and one full-size face.
"""
import numpy as np

F32 = np.float32
INT_MIN = -2 ** 31


def _int_cast(v):
    """(int) of a double on x86-64 (cvttsd2si): truncation; NaN / out of range -> INT_MIN."""
    if not np.isfinite(v) or v >= 2.0 ** 31 or v < -2.0 ** 31:
        return INT_MIN
    return int(v)  # truncates toward zero


def _mn(a, b):  # render_depth_op.h:15  #define min(a,b) (((a) < (b)) ? (a) : (b))
    return a if a < b else b


def _mx(a, b):  # render_depth_op.h:16
    return a if a > b else b


def render_depth_bruteforce(vertex, tri, texture, H, W, pit_batch):
    """vertex [B,3,N] f32, tri [3,T] f32, texture [Bt,3,N] f32 -> (depth, texture_image, normal, tri_ind).
    pit_batch(P[n,8] float64) -> bool[n] must be the REFERENCE's PointInTri."""
    vertex = np.ascontiguousarray(vertex, F32)
    tri = np.ascontiguousarray(tri, F32)
    texture = np.ascontiguousarray(texture, F32)
    if texture.ndim == 2:
        texture = texture[None]
    B, _, nver = vertex.shape
    ntri = tri.shape[1]
    depth = np.empty((B, H, W, 1), F32)
    teximg = np.zeros((B, H, W, 3), F32)
    normal = np.zeros((B, H, W, 3), F32)
    tind = np.empty((B, H, W, 1), F32)
    three = F32(3.0)
    for b in range(B):
        depth[b] = F32(-99999999999999.0)   # :186
        tind[b] = F32(-1.0)                 # :190
        tex_b = texture[b if texture.shape[0] > 1 else 0]
        cand = []   # (triangle, x, y) in the reference's visiting order
        rows = []
        setup = {}
        for i in range(ntri):
            p = [_int_cast(float(tri[k, i])) for k in range(3)]                      # :204-206
            if any(q < 0 or q >= nver for q in p):
                continue   # deviation 3: the reference would read out of bounds
            p1, p2, p3 = p
            x = [float(vertex[b, 0, q]) for q in p]                                  # :208-213 widened to double
            y = [float(vertex[b, 1, q]) for q in p]
            x_min = _int_cast(np.ceil(_mn(_mn(x[0], x[1]), x[2])))                   # :276-280
            x_max = _int_cast(np.floor(_mx(_mx(x[0], x[1]), x[2])))
            y_min = _int_cast(np.ceil(_mn(_mn(y[0], y[1]), y[2])))
            y_max = _int_cast(np.floor(_mx(_mx(y[0], y[1]), y[2])))
            if x_max < x_min or y_max < y_min or x_max > W - 1 or x_min < 0 or y_max > H - 1 or y_min < 0:   # :282
                continue
            z = vertex[b, 2, p]
            h = F32(F32(F32(z[0] + z[1]) + z[2]) / three)                            # :217 (fp32, left to right)
            tt = tex_b[:, p]
            tritex = (F32(tt[:, 0] + tt[:, 1]) + tt[:, 2]).astype(F32) / three       # :223
            d12 = (vertex[b, :, p1] - vertex[b, :, p2]).astype(np.float64)           # :227-232 fp32 differences, widened
            d13 = (vertex[b, :, p1] - vertex[b, :, p3]).astype(np.float64)
            nrm = np.array([d12[1] * d13[2] - d12[2] * d13[1],                       # :234-236
                            d12[2] * d13[0] - d12[0] * d13[2],
                            d12[0] * d13[1] - d12[1] * d13[0]]).astype(F32)
            setup[i] = (h, tritex.astype(F32), nrm)
            for xx in range(x_min, x_max + 1):                                       # :285-288 (x outer, y inner)
                for yy in range(y_min, y_max + 1):
                    cand.append((i, xx, yy))
                    rows.append((xx, yy, x[0], y[0], x[1], y[1], x[2], y[2]))
        inside = pit_batch(np.array(rows, np.float64).reshape(-1, 8)) if rows else np.zeros(0, bool)
        for (i, xx, yy), ins in zip(cand, inside):
            h, tritex, nrm = setup[i]
            if float(depth[b, yy, xx, 0]) < float(h) and ins:                        # :295
                depth[b, yy, xx, 0] = h
                teximg[b, yy, xx] = tritex
                normal[b, yy, xx] = nrm
                tind[b, yy, xx, 0] = F32(i)
    return depth, teximg, normal, tind


def render_depth_grad_bruteforce(depth_grad, tri, tri_ind, nver):
    """TEST-SIDE restatement of RenderDepthGrad(CPUDevice) as a WHOLE function (render_depth_op.cc:344-366): plain Python,
    pixels in row-major order (b, then j = row, then i = column: :344-346), `depth_grad_ * 1.0f / 3.0f` evaluated in fp32
    left to right (:361-363) and added with a sequential fp32 `+=` to the z row of the triangle's three vertices, in the
    order p1, p2, p3.  Shares no code with oracle/fr_oracle.c.  Under the two stated deviations:
      1. vertex_grad starts from zero (the reference adds into an uninitialised output);
      2. pixels with tri_ind outside [0, ntri) are skipped (the reference indexes tri(., -1));
      3. triangles with a vertex id outside [0, nver) are skipped (the reference would write out of bounds).
    depth_grad [B,H,W,1] f32, tri [3,T] f32, tri_ind [B,H,W,1] f32 -> vertex_grad [B,3,nver] f32."""
    g = np.ascontiguousarray(depth_grad, F32)
    tri = np.ascontiguousarray(tri, F32)
    ti = np.ascontiguousarray(tri_ind, F32)
    B, H, W = g.shape[:3]
    ntri = tri.shape[1]
    out = np.zeros((B, 3, nver), F32)
    one, three = F32(1.0), F32(3.0)
    with np.errstate(all="ignore"):
        for b in range(B):
            gz = out[b, 2]
            for j in range(H):
                for i in range(W):
                    t = _int_cast(float(ti[b, j, i, 0]))          # int tri_ind_ = tri_ind(b,j,i,0)   (:350)
                    if t < 0 or t >= ntri:
                        continue
                    p = [_int_cast(float(tri[k, t])) for k in range(3)]   # :351-353
                    if any(q < 0 or q >= nver for q in p):
                        continue
                    c = F32(F32(g[b, j, i, 0] * one) / three)     # depth_grad_ * 1.0f / 3.0f, fp32, left to right
                    for q in p:                                   # :361-363, p1 then p2 then p3
                        gz[q] = F32(gz[q] + c)
    return out
