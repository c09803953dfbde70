/*
 * fr_oracle.c -- CPU ORACLE for the render_depth + 3DMM-decode hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker / reported CPU
 * baseline.  The product path (3dfacerecon_amd/) never imports, links or calls anything here.
 *
 * It is a clean-room restatement (no reference source text) of:
 *   - rendering_layer/ops_src/render_depth_op.cc:76-122   PointInTri          -> fr_oracle_point_in_tri_op
 *   - rendering_layer/ops_src/render_depth_op.cc:132-322  RenderDepth (CPU)   -> fr_oracle_render_depth_forward
 *   - rendering_layer/ops_src/render_depth_op.cc:325-368  RenderDepthGrad     -> fr_oracle_render_depth_backward
 *   - prepare_data/ZBuffer/ModalAndRef.cpp:3-94, 96-142   MM3D::ZBuffer/PointInTri -> fr_oracle_zbuffer
 *   - nets/network.py:140-171, 253-297                    vertices_transform + rotation -> fr_oracle_decode_3dmm
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - PointInTri is checked bit-for-bit against the reference's own function compiled from
 *     /root/reference by oracle/Makefile (oracle/_ref/libref_pit.so) -- tests/test_oracle_ref.py.
 *   - rotation_matrix / get_random_params are checked against fixtures produced by executing the
 *     reference's own numpy functions (tests/golden/make_golden.py).
 *   - RenderDepth / RenderDepthGrad / ZBuffer as whole functions need TensorFlow / OpenCV / MEX headers
 *     that are absent here, so the reference cannot be built; they are pinned only by the known-answer
 *     cases K1-K6 recorded in SURVEY.md section 8(a).  The decode's matmul order inside TF is unknowable:
 *     PARITY UNPINNED at that boundary (we define the summation spec below).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (x86-64 baseline, no FMA contraction), as the
 * reference builds its op with plain "g++ -std=c++11 -O2" (rendering_layer/ops.py:51).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

/* render_depth_op.h:15-16 defines min/max as these macros; NaN handling follows from them. */
#define FR_MIN(a, b) ((a) < (b) ? (a) : (b))
#define FR_MAX(a, b) ((a) > (b) ? (a) : (b))

/* (int) of an out-of-range / NaN double or float is UB in C; the reference runs on x86-64 where
 * cvttsd2si/cvttss2si return INT_MIN ("integer indefinite").  Make that explicit. */
static int fr_d2i(double d) {
    if (!(d > -2147483649.0 && d < 2147483648.0)) return INT_MIN;
    return (int)d;
}
static int fr_f2i(float f) {
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}

/* Background depth: the literal -99999999999999 stored to float (render_depth_op.cc:186). */
#define FR_BG_DEPTH ((float)(-99999999999999.0))

/* ---- R6: PointInTri, op flavour (u+v < 1), render_depth_op.cc:76-122 ------------------- */
int fr_oracle_point_in_tri_op(double px, double py, double x1, double y1, double x2, double y2,
                              double x3, double y3) {
    double v0x = x3 - x1, v0y = y3 - y1;
    double v1x = x2 - x1, v1y = y2 - y1;
    double v2x = px - x1, v2y = py - y1;
    double dot00 = v0x * v0x + v0y * v0y;
    double dot01 = v0x * v1x + v0y * v1y;
    double dot02 = v0x * v2x + v0y * v2y;
    double dot11 = v1x * v1x + v1y * v1y;
    double dot12 = v1x * v2x + v1y * v2y;
    double inv = 0;
    if ((dot00 * dot11 - dot01 * dot01) == 0)
        inv = 0;
    else
        inv = 1 / (dot00 * dot11 - dot01 * dot01);
    double u = (dot11 * dot02 - dot01 * dot12) * inv;
    if (u < 0 || u > 1) return 0;
    double v = (dot00 * dot12 - dot01 * dot02) * inv;
    if (v < 0 || v > 1) return 0;
    return u + v < 1;
}

/* ---- MEX flavour (u+v <= 1), ModalAndRef.cpp:96-142 ------------------------------------- */
int fr_oracle_point_in_tri_mex(double px, double py, double x1, double y1, double x2, double y2,
                               double x3, double y3) {
    double v0x = x3 - x1, v0y = y3 - y1;
    double v1x = x2 - x1, v1y = y2 - y1;
    double v2x = px - x1, v2y = py - y1;
    double dot00 = v0x * v0x + v0y * v0y;
    double dot01 = v0x * v1x + v0y * v1y;
    double dot02 = v0x * v2x + v0y * v2y;
    double dot11 = v1x * v1x + v1y * v1y;
    double dot12 = v1x * v2x + v1y * v2y;
    double inv = 0;
    if ((dot00 * dot11 - dot01 * dot01) == 0)
        inv = 0;
    else
        inv = 1 / (dot00 * dot11 - dot01 * dot01);
    double u = (dot11 * dot02 - dot01 * dot12) * inv;
    if (u < 0 || u > 1) return 0;
    double v = (dot00 * dot12 - dot01 * dot02) * inv;
    if (v < 0 || v > 1) return 0;
    return u + v <= 1;
}

/* ---- R3-R5: RenderDepth CPU functor, render_depth_op.cc:132-322 --------------------------
 * vertex [B,3,nver], tri [3,ntri] (float indices), texture [tex_batch,3,nver] with tex_batch in {1,B},
 * outputs depth [B,H,W,1], tex_img [B,H,W,3], normal [B,H,W,3], tri_ind [B,H,W,1].
 * Deviations from the literal reference (SURVEY.md 8a "Deviations" 3, 7): a triangle with a vertex id
 * outside [0,nver) is skipped (the reference reads out of bounds); scratch is heap, not 1 GB static.
 * Returns 0, or -1 on invalid sizes, -2 on allocation failure. */
int fr_oracle_render_depth_forward(const float* vertex, const float* tri, const float* texture, int B,
                                   int nver, int ntri, int H, int W, int C, int tex_batch, float* depth,
                                   float* tex_img, float* normal, float* tri_ind) {
    if (B < 0 || nver < 0 || ntri < 0 || H < 0 || W < 0 || C != 3) return -1;
    if (tex_batch != 1 && tex_batch != B) return -1;
    size_t nt = (size_t)(ntri > 0 ? ntri : 1);
    double* pt = (double*)malloc(nt * 6 * sizeof(double));      /* point1/2/3 xy   cc:126-128 */
    double* hh = (double*)malloc(nt * sizeof(double));          /* h               cc:129 */
    double* tritex = (double*)malloc(nt * 3 * sizeof(double));  /* tritex          cc:130 */
    double* trinrm = (double*)malloc(nt * 3 * sizeof(double));  /* tri_normal      cc:131 */
    unsigned char* ok = (unsigned char*)malloc(nt);
    if (!pt || !hh || !tritex || !trinrm || !ok) {
        free(pt); free(hh); free(tritex); free(trinrm); free(ok);
        return -2;
    }
    for (int b = 0; b < B; b++) {
        const float* vx = vertex + ((size_t)b * 3 + 0) * nver;
        const float* vy = vertex + ((size_t)b * 3 + 1) * nver;
        const float* vz = vertex + ((size_t)b * 3 + 2) * nver;
        const float* tx = texture + (size_t)(tex_batch == 1 ? 0 : b) * 3 * nver;
        float* d_b = depth + (size_t)b * H * W;
        float* t_b = tex_img + (size_t)b * H * W * 3;
        float* n_b = normal + (size_t)b * H * W * 3;
        float* i_b = tri_ind + (size_t)b * H * W;
        /* init, cc:182-192 and cc:255-261 */
        for (size_t q = 0; q < (size_t)H * W; q++) {
            d_b[q] = FR_BG_DEPTH;
            i_b[q] = -1;
            n_b[3 * q] = n_b[3 * q + 1] = n_b[3 * q + 2] = 0;
            t_b[3 * q] = t_b[3 * q + 1] = t_b[3 * q + 2] = 0;
        }
        /* per-triangle setup, cc:201-246 */
        for (int i = 0; i < ntri; i++) {
            int p1 = fr_f2i(tri[i]), p2 = fr_f2i(tri[(size_t)ntri + i]), p3 = fr_f2i(tri[2 * (size_t)ntri + i]);
            ok[i] = (p1 >= 0 && p1 < nver && p2 >= 0 && p2 < nver && p3 >= 0 && p3 < nver);
            if (!ok[i]) continue;
            pt[6 * (size_t)i + 0] = vx[p1]; pt[6 * (size_t)i + 1] = vy[p1];
            pt[6 * (size_t)i + 2] = vx[p2]; pt[6 * (size_t)i + 3] = vy[p2];
            pt[6 * (size_t)i + 4] = vx[p3]; pt[6 * (size_t)i + 5] = vy[p3];
            /* h is computed in fp32 then widened, cc:217 */
            hh[i] = (double)((vz[p1] + vz[p2] + vz[p3]) / 3.0f);
            for (int j = 0; j < 3; j++) {
                const float* tj = tx + (size_t)j * nver;
                tritex[3 * (size_t)i + j] = (tj[p1] + tj[p2] + tj[p3]) / 3.0f; /* fp32, cc:223 */
            }
            /* differences in fp32, cross product in double, cc:227-236 */
            double ax = vx[p1] - vx[p2], ay = vy[p1] - vy[p2], az = vz[p1] - vz[p2];
            double bx = vx[p1] - vx[p3], by = vy[p1] - vy[p3], bz = vz[p1] - vz[p3];
            trinrm[3 * (size_t)i + 0] = ay * bz - az * by;
            trinrm[3 * (size_t)i + 1] = az * bx - ax * bz;
            trinrm[3 * (size_t)i + 2] = ax * by - ay * bx;
        }
        /* raster + z-test in triangle order, cc:263-316 */
        for (int i = 0; i < ntri; i++) {
            if (!ok[i]) continue;
            const double* p = pt + 6 * (size_t)i;
            int x_min = fr_d2i(ceil((double)FR_MIN(FR_MIN(p[0], p[2]), p[4])));
            int x_max = fr_d2i(floor((double)FR_MAX(FR_MAX(p[0], p[2]), p[4])));
            int y_min = fr_d2i(ceil((double)FR_MIN(FR_MIN(p[1], p[3]), p[5])));
            int y_max = fr_d2i(floor((double)FR_MAX(FR_MAX(p[1], p[3]), p[5])));
            if (x_max < x_min || y_max < y_min || x_max > W - 1 || x_min < 0 || y_max > H - 1 || y_min < 0)
                continue;
            for (int x = x_min; x <= x_max; x++) {
                for (int y = y_min; y <= y_max; y++) {
                    size_t q = (size_t)y * W + x;
                    if ((double)d_b[q] < hh[i] &&
                        fr_oracle_point_in_tri_op((double)x, (double)y, p[0], p[1], p[2], p[3], p[4], p[5])) {
                        d_b[q] = (float)hh[i];
                        for (int j = 0; j < 3; j++) t_b[3 * q + j] = (float)tritex[3 * (size_t)i + j];
                        for (int j = 0; j < 3; j++) n_b[3 * q + j] = (float)trinrm[3 * (size_t)i + j];
                        i_b[q] = (float)i;
                    }
                }
            }
        }
    }
    free(pt); free(hh); free(tritex); free(trinrm); free(ok);
    return 0;
}

/* ---- R7: RenderDepthGrad CPU functor, render_depth_op.cc:325-368 --------------------------
 * vertex_grad [B,3,nver] = zeros, then per pixel in row-major order += depth_grad*1.0f/3.0f on the
 * z row of the three vertices of tri_ind.  Deviations 1-2: the output is zeroed first
 * (the reference accumulates into uninitialised memory) and pixels with tri_ind < 0 (or an id that is
 * out of range) are skipped (the reference indexes tri(k,-1)). */
int fr_oracle_render_depth_backward(const float* depth_grad, const float* tri, const float* tri_ind,
                                    int B, int nver, int ntri, int H, int W, float* vertex_grad) {
    if (B < 0 || nver < 0 || ntri < 0 || H < 0 || W < 0) return -1;
    memset(vertex_grad, 0, (size_t)B * 3 * nver * sizeof(float));
    for (int b = 0; b < B; b++) {
        float* gz = vertex_grad + ((size_t)b * 3 + 2) * nver;
        for (int j = 0; j < H; j++) {
            for (int i = 0; i < W; i++) {
                size_t q = ((size_t)b * H + j) * W + i;
                float g = depth_grad[q];
                int t = fr_f2i(tri_ind[q]);
                if (t < 0 || t >= ntri) continue;
                int p1 = fr_f2i(tri[t]), p2 = fr_f2i(tri[(size_t)ntri + t]), p3 = fr_f2i(tri[2 * (size_t)ntri + t]);
                if (p1 < 0 || p1 >= nver || p2 < 0 || p2 >= nver || p3 < 0 || p3 >= nver) continue;
                gz[p1] += g * 1.0f / 3.0f;
                gz[p2] += g * 1.0f / 3.0f;
                gz[p3] += g * 1.0f / 3.0f;
            }
        }
    }
    return 0;
}

/* ---- Z1: MM3D::ZBuffer, ModalAndRef.cpp:3-94 (the CPU baseline the bench times) -------------
 * All double, MATLAB column-major layouts: vertex[3p+k], tri[3i+k], texture[C p + j],
 * depth buffer imgh[x*H + y], img[j*W*H + x*H + y], tri_ind[x*H + y]; background = copy of src_img;
 * edge rule u+v <= 1; heap scratch per call. */
int fr_oracle_zbuffer(const double* vertex, const double* tri, const double* texture, int nver, int ntri,
                      const double* src_img, int W, int H, int C, double* img, double* tri_ind) {
    if (nver < 0 || ntri < 0 || W < 0 || H < 0 || C < 0) return -1;
    size_t nt = (size_t)(ntri > 0 ? ntri : 1);
    double* pt = (double*)malloc(nt * 6 * sizeof(double));
    double* hh = (double*)malloc(nt * sizeof(double));
    double* imgh = (double*)malloc(((size_t)W * H + 1) * sizeof(double));
    double* tritex = (double*)malloc(nt * (size_t)(C > 0 ? C : 1) * sizeof(double));
    unsigned char* ok = (unsigned char*)malloc(nt);
    if (!pt || !hh || !imgh || !tritex || !ok) {
        free(pt); free(hh); free(imgh); free(tritex); free(ok);
        return -2;
    }
    for (size_t q = 0; q < (size_t)W * H; q++) {
        imgh[q] = -99999999999999.0;
        tri_ind[q] = -1;
    }
    for (int i = 0; i < ntri; i++) {
        int p1 = fr_d2i(tri[3 * (size_t)i]), p2 = fr_d2i(tri[3 * (size_t)i + 1]), p3 = fr_d2i(tri[3 * (size_t)i + 2]);
        ok[i] = (p1 >= 0 && p1 < nver && p2 >= 0 && p2 < nver && p3 >= 0 && p3 < nver);
        if (!ok[i]) continue;
        pt[6 * (size_t)i + 0] = vertex[3 * (size_t)p1]; pt[6 * (size_t)i + 1] = vertex[3 * (size_t)p1 + 1];
        pt[6 * (size_t)i + 2] = vertex[3 * (size_t)p2]; pt[6 * (size_t)i + 3] = vertex[3 * (size_t)p2 + 1];
        pt[6 * (size_t)i + 4] = vertex[3 * (size_t)p3]; pt[6 * (size_t)i + 5] = vertex[3 * (size_t)p3 + 1];
        hh[i] = (vertex[3 * (size_t)p1 + 2] + vertex[3 * (size_t)p2 + 2] + vertex[3 * (size_t)p3 + 2]) / 3;
        for (int j = 0; j < C; j++)
            tritex[(size_t)C * i + j] =
                (texture[(size_t)C * p1 + j] + texture[(size_t)C * p2 + j] + texture[(size_t)C * p3 + j]) / 3;
    }
    for (size_t q = 0; q < (size_t)W * H * C; q++) img[q] = src_img[q];
    for (int i = 0; i < ntri; i++) {
        if (!ok[i]) continue;
        const double* p = pt + 6 * (size_t)i;
        int x_min = fr_d2i(ceil(FR_MIN(FR_MIN(p[0], p[2]), p[4])));
        int x_max = fr_d2i(floor(FR_MAX(FR_MAX(p[0], p[2]), p[4])));
        int y_min = fr_d2i(ceil(FR_MIN(FR_MIN(p[1], p[3]), p[5])));
        int y_max = fr_d2i(floor(FR_MAX(FR_MAX(p[1], p[3]), p[5])));
        if (x_max < x_min || y_max < y_min || x_max > W - 1 || x_min < 0 || y_max > H - 1 || y_min < 0) continue;
        for (int x = x_min; x <= x_max; x++) {
            for (int y = y_min; y <= y_max; y++) {
                size_t q = (size_t)x * H + y;
                if (imgh[q] < hh[i] &&
                    fr_oracle_point_in_tri_mex((double)x, (double)y, p[0], p[1], p[2], p[3], p[4], p[5])) {
                    imgh[q] = hh[i];
                    for (int j = 0; j < C; j++) img[(size_t)j * W * H + q] = tritex[(size_t)C * i + j];
                    tri_ind[q] = i;
                }
            }
        }
    }
    free(pt); free(hh); free(imgh); free(tritex); free(ok);
    return 0;
}

/* ---- D3: rotation_matrix, nets/network.py:266-291 ------------------------------------------
 * angles are fp32 values widened to double (math.cos/sin on Python floats), R = (R_pitch . R_yaw) . R_roll
 * as plain 3-term dot products in float64 (no FMA), rounded once to fp32. */
static void fr_mat3_mul(const double* A, const double* Bm, double* Cm) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Cm[3 * i + j] = (A[3 * i + 0] * Bm[0 + j] + A[3 * i + 1] * Bm[3 + j]) + A[3 * i + 2] * Bm[6 + j];
}
void fr_oracle_rotation_matrix(float phi_f, float gamma_f, float theta_f, float* R9) {
    double phi = phi_f, gamma = gamma_f, theta = theta_f;
    double cp = cos(phi), sp = sin(phi), cy = cos(gamma), sy = sin(gamma), ct = cos(theta), st = sin(theta);
    double Rp[9] = {1, 0, 0, 0, cp, sp, 0, -sp, cp};
    double Ry[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
    double Rr[9] = {ct, st, 0, -st, ct, 0, 0, 0, 1};
    double PY[9], Rm[9];
    fr_mat3_mul(Rp, Ry, PY);
    fr_mat3_mul(PY, Rr, Rm);
    for (int i = 0; i < 9; i++) R9[i] = (float)Rm[i];
}

/* ---- D2+D4: vertices_transform, nets/network.py:140-171 --------------------------------------
 * params [B, 7+ns+ne] = [phi,gamma,theta,tx,ty,tz,f | alpha(ns) | beta(ne)] (network.py:253-263),
 * mu [3N] blocked (element r is coordinate r/N of vertex r%N, network.py:157), pc_shape [3N,ns],
 * pc_exp [3N,ne] row-major.  R_override (may be NULL) is a host-computed [B,3,3] fp32 rotation.
 *
 * Written numerical spec (TF's matmul order is unknowable; this is OUR definition, identical on CPU and
 * on the gfx950 f32 MFMA path, which is bit-for-bit a k-ordered fmaf chain):
 *   S_r = fmaf-chain over k = 0..ns-1 of pc_shape[r,k]*alpha[k], starting from +0
 *   E_r = fmaf-chain over k = 0..ne-1 of pc_exp[r,k]*beta[k],   starting from +0
 *   v_r = (mu_r + S_r) + E_r                                   (network.py:159)
 *   M   = f * R  (fp32 elementwise, network.py:165)
 *   p_i = fmaf(M_i2, vz, fmaf(M_i1, vy, M_i0*vx)) + t_i        (matmul then + t3d_expand, network.py:165)
 *   y   = (im_size - y) - 1                                    (network.py:168)
 * Output vertex_proj [B,3,N]. */
/* The body is compiled twice: once for x86-64 baseline (fmaf -> libm, correct but slow) and once with the
 * FMA ISA enabled so fmaf() is a single vfmadd; both give identical bits (fmaf is exactly rounded either way).
 * Four vertices x three coordinates are carried as 12 independent chains purely for ILP; each chain is still
 * strictly k-ordered. */
static inline __attribute__((always_inline)) int fr_decode_body(const float* params, const float* mu,
                                                                const float* pc_shape, const float* pc_exp,
                                                                const float* R_override, int B, int N, int ns,
                                                                int ne, float im_size, float* vertex_proj) {
    if (B < 0 || N < 0 || ns < 0 || ne < 0) return -1;
    int nd = 7 + ns + ne;
    for (int b = 0; b < B; b++) {
        const float* pr = params + (size_t)b * nd;
        const float* alpha = pr + 7;
        const float* beta = pr + 7 + ns;
        float R[9], M[9];
        if (R_override)
            memcpy(R, R_override + 9 * (size_t)b, sizeof(R));
        else
            fr_oracle_rotation_matrix(pr[0], pr[1], pr[2], R);
        float f = pr[6];
        for (int i = 0; i < 9; i++) M[i] = f * R[i];
        float* ox = vertex_proj + ((size_t)b * 3 + 0) * N;
        float* oy = vertex_proj + ((size_t)b * 3 + 1) * N;
        float* oz = vertex_proj + ((size_t)b * 3 + 2) * N;
        for (int p0 = 0; p0 < N; p0 += 4) {
            int np = N - p0 < 4 ? N - p0 : 4;
            float S[12], E[12];
            const float* ws[12];
            const float* we[12];
            for (int j = 0; j < 12; j++) {
                int c = j / 4, q = j % 4;
                size_t r = (size_t)c * N + p0 + (q < np ? q : 0);
                ws[j] = pc_shape + r * ns;
                we[j] = pc_exp + r * ne;
                S[j] = 0.0f;
                E[j] = 0.0f;
            }
            for (int k = 0; k < ns; k++) {
                float a = alpha[k];
                for (int j = 0; j < 12; j++) S[j] = fmaf(ws[j][k], a, S[j]);
            }
            for (int k = 0; k < ne; k++) {
                float a = beta[k];
                for (int j = 0; j < 12; j++) E[j] = fmaf(we[j][k], a, E[j]);
            }
            for (int q = 0; q < np; q++) {
                int p = p0 + q;
                float vx = (mu[p] + S[q]) + E[q];
                float vy = (mu[(size_t)N + p] + S[4 + q]) + E[4 + q];
                float vz = (mu[2 * (size_t)N + p] + S[8 + q]) + E[8 + q];
                float px = fmaf(M[2], vz, fmaf(M[1], vy, M[0] * vx)) + pr[3];
                float py = fmaf(M[5], vz, fmaf(M[4], vy, M[3] * vx)) + pr[4];
                float pz = fmaf(M[8], vz, fmaf(M[7], vy, M[6] * vx)) + pr[5];
                ox[p] = px;
                oy[p] = (im_size - py) - 1.0f;
                oz[p] = pz;
            }
        }
    }
    return 0;
}
__attribute__((target("fma"))) static int fr_decode_fma(const float* params, const float* mu, const float* pc_shape,
                                                        const float* pc_exp, const float* R_override, int B, int N,
                                                        int ns, int ne, float im_size, float* vertex_proj) {
    return fr_decode_body(params, mu, pc_shape, pc_exp, R_override, B, N, ns, ne, im_size, vertex_proj);
}
static int fr_decode_generic(const float* params, const float* mu, const float* pc_shape, const float* pc_exp,
                             const float* R_override, int B, int N, int ns, int ne, float im_size,
                             float* vertex_proj) {
    return fr_decode_body(params, mu, pc_shape, pc_exp, R_override, B, N, ns, ne, im_size, vertex_proj);
}
int fr_oracle_decode_3dmm(const float* params, const float* mu, const float* pc_shape, const float* pc_exp,
                          const float* R_override, int B, int N, int ns, int ne, float im_size,
                          float* vertex_proj) {
    if (__builtin_cpu_supports("fma"))
        return fr_decode_fma(params, mu, pc_shape, pc_exp, R_override, B, N, ns, ne, im_size, vertex_proj);
    return fr_decode_generic(params, mu, pc_shape, pc_exp, R_override, B, N, ns, ne, im_size, vertex_proj);
}
/* force the slow path (used by a test to show both bodies give the same bits) */
int fr_oracle_decode_3dmm_nofma(const float* params, const float* mu, const float* pc_shape, const float* pc_exp,
                                const float* R_override, int B, int N, int ns, int ne, float im_size,
                                float* vertex_proj) {
    return fr_decode_generic(params, mu, pc_shape, pc_exp, R_override, B, N, ns, ne, im_size, vertex_proj);
}

/* fp64 evaluation of the same formula (the "truth" the fp32 spec is compared to with a tolerance). */
int fr_oracle_decode_3dmm_f64(const float* params, const float* mu, const float* pc_shape, const float* pc_exp,
                              int B, int N, int ns, int ne, double im_size, double* vertex_proj) {
    if (B < 0 || N < 0 || ns < 0 || ne < 0) return -1;
    int nd = 7 + ns + ne;
    for (int b = 0; b < B; b++) {
        const float* pr = params + (size_t)b * nd;
        double phi = pr[0], gamma = pr[1], theta = pr[2];
        double cp = cos(phi), sp = sin(phi), cy = cos(gamma), sy = sin(gamma), ct = cos(theta), st = sin(theta);
        double Rp[9] = {1, 0, 0, 0, cp, sp, 0, -sp, cp};
        double Ry[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
        double Rr[9] = {ct, st, 0, -st, ct, 0, 0, 0, 1};
        double PY[9], R[9];
        fr_mat3_mul(Rp, Ry, PY);
        fr_mat3_mul(PY, Rr, R);
        double f = pr[6];
        for (int p = 0; p < N; p++) {
            double v[3];
            for (int c = 0; c < 3; c++) {
                size_t r = (size_t)c * N + p;
                double S = 0, E = 0;
                for (int k = 0; k < ns; k++) S += (double)pc_shape[r * ns + k] * (double)pr[7 + k];
                for (int k = 0; k < ne; k++) E += (double)pc_exp[r * ne + k] * (double)pr[7 + ns + k];
                v[c] = (double)mu[r] + S + E;
            }
            double q[3];
            for (int i = 0; i < 3; i++)
                q[i] = f * R[3 * i] * v[0] + f * R[3 * i + 1] * v[1] + f * R[3 * i + 2] * v[2] + (double)pr[3 + i];
            vertex_proj[((size_t)b * 3 + 0) * N + p] = q[0];
            vertex_proj[((size_t)b * 3 + 1) * N + p] = im_size - q[1] - 1.0;
            vertex_proj[((size_t)b * 3 + 2) * N + p] = q[2];
        }
    }
    return 0;
}

/* ---- decode backward: gradient of the 235-d parameters -------------------------------------
 * What TF autodiff produces for nets/network.py:140-171 given g = dL/d vertex_proj [B,3,N]:
 *   dq = (g_x, -g_y, g_z)                         (the y row is (im_size - q_1) - 1, network.py:168)
 *   dt3d_i = sum_p dq_i                           (t3d_expand, :164-165)
 *   df     = sum_p sum_i (R v)_i dq_i             (f_expand * R, :163-165)
 *   dv_c   = sum_i (f R)_ic dq_i;  dalpha_k = sum_r pc_shape[r,k] dv_r;  dbeta_k = sum_r pc_exp[r,k] dv_r   (:153-159)
 *   the three angles get NO gradient: R comes out of tf.py_func (:150), which has no registered gradient.
 * Evaluated in float64 (the tolerance reference for the fp32 HIP kernels).  grad_params [B, 7+ns+ne]. */
int fr_oracle_decode_3dmm_backward_f64(const float* grad_vertex_proj, const float* params, const float* mu,
                                       const float* pc_shape, const float* pc_exp, int B, int N, int ns, int ne,
                                       double* grad_params) {
    if (B < 0 || N < 0 || ns < 0 || ne < 0) return -1;
    int nd = 7 + ns + ne;
    double* dv = (double*)malloc(sizeof(double) * 3 * (size_t)(N > 0 ? N : 1));
    if (!dv) return -2;
    for (int b = 0; b < B; b++) {
        const float* pr = params + (size_t)b * nd;
        double* gp = grad_params + (size_t)b * nd;
        for (int i = 0; i < nd; i++) gp[i] = 0.0;
        float Rf[9];
        fr_oracle_rotation_matrix(pr[0], pr[1], pr[2], Rf);
        double f = pr[6];
        const float* gx = grad_vertex_proj + ((size_t)b * 3 + 0) * N;
        const float* gy = grad_vertex_proj + ((size_t)b * 3 + 1) * N;
        const float* gz = grad_vertex_proj + ((size_t)b * 3 + 2) * N;
        for (int p = 0; p < N; p++) {
            double dq[3] = {(double)gx[p], -(double)gy[p], (double)gz[p]};
            double v[3];
            for (int c = 0; c < 3; c++) {
                size_t r = (size_t)c * N + p;
                double S = 0, E = 0;
                for (int k = 0; k < ns; k++) S += (double)pc_shape[r * ns + k] * (double)pr[7 + k];
                for (int k = 0; k < ne; k++) E += (double)pc_exp[r * ne + k] * (double)pr[7 + ns + k];
                v[c] = (double)mu[r] + S + E;
            }
            for (int i = 0; i < 3; i++) {
                gp[3 + i] += dq[i];
                gp[6] += ((double)Rf[3 * i] * v[0] + (double)Rf[3 * i + 1] * v[1] + (double)Rf[3 * i + 2] * v[2]) * dq[i];
            }
            for (int c = 0; c < 3; c++)
                dv[(size_t)c * N + p] = f * ((double)Rf[c] * dq[0] + (double)Rf[3 + c] * dq[1] + (double)Rf[6 + c] * dq[2]);
        }
        for (size_t r = 0; r < (size_t)3 * N; r++) {
            double d = dv[r];
            if (d == 0.0) continue;
            for (int k = 0; k < ns; k++) gp[7 + k] += (double)pc_shape[r * ns + k] * d;
            for (int k = 0; k < ne; k++) gp[7 + ns + k] += (double)pc_exp[r * ne + k] * d;
        }
    }
    free(dv);
    return 0;
}

/* ---- Q30 decode: the exact specification of the product's fixed-point basis blend ---------------------------------
 * NOT the reference's arithmetic (the reference runs two fp32 tf.matmuls whose summation order is unknowable,
 * nets/network.py:153-156): this is the written spec of the gfx950 int8-MFMA decode, restated here so that the HIP
 * kernel can be held to it bit for bit, and so that its distance from the fp32 spec above and from the float64
 * evaluation can be measured on the CPU (tests/test_decode_q30_cpu.py).  Everything is integer or exactly specified
 * float64 arithmetic, hence order-independent:
 *   A = [pc_shape | pc_exp]  (3N x K),  x = [alpha | beta]  (K)
 *   ce_k  = frexp exponent of max_r |A[r,k]|           (0 for an all-zero column)      -> A'[r,k] = A[r,k] 2^-ce_k, |A'| < 1
 *   re_r  = frexp exponent of max_k |A'[r,k]|          (0 for an all-zero row; <= 0)
 *   qA    = rint(A'[r,k] 2^(30-re_r))                  (|qA| <= 2^30; entries within 2^-6 of the row maximum are exact)
 *   be    = max_k (frexp exponent of x_k) + ce_k       (0 if x == 0)                   -> x'_k = x_k 2^ce_k
 *   qB    = rint(x'_k 2^(30-be))
 *   I     = sum_k qA qB   evaluated as level sums L_s = sum_k sum_{i+j=s} a_i b_j over the balanced base-256 digits of
 *           qA, qB (most significant first; digit product a_i b_j has weight 256^(6-i-j)), s = 0 .. LV-1, combined as
 *           h = fl64(h*256 + L_s) -- exact whenever the integer stays below 2^53, correctly rounded steps otherwise.
 *           LV = 7 keeps all sixteen digit products (the exact product of the two 31-bit operands); LV = 5 keeps the
 *           thirteen products with i + j <= 4 (what is dropped is below 2^-38 of a term's full scale: the fp32 result
 *           is unchanged in > 99.98 % of the cases); LV = 4 the ten with i + j <= 3 (dropped: below 2^-30 of full scale)
 *   v_r   = fl32( fl64( mu_r + h 2^(re_r+be-60+8(7-LV)) ) )    -- ONE rounding of mu + S + E instead of the chain's ~K
 *   then the pose product and y flip exactly as fr_decode_body.
 * A non-finite parameter makes the face's vertices NaN, a non-finite basis entry its row's. */
static int fr_q30_exp(double x) { int e; frexp(x, &e); return e; }
static void fr_q30_digits(int32_t q, int d[4]) {
    for (int t = 3; t > 0; t--) {
        int l = ((q + 128) & 255) - 128;
        d[t] = l;
        q = (q - l) >> 8;
    }
    d[0] = q;
}
int fr_oracle_decode_3dmm_q30_lv(const float* params, const float* mu, const float* pc_shape, const float* pc_exp,
                                 const float* R_override, int B, int N, int ns, int ne, float im_size, int levels,
                                 float* vertex_proj) {
    if (B < 0 || N < 0 || ns < 0 || ne < 0 || levels < 1 || levels > 7) return -1;
    const int LV = levels;
    const int K = ns + ne, nd = 7 + K;
    const size_t rows = (size_t)3 * N;
    int* ce = (int*)calloc(K > 0 ? K : 1, sizeof(int));
    int32_t* qB = (int32_t*)calloc((size_t)(B > 0 ? B : 1) * (K > 0 ? K : 1), sizeof(int32_t));
    int32_t* qA = (int32_t*)calloc(K > 0 ? K : 1, sizeof(int32_t));
    int* be = (int*)calloc(B > 0 ? B : 1, sizeof(int));
    char* badb = (char*)calloc(B > 0 ? B : 1, 1);
    float* v = (float*)malloc(sizeof(float) * 3 * (size_t)(B > 0 ? B : 1));
    if (!ce || !qB || !qA || !be || !badb || !v) return -2;
#define FR_A(r, k) ((k) < ns ? pc_shape[(r) * ns + (k)] : pc_exp[(r) * ne + ((k) - ns)])
    for (int k = 0; k < K; k++) {
        float m = 0.0f;
        for (size_t r = 0; r < rows; r++) {
            float a = fabsf(FR_A(r, k));
            if (isfinite(a) && a > m) m = a;
        }
        ce[k] = m > 0.0f ? fr_q30_exp((double)m) : 0;
    }
    for (int b = 0; b < B; b++) {
        const float* x = params + (size_t)b * nd + 7;
        int e = INT_MIN;
        for (int k = 0; k < K; k++) {
            if (!isfinite(x[k])) badb[b] = 1;
            else if (x[k] != 0.0f) { int ek = fr_q30_exp((double)x[k]) + ce[k]; if (ek > e) e = ek; }
        }
        be[b] = e == INT_MIN ? 0 : e;
        for (int k = 0; k < K; k++)
            qB[(size_t)b * K + k] = isfinite(x[k]) ? (int32_t)rint(ldexp((double)x[k], ce[k] + 30 - be[b])) : 0;
    }
    float* Ms = (float*)malloc(sizeof(float) * 9 * (size_t)(B > 0 ? B : 1));
    if (!Ms) return -2;
    for (int b = 0; b < B; b++) {
        const float* pr = params + (size_t)b * nd;
        float R[9];
        if (R_override) memcpy(R, R_override + 9 * (size_t)b, sizeof(R));
        else fr_oracle_rotation_matrix(pr[0], pr[1], pr[2], R);
        for (int i = 0; i < 9; i++) Ms[9 * (size_t)b + i] = pr[6] * R[i];
    }
    for (int p = 0; p < N; p++) {
        for (int c = 0; c < 3; c++) {
            const size_t r = (size_t)c * N + p;
            int re = INT_MIN, badr = 0;
            for (int k = 0; k < K; k++) {
                float a = FR_A(r, k);
                if (!isfinite(a)) badr = 1;
                else if (a != 0.0f) { int ek = fr_q30_exp((double)a) - ce[k]; if (ek > re) re = ek; }
            }
            if (re == INT_MIN) re = 0;
            for (int k = 0; k < K; k++) {
                float a = FR_A(r, k);
                qA[k] = isfinite(a) ? (int32_t)rint(ldexp((double)a, 30 - re - ce[k])) : 0;
            }
            for (int b = 0; b < B; b++) {
                const int32_t* qb = qB + (size_t)b * K;
                __int128 I = 0;
                if (LV == 7) {
                    for (int k = 0; k < K; k++) I += (__int128)((int64_t)qA[k] * (int64_t)qb[k]);
                } else {
                    /* kept products: a_i 256^(3-i) times the top LV-i digits of qB (T[i]; all four when LV-i >= 4) */
                    for (int k = 0; k < K; k++) {
                        int da[4], db[4];
                        fr_q30_digits(qA[k], da);
                        fr_q30_digits(qb[k], db);
                        for (int i = 0; i < 4 && i < LV; i++) {
                            int64_t T = 0;
                            for (int j = 0; j < 4 && i + j < LV; j++) T += (int64_t)db[j] << (8 * (3 - j));
                            I += (__int128)(((int64_t)da[i] << (8 * (3 - i))) * T);
                        }
                    }
                    /* every kept product is a multiple of 256^(7-LV) */
                    I /= ((__int128)1 << (8 * (7 - LV)));
                }
                double h;
                const __int128 lim = (__int128)1 << 52;
                if (I < lim && I > -lim) {
                    h = (double)(int64_t)I;   /* every partial of the level chain is exact: h == I */
                } else {                       /* the chain's own roundings, level by level */
                    int64_t L[7] = {0, 0, 0, 0, 0, 0, 0};
                    for (int k = 0; k < K; k++) {
                        int da[4], db[4];
                        fr_q30_digits(qA[k], da);
                        fr_q30_digits(qb[k], db);
                        for (int i = 0; i < 4; i++)
                            for (int j = 0; j < 4; j++) L[i + j] += (int64_t)da[i] * db[j];
                    }
                    h = (double)L[0];
                    for (int s = 1; s < LV; s++) h = h * 256.0 + (double)L[s];
                }
                double d = (double)mu[r] + ldexp(h, re + be[b] - 60 + 8 * (7 - LV));
                v[(size_t)b * 3 + c] = (badr || badb[b]) ? NAN : (float)d;
            }
        }
        for (int b = 0; b < B; b++) {
            const float* pr = params + (size_t)b * nd;
            const float* M = Ms + 9 * (size_t)b;
            const float vx = v[(size_t)b * 3], vy = v[(size_t)b * 3 + 1], vz = v[(size_t)b * 3 + 2];
            float px = fmaf(M[2], vz, fmaf(M[1], vy, M[0] * vx)) + pr[3];
            float py = fmaf(M[5], vz, fmaf(M[4], vy, M[3] * vx)) + pr[4];
            float pz = fmaf(M[8], vz, fmaf(M[7], vy, M[6] * vx)) + pr[5];
            vertex_proj[((size_t)b * 3 + 0) * N + p] = px;
            vertex_proj[((size_t)b * 3 + 1) * N + p] = (im_size - py) - 1.0f;
            vertex_proj[((size_t)b * 3 + 2) * N + p] = pz;
        }
    }
#undef FR_A
    free(ce); free(qB); free(qA); free(be); free(badb); free(v); free(Ms);
    return 0;
}

int fr_oracle_decode_3dmm_q30(const float* params, const float* mu, const float* pc_shape, const float* pc_exp,
                              const float* R_override, int B, int N, int ns, int ne, float im_size,
                              float* vertex_proj) {
    return fr_oracle_decode_3dmm_q30_lv(params, mu, pc_shape, pc_exp, R_override, B, N, ns, ne, im_size, 7, vertex_proj);
}
