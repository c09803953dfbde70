// 3DMM decode backward for gfx950 (MI355X): dL/d vertex_proj [B,3,N] -> dL/d params [B,235].
//
// What it computes: the gradient TF autodiff derives from FaceRecNet.vertices_transform (nets/network.py:140-171):
//   dq      = (g_x, -g_y, g_z)                                  y row is (im_size - q_1) - 1 (:168)
//   d t3d_i = sum_p dq_i                                        (:164-165)
//   d f     = sum_p (R v) . dq  = sum_p (q - t3d) . dq / f      (f_expand * R, :163-165)
//   dv      = (f R)^T dq ;  d alpha = pc_shape^T dv ;  d beta = pc_exp^T dv          (:153-159)
//   d angles = 0: R comes out of tf.py_func (:150), which has no gradient in the reference.
//
// How (reference-layout entry point; the packed entry point runs bwd_fused_kernel + bwd_reduce_kernel: see there): three
// launches, no float atomics (bit-reproducible):
//   bwd_prepass_kernel  elementwise: dq -> dv, written transposed into MFMA B-fragment order dvT4[row][16][4]
//                       (LDS tile transpose so both the read of g and the write of dvT4 are coalesced), plus
//                       per-workgroup partial sums for d t3d and d f;
//   bwd_gemm_kernel     the [228 x 3N].[3N x 64] reduction on the matrix cores (exact-f32 v_mfma_f32_16x16x4_f32):
//                       split over row chunks, one workgroup per chunk, wave w owns 64 coefficient slots (lane l of a
//                       k-step loads FOUR consecutive coefficients of row l>>4 with one 16-byte request -- 256
//                       contiguous bytes of the reference-layout row per 16 lanes -- and feeds them to four MFMAs,
//                       element i to MFMA i, so MFMA i's 16 output rows are coefficients 4 m + i), dv fragments are
//                       one dwordx4 per k-step shared through L1 by the waves; partial [slots x 64] slabs go to the
//                       workspace;
//   bwd_reduce_kernel   sums the slabs / partials in a fixed order and writes grad_params.
#include "fr_common.h"

namespace fr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BW_PV = 64;        // vertices per prepass workgroup
constexpr int BW_MAXWAVES = 8;   // gemm waves per workgroup: one per 64 coefficient slots (<= 512 slots)
// coefficient slots: shape columns padded to a multiple of 4, then expression columns padded to a multiple of 4
__host__ __device__ inline int bw_ns4(int ns) { return (ns + 3) & ~3; }
__host__ __device__ inline int bw_waves(int ns, int ne) { return (bw_ns4(ns) + ((ne + 3) & ~3) + 63) / 64; }

struct BwdArgs {
    const float* g;          // [B,3,N]
    const float* params;     // [B,nd]
    const float* vproj;      // [B,3,N] forward output
    const float* pc_shape;   // [3N,ns]
    const float* pc_exp;     // [3N,ne]
    const float* R_override; // [B,9] or null
    float* grad_params;      // [B,nd]
    float4* dvT4;            // [3N][16] float4: dv[row][batch = 16*mb + j] at [row][j].mb
    float* pose_part;        // [prepass blocks][64][4]  (dt_x, dt_y, dt_z, sum (q-t).dq)
    float* slab;             // [gemm blocks][64 * waves slots][64]
    int B, N, ns, ne, b0, nbatch;
    int pre_blocks, gemm_blocks, rows_per_block;
    float im_size;
    // K-major packed image (fr_decode_backward_pack_basis) and its geometry; null: the reference-layout kernel runs
    const float4* At;        // packed image [vertex groups of 16][3 coordinates][slot blocks][64 lanes] float4 (bwd_pack_kernel)
    int sbt, sbs;            // slot blocks of 16 coefficients in all / of the shape basis (expression blocks follow)
    int rbt, rb_per_block;   // reference-layout path: 16-row blocks in all / per gemm workgroup.  Packed path (bwd_fused_kernel):
                             // vertex groups of 16 in all / per workgroup
    int exp_slot0;           // first coefficient slot of the expression basis in the slabs
    int nslots;              // coefficient slots per partial slab (64 per wave of the GEMM workgroup)
    const float* mu;         // [3N] mean shape, or null.  Non-null (fr_decode_3dmm_backward_packed_mu): d f is formed WITHOUT the
                             // forward output -- sum_p (R v_p) . dq = sum_p v_p . dv_p / f with v = mu + S alpha + E beta, i.e.
                             // d f = (sum_p mu_p . dv_p + alpha . d alpha + beta . d beta) / f: the fused kernel reads 0.64 MB of mu
                             // (L2-resident) instead of 41 MB of vertex_proj -- which buys MEMORY (the forward output need not be
                             // kept for the backward), not time: 71.3 vs 71.1 us at 64 faces, 56.3 vs 53.2 at 32; each workgroup adds the parameters' dot product with
                             // ITS partial coefficient gradients to its pose partial (linear: the partials add up to the whole)
};

__device__ __forceinline__ void bwd_rotation(const BwdArgs& a, int b, float* R9) {
    const int nd = FR_N_POSE + a.ns + a.ne;
    const float* pr = a.params + (size_t)(a.b0 + b) * nd;
    if (a.R_override) {
#pragma unroll
        for (int i = 0; i < 9; i++) R9[i] = a.R_override[(size_t)(a.b0 + b) * 9 + i];
        return;
    }
    double sp, cp, sy, cy, st, ct;
    sincos((double)pr[0], &sp, &cp);
    sincos((double)pr[1], &sy, &cy);
    sincos((double)pr[2], &st, &ct);
    // (R_pitch . R_yaw) . R_roll in float64, 3-term dots without FMA, one rounding (network.py:276-290)
    const double Rp[9] = {1, 0, 0, 0, cp, sp, 0, -sp, cp};
    const double Ry[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
    const double Rr[9] = {ct, st, 0, -st, ct, 0, 0, 0, 1};
    double PY[9], Rm[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) PY[3 * i + j] = (Rp[3 * i] * Ry[j] + Rp[3 * i + 1] * Ry[3 + j]) + Rp[3 * i + 2] * Ry[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rm[3 * i + j] = (PY[3 * i] * Rr[j] + PY[3 * i + 1] * Rr[3 + j]) + PY[3 * i + 2] * Rr[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) R9[i] = (float)Rm[i];
}

// Sum over the 64 lanes of a wave in a FIXED order (pairs, quads, half rows, rows, then the row totals passed on with
// row_bcast:15 / row_bcast:31): six v_add_f32_dpp, result valid in lane 63.  All 64 lanes must be active.
#define FR_DPP_ADD(x, ctrl, rmask, bc) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, bc))
__device__ __forceinline__ float wave_sum_to_lane63(float x) {
    FR_DPP_ADD(x, 0x111, 0xf, true);    // row_shr:1
    FR_DPP_ADD(x, 0x112, 0xf, true);    // row_shr:2
    FR_DPP_ADD(x, 0x114, 0xf, true);    // row_shr:4
    FR_DPP_ADD(x, 0x118, 0xf, true);    // row_shr:8  -> lane 15 of every row holds the row's sum
    FR_DPP_ADD(x, 0x142, 0xa, false);   // row_bcast:15 into rows 1 and 3
    FR_DPP_ADD(x, 0x143, 0xc, false);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return x;
}
#undef FR_DPP_ADD

// ---- prepass: 64 vertices x 64 batch columns per workgroup -------------------------------------------------------------
__global__ __launch_bounds__(256) void bwd_prepass_kernel(BwdArgs a) {
    __shared__ float Mt[64][13];                 // f*R (9), t (3), 1/f or 0
    __shared__ float tile[3][BW_PV][64 + 1];      // dv[c][p][bslot]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nd = FR_N_POSE + a.ns + a.ne;
    const int N = a.N;
    if (tid < 64) {
        float m[13];
#pragma unroll
        for (int i = 0; i < 13; i++) m[i] = 0.f;
        if (tid < a.nbatch) {
            const float* pr = a.params + (size_t)(a.b0 + tid) * nd;
            float R[9];
            bwd_rotation(a, tid, R);
            const float f = pr[6];
#pragma unroll
            for (int i = 0; i < 9; i++) m[i] = f * R[i];
            m[9] = pr[3]; m[10] = pr[4]; m[11] = pr[5];
            m[12] = (f != 0.0f) ? 1.0f / f : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 13; i++) Mt[tid][i] = m[i];
    }
    __syncthreads();
    const int p = blockIdx.x * BW_PV + lane;
    const bool pok = p < N;
    // wave wv handles batch columns 16*wv .. 16*wv+15, lane = vertex: reads of g / vproj are 256-byte coalesced.  Four
    // columns per trip: their 24 loads are requested (unconditionally, clamped) before anything is computed
    const int pc = pok ? p : 0;
    for (int bq0 = 0; bq0 < 16; bq0 += 4) {
        float gq[4][3], vq[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < 3; c++) gq[u][c] = vq[u][c] = 0.f;
        if (16 * wv + bq0 < a.nbatch)   // (wave-uniform: a trip whose four columns are all dead requests nothing)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = min(16 * wv + bq0 + u, a.nbatch - 1);
            const float* gb = a.g + (size_t)(a.b0 + b) * 3 * N;
            const float* vb = a.vproj + (size_t)(a.b0 + b) * 3 * N;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                gq[u][c] = gb[(size_t)c * N + pc];
                vq[u][c] = vb[(size_t)c * N + pc];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = 16 * wv + bq0 + u;
            float dq0 = 0.f, dq1 = 0.f, dq2 = 0.f, fs = 0.f;
            if (pok && b < a.nbatch) {
                dq0 = gq[u][0];
                dq1 = -gq[u][1];
                dq2 = gq[u][2];
                // (q - t): q_0 = out_x, q_1 = (im - 1) - out_y, q_2 = out_z
                const float q0 = vq[u][0] - Mt[b][9];
                const float q1 = ((a.im_size - 1.0f) - vq[u][1]) - Mt[b][10];
                const float q2 = vq[u][2] - Mt[b][11];
                fs = __builtin_fmaf(q2, dq2, __builtin_fmaf(q1, dq1, q0 * dq0));
            }
            const float* m = Mt[b];
            const int bslot = (b & 15) * 4 + (b >> 4);
#pragma unroll
            for (int c = 0; c < 3; c++)
                tile[c][lane][bslot] = __builtin_fmaf(m[6 + c], dq2, __builtin_fmaf(m[3 + c], dq1, m[c] * dq0));
            // fixed-order wave reduction over the 64 vertices (six DPP adds per value, the total lands in lane 63; the
            // __shfl_xor butterfly was 24 dependent ds_bpermute trips per column)
            const float r0 = wave_sum_to_lane63(dq0), r1 = wave_sum_to_lane63(dq1), r2 = wave_sum_to_lane63(dq2),
                        r3 = wave_sum_to_lane63(fs);
            if (lane == 63) {
                float* pp = a.pose_part + ((size_t)blockIdx.x * 64 + b) * 4;
                pp[0] = r0; pp[1] = r1; pp[2] = r2; pp[3] = r3 * m[12];
            }
        }
    }
    __syncthreads();
    // flush: rows r = c*N + p, 64 floats (= 16 float4) per row, coalesced
    for (int i = tid; i < 3 * BW_PV * 16; i += 256) {
        const int j4 = i & 15, pl = (i >> 4) & (BW_PV - 1), c = i >> 10;
        const int pp = blockIdx.x * BW_PV + pl;
        if (pp < N) {
            const float* t = &tile[c][pl][j4 * 4];
            a.dvT4[((size_t)c * N + pp) * 16 + j4] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
    // the pad rows [3N, 16 * row blocks) of the last 16-row block (read as part of a 1 KiB fragment by the packed kernel): zeros
    if (blockIdx.x == 0) {
        const long long pad0 = 3ll * N * 16, pad1 = (long long)a.rbt * 256;
        for (long long i = pad0 + tid; i < pad1; i += 256) a.dvT4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- the reduction over vertices on the matrix cores -------------------------------------------------------------------
// D^T[coeff][batch] += sum_k basis[row k][coeff] * dv[row k][batch]:  A operand = basis^T (lane l: row r0+(l>>4),
// coefficient slot 4*(l&15)+i for MFMA i), B operand = dv (lane l: row r0+(l>>4), batch 16*mb+(l&15)).
// NB: live 16-column blocks of this pass (1..4): dead blocks cost neither MFMAs nor slab stores -- at the 32 faces per
// GPU of the reference's train loop that is half of the matrix work.  The row loop takes four k-steps per trip with all
// their operands requested before the first MFMA issues; every accumulation chain stays in row order.
typedef float f32x4u4 __attribute__((ext_vector_type(4), aligned(4)));
template <int NB>
__global__ __launch_bounds__(BW_MAXWAVES * 64) void bwd_gemm_kernel(BwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long rows = 3ll * a.N;
    const long long r_begin = (long long)blockIdx.x * a.rows_per_block;
    const long long r_end = min(rows, r_begin + a.rows_per_block);
    const int kq = lane >> 4, jn = lane & 15;
    // this lane's four coefficient slots: source array, row stride, first column, live columns (0..4)
    const int ns4 = bw_ns4(a.ns);
    const int slot0 = 64 * wave + 4 * jn;
    const bool in_shape = slot0 < ns4;
    const float* src = in_shape ? a.pc_shape : a.pc_exp;
    const int stride = in_shape ? a.ns : a.ne;
    const int col0 = in_shape ? slot0 : slot0 - ns4;
    const int live = max(0, min(4, stride - col0));
    f32x4 acc[4][NB];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int mb = 0; mb < NB; mb++) acc[i][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto bsel = [](const float4& v, int mb) { return mb == 0 ? v.x : mb == 1 ? v.y : mb == 2 ? v.z : v.w; };
    constexpr int UN = 4;   // k-steps (of 4 rows) per trip
    auto request = [&](long long r, float4 (&dv)[UN], f32x4 (&av)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long rr = r + 4 * u + kq;
            const bool rok = rr < r_end;
            dv[u] = rok ? a.dvT4[(size_t)rr * 16 + jn] : make_float4(0.f, 0.f, 0.f, 0.f);
            av[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (rok) {
                const float* ptr = src + (size_t)rr * stride + col0;
                if (live == 4) {
                    av[u] = *reinterpret_cast<const f32x4u4*>(ptr);   // (rows are only 4-byte aligned: 199 columns)
                } else {   // the array's last, partial quadruple: never read past the row
                    if (live > 0) av[u][0] = ptr[0];
                    if (live > 1) av[u][1] = ptr[1];
                    if (live > 2) av[u][2] = ptr[2];
                }
            }
        }
    };
    auto consume = [&](const float4 (&dv)[UN], const f32x4 (&av)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int mb = 0; mb < NB; mb++)
                    acc[i][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bsel(dv[u], mb), acc[i][mb], 0, 0, 0);
    };
    // (a second operand set, requested one trip ahead, measured no faster: 44.3 / 71.4 us against 45.1 / 68.0 us at
    // 32 / 64 faces, at 176 instead of 110 VGPRs)
    for (long long r = r_begin; r < r_end; r += 4 * UN) {
        float4 dv[UN];
        f32x4 av[UN];
        request(r, dv, av);
        consume(dv, av);
    }
    // D^T tile of MFMA i: row m = 4*(lane>>4) + reg is coefficient slot 64*wave + 4*m + i, column (batch within block) = lane & 15
    float* slab = a.slab + (size_t)blockIdx.x * a.nslots * 64;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int mb = 0; mb < NB; mb++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int slot = 64 * wave + 4 * (4 * kq + rg) + i;
                slab[(size_t)slot * 64 + 16 * mb + jn] = acc[i][mb][rg];
            }
}


// ---- the same reduction from a K-major packed image (round 3) ---------------------------------------------------------
// The kernel above reads the basis in its reference layout: 16-byte pieces at a 796-byte row stride, requested a trip at a
// time and waited for with the compiler's vmcnt(0) -- the two waves that share a SIMD run in lock-step, so every trip
// exposes a full memory round trip (47 / 70 us at 32 / 64 faces against a 33 us MFMA floor).  With the basis packed ONCE
// into MFMA A-fragment order (fr_decode_backward_pack_basis: [16-row block][16-coefficient block][lane] float4, element
// j = basis[16 rb + 4 j + (lane >> 4)][16 sb + (lane & 15)], zero padded) every operand fetch is a coalesced, aligned 1 KiB
// fragment -- the dv rows of a 16-row block are four such fragments as well -- and the fragments run through a ring of
// BR row blocks of registers filled by inline-asm loads that the compiler can neither reorder nor drain; the only waits are
// counted.  Accumulation order per output: rows ascending within the workgroup's row range (as above); the partial
// slabs are summed by bwd_reduce_kernel in its fixed order.
__global__ __launch_bounds__(256) void bwd_pack_kernel(const float* __restrict__ pc_shape, const float* __restrict__ pc_exp,
                                                       int N, int ns, int ne, int sbs, int sbt, long long ngroups,
                                                       float4* __restrict__ At) {
    // image [vertex group of 16][coordinate][16-coefficient block][lane] float4, element j = basis[c N + 16 grp + 4 j + (lane >> 4)]
    // [16 sb + (lane & 15)], zero padded: a vertex group's three 16-row blocks (x, y, z rows of the SAME sixteen vertices) sit
    // side by side, so that the workgroup that multiplies them can form their dv rows from ONE tile of the incoming gradient
    const long long total = ngroups * 3 * sbt * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long long q = i >> 6;
        const int sb = (int)(q % sbt);
        q /= sbt;
        const int c = (int)(q % 3);
        const long long grp = q / 3;
        const int slot = 16 * sb + (lane & 15);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long long pv = 16 * grp + 4 * j + (lane >> 4);
            float x = 0.f;
            if (pv < N) {
                const size_t row = (size_t)c * N + pv;
                if (sb < sbs) { if (slot < ns) x = pc_shape[row * ns + slot]; }
                else { const int cc = slot - 16 * sbs; if (cc < ne) x = pc_exp[row * ne + cc]; }
            }
            v[j] = x;
        }
        At[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

#define FRB_LD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
// ---- the fused backward: gradient tile -> dv rows (LDS) -> MFMA reduction, one launch ------------------------------------------
// Round 3 ran a prepass (g, vertex_proj -> dv rows in B-fragment order + the pose partial sums: 123 MB moved, 27.6 us) and then
// the reduction (153 MB of basis + the 41 MB of dv rows again).  Here ONE workgroup per CU walks its vertex groups of 16; per
// group its 256 staging threads -- thread = (batch column b, vertex quad q) -- load the 16 x 64 tile of g and of vertex_proj
// (six 16-byte loads per thread, inline asm, in the same counted-wait stream as the basis fragments), form the group's
// three dv row blocks  dv = (f R)^T dq  and the d t3d / d f partial sums in registers, and park the dv rows in LDS in MFMA
// B-fragment order (double buffered; ONE LDS-only barrier per group); the waves then multiply the group's three basis row
// blocks (twelve 1 KiB fragments per wave, two groups in flight) against them.  dv never exists in global memory: g and
// vertex_proj are read once, the 41 MB write + re-read and the prepass launch are gone.
// Summation order: per output, groups ascending, x / y / z row block, k-step -- fixed by the chunking (FR_BWD_CHUNKS) alone.
// CB: 16-coefficient blocks per wave.  CB = 2 (eight waves for the model's 15 blocks: two per SIMD, one multiplying while the
// other waits for its fragments) or 4 (four waves, one per SIMD; also what bases of more than 16 blocks take).
// Either way the packed path covers bases of at most 16 blocks (256 coefficients: the model has 228); larger ones take the
// reference-layout entry point (fr_decode_backward_basis_bytes answers 0 for them).
template <int NB, int CB>
__global__ __launch_bounds__(CB == 2 ? 512 : 256) void bwd_fused_kernel(BwdArgs a) {
    __shared__ float Mt[64][13];                                            // f*R (9), t (3), 1/f or 0
    __shared__ __attribute__((aligned(16))) float4 dvL[2][3][4][64];        // [buffer][coordinate][k-step][lane] (mb in .xyzw)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nd = FR_N_POSE + a.ns + a.ne;
    const int N = a.N;
    const bool muf = a.mu != nullptr;   // (uniform) d f from mu . dv: the tile's second operand is mu, not vertex_proj
    if (tid < 64) {
        float m[13];
#pragma unroll
        for (int i = 0; i < 13; i++) m[i] = 0.f;
        if (tid < a.nbatch) {
            const float* pr = a.params + (size_t)(a.b0 + tid) * nd;
            float R[9];
            bwd_rotation(a, tid, R);
            const float f = pr[6];
#pragma unroll
            for (int i = 0; i < 9; i++) m[i] = f * R[i];
            m[9] = pr[3]; m[10] = pr[4]; m[11] = pr[5];
            m[12] = (f != 0.0f) ? 1.0f / f : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 13; i++) Mt[tid][i] = m[i];
    }
    __syncthreads();
    // staging role (threads 0..255): batch column sb_ = tid >> 2 (its 16-column block is the wave index), vertex quad q
    const bool stager = tid < 256;
    const int sb_ = (tid >> 2) & 63, q = tid & 3;
    const bool blive = stager && sb_ < a.nbatch;
    float m[13];
#pragma unroll
    for (int i = 0; i < 13; i++) m[i] = Mt[sb_][i];
    const int bc = min(sb_, a.nbatch - 1);   // (dead columns load a live column's tile and discard it)
    const float* gx = a.g + (size_t)(a.b0 + bc) * 3 * N + 4 * q;
    const float* vx = muf ? a.mu + 4 * q : a.vproj + (size_t)(a.b0 + bc) * 3 * N + 4 * q;
    // MFMA role: this wave's CB 16-coefficient blocks
    const int kq = lane >> 4, jn = lane & 15;
    const int sbt = a.sbt;
    const int sb0 = CB * wave;
    const int nsb = max(0, min(CB, sbt - sb0));          // live ones (wave-uniform; 0: a staging-only wave)
    const bool stage_wave = wave < 4;                    // (wave-uniform: the 256 staging threads are waves 0..3)
    const long long g_begin = (long long)blockIdx.x * a.rb_per_block;              // (rb_per_block = vertex groups per workgroup)
    const long long g_end = min((long long)a.rbt, g_begin + a.rb_per_block);       // (rbt = vertex groups in all)
    const int n = (int)(g_end - g_begin);
    f32x4 acc[CB][NB];
#pragma unroll
    for (int i = 0; i < CB; i++)
#pragma unroll
        for (int mb = 0; mb < NB; mb++) acc[i][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[2][3][CB];    // basis fragments of two groups in flight: [slot][coordinate][coefficient block]
    f32x4 tg[3], tv[3];    // the tile: g and vertex_proj, x / y / z rows, this thread's four vertices
    float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f, ps3 = 0.f;   // pose partial sums of this thread's column: d t3d (3), sum (q - t) . dq
    // requests (a group past the end re-requests the last one and is never consumed; a dead coefficient block re-requests the
    // image's last block: every wave issues the same number of loads in the same order, which is what the counted waits count)
    auto req_tile = [&](long long grp) {
        if (!stage_wave) return;   // (a wave without staging threads requests no tile: its counted waits count accordingly)
        const long long gc = min(grp, (long long)a.rbt - 1);
        // (the LAST group's vertices may end before its sixteen: its loads are clamped to stay inside the rows; the values
        // of the missing vertices are zeroed by the mask below)
        const long long p0 = min(16 * gc, (long long)max(N - 16, 0));
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float* pg = gx + (size_t)c * N + p0;
            const float* pv = vx + (size_t)c * N + p0;
            FRB_LD(tg[c], pg);
            FRB_LD(tv[c], pv);
        }
    };
    auto req_basis = [&](int d, int c, long long grp) {
        const long long gc = min(grp, (long long)a.rbt - 1);
        const float4* ab = a.At + ((size_t)(gc * 3 + c) * sbt) * 64 + lane;
#pragma unroll
        for (int i = 0; i < CB; i++) {
            const float4* p = ab + (size_t)min(sb0 + i, sbt - 1) * 64;
            FRB_LD(ra[d][c][i], p);
        }
    };
    // counted waits: the tile is followed by the 3 CB fragments of its group; a coordinate's CB fragments by 5 CB fragments
    // (+ 12 tile loads in a staging wave)
    auto wait_tile = [&]() {
        if (stage_wave) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(tg[0]), "+v"(tg[1]), "+v"(tg[2]), "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]) : "n"(3 * CB));
    };
    // tile -> dv rows of group grp into LDS buffer `buf` + the pose partial sums (tile registers must have arrived)
    auto stage = [&](long long grp, int buf) {
        const long long p0 = min(16 * grp, (long long)max(N - 16, 0));   // first vertex the tile was loaded from
        const int shift = (int)(16 * grp - p0);                              // > 0 only in a clamped last group
        float dvr[3][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // vertex of this slot: 16 grp + 4 q + r; it sits at tile position 4 q + r + shift (past 16: it does not exist)
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f;
            bool ok = blive && (16 * grp + 4 * q + r) < N;
            if (shift == 0) {
                g0 = tg[0][r]; g1 = tg[1][r]; g2 = tg[2][r]; v0 = tv[0][r]; v1 = tv[1][r]; v2 = tv[2][r];
            } else {
                ok = false;   // (clamped group: taken by the slow path below)
            }
            const float dq0 = ok ? g0 : 0.f, dq1 = ok ? -g1 : 0.f, dq2 = ok ? g2 : 0.f;
            // (a slot without a vertex or a column without a face is exactly zero -- not 0 * m, which an infinite f would turn
            // into a NaN that the zero-padded basis rows then spread over the whole sum)
#pragma unroll
            for (int c = 0; c < 3; c++)
                dvr[c][r] = ok ? __builtin_fmaf(m[6 + c], dq2, __builtin_fmaf(m[3 + c], dq1, m[c] * dq0)) : 0.f;
            float fs;
            if (muf) {   // mu_p . dv_p (v0..v2 hold mu's x / y / z of this vertex)
                fs = ok ? __builtin_fmaf(v2, dvr[2][r], __builtin_fmaf(v1, dvr[1][r], v0 * dvr[0][r])) : 0.f;
            } else {
                const float q0 = v0 - m[9], q1 = ((a.im_size - 1.0f) - v1) - m[10], q2 = v2 - m[11];
                fs = ok ? __builtin_fmaf(q2, dq2, __builtin_fmaf(q1, dq1, q0 * dq0)) : 0.f;
            }
            ps0 += dq0; ps1 += dq1; ps2 += dq2; ps3 += fs;
        }
        if (shift != 0 && stager) {   // the one clamped group of the launch: guarded scalar loads (drains the wave's loads once)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const long long pvx = 16 * grp + 4 * q + r;
                float dq0 = 0.f, dq1 = 0.f, dq2 = 0.f, fs = 0.f;
                const bool okx = blive && pvx < N;
                float vv[3] = {0.f, 0.f, 0.f};
                if (okx) {
                    const float* gb = a.g + (size_t)(a.b0 + bc) * 3 * N + pvx;
                    const float* vb = muf ? a.mu + pvx : a.vproj + (size_t)(a.b0 + bc) * 3 * N + pvx;
                    dq0 = gb[0]; dq1 = -gb[N]; dq2 = gb[2 * (size_t)N];
                    vv[0] = vb[0]; vv[1] = vb[N]; vv[2] = vb[2 * (size_t)N];
                }
#pragma unroll
                for (int c = 0; c < 3; c++)
                    dvr[c][r] = okx ? __builtin_fmaf(m[6 + c], dq2, __builtin_fmaf(m[3 + c], dq1, m[c] * dq0)) : 0.f;
                if (okx) {
                    if (muf) {
                        fs = __builtin_fmaf(vv[2], dvr[2][r], __builtin_fmaf(vv[1], dvr[1][r], vv[0] * dvr[0][r]));
                    } else {
                        const float q0 = vv[0] - m[9], q1 = ((a.im_size - 1.0f) - vv[1]) - m[10], q2 = vv[2] - m[11];
                        fs = __builtin_fmaf(q2, dq2, __builtin_fmaf(q1, dq1, q0 * dq0));
                    }
                }
                ps0 += dq0; ps1 += dq1; ps2 += dq2; ps3 += fs;
            }
        }
        if (stager) {
            // B fragment of k-step j = q: lane 16 r + (column & 15) holds dv[row 4 q + r][column], component = column >> 4
            float* base = reinterpret_cast<float*>(&dvL[buf][0][q][0]) + (size_t)(sb_ & 15) * 4 + (sb_ >> 4);
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) base[(size_t)c * (4 * 64 * 4) + (size_t)(16 * r) * 4] = dvr[c][r];
        }
    };
    if (n > 0) {
        // prologue: group 0's tile and basis fragments, its dv rows into buffer 0; then group 1's tile and fragments
        req_tile(g_begin);
#pragma unroll
        for (int c = 0; c < 3; c++) req_basis(0, c, g_begin);
        wait_tile();
        stage(g_begin, 0);
        req_tile(g_begin + 1);
#pragma unroll
        for (int c = 0; c < 3; c++) req_basis(1, c, g_begin + 1);
        // (two trips per loop pass, fully unrolled: the ring slot / LDS buffer index d is a compile-time constant -- indexed at
        // run time the fragment ring lands in scratch memory; a trip past the last group still waits, requests and meets the
        // barrier like the others -- every wave runs the same count -- and only skips the staging and the MFMAs)
        for (int i0 = 0; i0 < n; i0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const int i = i0 + d;
                const long long grp = g_begin + i;
                // tile of group i + 1 (behind it: the twelve fragments of group i + 1)
                wait_tile();
                // every wave has finished reading buffer d ^ 1 (group i - 1); buffer d (group i, written last trip) is published
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (i + 1 < n) stage(grp + 1, d ^ 1);
                req_tile(grp + 2);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    // the CB fragments of (group i, coordinate c), the oldest loads in flight
                    if constexpr (CB == 4) {
                        if (stage_wave) asm volatile("s_waitcnt vmcnt(32)" : "+v"(ra[d][c][0]), "+v"(ra[d][c][1]), "+v"(ra[d][c][2]), "+v"(ra[d][c][3]));
                        else asm volatile("s_waitcnt vmcnt(20)" : "+v"(ra[d][c][0]), "+v"(ra[d][c][1]), "+v"(ra[d][c][2]), "+v"(ra[d][c][3]));
                    } else {
                        static_assert(CB == 2 || CB == 4, "coefficient blocks per wave");
                        if (stage_wave) asm volatile("s_waitcnt vmcnt(22)" : "+v"(ra[d][c][0]), "+v"(ra[d][c][1]));
                        else asm volatile("s_waitcnt vmcnt(10)" : "+v"(ra[d][c][0]), "+v"(ra[d][c][1]));
                    }
                    if (nsb > 0 && i < n) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {          // k-step: rows 4 j .. 4 j + 3 of the row block, ascending
                            const float4 bv = dvL[d][c][j][lane];
                            const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                            for (int ii = 0; ii < CB; ii++)
                                if (ii < nsb) {
#pragma unroll
                                    for (int mb = 0; mb < NB; mb++)
                                        acc[ii][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[d][c][ii][j], bq[mb], acc[ii][mb], 0, 0, 0);
                                }
                        }
                    }
                    req_basis(d, c, grp + 2);
                }
            }
        }
#pragma unroll
        for (int dd = 0; dd < 2; dd++)
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int ii = 0; ii < CB; ii++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[dd][c][ii]));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(tg[0]), "+v"(tg[1]), "+v"(tg[2]), "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]));
    }
    // mu form of d f: this workgroup's share of alpha . d alpha + beta . d beta -- the parameters against the PARTIAL coefficient
    // gradients it has just accumulated (the dot product is linear, so the shares of all workgroups add up to the whole; no
    // third launch, nothing read back).  Lane (kq, jn) of wave w holds slots 16 (CB w + i) + 4 kq + rg of columns 16 mb + jn:
    // its products go to xp[w][kq][column], and the column's 4 x waves partials are added in a fixed order below.
    float* xp = reinterpret_cast<float*>(&dvL[0][0][0][0]);   // (the dv buffers are dead: every wave is past its last MFMA)
    const int nwaves = (int)(blockDim.x >> 6);
    if (muf) {
        __syncthreads();
        float xd[NB];
#pragma unroll
        for (int mb = 0; mb < NB; mb++) xd[mb] = 0.f;
#pragma unroll
        for (int i = 0; i < CB; i++)
            if (i < nsb) {
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int slot = 16 * (CB * wave + i) + 4 * kq + rg;
                    // slot -> parameter: shape slots [0, ns), expression slots from 16 sbs; padding slots carry no parameter.
                    // (Gathered here, once per workgroup.  Staging the parameters by slot in LDS at kernel start instead --
                    // 62 KB, conflict-free reads -- measured 1 / 4 / 7 us SLOWER at 64 / 32 / 16 faces: tools/bwd_ab_probe.py.)
                    int k = -1;
                    if (slot < 16 * a.sbs) { if (slot < a.ns) k = slot; }
                    else if (slot - 16 * a.sbs < a.ne) k = a.ns + slot - 16 * a.sbs;
#pragma unroll
                    for (int mb = 0; mb < NB; mb++) {
                        const int col = 16 * mb + jn;
                        const float x = (k >= 0 && col < a.nbatch) ? a.params[(size_t)(a.b0 + col) * nd + FR_N_POSE + k] : 0.f;
                        xd[mb] = __builtin_fmaf(x, acc[i][mb][rg], xd[mb]);
                    }
                }
            }
#pragma unroll
        for (int mb = 0; mb < 4; mb++) xp[(wave * 4 + kq) * 64 + 16 * mb + jn] = mb < NB ? xd[mb < NB ? mb : 0] : 0.f;
        __syncthreads();
    }
    // pose partial sums: the four vertex quads of a column are the four lanes of a DPP quad -- (q0 + q1) + (q2 + q3), fixed
    if (stager) {
#define FR_QUAD_SUM(x)                                                                                                         \
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)); /* quad_perm [1,0,3,2] */       \
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true)); /* quad_perm [2,3,0,1] */
        FR_QUAD_SUM(ps0) FR_QUAD_SUM(ps1) FR_QUAD_SUM(ps2) FR_QUAD_SUM(ps3)
#undef FR_QUAD_SUM
        if (q == 0) {
            if (muf) {   // + this workgroup's share of alpha . d alpha + beta . d beta, waves ascending, kq ascending
                float t = 0.f;
                for (int w = 0; w < nwaves; w++)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) t += xp[(w * 4 + kk) * 64 + sb_];
                ps3 += t;
            }
            float* pp = a.pose_part + ((size_t)blockIdx.x * 64 + sb_) * 4;
            pp[0] = ps0; pp[1] = ps1; pp[2] = ps2; pp[3] = ps3 * m[12];
        }
    }
    // D tile of coefficient block i: row m = 4 * (lane >> 4) + reg is slot 16 (CB wave + i) + m, column (batch in block) = lane & 15
    float* slab = a.slab + (size_t)blockIdx.x * a.nslots * 64;
#pragma unroll
    for (int i = 0; i < CB; i++)
        if (i < nsb) {
#pragma unroll
            for (int mb = 0; mb < NB; mb++)
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int slot = 16 * (CB * wave + i) + 4 * kq + rg;
                    slab[(size_t)slot * 64 + 16 * mb + jn] = acc[i][mb][rg];
                }
        }
}
#undef FRB_LD

// ---- fixed-order reduction of the partials ---------------------------------------------------------------------------------
// One 1024-thread workgroup per 64 consecutive outputs (output i = what*64 + batch: what 0..3 = d t3d / d f, 4.. = the
// padded coefficients).  Wave w sums its contiguous 1/16 of the partials with eight independent loads in flight per
// lane (the slabs are contiguous in i, so every load is a 256-byte row); the 16 wave sums meet in LDS and are added in
// wave order.  The association is fixed by (gemm_blocks, pre_blocks) alone, so the result is bit-reproducible.
constexpr int RED_WAVES = 16;
__global__ __launch_bounds__(RED_WAVES * 64) void bwd_reduce_kernel(BwdArgs a) {
    __shared__ float part[RED_WAVES][64];
    const int nd = FR_N_POSE + a.ns + a.ne;
    const int nslots = a.nslots;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int what = blockIdx.x, b = lane;  // i = what * 64 + b
    const float* src;
    size_t kstride;
    int nk;
    if (what < 4) {
        src = a.pose_part + (size_t)b * 4 + what;
        kstride = 64 * 4;
        nk = a.pre_blocks;
    } else {
        src = a.slab + (size_t)(what - 4) * 64 + b;
        kstride = (size_t)nslots * 64;
        nk = a.gemm_blocks;
    }
    const int per = (nk + RED_WAVES - 1) / RED_WAVES;
    const int k0 = wave * per, k1 = min(nk, k0 + per);
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; u++) s[u] = 0.f;
    int k = k0;
    for (; k + 16 <= k1; k += 16) {   // sixteen loads in flight; s[u] still receives k0+u, k0+u+8, ... in that order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = src[(size_t)(k + u) * kstride];
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] += v[u];
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] += v[8 + u];
    }
    for (; k + 8 <= k1; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t)(k + u) * kstride];
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] += v[u];
    }
    for (; k < k1; k++) s[0] += src[(size_t)k * kstride];
    part[wave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (wave != 0 || b >= a.nbatch) return;
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < RED_WAVES; w++) tot += part[w][lane];
    float* gp = a.grad_params + (size_t)(a.b0 + b) * nd;
    if (what < 3) {
        gp[3 + what] = tot;
        gp[what] = 0.0f;  // angles: no gradient through tf.py_func (network.py:150)
    } else if (what == 3) {
        gp[6] = tot;
    } else {
        const int slot = what - 4, e0 = a.exp_slot0;   // shape slots [0, e0), expression slots from e0
        if (slot < e0) {
            if (slot < a.ns) gp[FR_N_POSE + slot] = tot;
        } else if (slot - e0 < a.ne) {
            gp[FR_N_POSE + a.ns + slot - e0] = tot;
        }
    }
}

struct BwdGeom {
    int pre_blocks, gemm_blocks, rows_per_block;
    size_t dv_bytes, pose_bytes, slab_bytes;
    // packed-image variant
    int sbs, sbt, rbt, rb_per_block, gemm_blocks_p, waves_p;
    int ngroups, groups_per_block, block_waves_p;   // packed path: vertex groups of 16, per workgroup; waves per workgroup
    int cb_p;                                       // packed path: 16-coefficient blocks per wave (2 or 4)
    size_t slab_bytes_p, at_bytes;
};
static BwdGeom bwd_geom(int N, int ns = 199, int ne = 29) {
    BwdGeom g;
    g.pre_blocks = (N + BW_PV - 1) / BW_PV;
    const long long rows = 3ll * N;
    long long want = 512;
    long long rpb = (rows + want - 1) / want;
    rpb = ((rpb + 3) / 4) * 4;
    if (rpb < 4) rpb = 4;
    g.rows_per_block = (int)rpb;
    g.gemm_blocks = (int)((rows + rpb - 1) / rpb);
    // (dv rows padded to whole 16-row blocks: the packed variant reads them as 1 KiB fragments)
    g.rbt = (int)((rows + 15) / 16);
    g.dv_bytes = (size_t)g.rbt * 16 * 16 * sizeof(float4);
    g.sbs = (ns + 15) / 16;
    g.sbt = g.sbs + (ne + 15) / 16;
    g.waves_p = (g.sbt + 3) / 4;
    // Packed path: row chunks = workgroups = partial slabs: FR_BWD_CHUNKS, at most 512.  Default 256 = one four-wave workgroup per
    // CU of an MI355X (a constant, not the device's CU count: the chunking fixes the association of the sums, so the gradient's
    // bits must not depend on the part).  Round 4, same-process A/B through the autograd surface (round 3's prepass + ring GEMM):
    // 512 chunks 108.2 us, 256 chunks 96.0 us at 64 faces; 80.2 -> 76.4 us at 32; 384 and 128 are slower than both.  The slab /
    // partial space is sized for 512 chunks whatever the knob says, so that a workspace sized before it changed still fits.
    int chunks = opt(OPT_BWD_CHUNKS);
    if (chunks < 1 || chunks > 512) chunks = 256;
    g.ngroups = (N + 15) / 16;
    g.groups_per_block = (g.ngroups + chunks - 1) / chunks;
    if (g.groups_per_block < 1) g.groups_per_block = 1;
    g.gemm_blocks_p = g.ngroups > 0 ? (g.ngroups + g.groups_per_block - 1) / g.groups_per_block : 0;
    // FR_BWD_CB: 16-coefficient blocks per wave -- 2 (eight waves, two per SIMD), 4 (four waves), 0 = by batch (below).
    // Same-box runs through the autograd surface: 64 faces 77 us (2) vs 80 (4); 32 faces 73 (2) vs 67 (4).
    g.cb_p = opt(OPT_BWD_CB) == 4 ? 4 : 2;
    g.block_waves_p = (g.sbt + g.cb_p - 1) / g.cb_p;
    if (g.block_waves_p < 4) g.block_waves_p = 4;      // (the 256 staging threads of bwd_fused_kernel)
    {
        const int max_blocks = g.ngroups < 512 ? g.ngroups : 512;
        g.slab_bytes_p = (size_t)max_blocks * 64 * g.block_waves_p * 64 * sizeof(float);
        const size_t pose_p = (size_t)max_blocks * 64 * 4 * sizeof(float);
        const size_t pose_r = (size_t)g.pre_blocks * 64 * 4 * sizeof(float);
        g.pose_bytes = (((pose_p > pose_r ? pose_p : pose_r)) + 15) & ~(size_t)15;
    }
    g.at_bytes = (size_t)g.ngroups * 3 * g.sbt * 64 * sizeof(float4);
    g.slab_bytes = (size_t)g.gemm_blocks * 64 * bw_waves(ns, ne) * 64 * sizeof(float);
    return g;
}

}  // namespace fr

size_t fr_decode_backward_workspace_impl(int N, int ns, int ne) {
    if (N <= 0) return 0;
    fr::BwdGeom g = fr::bwd_geom(N, ns, ne);
    return g.dv_bytes + g.pose_bytes + (g.slab_bytes > g.slab_bytes_p ? g.slab_bytes : g.slab_bytes_p);
}

// the packed path serves bases of at most 16 coefficient blocks (see bwd_fused_kernel)
static bool bwd_packed_supported(int ns, int ne) { return ns + ne > 0 && (ns + 15) / 16 + (ne + 15) / 16 <= 16; }
// ... and meshes of at least one whole 16-vertex tile: the fused kernel clamps a tile's origin to max(N - 16, 0) and always loads
// sixteen floats per row, which for N < 16 would run past the row (and, for the last batch's z row, past the tensor).  Smaller
// meshes take the reference-layout entry point (fr_decode_backward_basis_bytes answers 0 for them).
static bool bwd_packed_mesh_ok(int N) { return N >= 16; }

size_t fr_decode_backward_basis_bytes_impl(int N, int ns, int ne) {
    if (N <= 0 || !bwd_packed_supported(ns, ne) || !bwd_packed_mesh_ok(N)) return 0;
    return fr::bwd_geom(N, ns, ne).at_bytes;
}

int fr_launch_decode_backward_pack(const float* pc_shape, const float* pc_exp, int N, int ns, int ne, void* packed_t,
                                   hipStream_t stream) {
    using namespace fr;
    if (N <= 0 || ns + ne <= 0) return FR_OK;
    if (!bwd_packed_supported(ns, ne) || !bwd_packed_mesh_ok(N)) return FR_ERR_UNSUPPORTED;
    BwdGeom g = bwd_geom(N, ns, ne);
    hipLaunchKernelGGL(bwd_pack_kernel, dim3(2048), dim3(256), 0, stream, pc_shape, pc_exp, N, ns, ne, g.sbs, g.sbt,
                       (long long)g.ngroups, reinterpret_cast<float4*>(packed_t));
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

int fr_launch_decode_backward(const float* grad_vertex_proj, const float* params, const float* vertex_proj,
                              const float* pc_shape, const float* pc_exp, const float* R_override, int B, int N, int ns,
                              int ne, float im_size, float* grad_params, void* workspace, hipStream_t stream,
                              const void* packed_t, const float* mu) {
    using namespace fr;
    if (B == 0) return FR_OK;
    const int nd = FR_N_POSE + ns + ne;
    if (N == 0) return hipMemsetAsync(grad_params, 0, (size_t)B * nd * sizeof(float), stream) == hipSuccess ? FR_OK
                                                                                                             : FR_ERR_LAUNCH;
    BwdGeom g = bwd_geom(N, ns, ne);
    const bool packed = packed_t != nullptr;
    if (packed && (!bwd_packed_supported(ns, ne) || !bwd_packed_mesh_ok(N))) return FR_ERR_UNSUPPORTED;
    const int waves = packed ? g.waves_p : bw_waves(ns, ne);
    if (waves > BW_MAXWAVES) return FR_ERR_UNSUPPORTED;  // > 512 coefficient slots
    BwdArgs a;
    a.g = grad_vertex_proj; a.params = params; a.vproj = vertex_proj;
    a.pc_shape = pc_shape; a.pc_exp = pc_exp; a.R_override = R_override; a.grad_params = grad_params;
    char* ws = reinterpret_cast<char*>(workspace);
    a.dvT4 = reinterpret_cast<float4*>(ws);
    a.pose_part = reinterpret_cast<float*>(ws + g.dv_bytes);
    a.slab = reinterpret_cast<float*>(ws + g.dv_bytes + g.pose_bytes);
    a.B = B; a.N = N; a.ns = ns; a.ne = ne; a.im_size = im_size;
    a.pre_blocks = packed ? g.gemm_blocks_p : g.pre_blocks;   // (packed path: the fused kernel's workgroups write the pose partials)
    a.gemm_blocks = packed ? g.gemm_blocks_p : g.gemm_blocks; a.rows_per_block = g.rows_per_block;
    a.At = reinterpret_cast<const float4*>(packed_t);
    a.sbt = g.sbt; a.sbs = g.sbs;
    a.rbt = packed ? g.ngroups : g.rbt;
    a.rb_per_block = packed ? g.groups_per_block : g.rb_per_block;
    a.mu = packed ? mu : nullptr;   // (the mu form of d f exists on the packed path only)
    a.exp_slot0 = packed ? 16 * g.sbs : bw_ns4(ns);
    a.nslots = packed ? 16 * g.cb_p * g.block_waves_p : 64 * bw_waves(ns, ne);
    for (int b0 = 0; b0 < B; b0 += 64) {
        a.b0 = b0;
        a.nbatch = min(B - b0, 64);
        const int nbt = (a.nbatch + 15) / 16;
        if (packed) {   // ONE launch: gradient tile -> dv rows in LDS -> MFMA reduction + the pose partial sums
            if (opt(OPT_BWD_CB) == 0) {   // by batch: four blocks per wave up to 32 live columns, two beyond
                const int cb = nbt <= 2 ? 4 : 2;
                if (cb != g.cb_p) {
                    g.cb_p = cb;
                    g.block_waves_p = (g.sbt + cb - 1) / cb < 4 ? 4 : (g.sbt + cb - 1) / cb;
                    a.nslots = 16 * cb * g.block_waves_p;
                }
            }
            const dim3 gb(g.block_waves_p * 64);
#define FR_BWD_LAUNCH1(NBV, CBV) hipLaunchKernelGGL((bwd_fused_kernel<NBV, CBV>), dim3(a.gemm_blocks), gb, 0, stream, a);
#define FR_BWD_LAUNCH(NBV)                                                                                            \
    {                                                                                                                 \
        if (g.cb_p == 2) FR_BWD_LAUNCH1(NBV, 2)                                                                       \
        else FR_BWD_LAUNCH1(NBV, 4)                                                                                   \
    }
            if (nbt == 1) FR_BWD_LAUNCH(1)
            else if (nbt == 2) FR_BWD_LAUNCH(2)
            else if (nbt == 3) FR_BWD_LAUNCH(3)
            else FR_BWD_LAUNCH(4)
#undef FR_BWD_LAUNCH
#undef FR_BWD_LAUNCH1
        } else {
            hipLaunchKernelGGL(bwd_prepass_kernel, dim3(g.pre_blocks), dim3(256), 0, stream, a);
            if (waves > 0) {
                const dim3 gb(waves * 64);
                if (nbt == 1) hipLaunchKernelGGL(bwd_gemm_kernel<1>, dim3(g.gemm_blocks), gb, 0, stream, a);
                else if (nbt == 2) hipLaunchKernelGGL(bwd_gemm_kernel<2>, dim3(g.gemm_blocks), gb, 0, stream, a);
                else if (nbt == 3) hipLaunchKernelGGL(bwd_gemm_kernel<3>, dim3(g.gemm_blocks), gb, 0, stream, a);
                else hipLaunchKernelGGL(bwd_gemm_kernel<4>, dim3(g.gemm_blocks), gb, 0, stream, a);
            }
        }
        hipLaunchKernelGGL(bwd_reduce_kernel, dim3(4 + 64 * waves), dim3(RED_WAVES * 64), 0, stream, a);
    }
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}
