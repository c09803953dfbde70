// 3DMM decode backward for gfx950 (MI355X): dL/d vertex_proj [B,3,N] -> dL/d params [B,235].
//
// What it computes: the gradient TF autodiff derives from FaceRecNet.vertices_transform (nets/network.py:140-171):
//   dq      = (g_x, -g_y, g_z)                                  y row is (im_size - q_1) - 1 (:168)
//   d t3d_i = sum_p dq_i                                        (:164-165)
//   d f     = sum_p (R v) . dq  = sum_p (q - t3d) . dq / f      (f_expand * R, :163-165)
//   dv      = (f R)^T dq ;  d alpha = pc_shape^T dv ;  d beta = pc_exp^T dv          (:153-159)
//   d angles = 0: R comes out of tf.py_func (:150), which has no gradient in the reference.
//
// How: three launches, no float atomics (bit-reproducible):
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
    const float4* At;        // [row blocks][slot blocks][64 lanes] float4
    int sbt, sbs;            // slot blocks of 16 coefficients in all / of the shape basis (expression blocks follow)
    int rbt, rb_per_block;   // row blocks of 16 rows in all / per gemm workgroup
    int exp_slot0;           // first coefficient slot of the expression basis in the slabs
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
    const int nslots = 64 * (int)(blockDim.x >> 6);
    float* slab = a.slab + (size_t)blockIdx.x * nslots * 64;
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
                                                       int N, int ns, int ne, int sbs, int sbt, long long rbt,
                                                       float4* __restrict__ At) {
    const long long rows = 3ll * N;
    const long long total = rbt * sbt * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long long q = i >> 6;
        const int sb = (int)(q % sbt);
        const long long rb = q / sbt;
        const int slot = 16 * sb + (lane & 15);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long long row = 16 * rb + 4 * j + (lane >> 4);
            float x = 0.f;
            if (row < rows) {
                if (sb < sbs) { if (slot < ns) x = pc_shape[(size_t)row * ns + slot]; }
                else { const int c = slot - 16 * sbs; if (c < ne) x = pc_exp[(size_t)row * ne + c]; }
            }
            v[j] = x;
        }
        At[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

#define FRB_LD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
// BR: ring depth in 16-row blocks: 8 BR fragment loads (BR x 8 KiB) in flight per wave.  (Round 4, same-process A/B at one
// workgroup per CU: BR = 4 and 5 -- 32 / 40 KiB in flight per wave, 200 / 232 VGPRs -- measured 96-97 us against 96 at 64 faces.)
template <int NB, int BR = 3>
__global__ __launch_bounds__(BW_MAXWAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void bwd_gemm_ring_kernel(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kq = lane >> 4, jn = lane & 15;
    const int sbt = a.sbt;
    const int sb0 = 4 * wave;                              // this wave's four 16-coefficient blocks
    const int nsb = min(4, sbt - sb0);                     // live ones (wave-uniform, >= 1)
    const long long rb_begin = (long long)blockIdx.x * a.rb_per_block;
    const long long rb_end = min((long long)a.rbt, rb_begin + a.rb_per_block);
    f32x4 acc[4][NB];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int mb = 0; mb < NB; mb++) acc[i][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[BR][4], rbv[BR][4];   // ring: the row block's A fragments (one per coefficient block) and dv fragments (one per k-step)
    // fragment addresses of row block rb (clamped: a slot past the end re-requests the last block and is never consumed;
    // a dead coefficient block re-requests block sbt - 1)
    auto request = [&](int d, long long rb) {
        const long long rbc = min(rb, (long long)a.rbt - 1);
        const float4* ab = a.At + ((size_t)rbc * sbt) * 64 + lane;
        const float4* bb = a.dvT4 + (size_t)rbc * 256 + lane;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4* p = ab + (size_t)min(sb0 + i, sbt - 1) * 64;
            FRB_LD(ra[d][i], p);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4* p = bb + (size_t)j * 64;
            FRB_LD(rbv[d][j], p);
        }
    };
#pragma unroll
    for (int d = 0; d < BR; d++) request(d, rb_begin + d);
    for (long long rb0 = rb_begin; rb0 < rb_end; rb0 += BR) {
#pragma unroll
        for (int d = 0; d < BR; d++) {
            // the 8 loads of ring slot d are the oldest outstanding ones: all but the 8 (BR - 1) youngest must be back
            asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ra[d][0]), "+v"(ra[d][1]), "+v"(ra[d][2]), "+v"(ra[d][3]), "+v"(rbv[d][0]),
                         "+v"(rbv[d][1]), "+v"(rbv[d][2]), "+v"(rbv[d][3]) : "n"(8 * (BR - 1)));
            static_assert(8 * (BR - 1) <= 63, "vmcnt is a 6-bit counter");
            if (rb0 + d < rb_end) {
#pragma unroll
                for (int j = 0; j < 4; j++)        // k-step: rows 16 rb + 4 j .. + 3, ascending
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (i < nsb) {
#pragma unroll
                            for (int mb = 0; mb < NB; mb++)
                                acc[i][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[d][i][j], rbv[d][j][mb], acc[i][mb], 0, 0, 0);
                        }
            }
            request(d, rb0 + d + BR);
        }
    }
#pragma unroll
    for (int d = 0; d < BR; d++)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[d][0]), "+v"(ra[d][1]), "+v"(ra[d][2]), "+v"(ra[d][3]), "+v"(rbv[d][0]),
                     "+v"(rbv[d][1]), "+v"(rbv[d][2]), "+v"(rbv[d][3]));
    // D tile of coefficient block i: row m = 4 * (lane >> 4) + reg is slot 64 wave + 16 i + m, column (batch in block) = lane & 15
    // (staging the tile through LDS so that a store instruction writes four whole 256-byte slab rows instead of four 64-byte
    // pieces measured no faster: 59.2 vs 58.0 us at 64 faces -- the store tail is not what bounds the kernel)
    const int nslots = 64 * (int)(blockDim.x >> 6);
    float* slab = a.slab + (size_t)blockIdx.x * nslots * 64;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < nsb) {
#pragma unroll
            for (int mb = 0; mb < NB; mb++)
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int slot = 64 * wave + 16 * i + 4 * kq + rg;
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
    const int nslots = a.At ? 64 * ((a.sbt + 3) / 4) : 64 * bw_waves(a.ns, a.ne);
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
    // row chunks of the packed GEMM (one workgroup and one partial slab each): FR_BWD_CHUNKS, at most 512 (the slab space is
    // sized for 512 whatever the knob says, so that a workspace sized before the knob changed still fits)
    // Default 256 = one four-wave workgroup per CU of an MI355X (a constant, not the device's CU count: the chunking fixes the
    // association of the sums, so the gradient's bits must not depend on the part).  Round 4, same-process A/B through the
    // autograd surface: 512 chunks (two workgroups per CU, round 3) 108.2 us, 256 chunks 96.0 us at 64 faces; 80.2 -> 76.4 us
    // at 32; 384 and 128 are slower than both (uneven CU load / too little in flight).  Half the slabs also halve the
    // reduce kernel's reads.
    int chunks = opt(OPT_BWD_CHUNKS);
    if (chunks < 1 || chunks > 512) chunks = 256;
    g.rb_per_block = (int)((g.rbt + chunks - 1) / chunks);
    if (g.rb_per_block < 1) g.rb_per_block = 1;
    g.gemm_blocks_p = g.rbt > 0 ? (g.rbt + g.rb_per_block - 1) / g.rb_per_block : 0;
    {
        const int rpb512 = g.rbt > 512 ? (g.rbt + 511) / 512 : 1;
        const int blocks512 = g.rbt > 0 ? (g.rbt + rpb512 - 1) / rpb512 : 0;
        g.slab_bytes_p = (size_t)blocks512 * 64 * g.waves_p * 64 * sizeof(float);
    }
    g.at_bytes = (size_t)g.rbt * g.sbt * 64 * sizeof(float4);
    g.pose_bytes = (((size_t)g.pre_blocks * 64 * 4 * sizeof(float)) + 15) & ~(size_t)15;
    g.slab_bytes = (size_t)g.gemm_blocks * 64 * bw_waves(ns, ne) * 64 * sizeof(float);
    return g;
}

}  // namespace fr

size_t fr_decode_backward_workspace_impl(int N, int ns, int ne) {
    if (N <= 0) return 0;
    fr::BwdGeom g = fr::bwd_geom(N, ns, ne);
    return g.dv_bytes + g.pose_bytes + (g.slab_bytes > g.slab_bytes_p ? g.slab_bytes : g.slab_bytes_p);
}

size_t fr_decode_backward_basis_bytes_impl(int N, int ns, int ne) {
    if (N <= 0 || ns + ne <= 0) return 0;
    return fr::bwd_geom(N, ns, ne).at_bytes;
}

int fr_launch_decode_backward_pack(const float* pc_shape, const float* pc_exp, int N, int ns, int ne, void* packed_t,
                                   hipStream_t stream) {
    using namespace fr;
    if (N <= 0 || ns + ne <= 0) return FR_OK;
    BwdGeom g = bwd_geom(N, ns, ne);
    hipLaunchKernelGGL(bwd_pack_kernel, dim3(2048), dim3(256), 0, stream, pc_shape, pc_exp, N, ns, ne, g.sbs, g.sbt,
                       (long long)g.rbt, reinterpret_cast<float4*>(packed_t));
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

int fr_launch_decode_backward(const float* grad_vertex_proj, const float* params, const float* vertex_proj,
                              const float* pc_shape, const float* pc_exp, const float* R_override, int B, int N, int ns,
                              int ne, float im_size, float* grad_params, void* workspace, hipStream_t stream,
                              const void* packed_t) {
    using namespace fr;
    if (B == 0) return FR_OK;
    const int nd = FR_N_POSE + ns + ne;
    if (N == 0) return hipMemsetAsync(grad_params, 0, (size_t)B * nd * sizeof(float), stream) == hipSuccess ? FR_OK
                                                                                                             : FR_ERR_LAUNCH;
    BwdGeom g = bwd_geom(N, ns, ne);
    const bool packed = packed_t != nullptr;
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
    a.pre_blocks = g.pre_blocks; a.gemm_blocks = packed ? g.gemm_blocks_p : g.gemm_blocks; a.rows_per_block = g.rows_per_block;
    a.At = reinterpret_cast<const float4*>(packed_t);
    a.sbt = g.sbt; a.sbs = g.sbs; a.rbt = g.rbt; a.rb_per_block = g.rb_per_block;
    a.exp_slot0 = packed ? 16 * g.sbs : bw_ns4(ns);
    for (int b0 = 0; b0 < B; b0 += 64) {
        a.b0 = b0;
        a.nbatch = min(B - b0, 64);
        hipLaunchKernelGGL(bwd_prepass_kernel, dim3(g.pre_blocks), dim3(256), 0, stream, a);
        const int nbt = (a.nbatch + 15) / 16;
        if (waves > 0 && packed) {
            const dim3 gb(waves * 64);
            if (nbt == 1) hipLaunchKernelGGL(bwd_gemm_ring_kernel<1>, dim3(a.gemm_blocks), gb, 0, stream, a);
            else if (nbt == 2) hipLaunchKernelGGL(bwd_gemm_ring_kernel<2>, dim3(a.gemm_blocks), gb, 0, stream, a);
            else if (nbt == 3) hipLaunchKernelGGL(bwd_gemm_ring_kernel<3>, dim3(a.gemm_blocks), gb, 0, stream, a);
            else hipLaunchKernelGGL(bwd_gemm_ring_kernel<4>, dim3(a.gemm_blocks), gb, 0, stream, a);
        } else if (waves > 0) {
            const dim3 gb(waves * 64);
            if (nbt == 1) hipLaunchKernelGGL(bwd_gemm_kernel<1>, dim3(g.gemm_blocks), gb, 0, stream, a);
            else if (nbt == 2) hipLaunchKernelGGL(bwd_gemm_kernel<2>, dim3(g.gemm_blocks), gb, 0, stream, a);
            else if (nbt == 3) hipLaunchKernelGGL(bwd_gemm_kernel<3>, dim3(g.gemm_blocks), gb, 0, stream, a);
            else hipLaunchKernelGGL(bwd_gemm_kernel<4>, dim3(g.gemm_blocks), gb, 0, stream, a);
        }
        hipLaunchKernelGGL(bwd_reduce_kernel, dim3(4 + 64 * waves), dim3(RED_WAVES * 64), 0, stream, a);
    }
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}
