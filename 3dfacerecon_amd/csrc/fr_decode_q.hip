// 3DMM decode for gfx950, Q30 arithmetic: the basis blend  [3N x (ns+ne)] . [(ns+ne) x B]  of network.py:153-159 as an
// EXACT fixed-point dot product on the int8 matrix cores (v_mfma_i32_16x16x64_i8, 32x the rate of the f32-input MFMA),
// rounded to fp32 once.
//
// Why: the exact-f32 MFMA (fr_decode.hip) runs at the fp32 VECTOR rate -- 30 us of matrix pipe per 64 faces -- while the
// 153 MB basis needs ~25 us of HBM.  Integer arithmetic is the only other arithmetic that is order-independent and hence
// exactly re-statable on a CPU (the bf16/fp8 MFMAs accumulate in an unspecified internal order), and the int8 cores give
// 16 digit products in the time of one f32 step, which is enough for 31-bit operands:
//   qA[r,k] = rint(A[r,k] 2^(30 - re_r - ce_k))   (per-row and per-column power-of-two scales; |qA| <= 2^30)
//   qB[k,b] = rint(x[b,k] 2^(30 - be_b + ce_k))
//   v[r,b]  = fl32( mu_r + (sum_k qA qB) 2^(re_r + be_b - 60) )
// with qA, qB split into four balanced base-256 digits each and the digit products a_i b_j (weight 256^(6-i-j)) accumulated in
// int32 level sums by the MFMA.  LV (template argument of the kernels, `levels` of the entry points) = how many of the seven
// levels i + j = 0 .. 6 are kept: 7 = all sixteen products (the exact product), 5 = thirteen, 4 = ten -- what is dropped lies
// below 2^-38 / 2^-30 of a term's full scale; round 5: the ten-product kernel is 13 us shorter than the f32 kernel (rocprofv3).
// Entries within 2^-6 of their row's maximum are represented exactly; the quantisation error of the rest is below
// 2^-31 of the row / column maximum, so the result is the correctly rounded fp32 value of the real-number blend in
// 99.7 % of the cases on the model's data (mean error 0.25 ulp against the f32 chain's 0.5; tests/test_decode_q30_*.py).
// The specification is restated on the CPU by the test oracle ("Q30 decode") and the kernel is held to it BIT FOR BIT.
//
// Packed image (built once by fr_decode_pack_basis, after the f32 image): int ce[S*64] | per 16-vertex tile:
//   fragment (c, sidx, i): coordinate c, k-step s = (sidx == 0 ? S-1 : sidx-1), digit i; lane l holds 16 bytes:
//   row 16*tile + (l & 15), k = 64 s + 16 (l >> 4) + t.  Only the KB = ceil(K/16) live 16-k groups are stored, so the
//   LAST k-step's fragments are short (256 * (KB - 4 (S-1)) bytes) -- they come FIRST in a coordinate, the lanes past
//   their end read the bytes that follow (harmless: the matching parameter digits are zero), and the 256 bytes that
//   follow the very first fragment of a tile are its payload: per row {mu_x, mu_y, mu_z, -re_x | -re_y << 10 | -re_z << 20}.
//   With KB % 4 != 0 the payload therefore arrives in lanes 48..63 of the tile's first fragment load, free of charge.
#include "fr_decode_shared.h"

namespace fr {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct QShape {
    int K, KB, S, ngl;           // coefficients, live 16-k groups, k-steps of 64, groups in the last k-step
    size_t coord_bytes, tile_bytes, hdr_bytes;
};
__host__ __device__ inline QShape q_shape(int ns, int ne) {
    QShape q;
    q.K = ns + ne;
    q.KB = (q.K + 15) / 16;
    if (q.KB == 0) q.KB = 1;
    q.S = (q.KB + 3) / 4;
    q.ngl = q.KB - 4 * (q.S - 1);
    q.coord_bytes = (size_t)1024 * q.KB;
    q.tile_bytes = 3 * q.coord_bytes + 256;
    q.hdr_bytes = ((size_t)q.S * 64 * sizeof(int) + 255) & ~(size_t)255;
    return q;
}
// byte offset of fragment (c, sidx, i) inside a tile
__host__ __device__ inline size_t q_frag_off(const QShape& q, int c, int sidx, int i) {
    size_t o = (size_t)c * q.coord_bytes;
    o += sidx == 0 ? (size_t)i * 256 * q.ngl : (size_t)4 * 256 * q.ngl + (size_t)(sidx - 1) * 4096 + (size_t)i * 1024;
    return o + ((c | sidx | i) ? 256 : 0);
}

__device__ __forceinline__ int q_exp_of(double x) {  // frexp exponent: x = m 2^e, 0.5 <= |m| < 1
    int e;
    (void)frexp(x, &e);
    return e;
}
__device__ __forceinline__ int q_digit(int q, int i) {  // balanced base-256 digit i (0 = most significant) of q
    int d3 = ((q + 128) & 255) - 128;
    q = (q - d3) >> 8;
    int d2 = ((q + 128) & 255) - 128;
    q = (q - d2) >> 8;
    int d1 = ((q + 128) & 255) - 128;
    q = (q - d1) >> 8;
    return i == 0 ? q : i == 1 ? d1 : i == 2 ? d2 : d3;
}

// ---- pack -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float q_basis(const float* pc_shape, const float* pc_exp, size_t row, int k, int ns, int ne) {
    return k < ns ? pc_shape[row * ns + k] : pc_exp[row * ne + (k - ns)];
}
// column maxima (bit patterns of |a|, finite entries only): one thread per column walks a chunk of rows
__global__ __launch_bounds__(256) void q_colmax_kernel(const float* __restrict__ pc_shape, const float* __restrict__ pc_exp,
                                                       int N, int ns, int ne, unsigned* __restrict__ colmax) {
    const int K = ns + ne;
    const size_t rows = (size_t)3 * N;
    const size_t chunk = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        unsigned m = 0;
        for (size_t r = r0; r < r1; r++) {
            const unsigned u = __float_as_uint(q_basis(pc_shape, pc_exp, r, k, ns, ne)) & 0x7FFFFFFFu;
            if (u < 0x7F800000u && u > m) m = u;
        }
        if (m) atomicMax(&colmax[k], m);
    }
}
// (in place: the header's ce[] slots first collect the column maxima, then each thread turns its own slot into the exponent)
__global__ void q_ce_kernel(int K, int KP, int* ce) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= KP) return;
    const unsigned m = k < K ? (unsigned)ce[k] : 0u;
    ce[k] = m ? q_exp_of((double)__uint_as_float(m)) : 0;
}
// per-vertex payload: mu and the three row exponents
__global__ __launch_bounds__(256) void q_payload_kernel(const float* __restrict__ mu, const float* __restrict__ pc_shape,
                                                        const float* __restrict__ pc_exp, int N, int ns, int ne,
                                                        const int* __restrict__ ce, char* __restrict__ tiles_base,
                                                        QShape qs) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long NP = (long long)tiles_of(N) * TILE_V;
    if (p >= NP) return;
    float m3[3] = {0.f, 0.f, 0.f};
    unsigned ex = 0;
    if (p < N) {
        for (int c = 0; c < 3; c++) {
            const size_t row = (size_t)c * N + p;
            m3[c] = mu[row];
            int re = INT_MIN;
            bool bad = false;
            for (int k = 0; k < qs.K; k++) {
                const float a = q_basis(pc_shape, pc_exp, row, k, ns, ne);
                if (!isfinite(a)) bad = true;
                else if (a != 0.f) re = max(re, q_exp_of((double)a) - ce[k]);
            }
            if (re == INT_MIN) re = 0;
            const unsigned e10 = bad ? 1023u : (unsigned)min(1022, max(0, -re));   // re <= 0 by construction
            ex |= e10 << (10 * c);
        }
    }
    uint4 w;
    w.x = __float_as_uint(m3[0]);
    w.y = __float_as_uint(m3[1]);
    w.z = __float_as_uint(m3[2]);
    w.w = ex;
    char* tb = tiles_base + (size_t)(p / TILE_V) * qs.tile_bytes;
    *reinterpret_cast<uint4*>(tb + (size_t)256 * qs.ngl + (size_t)(p % TILE_V) * 16) = w;
}
// digit fragments: one thread per (tile, c, sidx, i, lane)
__global__ __launch_bounds__(256) void q_pack_kernel(const float* __restrict__ pc_shape, const float* __restrict__ pc_exp,
                                                     int N, int ns, int ne, const int* __restrict__ ce,
                                                     char* __restrict__ tiles_base, QShape qs) {
    const long long per_tile = (long long)3 * qs.S * 4 * 64;
    const long long total = (long long)tiles_of(N) * per_tile;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += step) {
        const int lane = (int)(it & 63);
        long long r = it >> 6;
        const int i = (int)(r & 3);
        r >>= 2;
        const int sidx = (int)(r % qs.S);
        r /= qs.S;
        const int c = (int)(r % 3);
        const long long tile = r / 3;
        const int s = sidx == 0 ? qs.S - 1 : sidx - 1;
        const int g = lane >> 4;
        if (sidx == 0 && g >= qs.ngl) continue;   // not stored
        char* tb = tiles_base + (size_t)tile * qs.tile_bytes;
        const long long p = tile * TILE_V + (lane & 15);
        unsigned w[4] = {0u, 0u, 0u, 0u};
        if (p < N) {
            const uint4 pay = *reinterpret_cast<const uint4*>(tb + (size_t)256 * qs.ngl + (size_t)(lane & 15) * 16);
            const unsigned e10 = (pay.w >> (10 * c)) & 1023u;
            const int re = -(int)e10;   // (a flagged row: its digits are irrelevant, the epilogue writes NaN)
            const size_t row = (size_t)c * N + p;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int k = 64 * s + 16 * g + t;
                int d = 0;
                if (k < qs.K) {
                    const float a = q_basis(pc_shape, pc_exp, row, k, ns, ne);
                    if (isfinite(a) && e10 != 1023u) d = q_digit((int)rint(ldexp((double)a, 30 - re - ce[k])), i);
                }
                w[t >> 2] |= (unsigned)(d & 255) << (8 * (t & 3));
            }
        }
        *reinterpret_cast<uint4*>(tb + q_frag_off(qs, c, sidx, i) + (size_t)lane * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- decode -----------------------------------------------------------------------------------------------------------
struct DecodeQArgs {
    DecodeArgs d;          // params, R_override, out, B, N, ns, ne, b0, im_size (A / mu_p / halves unused)
    const char* tiles;     // first tile of the Q image
    const int* ce;         // column exponents [S*64]
    QShape qs;
    const char* stage;     // this pass's staged parameters (q_stage_kernel)
    int cb0;               // generic kernel: first 16-column block of this launch (0, or 2 for the second half of a pass)
};

constexpr int Q_BE_BAD = 0x7FFFFFFF;

// Staging kernel, one launch per pass of <= 64 faces, one workgroup per column b: the face's parameters become digit
// fragments (B operand), its pose becomes Mt = f.R | t3d (float64 rotation, network.py:266-297), ONCE -- the decode
// kernel's 256 workgroups then only copy the 64 + 3 KiB image into their LDS (quantising in every workgroup cost 12 us
// of every CU's time).  Image: [(s*4 + j)*4 + nb][lane] x 16 bytes; lane l of column block nb holds column
// 16 nb + (l & 15), k = 64 s + 16 (l >> 4) + t; then Mt [64][12] floats, then be [64] ints (the column's exponent;
// Q_BE_BAD: a non-finite parameter).
__host__ __device__ inline size_t q_stage_bytes(int S) { return (size_t)S * 16384 + MAXB * 12 * sizeof(float) + MAXB * sizeof(int); }
__global__ __launch_bounds__(128) void q_stage_kernel(DecodeQArgs a, char* __restrict__ stage) {
    __shared__ int be_sh;
    __shared__ double sc_sh[6];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int K = a.qs.K, S = a.qs.S;
    const int nd = FR_N_POSE + a.d.ns + a.d.ne;
    const int nbatch = min(a.d.B - a.d.b0, MAXB);
    const bool live = b < nbatch;
    const float* pr = a.d.params + (size_t)(a.d.b0 + (live ? b : 0)) * nd;
    if (tid == 0) be_sh = INT_MIN;
    __syncthreads();
    // thread tid owns k = 4 tid .. 4 tid + 3 (one dword of each digit's 16-byte slot)
    float x[4];
    int cek[4];
    const bool mine = 4 * tid < 64 * S;
    int e = INT_MIN;
    bool bad = false;
    // the eight loads of a thread are unconditional (clamped) and issued together: written as `cond ? load : 0` each becomes a
    // branch with its own s_waitcnt vmcnt(0) -- eight dependent round trips, most of what this kernel used to take
    float pose[FR_N_POSE];   // (the column's seven pose parameters: the same round trip, not one more behind each branch below)
    {
        const int kmax = K > 0 ? K - 1 : 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int kc = min(4 * tid + t, kmax);
            x[t] = pr[K > 0 ? FR_N_POSE + kc : 0];   // (no coefficients at all: the row ends at the pose, stay inside it)
            cek[t] = a.ce[kc];
        }
#pragma unroll
        for (int i = 0; i < FR_N_POSE; i++) pose[i] = pr[i];
#pragma unroll
        for (int t = 0; t < 4; t++) asm volatile("" : "+v"(x[t]), "+v"(cek[t]));
#pragma unroll
        for (int i = 0; i < FR_N_POSE; i++) asm volatile("" : "+v"(pose[i]));
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int k = 4 * tid + t;
        if (!(live && k < K)) x[t] = 0.f;
        if (!(mine && k < K)) cek[t] = 0;
        if (!isfinite(x[t])) bad = true;
        else if (x[t] != 0.f) e = max(e, __builtin_amdgcn_frexp_expf(x[t]) + cek[t]);
    }
    if (bad) atomicMax(&be_sh, Q_BE_BAD);
    else if (e != INT_MIN) atomicMax(&be_sh, e);
    if (tid < 3 && live && !a.d.R_override) {
        double sn, cs;
        sincos((double)(tid == 0 ? pose[0] : tid == 1 ? pose[1] : pose[2]), &sn, &cs);
        sc_sh[2 * tid] = sn;
        sc_sh[2 * tid + 1] = cs;
    }
    __syncthreads();
    int be = be_sh;
    if (be == INT_MIN) be = 0;
    if (mine) {
        unsigned w[4] = {0u, 0u, 0u, 0u};
        if (live && be != Q_BE_BAD) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (4 * tid + t < K) {
                    const int qv = (int)rint(ldexp((double)x[t], cek[t] + 30 - be));
                    int d3 = ((qv + 128) & 255) - 128;
                    int q1 = (qv - d3) >> 8;
                    int d2 = ((q1 + 128) & 255) - 128;
                    q1 = (q1 - d2) >> 8;
                    int d1 = ((q1 + 128) & 255) - 128;
                    q1 = (q1 - d1) >> 8;
                    w[0] |= (unsigned)(q1 & 255) << (8 * t);
                    w[1] |= (unsigned)(d1 & 255) << (8 * t);
                    w[2] |= (unsigned)(d2 & 255) << (8 * t);
                    w[3] |= (unsigned)(d3 & 255) << (8 * t);
                }
            }
        }
        const int u = tid >> 2, s = u >> 2, g = u & 3, nb = b >> 4;
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<unsigned*>(stage + ((size_t)((s * 4 + j) * 4 + nb) * 64 + g * 16 + (b & 15)) * 16 + 4 * (tid & 3)) = w[j];
    }
    if (tid == 0) {
        float* Mt = reinterpret_cast<float*>(stage + (size_t)S * 16384) + b * 12;
        int* be_o = reinterpret_cast<int*>(stage + (size_t)S * 16384 + MAXB * 12 * sizeof(float));
        float m[12];
#pragma unroll
        for (int i = 0; i < 12; i++) m[i] = 0.f;
        if (live) {
            float R[9];
            if (a.d.R_override) {
#pragma unroll
                for (int i = 0; i < 9; i++) R[i] = a.d.R_override[(size_t)(a.d.b0 + b) * 9 + i];
            } else {
                rotation_from_sincos(sc_sh[0], sc_sh[1], sc_sh[2], sc_sh[3], sc_sh[4], sc_sh[5], R);
            }
            const float f = pose[6];
#pragma unroll
            for (int i = 0; i < 9; i++) m[i] = f * R[i];  // f (.) R elementwise, network.py:163-165
            m[9] = pose[3];
            m[10] = pose[4];
            m[11] = pose[5];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) Mt[i] = m[i];
        be_o[b] = live ? be : 0;
    }
}
// staged image -> LDS (the decode kernels' whole prologue); ends with a barrier
template <int DEC_BLOCK>
__device__ __forceinline__ void q_copy_stage(const char* __restrict__ stage, char* lds, int S, int tid) {
    const int n16 = (int)(q_stage_bytes(S) / 16);
    for (int i = tid; i < n16; i += DEC_BLOCK)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(stage)[i];
    __syncthreads();
}

// the model's basis shape (199 + 29 coefficients): 15 live 16-k groups, 4 k-steps of 64, 3 groups in the last one, 48 fragments
constexpr int QR_KB = 15, QR_S = 4, QR_NGL = 3, QR_F = 48;
constexpr size_t QR_CB = (size_t)1024 * QR_KB, QR_TB = 3 * QR_CB + 256;

// One coordinate's LV level sums -> fl32(mu + h 2^(be - e_r - 60 + 8 (7 - LV))) for this lane's 4 rows x NBW columns, h the
// written chain h = fl64(h * 256 + L_s).  Two steps are taken in cheaper, bit-identical forms: the leading digits of the
// 31-bit operands are at most 64 in magnitude, so with K <= 512 coefficients |L_0| <= 2^21 and |L_1| <= 2^23 and
// L_0 * 256 + L_1 is exact in int32 (one integer instruction instead of two conversions and a float64 multiply-add); and
// h * 2^e is exact (a power of two, far from the exponent range's ends), so fma(h, 2^e, mu) == mu + h * 2^e to the bit.
template <int LV, int NBW>
__device__ __forceinline__ void q_finish(const i32x4 (&acc)[LV][NBW], const float (&mu4)[4], const int (&e4)[4],
                                         const int (&be)[NBW], f32x4 (&out)[NBW]) {
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double h;
            if constexpr (LV >= 2) {
                h = (double)(acc[0][nb][r] * 256 + acc[1][nb][r]);
#pragma unroll
                for (int s = 2; s < LV; s++) h = __builtin_fma(h, 256.0, (double)acc[s][nb][r]);
            } else {
                h = (double)acc[0][nb][r];
            }
            const bool bad = e4[r] == 1023 || be[nb] == Q_BE_BAD;
            const int e = bad ? 0 : be[nb] - e4[r] - 60 + 8 * (7 - LV);
            const double sc = __longlong_as_double((long long)(1023 + e) << 52);
            const double d = __builtin_fma(h, sc, (double)mu4[r]);
            out[nb][r] = bad ? __uint_as_float(0x7FC00000u) : (float)d;
        }
    }
}

// Generic kernel (any basis shape with ns + ne <= 512): one wave per 16-vertex tile, NBW column blocks, plain loads.
template <int NBW, int DEC_WAVES, int LV>
__global__ __launch_bounds__(DEC_WAVES * 64) void decode_q_kernel(DecodeQArgs a) {
    constexpr int DEC_BLOCK = DEC_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char qsmem[];
    const QShape qs = a.qs;
    char* Bimg = qsmem;                                                   // S * 16 KiB
    float* Mt = reinterpret_cast<float*>(Bimg + (size_t)qs.S * 16384);     // [64][12]
    int* be_s = reinterpret_cast<int*>(Mt + MAXB * 12);                    // [64]
    const int tid = threadIdx.x;
    const int nd = FR_N_POSE + a.d.ns + a.d.ne;
    const int nbatch = min(a.d.B - a.d.b0, MAXB);
    q_copy_stage<DEC_BLOCK>(a.stage, qsmem, qs.S, tid);

    const int lane = tid & 63, wave = tid >> 6;
    const int tiles = tiles_of(a.d.N);
    int be[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) be[nb] = be_s[16 * (a.cb0 + nb) + (lane & 15)];
    for (int tile = (int)blockIdx.x * DEC_WAVES + wave; tile < tiles; tile += (int)gridDim.x * DEC_WAVES) {
        const char* tb = a.tiles + (size_t)tile * qs.tile_bytes;
        uint4 pay[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            pay[r] = *reinterpret_cast<const uint4*>(tb + (size_t)256 * qs.ngl + (size_t)(4 * (lane >> 4) + r) * 16);
        f32x4 v[3][NBW];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            i32x4 acc[LV][NBW];
#pragma unroll
            for (int s = 0; s < LV; s++)
#pragma unroll
                for (int nb = 0; nb < NBW; nb++) acc[s][nb] = (i32x4){0, 0, 0, 0};
            for (int sidx = 0; sidx < qs.S; sidx++) {
                const int s = sidx == 0 ? qs.S - 1 : sidx - 1;
                i32x4 af[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    af[i] = *reinterpret_cast<const i32x4*>(tb + q_frag_off(qs, c, sidx, i) + (size_t)lane * 16);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j >= LV) continue;   // (no kept product uses this parameter digit)
                    i32x4 bf[NBW];
#pragma unroll
                    for (int nb = 0; nb < NBW; nb++)
                        bf[nb] = *reinterpret_cast<const i32x4*>(Bimg + ((size_t)((s * 4 + j) * 4 + a.cb0 + nb) * 64 + lane) * 16);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (i + j >= LV) continue;   // kept products: i + j < LV
#pragma unroll
                        for (int nb = 0; nb < NBW; nb++)
                            acc[i + j][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[nb], acc[i + j][nb], 0, 0, 0);
                    }
                }
            }
            float mu4[4];
            int e4[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                mu4[r] = __uint_as_float(c == 0 ? pay[r].x : c == 1 ? pay[r].y : pay[r].z);
                e4[r] = (int)((pay[r].w >> (10 * c)) & 1023u);
            }
            q_finish<LV, NBW>(acc, mu4, e4, be, v[c]);
        }
        decode_store<NBW>(a.d, v[0], v[1], v[2], Mt, tile, a.cb0 / NBW, lane, nbatch, a.d.N);
    }
}

// ---- streaming variant for the model's basis shape (199 + 29 coefficients: KB = 15, S = 4, last k-step 3 groups) -------
// Same arithmetic, different schedule (the one fr_decode.hip's ring kernel uses): a wave owns whole tiles (all NBW column
// blocks, so every basis byte is requested by exactly one wave of the chip) and treats its tiles as ONE stream of 1 KiB
// fragment requests through a ring of R registers quadruples, consumed four at a time (the four digits of a k-step, each
// used against the four parameter digits: 16 MFMAs x NBW per ring group).  The requests are inline asm so that the
// compiler can neither reorder nor drain them; the only waits are counted s_waitcnt vmcnt(R-4).  The tile's payload
// rides in lanes 48..63 of its first (768-byte) fragment and is parked in LDS; finished x / y rows wait in LDS for z.
#define FRQ_LD(dst, sbase, voff)                                                                                  \
    {                                                                                                             \
        if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase)); \
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));                 \
    }
template <int R>
__device__ __forceinline__ void q_ring_wait4(i32x4& a0, i32x4& a1, i32x4& a2, i32x4& a3) {
    static_assert(R == 8 || R == 12 || R == 16, "ring size");
    if constexpr (R == 8) asm volatile("s_waitcnt vmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    else if constexpr (R == 12) asm volatile("s_waitcnt vmcnt(8)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    else asm volatile("s_waitcnt vmcnt(12)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}
__host__ __device__ constexpr size_t qr_off(int f) {
    return (size_t)(f / 16) * QR_CB +
           (((f % 16) / 4) == 0 ? (size_t)(f % 4) * 256 * QR_NGL
                                : (size_t)4 * 256 * QR_NGL + (size_t)(((f % 16) / 4) - 1) * 4096 + (size_t)(f % 4) * 1024) +
           (f ? 256 : 0);
}

// LV: digit-product levels kept (7 = all sixteen products; 5 = thirteen; 4 = ten -- the spec's LV).
// H2: waves that share a tile (1: a wave owns all of a tile's NBW column blocks; 2: NBW = 2 and the tile's two 32-column halves
// go to neighbouring waves, which stream the same fragments at the same time -- the second request is an L1 / L2 hit -- so that
// the accumulators of one wave (LV x NBW x 4 registers) leave room for four waves per SIMD).
template <int R, int NBW, int DEC_WAVES, int WPE, bool NT, int LV, int H2>
__global__ __launch_bounds__(DEC_WAVES * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void decode_q_ring_kernel(DecodeQArgs a) {
    constexpr int DEC_BLOCK = DEC_WAVES * 64;
    constexpr int F = QR_F;
    static_assert(F % R == 0 && R % 4 == 0, "the ring must close on an item boundary, in whole groups");
    static_assert(NBW * H2 <= 4 && DEC_WAVES % H2 == 0, "a pass holds four 16-column blocks");
    extern __shared__ __attribute__((aligned(16))) char qsmem[];
    char* Bimg = qsmem;                                                        // 4 k-steps x 16 KiB
    float* Mt = reinterpret_cast<float*>(Bimg + (size_t)QR_S * 16384);          // [64][12]
    int* be_s = reinterpret_cast<int*>(Mt + MAXB * 12);                         // [64]
    uint4* pay_s = reinterpret_cast<uint4*>(be_s + MAXB);                       // [DEC_WAVES][16]
    f32x4* park_s = reinterpret_cast<f32x4*>(pay_s + DEC_WAVES * 16);           // [DEC_WAVES][2][NBW][64]
    const int tid = threadIdx.x;
    const int nbatch = min(a.d.B - a.d.b0, MAXB);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_of(a.d.N);
    const int N = a.d.N;
    const int slot = wave / H2, hf = wave - slot * H2;
    const TileWalk tw = tile_walk(slot, DEC_WAVES / H2, (int)blockIdx.x, (int)gridDim.x);
    const int tile0 = tw.first, tstride = tw.stride;
    const unsigned voffA = (unsigned)lane * 16u;
    const char* Tb = a.tiles;

    i32x4 ring[R];
#define FRQ_REQ(slot_, f_, t_)                                                   \
    {                                                                            \
        const char* sb_ = Tb + (size_t)(t_) * QR_TB + qr_off(f_);                \
        FRQ_LD(ring[slot_], sb_, voffA);                                         \
    }
    {
        const int t0c = tile0 < tiles ? tile0 : 0;
#pragma unroll
        for (int f = 0; f < R; f++) FRQ_REQ(f, f, t0c)
    }
    q_copy_stage<DEC_BLOCK>(a.stage, qsmem, QR_S, tid);
    if (tile0 >= tiles) {
#pragma unroll
        for (int f = 0; f < R; f++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[f]));
        return;
    }
    int be[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) be[nb] = be_s[16 * (hf * NBW + nb) + (lane & 15)];
    uint4* pay_w = pay_s + wave * 16;
    f32x4* park_w = park_s + (size_t)wave * 2 * NBW * 64 + lane;
    const char* Bl = Bimg + (size_t)(hf * NBW) * 1024 + (size_t)lane * 16;

    for (int ct = tile0; ct < tiles; ct += tstride) {
        int nt = ct + tstride;
        if (nt >= tiles) nt = tile0;   // past the end: harmless re-request of a valid address, never consumed
        i32x4 acc[LV][NBW];
#pragma unroll
        for (int g = 0; g < F / 4; g++) {
            const int f0 = 4 * g, c = g / 4, sidx = g % 4;
            const int s = sidx == 0 ? QR_S - 1 : sidx - 1;
            if (sidx == 0) {
#pragma unroll
                for (int l = 0; l < LV; l++)
#pragma unroll
                    for (int nb = 0; nb < NBW; nb++) acc[l][nb] = (i32x4){0, 0, 0, 0};
            }
            q_ring_wait4<R>(ring[f0 % R], ring[(f0 + 1) % R], ring[(f0 + 2) % R], ring[(f0 + 3) % R]);
            if (g == 0) {
                if (lane >= 48) {
                    const i32x4 p = ring[0];
                    pay_w[lane - 48] = make_uint4((unsigned)p[0], (unsigned)p[1], (unsigned)p[2], (unsigned)p[3]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j >= LV) continue;   // (no kept product uses this parameter digit)
                i32x4 bf[NBW];
#pragma unroll
                for (int nb = 0; nb < NBW; nb++)
                    bf[nb] = *reinterpret_cast<const i32x4*>(Bl + (size_t)((s * 4 + j) * 4 + nb) * 1024);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (i + j >= LV) continue;   // kept products: i + j < LV
#pragma unroll
                    for (int nb = 0; nb < NBW; nb++)
                        acc[i + j][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[(f0 + i) % R], bf[nb], acc[i + j][nb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (f0 + i + R < F) FRQ_REQ((f0 + i) % R, f0 + i + R, ct)
                else FRQ_REQ((f0 + i) % R, f0 + i + R - F, nt)
            }
            if (sidx == 3) {   // coordinate c complete
                float mu4[4];
                int e4[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint4 row = pay_w[4 * (lane >> 4) + r];
                    mu4[r] = __uint_as_float(c == 0 ? row.x : c == 1 ? row.y : row.z);
                    e4[r] = (int)((row.w >> (10 * c)) & 1023u);
                }
                f32x4 vz[NBW];
                q_finish<LV, NBW>(acc, mu4, e4, be, vz);
                if (c < 2) {
#pragma unroll
                    for (int nb = 0; nb < NBW; nb++) park_w[(size_t)(c * NBW + nb) * 64] = vz[nb];
                } else {
                    f32x4 vx[NBW], vy[NBW];
#pragma unroll
                    for (int nb = 0; nb < NBW; nb++) {
                        vx[nb] = park_w[(size_t)nb * 64];
                        vy[nb] = park_w[(size_t)(NBW + nb) * 64];
                    }
                    decode_store<NBW>(a.d, vx, vy, vz, Mt, ct, hf, lane, nbatch, N);
                }
            }
        }
    }
#pragma unroll
    for (int f = 0; f < R; f++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[f]));
#undef FRQ_REQ
}

}  // namespace fr

// ---- host side ----------------------------------------------------------------------------------------------------------
size_t fr_packed_q_bytes(int N, int n_shape, int n_exp) {
    using namespace fr;
    const QShape qs = q_shape(n_shape, n_exp);
    return qs.hdr_bytes + (size_t)tiles_of(N) * qs.tile_bytes + 1024;   // + slack: the last short fragment's over-read
}

// Staging buffer of the decode (68 KiB for the model's shape): caller-owned, one per stream in flight -- launches on one
// stream are ordered, so its passes can share one buffer; different streams must not.
size_t fr_decode_q_workspace_bytes_impl(int n_shape, int n_exp) {
    return fr_decode_q_supported(n_shape, n_exp) ? fr::q_stage_bytes(fr::q_shape(n_shape, n_exp).S) : 0;
}

int fr_launch_pack_q(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp,
                     void* qimage, hipStream_t stream) {
    using namespace fr;
    if (N == 0) return FR_OK;
    const QShape qs = q_shape(n_shape, n_exp);
    if (qs.K > 512) return FR_OK;   // the Q30 kernels do not take this shape; the f32 image serves it
    int* ce = reinterpret_cast<int*>(qimage);
    char* tb = reinterpret_cast<char*>(qimage) + qs.hdr_bytes;
    const int KP = qs.S * 64;
    if (hipMemsetAsync(qimage, 0, fr_packed_q_bytes(N, n_shape, n_exp), stream) != hipSuccess) return FR_ERR_LAUNCH;
    if (qs.K > 0)   // the column maxima are collected in the header's own ce[] slots (no scratch allocation)
        hipLaunchKernelGGL(q_colmax_kernel, dim3(1024), dim3(256), 0, stream, pc_shape, pc_exp, N, n_shape, n_exp,
                           reinterpret_cast<unsigned*>(ce));
    hipLaunchKernelGGL(q_ce_kernel, dim3((KP + 255) / 256), dim3(256), 0, stream, qs.K, KP, ce);
    const long long NP = (long long)tiles_of(N) * TILE_V;
    hipLaunchKernelGGL(q_payload_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, stream, mu, pc_shape, pc_exp, N,
                       n_shape, n_exp, ce, tb, qs);
    hipLaunchKernelGGL(q_pack_kernel, dim3(4096), dim3(256), 0, stream, pc_shape, pc_exp, N, n_shape, n_exp, ce, tb, qs);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

bool fr_decode_q_supported(int n_shape, int n_exp) { return n_shape + n_exp <= 512; }

template <int NBW, int LV>
static int launch_q_generic(const fr::DecodeQArgs& a, size_t lds, int cus, hipStream_t stream) {
    static fr_lds_flags_t lds_ok[64];
    if (fr_allow_full_lds(reinterpret_cast<const void*>(&fr::decode_q_kernel<NBW, 8, LV>), lds_ok) != hipSuccess)
        return FR_ERR_LAUNCH;
    const int tiles = fr::tiles_of(a.d.N);
    const int grid = (int)min((long long)cus, (long long)(tiles + 7) / 8);
    hipLaunchKernelGGL((fr::decode_q_kernel<NBW, 8, LV>), dim3(grid), dim3(512), lds, stream, a);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

template <int R, int NBW, int WAVES, int WPE, int LV, int H2>
static int launch_q_ring(const fr::DecodeQArgs& a, int cus, hipStream_t stream) {
    static fr_lds_flags_t lds_ok[64];
    const void* k = reinterpret_cast<const void*>(&fr::decode_q_ring_kernel<R, NBW, WAVES, WPE, true, LV, H2>);
    if (fr_allow_full_lds(k, lds_ok) != hipSuccess) return FR_ERR_LAUNCH;
    const size_t lds = fr::q_stage_bytes(fr::QR_S) + (size_t)WAVES * 256 + (size_t)WAVES * 2 * NBW * 1024;
    const int tiles = fr::tiles_of(a.d.N);
    const int slots = WAVES / H2;
    const int grid = (int)min((long long)cus, (long long)(tiles + slots - 1) / slots);
    hipLaunchKernelGGL((fr::decode_q_ring_kernel<R, NBW, WAVES, WPE, true, LV, H2>), dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

// One pass (<= 64 columns, nbt live 16-column blocks) of the model's basis shape through the streaming schedule.
//   sched 0 (default): 8 waves per CU, a wave owns a tile's nbt column blocks (every basis byte is requested once on the chip),
//                      16-deep fragment ring, behind the staging launch (q_stage_kernel)
//   sched 1: 16 waves per CU at <= 128 registers (12 at <= 168 with all seven levels), 8-deep ring, a full pass cut into two
//            32-column halves per tile taken by neighbouring waves (the f32 kernel's arrangement; measured 2-7 us slower: the
//            partner's request of a non-temporal fragment misses L2 too often)
// Measured and not kept (profiles/round5_probes/r5b): the staging WITHOUT its launch -- every workgroup building the whole
// parameter image for itself (integer / fp32 forms of the same steps): equal one batch at a time, 2.5 us slower with two batches
// in flight; the kernel's first 64 workgroups staging one column each and all 256 waiting for them on an agent-scope counter
// (release / acquire hand-off): +6 us -- the hand-off costs more than the launch gap it removes; a decode that leaves half of
// every CU to the other stream's emit workgroups: +9 us in flight; write-through (sc0 sc1) stores of the vertex rows: 0, non-
// temporal ones: +4 us (the emit kernel's event bracket is ~2 us longer behind this kernel than behind the f32 one under every
// store policy, while rocprofv3 shows the same 40.5 us emit kernel: the difference sits in the kernel boundary).
template <int LV>
static int launch_q_pass(const fr::DecodeQArgs& a, int nbt, int sched, int cus, hipStream_t stream) {
    if (sched != 1 || nbt <= 2)
        return nbt == 1 ? launch_q_ring<16, 1, 8, 2, LV, 1>(a, cus, stream)
               : nbt == 2 ? launch_q_ring<16, 2, 8, 2, LV, 1>(a, cus, stream)
                          : launch_q_ring<16, 4, 8, 2, LV, 1>(a, cus, stream);
    // (all seven levels: 56 accumulators + the ring do not fit 128 registers -- twelve waves at <= 168)
    if constexpr (LV == 7) return launch_q_ring<8, 2, 12, 3, LV, 2>(a, cus, stream);
    else return launch_q_ring<8, 2, 16, 4, LV, 2>(a, cus, stream);
}
template <int LV>
static int launch_q_generic_pass(fr::DecodeQArgs& a, int nbt, size_t lds, int cus, hipStream_t stream) {
    // (three or four column blocks: two passes of two -- one wave holding four blocks' accumulators spills)
    int rc = nbt == 1 ? launch_q_generic<1, LV>(a, lds, cus, stream) : launch_q_generic<2, LV>(a, lds, cus, stream);
    if (rc == FR_OK && nbt > 2) {
        a.cb0 = 2;
        rc = launch_q_generic<2, LV>(a, lds, cus, stream);
        a.cb0 = 0;
    }
    return rc;
}

bool fr_decode_q_levels_ok(int levels) { return levels == 7 || levels == 5 || levels == 4; }

int fr_launch_decode_q(const float* params, const void* qimage, const float* R_override, int B, int N, int n_shape,
                       int n_exp, float im_size, float* vertex_proj, int pitch, int levels, void* workspace, size_t ws_bytes,
                       hipStream_t stream) {
    using namespace fr;
    if (!fr_decode_q_levels_ok(levels)) return FR_ERR_INVALID_ARG;
    if (B == 0 || N == 0) return FR_OK;
    DecodeQArgs a;
    a.qs = q_shape(n_shape, n_exp);
    a.ce = reinterpret_cast<const int*>(qimage);
    a.tiles = reinterpret_cast<const char*>(qimage) + a.qs.hdr_bytes;
    a.d.params = params;
    a.d.A = nullptr;
    a.d.mu_p = nullptr;
    a.d.R_override = R_override;
    a.d.out = vertex_proj;
    a.d.B = B; a.d.N = N; a.d.ns = n_shape; a.d.ne = n_exp;
    a.d.halves = 1;
    a.d.im_size = im_size;
    a.d.pitch = pitch;
    const size_t lds = q_stage_bytes(a.qs.S);
    if (lds > 160 * 1024) return FR_ERR_UNSUPPORTED;
    if (!workspace || ws_bytes < lds || ((uintptr_t)workspace & 15)) return FR_ERR_WORKSPACE;
    char* stage = reinterpret_cast<char*>(workspace);
    a.stage = stage;
    a.cb0 = 0;
    const int cus = fr_device_cu_count();
    const bool loop_env = opt(OPT_DECODE_IMPL) == 1;
    const int sched = opt(OPT_Q30_SCHED);
    for (int b0 = 0; b0 < B; b0 += MAXB) {
        a.d.b0 = b0;
        const int nbt = (min(B - b0, MAXB) + 15) / 16;
        const bool ring = !loop_env && a.qs.KB == QR_KB;
        hipLaunchKernelGGL(q_stage_kernel, dim3(MAXB), dim3(128), 0, stream, a, stage);
        int rc;
        if (ring)   // the model's basis shape: streaming schedule
            rc = levels == 7 ? launch_q_pass<7>(a, nbt, sched, cus, stream)
                 : levels == 5 ? launch_q_pass<5>(a, nbt, sched, cus, stream)
                               : launch_q_pass<4>(a, nbt, sched, cus, stream);
        else
            rc = levels == 7 ? launch_q_generic_pass<7>(a, nbt, lds, cus, stream)
                 : levels == 5 ? launch_q_generic_pass<5>(a, nbt, lds, cus, stream)
                               : launch_q_generic_pass<4>(a, nbt, lds, cus, stream);
        if (rc != FR_OK) return rc;
    }
    return FR_OK;
}
