// 3DMM decode for gfx950 (MI355X): 235-d parameters -> projected, y-flipped vertices [B,3,N].
//
// What it computes: FaceRecNet.vertices_transform of the reference (nets/network.py:140-171) with
// parse_pose_params (:253-263) and rotation_matrix_batch (:266-297) folded in -- the reference round-trips to
// the host through tf.py_func for the rotation (:150); here it is evaluated in-kernel in float64.
//
// How: the basis blend is a skinny fp32 GEMM  [3N x (199+29)] . [(199+29) x B]  that really is dense, so it
// runs on the matrix cores with the exact-f32 MFMA v_mfma_f32_16x16x4_f32 (bit-for-bit a k-ordered fmaf chain
// on gfx950, so the CPU oracle reproduces it exactly).  One wave owns 16 vertices x 3 coordinates x 64 batch
// columns: 12 accumulator tiles.  The basis is pre-packed ONCE (fr_decode_pack_basis) into MFMA A-fragment
// order so that every operand fetch is a fully coalesced 1 KiB global_load_dwordx4 per wave with no LDS
// staging; the parameters (B operand) sit in LDS in fragment order (one ds_read_b128 per k-step) and are
// shared by the workgroup's 8 waves.  Because a lane ends up holding x, y and z of the same (vertex, batch),
// the epilogue (+mu, 3x3 f*R transform, +t3d, y-flip) is fused and the result is written straight to
// [B,3,N] -- none of the [B,N,3] temporaries / transposes of network.py:153-169 exist.
// Bound: fp32 MFMA rate at B = 64 (72.8 MFLOP per face; 157 TF peak) and the 146 MB basis stream below that.
#include "fr_common.h"

namespace fr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int TILE_V = 16;      // vertices per wave tile (MFMA M)
constexpr int KGROUP = 16;      // k values per packed group (4 k-steps of 4)
constexpr int MAXB = 64;        // batch columns per pass (4 MFMA column blocks of 16)
constexpr int DEC_WAVES = 8;    // waves per workgroup
constexpr int DEC_BLOCK = DEC_WAVES * 64;

__host__ __device__ inline int groups_of(int n) { return (n + KGROUP - 1) / KGROUP; }
__host__ __device__ inline int tiles_of(int N) { return (N + TILE_V - 1) / TILE_V; }

// Packed image:  A[tile][g][c][lane] as float4 (4 consecutive k-steps), then mu[tile][c][16].
//   A[tile][g][c][lane][j] = basis_c[16*tile + (lane&15)][16*g' + 4*j + (lane>>4)]   (0 when out of range)
// where groups 0..GS-1 come from pc_shape and GS..GS+GE-1 from pc_exp.
__global__ __launch_bounds__(256) void pack_basis_kernel(const float* __restrict__ mu, const float* __restrict__ pc_shape,
                                                         const float* __restrict__ pc_exp, int N, int ns, int ne,
                                                         float4* __restrict__ A, float* __restrict__ mu_p) {
    const int GS = groups_of(ns), GE = groups_of(ne), G = GS + GE;
    const long long tiles = tiles_of(N);
    const long long totalA = tiles * G * 3 * 64;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < totalA; i += step) {
        int lane = (int)(i & 63);
        long long r = i >> 6;
        int c = (int)(r % 3);
        r /= 3;
        int g = (int)(r % G);
        long long tile = r / G;
        long long p = tile * TILE_V + (lane & 15);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (p < N) {
            size_t row = (size_t)c * N + p;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (g < GS) {
                    int k = KGROUP * g + 4 * j + (lane >> 4);
                    if (k < ns) v[j] = pc_shape[row * ns + k];
                } else {
                    int k = KGROUP * (g - GS) + 4 * j + (lane >> 4);
                    if (k < ne) v[j] = pc_exp[row * ne + k];
                }
            }
        }
        A[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    const long long totalM = tiles * 3 * TILE_V;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < totalM; i += step) {
        int q = (int)(i % TILE_V);
        long long r = i / TILE_V;
        int c = (int)(r % 3);
        long long tile = r / 3;
        long long p = tile * TILE_V + q;
        mu_p[i] = (p < N) ? mu[(size_t)c * N + p] : 0.f;
    }
}

struct DecodeArgs {
    const float* params;      // [B, 7+ns+ne]
    const float4* A;          // packed basis
    const float* mu_p;        // packed mu
    const float* R_override;  // [B,9] or null
    float* out;               // [B,3,N]
    int B, N, ns, ne;
    int b0;                   // first batch column of this pass
    float im_size;
};

// rotation in float64 exactly as network.py:276-290: R = (R_pitch . R_yaw) . R_roll, 3-term dots, no FMA.
__device__ __forceinline__ void mat3_mul(const double* A, const double* Bm, double* C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = (A[3 * i + 0] * Bm[0 + j] + A[3 * i + 1] * Bm[3 + j]) + A[3 * i + 2] * Bm[6 + j];
}
__device__ void rotation_f64(float phi_f, float gamma_f, float theta_f, float* R9) {
    double sp, cp, sy, cy, st, ct;
    sincos((double)phi_f, &sp, &cp);
    sincos((double)gamma_f, &sy, &cy);
    sincos((double)theta_f, &st, &ct);
    double Rp[9] = {1, 0, 0, 0, cp, sp, 0, -sp, cp};
    double Ry[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
    double Rr[9] = {ct, st, 0, -st, ct, 0, 0, 0, 1};
    double PY[9], Rm[9];
    mat3_mul(Rp, Ry, PY);
    mat3_mul(PY, Rr, Rm);
#pragma unroll
    for (int i = 0; i < 9; i++) R9[i] = (float)Rm[i];
}

// NB = number of 16-column batch blocks handled (1..4)
template <int NB>
__global__ __launch_bounds__(DEC_BLOCK) void decode_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int GS = groups_of(a.ns), GE = groups_of(a.ne), G = GS + GE;
    // LDS: Pl[(k)*16 + j] as float4 over nb (k = padded coefficient index, j = column within block), then Mt[64][12]
    f32x4* Pl = reinterpret_cast<f32x4*>(smem);
    float* Mt = smem + (size_t)G * KGROUP * 16 * 4;
    const int tid = threadIdx.x;
    const int nd = FR_N_POSE + a.ns + a.ne;
    const int nbatch = min(a.B - a.b0, 16 * NB);

    for (int i = tid; i < G * KGROUP * 16; i += DEC_BLOCK) {
        int j = i & 15, k = i >> 4;
        int g = k / KGROUP, kk = k - g * KGROUP;
        int col = -1;  // column in the parameter row
        if (g < GS) {
            int ks = g * KGROUP + kk;
            if (ks < a.ns) col = FR_N_POSE + ks;
        } else {
            int ke = (g - GS) * KGROUP + kk;
            if (ke < a.ne) col = FR_N_POSE + a.ns + ke;
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            int bb = 16 * nb + j;
            if (col >= 0 && bb < nbatch) v[nb] = a.params[(size_t)(a.b0 + bb) * nd + col];
        }
        Pl[i] = v;
    }
    if (tid < 64) {
        float m[12];
#pragma unroll
        for (int i = 0; i < 12; i++) m[i] = 0.f;
        if (tid < nbatch) {
            const float* pr = a.params + (size_t)(a.b0 + tid) * nd;
            float R[9];
            if (a.R_override) {
#pragma unroll
                for (int i = 0; i < 9; i++) R[i] = a.R_override[(size_t)(a.b0 + tid) * 9 + i];
            } else {
                rotation_f64(pr[0], pr[1], pr[2], R);
            }
            float f = pr[6];
#pragma unroll
            for (int i = 0; i < 9; i++) m[i] = f * R[i];  // f (.) R elementwise, network.py:163-165
            m[9] = pr[3];
            m[10] = pr[4];
            m[11] = pr[5];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) Mt[tid * 12 + i] = m[i];
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int tiles = tiles_of(a.N);
    const int N = a.N;
    for (int tile = blockIdx.x * DEC_WAVES + wave; tile < tiles; tile += gridDim.x * DEC_WAVES) {
        const float4* Ap = a.A + (size_t)tile * G * 3 * 64 + lane;
        f32x4 acc[3][NB];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++) acc[c][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- S = pc_shape . alpha : fmaf chain over k, from +0 ----
        for (int g = 0; g < GS; g++) {
            float4 a0 = Ap[(size_t)(g * 3 + 0) * 64];
            float4 a1 = Ap[(size_t)(g * 3 + 1) * 64];
            float4 a2 = Ap[(size_t)(g * 3 + 2) * 64];
            const float av[3][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}, {a2.x, a2.y, a2.z, a2.w}};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4 bq = Pl[(size_t)(g * 4 + j) * 64 + lane];
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
                        acc[c][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][j], bq[nb], acc[c][nb], 0, 0, 0);
            }
        }
        // v = mu + S   (network.py:159, first add)
        f32x4 v[3][NB];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            f32x4 m4 = *reinterpret_cast<const f32x4*>(a.mu_p + ((size_t)tile * 3 + c) * TILE_V + 4 * (lane >> 4));
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                v[c][nb] = m4 + acc[c][nb];
                acc[c][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // ---- E = pc_exp . beta ----
        for (int g = GS; g < G; g++) {
            float4 a0 = Ap[(size_t)(g * 3 + 0) * 64];
            float4 a1 = Ap[(size_t)(g * 3 + 1) * 64];
            float4 a2 = Ap[(size_t)(g * 3 + 2) * 64];
            const float av[3][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}, {a2.x, a2.y, a2.z, a2.w}};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4 bq = Pl[(size_t)(g * 4 + j) * 64 + lane];
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
                        acc[c][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][j], bq[nb], acc[c][nb], 0, 0, 0);
            }
        }
        // ---- fused epilogue: (+E), 3x3 (f.R) transform, +t3d, y flip, store [B,3,N] ----
        const int p0 = tile * TILE_V + 4 * (lane >> 4);  // first of this lane's 4 vertices
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const int bb = 16 * nb + (lane & 15);
            if (bb >= nbatch) continue;
            const float* m = Mt + bb * 12;
            f32x4 px, py, pz;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float vx = v[0][nb][r] + acc[0][nb][r];  // (mu + S) + E
                float vy = v[1][nb][r] + acc[1][nb][r];
                float vz = v[2][nb][r] + acc[2][nb][r];
                float qx = __builtin_fmaf(m[2], vz, __builtin_fmaf(m[1], vy, m[0] * vx)) + m[9];
                float qy = __builtin_fmaf(m[5], vz, __builtin_fmaf(m[4], vy, m[3] * vx)) + m[10];
                float qz = __builtin_fmaf(m[8], vz, __builtin_fmaf(m[7], vy, m[6] * vx)) + m[11];
                px[r] = qx;
                py[r] = (a.im_size - qy) - 1.0f;  // network.py:168
                pz[r] = qz;
            }
            float* ox = a.out + ((size_t)(a.b0 + bb) * 3) * N + p0;
            float* oy = ox + N;
            float* oz = oy + N;
            if (p0 + 3 < N) {
                *reinterpret_cast<f32x4u*>(ox) = px;
                *reinterpret_cast<f32x4u*>(oy) = py;
                *reinterpret_cast<f32x4u*>(oz) = pz;
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (p0 + r < N) {
                        ox[r] = px[r];
                        oy[r] = py[r];
                        oz[r] = pz[r];
                    }
                }
            }
        }
    }
}

}  // namespace fr

size_t fr_packed_basis_bytes(int N, int n_shape, int n_exp) {
    using namespace fr;
    size_t tiles = (size_t)tiles_of(N);
    size_t G = (size_t)groups_of(n_shape) + groups_of(n_exp);
    return tiles * G * 3 * 64 * sizeof(float4) + tiles * 3 * TILE_V * sizeof(float);
}

int fr_launch_pack_basis(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp,
                         void* packed, hipStream_t stream) {
    using namespace fr;
    if (N == 0) return FR_OK;
    size_t tiles = (size_t)tiles_of(N);
    size_t G = (size_t)groups_of(n_shape) + groups_of(n_exp);
    float4* A = reinterpret_cast<float4*>(packed);
    float* mu_p = reinterpret_cast<float*>(A + tiles * G * 3 * 64);
    hipLaunchKernelGGL(pack_basis_kernel, dim3(2048), dim3(256), 0, stream, mu, pc_shape, pc_exp, N, n_shape, n_exp, A,
                       mu_p);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

template <int NB>
static int launch_decode_nb(const fr::DecodeArgs& a, size_t lds, int grid, hipStream_t stream) {
    static unsigned char lds_ok[64];
    if (fr_allow_full_lds(reinterpret_cast<const void*>(&fr::decode_kernel<NB>), lds_ok) != hipSuccess)
        return FR_ERR_LAUNCH;
    hipLaunchKernelGGL(fr::decode_kernel<NB>, dim3(grid), dim3(fr::DEC_BLOCK), lds, stream, a);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

int fr_launch_decode(const float* params, const void* packed, const float* R_override, int B, int N, int n_shape,
                     int n_exp, float im_size, float* vertex_proj, hipStream_t stream) {
    using namespace fr;
    if (B == 0 || N == 0) return FR_OK;
    size_t tiles = (size_t)tiles_of(N);
    size_t G = (size_t)groups_of(n_shape) + groups_of(n_exp);
    size_t lds = G * KGROUP * 16 * sizeof(float4) + 64 * 12 * sizeof(float);
    if (lds > 160 * 1024) return FR_ERR_UNSUPPORTED;
    DecodeArgs a;
    a.params = params;
    a.A = reinterpret_cast<const float4*>(packed);
    a.mu_p = reinterpret_cast<const float*>(a.A + tiles * G * 3 * 64);
    a.R_override = R_override;
    a.out = vertex_proj;
    a.B = B; a.N = N; a.ns = n_shape; a.ne = n_exp;
    a.im_size = im_size;
    int grid = (int)((tiles + DEC_WAVES - 1) / DEC_WAVES);
    if (grid > 512) grid = 512;
    for (int b0 = 0; b0 < B; b0 += MAXB) {
        a.b0 = b0;
        int nb = (min(B - b0, MAXB) + 15) / 16;
        int rc;
        switch (nb) {
            case 1: rc = launch_decode_nb<1>(a, lds, grid, stream); break;
            case 2: rc = launch_decode_nb<2>(a, lds, grid, stream); break;
            case 3: rc = launch_decode_nb<3>(a, lds, grid, stream); break;
            default: rc = launch_decode_nb<4>(a, lds, grid, stream); break;
        }
        if (rc != FR_OK) return rc;
    }
    return FR_OK;
}
