// Pieces shared by the two decode arithmetic paths (fr_decode.hip: exact-f32 MFMA chain; fr_decode_q.hip: Q30 fixed
// point on the int8 MFMA): argument block, in-kernel float64 rotation, work distribution, pose prologue, fused epilogue.
#pragma once
#include "fr_common.h"

namespace fr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int TILE_V = 16;      // vertices per wave tile (MFMA M)
constexpr int MAXB = 64;        // batch columns per pass (4 MFMA column blocks of 16)
__host__ __device__ inline int tiles_of(int N) { return (N + TILE_V - 1) / TILE_V; }

struct DecodeArgs {
    const float* params;      // [B, 7+ns+ne]
    const float4* A;          // packed basis
    const float* mu_p;        // packed mu
    const float* R_override;  // [B,9] or null
    float* out;               // [B,3,N]
    int B, N, ns, ne;
    int b0;                   // first batch column of this pass
    int halves;               // column-block groups per tile: a work item is (tile, half)
    float im_size;
    int pitch;                // floats between consecutive coordinate rows of `out` (>= N; N for the dense [B,3,N] tensor of
                              // the op surface; the fused decode -> render entry point pads it to a multiple of 32 so that
                              // every 16-vertex tile piece is an aligned 64-byte half of a 128-byte line)
};

// rotation in float64 exactly as network.py:276-290: R = (R_pitch . R_yaw) . R_roll, 3-term dots, no FMA.
__device__ __forceinline__ void mat3_mul(const double* A, const double* Bm, double* C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = (A[3 * i + 0] * Bm[0 + j] + A[3 * i + 1] * Bm[3 + j]) + A[3 * i + 2] * Bm[6 + j];
}
__device__ __forceinline__ void rotation_from_sincos(double sp, double cp, double sy, double cy, double st, double ct,
                                                     float* R9) {
    double Rp[9] = {1, 0, 0, 0, cp, sp, 0, -sp, cp};
    double Ry[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
    double Rr[9] = {ct, st, 0, -st, ct, 0, 0, 0, 1};
    double PY[9], Rm[9];
    mat3_mul(Rp, Ry, PY);
    mat3_mul(PY, Rr, Rm);
#pragma unroll
    for (int i = 0; i < 9; i++) R9[i] = (float)Rm[i];
}

// Work distribution shared by the decode kernels.  A workgroup works on `slots` tiles at a time (slot = wave / halves).
// Tiles are dealt in adjacent PAIRS: slots 2s and 2s+1 of a workgroup take tiles 2P and 2P+1, so the two 64-byte pieces
// a pair writes into each output row come from one CU at about the same time; pair P goes to workgroup perm(P % grid) of
// round P / grid, where perm keeps consecutive pairs on the same XCD (workgroup b runs on XCD b % 8), so the partial
// cache lines at the seams still meet in one L2.  Dealing pairs round-robin keeps the last, partial round to at most one
// extra pair per CU.
struct TileWalk {
    int first, stride;
};
__device__ __forceinline__ TileWalk tile_walk(int slot, int slots, int b, int grid) {
    const int pb = (grid & 7) == 0 ? (b & 7) * (grid >> 3) + (b >> 3) : b;
    TileWalk w;
    if (slots & 1) {  // odd slot count (not used by the launchers): plain round-robin over tiles
        w.first = slot * grid + pb;
        w.stride = slots * grid;
    } else {
        w.first = 2 * ((slot >> 1) * grid + pb) + (slot & 1);
        w.stride = slots * grid;  // = 2 * (slots / 2) * grid
    }
    return w;
}

// (Round 6, measured and rejected -- profiles/round6_probes/r6a, r6b: with 3,326 tiles on 256 CUs x 8 slots the second, partial
// round gives the CUs of XCDs 0-3 three more tile pairs and those of XCDs 4-7 two -- 7 against 6 half-tile items on every SIMD,
// and the stamped kernel shows XCDs 4-7 idle for the last 6 us of the launch.  Dealing the partial round evenly -- 13 whole
// tiles on every CU of every XCD -- makes all eight XCDs exit together and the kernel no faster: 52.4 against 52.6 us back to
// back, 56.3 against 54.4 behind a flush, the serial step 112.1 against 111.9, the in-flight step +0.3 ... +2.3 and the Q30
// in-flight step +3: the duration is a chip-wide rate of the memory side, and in flight the XCDs that finish early take the other
// batch's workgroups early.)

// Pose part of the per-CU prologue: Mt[b] = f.R | t3d for the pass's MB columns (float64 rotation, network.py:266-297).
// Needs >= 3*MB threads; ends with a workgroup barrier (so it also publishes whatever the caller staged before it).
template <int MB>
__device__ __forceinline__ void pose_prologue(const DecodeArgs& a, float* Mt, double* SC, int tid, int nd, int nbatch) {
    // pose: the 3*MB float64 sincos evaluations are spread over 3*MB threads, then MB threads assemble f*R and t
    if (tid < 3 * MB && !a.R_override) {
        const int b = tid / 3, ang = tid - 3 * b;
        double sn = 0.0, cs = 1.0;
        if (b < nbatch) sincos((double)a.params[(size_t)(a.b0 + b) * nd + ang], &sn, &cs);
        SC[(b * 3 + ang) * 2 + 0] = sn;
        SC[(b * 3 + ang) * 2 + 1] = cs;
    }
    __syncthreads();
    if (tid < MB) {
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
                const double* sc = SC + tid * 6;
                rotation_from_sincos(sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], R);
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

}

// Fused epilogue of one work item: 3x3 (f.R) transform, +t3d, y flip, store to [B,3,pitch] rows.
// (Measured and not adopted, profiles/round3_probes: regrouping the lanes with ds_bpermute so that four ADJACENT lanes hold
// one row's 64 contiguous bytes: no change; a tile-blocked hand-off layout [N/32][3][B][32]: -3 us, [N/16][3][B][16] with
// lane-linear 1 KiB stores: -5 us for the decode, at the price of half-used cache lines in the emit kernel's gathers.)
template <int NBW>
__device__ __forceinline__ void decode_store(const DecodeArgs& a, const f32x4 (&s0)[NBW], const f32x4 (&s1)[NBW],
                                             const f32x4 (&s2)[NBW], const float* Mt, int tile, int hf, int lane,
                                             int nbatch, int N) {
    const int p0v = tile * TILE_V + 4 * (lane >> 4);  // first of this lane's 4 vertices
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) {
        const int bb = 16 * (hf * NBW + nb) + (lane & 15);
        if (bb >= nbatch) continue;
        const float* m = Mt + bb * 12;
        f32x4 px, py, pz;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float vx = s0[nb][r], vy = s1[nb][r], vz = s2[nb][r];
            const float qx = __builtin_fmaf(m[2], vz, __builtin_fmaf(m[1], vy, m[0] * vx)) + m[9];
            const float qy = __builtin_fmaf(m[5], vz, __builtin_fmaf(m[4], vy, m[3] * vx)) + m[10];
            const float qz = __builtin_fmaf(m[8], vz, __builtin_fmaf(m[7], vy, m[6] * vx)) + m[11];
            px[r] = qx;
            py[r] = (a.im_size - qy) - 1.0f;  // network.py:168
            pz[r] = qz;
        }
        float* ox = a.out + ((size_t)(a.b0 + bb) * 3) * a.pitch + p0v;
        float* oy = ox + a.pitch;
        float* oz = oy + a.pitch;
        if (p0v + 3 < N) {
            *reinterpret_cast<f32x4u*>(ox) = px;
            *reinterpret_cast<f32x4u*>(oy) = py;
            *reinterpret_cast<f32x4u*>(oz) = pz;
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (p0v + r < N) {
                    ox[r] = px[r];
                    oy[r] = py[r];
                    oz[r] = pz[r];
                }
            }
        }
    }
}

// The same epilogue for TRANSPOSED accumulators (mfma_step<NBW, true>): register r of lane l is (vertex l & 15, batch
// 16 * block + 4 * (l >> 4) + r).  One dword per lane per store: the sixteen lanes of a quarter wave write one row's 64
// contiguous bytes.
template <int NBW>
__device__ __forceinline__ void decode_store_tr(const DecodeArgs& a, const f32x4 (&s0)[NBW], const f32x4 (&s1)[NBW],
                                                const f32x4 (&s2)[NBW], const float* Mt, int tile, int hf, int lane,
                                                int nbatch, int N) {
    const int p = tile * TILE_V + (lane & 15);
    if (p >= N) return;
    const int q = lane >> 4;
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int bb = 16 * (hf * NBW + nb) + 4 * q + r;
            if (bb >= nbatch) continue;
            const float* m = Mt + bb * 12;
            const float vx = s0[nb][r], vy = s1[nb][r], vz = s2[nb][r];
            const float qx = __builtin_fmaf(m[2], vz, __builtin_fmaf(m[1], vy, m[0] * vx)) + m[9];
            const float qy = __builtin_fmaf(m[5], vz, __builtin_fmaf(m[4], vy, m[3] * vx)) + m[10];
            const float qz = __builtin_fmaf(m[8], vz, __builtin_fmaf(m[7], vy, m[6] * vx)) + m[11];
            float* ox = a.out + ((size_t)(a.b0 + bb) * 3) * a.pitch + p;
            ox[0] = qx;
            ox[a.pitch] = (a.im_size - qy) - 1.0f;  // network.py:168
            ox[2 * (size_t)a.pitch] = qz;
        }
    }
}

// PROBE ONLY (tools/decode_probe.hip, ablation bit 64; the product never instantiates it): the item's 6 KiB leave as 24
// one-dword-per-lane stores that each write TWO WHOLE 128-byte lines -- the store stream a transposed v_mfma_f32_32x32x2_f32
// item (lane = vertex of 32, registers = batches) would issue.  The two tiles of a pair take alternate (batch, coordinate)
// rows, so every line of the output is written exactly once; the VALUES land in the wrong places: timing only.
template <int NBW>
__device__ __forceinline__ void decode_store_fullline_probe(const DecodeArgs& a, const f32x4 (&s0)[NBW], const f32x4 (&s1)[NBW],
                                                            const f32x4 (&s2)[NBW], int tile, int hf, int lane, int nbatch, int N) {
    // rows of the output are indexed (batch * 3 + coordinate) linearly: the item's 96 rows are 96 consecutive row indices, the
    // tile of parity `par` takes the row pairs p = 2 s + par, the half wave `half` row 2 p + half -- one pointer per lane, a
    // constant stride per store, no index arithmetic beside the stores (a first version derived batch / coordinate per store
    // and measured its own divisions: 129 us)
    const int par = tile & 1, half = lane >> 5;
    const size_t v = (size_t)(tile & ~1) * TILE_V + (lane & 31);   // the tile pair's 32 vertices
    if (v >= (size_t)N || nbatch < MAXB) return;                   // (the probe runs full 64-column passes only)
    float* p = a.out + ((size_t)(a.b0 * 3 + 96 * hf) + 2 * par + half) * a.pitch + v;
    const size_t stride = 4 * (size_t)a.pitch;
#pragma unroll
    for (int s = 0; s < 24; s++) {
        const int cc = s % 3;
        const float val = (cc == 0 ? s0 : cc == 1 ? s1 : s2)[(s >> 2) % NBW][s & 3];
        p[(size_t)s * stride] = val;
    }
}

}  // namespace fr
