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
#include "fr_decode_shared.h"

namespace fr {

constexpr int KGROUP = 16;      // k values per packed group (4 k-steps of 4)
// waves per persistent workgroup (one per CU): 16 (4 per SIMD, <= 128 VGPRs) for 32-column work items, 12 (3 per
// SIMD, <= 168 VGPRs) for 64-column work items

__host__ __device__ inline int groups_of(int n) { return (n + KGROUP - 1) / KGROUP; }

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

// One MFMA k-group: 4 k-steps of v_mfma_f32_16x16x4_f32 for NBW column blocks of one coordinate row-block.
// The A fragments of the 4 k-steps are passed by value (stay in registers); the B fragments come from LDS.
template <int NBW>
struct BFrag;
template <>
struct BFrag<1> { typedef float type; };
template <>
struct BFrag<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <>
struct BFrag<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int NBW>
__device__ __forceinline__ float bsel(const typename BFrag<NBW>::type& b, int i) {
    if constexpr (NBW == 1) return b;
    else return b[i];
}
// TR = false: D[vertex][batch] (a lane holds 4 consecutive vertices of one batch column).  TR = true: the two operands are
// swapped, D'[batch][vertex] = D^T with the same registers -- lane l then holds vertex l & 15 of batches 4 (l >> 4) + r --
// bit for bit the same sums (same k order, the products commute).
template <int NBW, bool TR = false>
__device__ __forceinline__ void mfma_step(float a, const typename BFrag<NBW>::type& bq, f32x4 (&acc)[NBW]) {
#pragma unroll
    for (int nb = 0; nb < NBW; nb++) {
        if constexpr (TR) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bsel<NBW>(bq, nb), a, acc[nb], 0, 0, 0);
        else acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bsel<NBW>(bq, nb), acc[nb], 0, 0, 0);
    }
}
// LDS image of the parameters (B operand): element (k, column j of 16, block nb) sits at k*16*NBW + pcol(k, j)*NBW + nb.
// For NBW <= 2 the column is XOR-swizzled with the k index: the prologue stores the image with 16 consecutive k per
// quarter-wave (stride 16*NBW floats = the same bank for every k un-swizzled: a 16-way conflict on every ds_write,
// PMC: 1.85 M conflict cycles per launch), swizzled they spread over 16 banks; a k-step's fragment read stays one
// conflict-free ds_read (the 16 lanes of a k row read a permutation of the row).
template <int NBW>
__device__ __forceinline__ int pcol(int k, int j) {
    if constexpr (NBW <= 2) return j ^ (k & 15);
    else return j;
}
// Plg points at the group's first k-step (wave-uniform base + hf); sw[j] is this lane's element offset for k-step j.
template <int NBW>
__device__ __forceinline__ typename BFrag<NBW>::type ldb(const float* Plg, const int (&sw)[4], int j) {
    return *reinterpret_cast<const typename BFrag<NBW>::type*>(Plg + sw[j]);
}
template <int NBW>
__device__ __forceinline__ void lane_swizzle(int lane, int (&sw)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) sw[j] = (j * 64 + (lane & 48) + pcol<NBW>(4 * j + (lane >> 4), lane & 15)) * NBW;
}
template <int NBW>
__device__ __forceinline__ void mfma_group3(float4 a0, float4 a1, float4 a2, const float* Plg, const int (&sw)[4],
                                            f32x4 (&acc0)[NBW], f32x4 (&acc1)[NBW], f32x4 (&acc2)[NBW]) {
    typename BFrag<NBW>::type bq = ldb<NBW>(Plg, sw, 0);
    mfma_step<NBW>(a0.x, bq, acc0); mfma_step<NBW>(a1.x, bq, acc1); mfma_step<NBW>(a2.x, bq, acc2);
    bq = ldb<NBW>(Plg, sw, 1);
    mfma_step<NBW>(a0.y, bq, acc0); mfma_step<NBW>(a1.y, bq, acc1); mfma_step<NBW>(a2.y, bq, acc2);
    bq = ldb<NBW>(Plg, sw, 2);
    mfma_step<NBW>(a0.z, bq, acc0); mfma_step<NBW>(a1.z, bq, acc1); mfma_step<NBW>(a2.z, bq, acc2);
    bq = ldb<NBW>(Plg, sw, 3);
    mfma_step<NBW>(a0.w, bq, acc0); mfma_step<NBW>(a1.w, bq, acc1); mfma_step<NBW>(a2.w, bq, acc2);
}
template <int NBW>
__device__ __forceinline__ void mfma_group1(float4 a0, const float* Plg, const int (&sw)[4], f32x4 (&acc0)[NBW]) {
    mfma_step<NBW>(a0.x, ldb<NBW>(Plg, sw, 0), acc0);
    mfma_step<NBW>(a0.y, ldb<NBW>(Plg, sw, 1), acc0);
    mfma_step<NBW>(a0.z, ldb<NBW>(Plg, sw, 2), acc0);
    mfma_step<NBW>(a0.w, ldb<NBW>(Plg, sw, 3), acc0);
}

// Persistent kernel: one workgroup per CU (16 waves).  The parameters are laid into LDS once per CU; then every wave
// walks work items (tile of 16 vertices, group of NBW batch-column blocks).  With B = 64 an item is half a tile
// (NBW = 2): 6,652 items over 1,024 SIMDs balance to within 8 % of the MFMA floor, where whole tiles (3,326) would leave
// a quarter of the matrix pipes idle in the last wave-round.  The two halves of a tile are taken by neighbouring waves
// of the same workgroup at the same time, so the second read of the tile's A fragments is an L1/L2 hit.
//
// LDS image of the parameters (B operand), per half hf: P[hf][k][j][NBW] floats -- lane l of k-step s reads the NBW
// consecutive floats at (s*64 + l)*NBW, i.e. one conflict-free ds_read_b32/b64 per k-step.
// Per-CU prologue shared by the decode kernels: the parameters go to LDS in B-fragment order, the pose to Mt = f.R | t3d.
template <int NBW, int NT_STAGE, int MB = 64>
__device__ __forceinline__ void stage_params(const DecodeArgs& a, float* smem, int GS, int GE, size_t half_floats, int tid,
                                             int nd, int nbatch) {
    // parameters -> LDS in B-fragment order: TPR threads per batch row, each walks the row with stride
    // TPR; loads are unconditional (clamped) and batched so they pipeline; padding slots and absent rows are 0.
    {
        static_assert(NT_STAGE >= MB, "at least one staging thread per batch row");
        constexpr int TPR = NT_STAGE / MB;  // threads per batch row (threads beyond TPR * MB have no row)
        const int bb = tid / TPR, sub = tid - bb * TPR;
        if (bb >= MB) return;
        const bool rowok = bb < nbatch;
        const float* prow = a.params + (size_t)(a.b0 + (rowok ? bb : 0)) * nd + FR_N_POSE;
        const int nbk = bb >> 4;
        float* dst = smem + (size_t)(nbk / NBW) * half_floats + (nbk % NBW);   // + k*16*NBW + pcol(k, bb & 15)*NBW
        const int jb = bb & 15;
        const int KS = GS * KGROUP, KE = GE * KGROUP;
        if (a.ns > 0) {
            for (int k0 = sub; k0 < KS; k0 += 8 * TPR) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = prow[min(k0 + TPR * u, a.ns - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + TPR * u;
                    if (k < KS) dst[(size_t)k * 16 * NBW + pcol<NBW>(k, jb) * NBW] = (rowok && k < a.ns) ? v[u] : 0.f;
                }
            }
        }
        if (a.ne > 0) {
            for (int k0 = sub; k0 < KE; k0 += 8 * TPR) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = prow[a.ns + min(k0 + TPR * u, a.ne - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + TPR * u;
                    if (k < KE)
                        dst[(size_t)(KS + k) * 16 * NBW + pcol<NBW>(KS + k, jb) * NBW] = (rowok && k < a.ne) ? v[u] : 0.f;
                }
            }
        }
    }
}
template <int NBW, int DEC_BLOCK, int MB = 64>
__device__ __forceinline__ void decode_prologue(const DecodeArgs& a, float* smem, float* Mt, double* SC, int GS, int GE,
                                                size_t half_floats, int tid, int nd, int nbatch) {
    static_assert(DEC_BLOCK % MB == 0 && DEC_BLOCK >= 3 * MB, "threads per batch row / sincos threads");
    stage_params<NBW, DEC_BLOCK, MB>(a, smem, GS, GE, half_floats, tid, nd, nbatch);
    pose_prologue<MB>(a, Mt, SC, tid, nd, nbatch);
}

template <int NBW, int DEC_WAVES>
__global__ __launch_bounds__(DEC_WAVES * 64) void decode_kernel(DecodeArgs a) {
    constexpr int DEC_BLOCK = DEC_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int GS = groups_of(a.ns), GE = groups_of(a.ne), G = GS + GE;
    const int KP = G * KGROUP;                      // padded coefficient count
    const size_t half_floats = (size_t)KP * 16 * NBW;
    float* Mt = smem + (size_t)KP * 16 * 4;          // [64][12] after the 4 column blocks' parameters
    double* SC = reinterpret_cast<double*>(Mt + 64 * 12);  // [64][3][2] sin/cos of the pose angles
    const int tid = threadIdx.x;
    const int nd = FR_N_POSE + a.ns + a.ne;
    const int nbatch = min(a.B - a.b0, 64);

    decode_prologue<NBW, DEC_BLOCK>(a, smem, Mt, SC, GS, GE, half_floats, tid, nd, nbatch);

    const int lane = tid & 63, wave = tid >> 6;
    const int tiles = tiles_of(a.N);
    const int N = a.N;
    // Work distribution: a tile's column-block halves go to neighbouring waves of one workgroup (they stream the same A
    // fragments at the same time, so the second read is an L1/L2 hit); tiles are dealt as tile_walk describes.
    const int H2 = a.halves;              // 1 or 2
    const int slots = DEC_WAVES / H2;     // tiles a workgroup works on at a time
    const int slot = wave / H2;
    const int hf = wave - slot * H2;
    const TileWalk tw = tile_walk(slot, slots, (int)blockIdx.x, (int)gridDim.x);
    for (int tile = tw.first; tile < tiles; tile += tw.stride) {
        const float4* Ap = a.A + (size_t)tile * G * 3 * 64 + lane;   // group g, coordinate c at Ap[(g*3+c)*64]
        const float* Pll = smem + (size_t)hf * half_floats;  // this half's parameter image; lane offsets in sw[]
        int sw[4];
        lane_swizzle<NBW>(lane, sw);
        f32x4 s0[NBW], s1[NBW], s2[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; nb++) s0[nb] = s1[nb] = s2[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- S = pc_shape . alpha : fmaf chain over k from +0.  The A fragments of the next two groups (6 KiB per
        //      wave, ~96 KiB per CU) are in flight while the current group's MFMAs issue ----
        float4 c0 = Ap[0], c1 = Ap[64], c2 = Ap[128];
        const int g1 = G > 1 ? 1 : 0;
        float4 d0 = Ap[(size_t)(g1 * 3 + 0) * 64], d1 = Ap[(size_t)(g1 * 3 + 1) * 64], d2 = Ap[(size_t)(g1 * 3 + 2) * 64];
        for (int g = 0; g < GS; g++) {
            const int gn = g + 2 < G ? g + 2 : G - 1;  // the last S groups prefetch the first E groups
            const float4 n0 = Ap[(size_t)(gn * 3 + 0) * 64], n1 = Ap[(size_t)(gn * 3 + 1) * 64],
                         n2 = Ap[(size_t)(gn * 3 + 2) * 64];
            mfma_group3<NBW>(c0, c1, c2, Pll + (size_t)g * 256 * NBW, sw, s0, s1, s2);
            c0 = d0; c1 = d1; c2 = d2;
            d0 = n0; d1 = n1; d2 = n2;
        }
        // v = mu + S   (network.py:159, first add)
        {
            const float* mp = a.mu_p + (size_t)tile * 3 * TILE_V + 4 * (lane >> 4);
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(mp);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(mp + TILE_V);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(mp + 2 * TILE_V);
#pragma unroll
            for (int nb = 0; nb < NBW; nb++) {
                s0[nb] = m0 + s0[nb];
                s1[nb] = m1 + s1[nb];
                s2[nb] = m2 + s2[nb];
            }
        }
        // ---- E = pc_exp . beta, one coordinate at a time (only NBW extra accumulators live), then
        //      v = (mu + S) + E.  Groups GS and GS+1 are already in c* / d*. ----
        {
            f32x4 e[NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; nb++) e[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (GE > 0) mfma_group1<NBW>(c0, Pll + (size_t)GS * 256 * NBW, sw, e);
            if (GE > 1) mfma_group1<NBW>(d0, Pll + (size_t)(GS + 1) * 256 * NBW, sw, e);
            for (int g = GS + 2; g < G; g++) mfma_group1<NBW>(Ap[(size_t)(g * 3 + 0) * 64], Pll + (size_t)g * 256 * NBW, sw, e);
#pragma unroll
            for (int nb = 0; nb < NBW; nb++) { s0[nb] = s0[nb] + e[nb]; e[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            if (GE > 0) mfma_group1<NBW>(c1, Pll + (size_t)GS * 256 * NBW, sw, e);
            if (GE > 1) mfma_group1<NBW>(d1, Pll + (size_t)(GS + 1) * 256 * NBW, sw, e);
            for (int g = GS + 2; g < G; g++) mfma_group1<NBW>(Ap[(size_t)(g * 3 + 1) * 64], Pll + (size_t)g * 256 * NBW, sw, e);
#pragma unroll
            for (int nb = 0; nb < NBW; nb++) { s1[nb] = s1[nb] + e[nb]; e[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            if (GE > 0) mfma_group1<NBW>(c2, Pll + (size_t)GS * 256 * NBW, sw, e);
            if (GE > 1) mfma_group1<NBW>(d2, Pll + (size_t)(GS + 1) * 256 * NBW, sw, e);
            for (int g = GS + 2; g < G; g++) mfma_group1<NBW>(Ap[(size_t)(g * 3 + 2) * 64], Pll + (size_t)g * 256 * NBW, sw, e);
#pragma unroll
            for (int nb = 0; nb < NBW; nb++) s2[nb] = s2[nb] + e[nb];
        }
        decode_store<NBW>(a, s0, s1, s2, Mt, tile, hf, lane, nbatch, N);
    }
}


// ---- streaming variant for a compile-time basis shape ---------------------------------------------------------------
// Same arithmetic as decode_kernel, different schedule.  With the group counts known at compile time (the model's
// 199 + 29 coefficients are 13 + 2 groups) an item is a fixed sequence of F = 3*(G+1) A fragments (g-major, coordinate
// minor; the last three are the item's mu values).  A wave treats ALL of its items as one fragment stream that runs
// through a ring of R float4 registers: fragment i is consumed from slot i % R and the slot is immediately re-requested
// with fragment i + R -- of the next item when the current one runs out, so the pipeline is never refilled per item
// and every fragment is requested R fragments (R/3 groups, ~8*R*NBW MFMAs of this wave and ~4x that of the SIMD)
// before its MFMAs issue.  F % R == 0 keeps the slot assignment identical for every item, so the item body is fully
// unrolled straight-line code with no cursor branches.  The requests are inline asm so that the compiler can neither
// reorder them nor drain them with its own vmcnt(0); the only waits are the counted s_waitcnt vmcnt(R-1) on the oldest
// slot (FR_RING_WAIT ties the slot register to the wait).
// Development hooks: the product instantiates the kernel with NoProbe only (every hook is an empty inline or a
// compile-time-false branch); tools/decode_probe.hip supplies policies with ablation bits
// (1 = no MFMA, 2 = requests hit 256 tiles, 4 = no prologue, 8 = no A requests, 16 = no LDS B reads, 32 = (almost) no
// stores, 64 = the stores issued as whole 128-byte lines: timing only) and in-kernel s_memtime stamps.
struct NoProbe {
    static constexpr int bits = 0;
    __device__ __forceinline__ void begin() {}
    template <int ID> __device__ __forceinline__ void stamp() {}
    __device__ __forceinline__ void item_begin() {}
    __device__ __forceinline__ void item_mfma_done() {}
    __device__ __forceinline__ void item_end() {}
    __device__ __forceinline__ void finish(int, int, int) {}
};
#define FR_RING_LD(dst, sbase, voff)                                                                   \
    {                                                                                                  \
        if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase)); \
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));       \
    }

template <int R>
__device__ __forceinline__ void ring_wait(f32x4& slot) {
    static_assert(R == 4 || R == 6 || R == 8, "ring size");  // (12 slots were measured too: slower, and they spill)
    if constexpr (R == 4) asm volatile("s_waitcnt vmcnt(3)" : "+v"(slot));
    else if constexpr (R == 6) asm volatile("s_waitcnt vmcnt(5)" : "+v"(slot));
    else asm volatile("s_waitcnt vmcnt(7)" : "+v"(slot));
}
// (Round 6, measured and rejected -- profiles/round6_probes/r6a: on gfx950 stores count in vmcnt with the loads, in issue
// order, so behind an epilogue of six store instructions vmcnt(R-1) on fragments 2 .. R-1 of the NEXT item waits for the
// acknowledgement of those stores, not for the fragment.  "Store-aware" waits -- vmcnt(R-1+6) on the first R fragments of
// every item but a wave's first -- keep the ring R deep across the epilogue and make the kernel SLOWER when the basis comes
// from HBM: 57.1 against 54.0 us behind a 512 MiB flush, 55.0 against 52.2 back to back, 45.0 against 44.4 with the basis
// resident in the Infinity Cache; in the bench 56.6 against 54.4 us.  The accidental pause after every epilogue is a
// throttle the memory side likes: fewer requests in flight while the stores drain.)
// WPE: waves per SIMD the register allocation is sized for (>= DEC_WAVES / 4).  A workgroup of fewer waves than that
// leaves VGPRs free on purpose: the 8-wave form (WPE = 4: <= 128 VGPRs, half the register file) lets the render
// kernels of the previous batch share the CU with the decode of the next one (pipeline.py, PipelinedPlan).
// NT: the basis stream is requested with the non-temporal hint (read once per launch by one CU pair of waves).
// (Measured and rejected in round 3, profiles/round3_probes/r3s12_decode_ab_balanced_tail.json: dealing the last, partial
// round of tile pairs as single tiles cut into four 16-column quarter items, one per SIMD -- 6.5 items on every SIMD instead
// of 7 on half of the CUs and 6 on the others: 55.1 vs 55.0 us, the matrix pipe's balance is not what bounds the kernel.)
// PRIO: static issue priorities for the four waves that share a SIMD (waves w, w+4, w+8, w+12 get 0..3): with equal
// priorities the sixteen waves of a CU advance in lock-step and reach their epilogues together; ranked, a SIMD tends to run
// them one after the other, which spreads the epilogues and their stores (decode -1 us dense, -2 us with aligned rows).
// TR: transposed accumulators (mfma_step) -- the epilogue then stores one dword per lane, and the sixteen lanes of a quarter
// wave write one batch row's 64 contiguous bytes (decode_store_tr), where the default form's store instruction scatters 16
// bytes per lane over sixteen rows.
template <int GS, int GE, int R, int NBW, int DEC_WAVES, int MB = 64, int WPE = DEC_WAVES / 4, bool NT = false, class PR = NoProbe,
          bool PRIO = false, bool TR = false>
__global__ __launch_bounds__(DEC_WAVES * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void decode_ring_kernel(DecodeArgs a) {
    PR pr;
    pr.begin();
    constexpr int DEC_BLOCK = DEC_WAVES * 64;
    constexpr int G = GS + GE;
    constexpr int F = 3 * (G + 1);  // fragments per item
    static_assert(F % R == 0, "the ring must close on an item boundary");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = G * KGROUP;
    constexpr size_t half_floats = (size_t)KP * 16 * NBW;
    float* Mt = smem + (size_t)KP * MB;              // [MB][12] after the MB / 16 column blocks' parameters
    double* SC = reinterpret_cast<double*>(Mt + MB * 12);  // [MB][3][2] sin/cos of the pose angles
    const int tid = threadIdx.x;
    const int nd = FR_N_POSE + a.ns + a.ne;
    const int nbatch = min(a.B - a.b0, MB);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_of(a.N);
    const int N = a.N;
    const int H2 = a.halves;
    const int slots = DEC_WAVES / H2;
    const int slot = wave / H2;
    const int hf = wave - slot * H2;
    const TileWalk tw = tile_walk(slot, slots, (int)blockIdx.x, (int)gridDim.x);
    const int tile0 = tw.first, tstride = tw.stride;
    const float* Pll = smem + (size_t)hf * half_floats;
    int sw[4];
    lane_swizzle<NBW>(lane, sw);
    const unsigned voffA = (unsigned)lane * 16u;         // this lane's 16 bytes of a 1 KiB A fragment
    // this lane's 4 vertices of a 64-byte mu row (TR: the aligned quad that holds ITS vertex, lane & 15)
    const unsigned voffM = TR ? (unsigned)(lane & 12) * 4u : (unsigned)(lane >> 4) * 16u;
    const char* Ab = reinterpret_cast<const char*>(a.A);
    const char* Mb = reinterpret_cast<const char*>(a.mu_p);
    constexpr size_t tile_bytes = (size_t)G * 3 * 1024;

    f32x4 ring[R];
    if constexpr (PR::bits & 8) {
#pragma unroll
        for (int f = 0; f < R; f++) ring[f] = (f32x4){1.f, 2.f, 3.f, 4.f};
    }
    // fragment f (0..F-1) of tile t
#define FR_REQ(slot_, f_, t_)                                                                        \
    {                                                                                                \
        if ((f_) < 3 * G) {                                                                          \
            const char* sb_ = Ab + (size_t)((PR::bits & 2) ? (t_) & 255 : (t_)) * tile_bytes + (size_t)(f_) * 1024; \
            FR_RING_LD(ring[slot_], sb_, voffA);                                                     \
        } else {                                                                                     \
            const char* sb_ = Mb + (size_t)(t_) * (3 * TILE_V * 4) + (size_t)((f_) - 3 * G) * (TILE_V * 4); \
            FR_RING_LD(ring[slot_], sb_, voffM);                                                     \
        }                                                                                            \
    }
    // the first R fragments are requested before the per-CU prologue, so their HBM latency overlaps the parameter staging
    // (a wave without work requests tile 0 and drops it)
    {
        const int t0c = tile0 < tiles ? tile0 : 0;
#pragma unroll
        for (int f = 0; f < R; f++) FR_REQ(f, f, t0c)
    }
    pr.template stamp<0>();
    // per-CU prologue: parameters into LDS (B-fragment order), float64 pose of the pass's columns -> Mt, two barriers.
    // (Measured and rejected in round 3, A/B in one process: the sincos evaluated by three waves WHILE the other thirteen
    // stage the parameters and the 3x3 assembly done by one wave behind a single barrier, Mt published through an LDS flag
    // that every wave checks before its first epilogue: +4.5 us per launch -- thirteen threads per parameter row stage
    // slower than sixteen, and the assembling wave crawls beside its SIMD's MFMA streams while every first epilogue waits.)
    if constexpr (!(PR::bits & 4)) decode_prologue<NBW, DEC_BLOCK, MB>(a, smem, Mt, SC, GS, GE, half_floats, tid, nd, nbatch);
    pr.template stamp<1>();
    if constexpr (PRIO) {
        const int pw = wave >> 2;
        if (pw == 1) __builtin_amdgcn_s_setprio(1);
        else if (pw == 2) __builtin_amdgcn_s_setprio(2);
        else if (pw >= 3) __builtin_amdgcn_s_setprio(3);
    }
    if (tile0 >= tiles) {
#pragma unroll
        for (int f = 0; f < R; f++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[f]));
        pr.finish((int)blockIdx.x, wave, 0);
        return;
    }
    f32x4 c[3][NBW], sv[3][NBW];
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
#pragma unroll
        for (int nb = 0; nb < NBW; nb++) c[cc][nb] = sv[cc][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // live k-steps (of 4) in the last shape / expression group
    const int ks_s = GS > 0 ? (a.ns - KGROUP * (GS - 1) + 3) / 4 : 4;
    const int ks_e = GE > 0 ? (a.ne - KGROUP * (GE - 1) + 3) / 4 : 4;
    for (int ct = tile0; ct < tiles; ct += tstride) {
        int nt = ct + tstride;  // tile whose fragments are requested once this item's run out
        if (nt >= tiles) nt = tile0;  // past the end: harmless re-request of a valid address, never consumed
        typename BFrag<NBW>::type bq[4];
        pr.item_begin();
#pragma unroll
        for (int f = 0; f < F; f++) {
            const int g = f / 3, cc = f % 3;
            if constexpr (!(PR::bits & 8)) ring_wait<R>(ring[f % R]);
            else asm volatile("" : "+v"(ring[f % R]));
            if (f < 3 * G) {
                if (cc == 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((PR::bits & 16) == 0 || (ct == tile0 && g == 0)) bq[j] = ldb<NBW>(Pll + (size_t)g * 256 * NBW, sw, j);
                    if (g == GS) {  // S finished: park it, restart the fmaf chain from +0 for E
#pragma unroll
                        for (int c2 = 0; c2 < 3; c2++)
#pragma unroll
                            for (int nb = 0; nb < NBW; nb++) {
                                sv[c2][nb] = c[c2][nb];
                                c[c2][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            }
                    }
                }
                const f32x4 af = ring[f % R];
                if constexpr (PR::bits & 1) {
                    asm volatile("" ::"v"(af), "v"(bq[0]), "v"(bq[3]));  // keeps the fragment and the LDS reads live
                } else {
                    // k-steps that hold only padding (199 = 12*16 + 7 leaves two of the last shape group's four) are
                    // skipped: their products are +0 (zero basis x zero parameter) and the chains, started from +0, are
                    // never -0, so adding them changes nothing -- and the oracle's chains do not contain them
                    const int live = (g == GS - 1) ? ks_s : (g == G - 1) ? ks_e : 4;
                    mfma_step<NBW, TR>(af.x, bq[0], c[cc]);
                    if (live > 1) mfma_step<NBW, TR>(af.y, bq[1], c[cc]);
                    if (live > 2) mfma_step<NBW, TR>(af.z, bq[2], c[cc]);
                    if (live > 3) mfma_step<NBW, TR>(af.w, bq[3], c[cc]);
                }
            } else {  // mu fragment of coordinate cc: v = (mu + S) + E   (network.py:159)
                f32x4 m = ring[f % R];
                if constexpr (TR) {   // one vertex per lane: its mu, for all four batches in the registers
                    const int sel = lane & 3;
                    const float mv = sel == 0 ? m.x : sel == 1 ? m.y : sel == 2 ? m.z : m.w;
                    m = (f32x4){mv, mv, mv, mv};
                }
#pragma unroll
                for (int nb = 0; nb < NBW; nb++) c[cc][nb] = (m + sv[cc][nb]) + c[cc][nb];
            }
            // re-request the slot: fragment f + R of this item, or of the next one
            if constexpr (!(PR::bits & 8)) {
                if (f + R < F) FR_REQ(f % R, f + R, ct)
                else FR_REQ(f % R, f + R - F, nt)
            }
        }
        pr.item_mfma_done();
        if ((PR::bits & 32) == 0 || c[0][0][0] + c[1][0][1] + c[2][NBW - 1][2] == 12345.678f) {
            if constexpr ((PR::bits & 64) != 0) decode_store_fullline_probe<NBW>(a, c[0], c[1], c[2], ct, hf, lane, nbatch, N);
            else if constexpr (TR) decode_store_tr<NBW>(a, c[0], c[1], c[2], Mt, ct, hf, lane, nbatch, N);
            else decode_store<NBW>(a, c[0], c[1], c[2], Mt, ct, hf, lane, nbatch, N);
        }
        pr.item_end();
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int nb = 0; nb < NBW; nb++) c[cc][nb] = sv[cc][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int f = 0; f < R; f++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[f]));
    pr.finish((int)blockIdx.x, wave, 1);
#undef FR_REQ
}

// Measurement hook (bench.py `clock_GHz_held`): every wave of a 16-wave workgroup per CU issues `iters` rounds of six independent
// v_mfma_f32_16x16x4_f32 -- the decode's matrix instruction at the decode's occupancy -- and lane 0 of each workgroup records how
// many shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) its loop took: their quotient is the clock the chip
// holds under that load (MI355X_MICROARCH.md, DVFS give-back (6)).  The stamps go to a buffer of their own.
__global__ __launch_bounds__(1024) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters, float a0, float b0) {
    f32x4 acc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = a0 + (float)threadIdx.x, b = b0 + 0.5f * (float)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) sink += acc[i][0] + acc[i][3];
    asm volatile("" ::"v"(sink));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

}  // namespace fr

// The packed buffer holds the f32 A-fragment image of this file (the Q30 digit image of fr_decode_q.hip is a separate,
// opt-in buffer: fr_decode_q30_*).
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

int fr_launch_clock_probe(unsigned long long* out, int blocks, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(fr::clock_probe_kernel, dim3(blocks), dim3(1024), 0, stream, out, iters, 1.0f, 2.0f);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

int fr_device_cu_count() {
    static std::atomic<int> cus[64];   // (per-device cache; racing threads store the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

template <int NBW, int WAVES>
static int launch_decode_nbw(const fr::DecodeArgs& a, size_t lds, int cus, size_t tiles, hipStream_t stream) {
    static fr_lds_flags_t lds_ok[64];
    if (fr_allow_full_lds(reinterpret_cast<const void*>(&fr::decode_kernel<NBW, WAVES>), lds_ok) != hipSuccess)
        return FR_ERR_LAUNCH;
    const int slots = WAVES / a.halves;
    const int grid = (int)min((long long)cus, (long long)(tiles + slots - 1) / slots);
    hipLaunchKernelGGL((fr::decode_kernel<NBW, WAVES>), dim3(grid), dim3(WAVES * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

template <int GS, int GE, int R, int NBW, int WAVES, int MB = 64, int WPE = WAVES / 4, bool NT = false, bool PRIO = false,
          bool TR = false>
static int launch_decode_ring(const fr::DecodeArgs& a, size_t lds, int cus, size_t tiles, hipStream_t stream) {
    static fr_lds_flags_t lds_ok[64];
    if (fr_allow_full_lds(reinterpret_cast<const void*>(&fr::decode_ring_kernel<GS, GE, R, NBW, WAVES, MB, WPE, NT, fr::NoProbe, PRIO, TR>),
                          lds_ok) != hipSuccess)
        return FR_ERR_LAUNCH;
    const int slots = WAVES / a.halves;
    const int grid = (int)min((long long)cus, (long long)(tiles + slots - 1) / slots);
    hipLaunchKernelGGL((fr::decode_ring_kernel<GS, GE, R, NBW, WAVES, MB, WPE, NT, fr::NoProbe, PRIO, TR>), dim3(grid), dim3(WAVES * 64),
                       lds, stream, a);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

int fr_launch_decode(const float* params, const void* packed, const float* R_override, int B, int N, int n_shape,
                     int n_exp, float im_size, float* vertex_proj, int pitch, hipStream_t stream) {
    using namespace fr;
    if (B == 0 || N == 0) return FR_OK;
    size_t tiles = (size_t)tiles_of(N);
    size_t G = (size_t)groups_of(n_shape) + groups_of(n_exp);
    size_t lds = G * KGROUP * 16 * sizeof(float4) + 64 * 12 * sizeof(float) + 64 * 3 * 2 * sizeof(double);
    if (lds > 160 * 1024) return FR_ERR_UNSUPPORTED;
    DecodeArgs a;
    a.params = params;
    a.A = reinterpret_cast<const float4*>(packed);
    a.mu_p = reinterpret_cast<const float*>(a.A + tiles * G * 3 * 64);
    a.R_override = R_override;
    a.out = vertex_proj;
    a.B = B; a.N = N; a.ns = n_shape; a.ne = n_exp;
    a.im_size = im_size;
    a.pitch = pitch;
    const int cus = fr_device_cu_count();
    const bool loop_env = opt(OPT_DECODE_IMPL) == 1;
    const bool wide_off = opt(OPT_DECODE_WIDE) == 0;
    const int nbw_env = opt(OPT_DECODE_NBW), waves_env = opt(OPT_DECODE_WAVES);
    const int nt_opt = opt(OPT_DECODE_NT);
    const bool ring_shape = !loop_env && groups_of(n_shape) == 13 && groups_of(n_exp) == 2;
    // 128 columns per pass (64-column items on 12 waves) when more than 64 remain: the basis is streamed once per 128
    // faces instead of once per 64 (102 vs 110 us at B = 128; FR_DECODE_WIDE=0 turns it off)
    const size_t lds_wide = G * KGROUP * 32 * sizeof(float4) + 128 * 12 * sizeof(float) + 128 * 3 * 2 * sizeof(double);
    for (int b0 = 0; b0 < B;) {
        a.b0 = b0;
        if (ring_shape && !wide_off && B - b0 > MAXB && lds_wide <= 160 * 1024) {
            a.halves = 2;  // two 64-column items per tile (32-column items on 16 waves measured the same)
            int rc = launch_decode_ring<13, 2, 8, 4, 12, 128>(a, lds_wide, cus, tiles, stream);
            if (rc != FR_OK) return rc;
            b0 += 2 * MAXB;
            continue;
        }
        const int nbt = (min(B - b0, MAXB) + 15) / 16;  // 16-column blocks in this pass (1..4)
        int nbw = nbt == 1 ? 1 : 2;                     // column blocks per work item
        if (nbw_env == 4 && nbt > 2) nbw = 4;
        if (nbw_env == 1) nbw = 1;                      // quarter-tile items: 13,304 at B = 64 (12.99 per SIMD)
        a.halves = (nbt + nbw - 1) / nbw;
        // the model's own basis shape (199 + 29 coefficients = 13 + 2 groups) takes the fully unrolled ring schedule
        const bool ring = ring_shape && nbw <= 2;
        int rc;
        if (ring && nbw == 1) rc = launch_decode_ring<13, 2, 8, 1, 16, 64, 4, true>(a, lds, cus, tiles, stream);
        else if (ring && waves_env == 8) rc = launch_decode_ring<13, 2, 8, 2, 8, 64, 4>(a, lds, cus, tiles, stream);
        else if (ring && (nt_opt == 0 || (nt_opt < 0 && B - b0 < MAXB)))
            // default cache policy for the basis stream of a pass below 64 faces: its working set (153 MB of basis + 3.7 MB of
            // vertices, records and planes per face) leaves the basis a share of the 256 MiB Infinity Cache worth having --
            // one batch at a time +4.1 % at 32 faces and +2.0 % at 48, two in flight +3.1 % at 48, nothing either way at 16
            // (profiles/round6_probes/r6a); at 64 faces the non-temporal hint below wins (r2f, r5m)
            rc = launch_decode_ring<13, 2, 8, 2, 16, 64, 4, false, true>(a, lds, cus, tiles, stream);
        else if (ring && opt(OPT_DECODE_STORE) == 1)
            rc = launch_decode_ring<13, 2, 8, 2, 16, 64, 4, true, true, true>(a, lds, cus, tiles, stream);  // A/B knob: transposed accumulators
        else if (ring)
            // the basis stream carries the non-temporal hint: it is read once per launch, and keeping its 153 MB out of
            // the way leaves the L2 / Infinity Cache to the vertices and hit records the render kernels re-read (measured
            // in the pipeline: decode +2 us, emit -1.5 us, resolve -5 us per 64-face step)
            rc = launch_decode_ring<13, 2, 8, 2, 16, 64, 4, true, true>(a, lds, cus, tiles, stream);
        else
            rc = nbw == 1   ? launch_decode_nbw<1, 16>(a, lds, cus, tiles, stream)
                 : nbw == 2 ? launch_decode_nbw<2, 16>(a, lds, cus, tiles, stream)
                            : launch_decode_nbw<4, 12>(a, lds, cus, tiles, stream);
        if (rc != FR_OK) return rc;
        b0 += MAXB;
    }
    return FR_OK;
}
