// render_depth forward / backward for gfx950 (MI355X).
//
// What it computes: the CPU functor of the reference, rendering_layer/ops_src/render_depth_op.cc:132-322
// (forward) and :325-368 (backward) -- NOT the reference's CUDA kernels (render_depth_op.cu.cc:176-237 race
// on the z-test and are not the numerical spec).
//
// How (MI355X-first, nothing like the reference's three-kernel + 705 MB fp64 scratch pipeline).  At BFM density
// triangles are sub-pixel (about half have no pixel centre in their bbox, the rest test ~1 pixel), so the work is
// "gather 9 floats, maybe emit one hit", then "per-pixel max":
//
//   raster_emit_kernel   512-triangle segments (13 k workgroups).  Phase A: each thread takes two triangles, loads
//                        the float ids (shared by all faces), gathers the nine vertex floats with 32-bit offsets and
//                        applies the reference's bbox / whole-triangle-reject rules; survivors (~half) are compacted
//                        into an LDS queue.  Phase B, on dense waves: fp32 centroid depth, packed key, the fp64
//                        barycentric test on the bbox's pixel centres, the un-normalised normal.  A triangle whose bbox
//                        fits an 8x4 window inside one screen strip emits ONE 16-byte record {key, x0|y0, 32-bit hit
//                        mask} (+ its normal in the companion slot); anything else (large or strip-straddling) emits a
//                        "big" record that the resolver rasterises itself.  Phase C counting-sorts the records by
//                        screen strip inside the segment through LDS counters -- no global atomics, no global counters
//                        to zero, fixed capacity.
//   resolve_write_kernel one workgroup per (face, strip-of-rows) screen bin; the bin's 64-bit keys live in LDS
//                        (rows*W*8 B).  It pulls only its own bucket of every segment as one flat list (bucket sizes
//                        prefix-summed in LDS, all record loads in flight), resolves with ds_max_u64 on
//                        key = orderable(h) << 32 | ~tri  (max == "largest h, ties to the lowest index": the serial
//                        semantics, order independent => deterministic, race free); a second pass over the same
//                        records stores the winners' normals; then depth / tri_ind / texture (per-triangle means from a
//                        table built once per batch) and the background normals stream out coalesced.
//   render_strip_kernel  the first-generation path (every bin scans every triangle); kept as the general fallback
//                        for shapes the binned path does not cover and for A/B runs (FR_RENDER_IMPL=scan).
//
// Bound: HBM -- 1.28 MB of output planes per face is the dominant algorithmic traffic.
#include "fr_common.h"

namespace fr {

struct RenderArgs {
    const float* vertex;   // [B,3,vpitch] rows (vpitch == nver: the dense tensor of the op surface)
    long long vpitch;      // floats between consecutive coordinate rows of `vertex`
    const float* tri;      // [3,ntri]
    const float* texture;  // [tex_batch,3,nver]
    float* depth;          // [B,H,W,1]
    float* tex_img;        // [B,H,W,3]
    float* normal;         // [B,H,W,3]
    float* tri_ind;        // [B,H,W,1]
    int B, nver, ntri, H, W;
    int rows;              // rows per strip
    int strips;            // strips per face
    long long tex_stride;  // 0 (shared texture) or 3*nver
    // binned path workspace
    uint4* recs;           // [B][nseg][SEG][2] 16-byte records, each followed by its un-normalised normal (xyz as a float4): the
                           // resolver reads both, so they share a 32-byte half line
    uint16_t* segoff;      // [B][OFF_STRIDE][nseg] bucket offsets, bucket-major: the resolver's thread = segment reads of one bucket
                           // are one coalesced line (segment-major, every thread touched its own 128-byte line)
    float4* tritex_ws;     // [tex_batch][ntri] per-triangle texture mean (one copy when the texture is shared)
    int nseg;
    // fused rendering-layer outputs (fr_rendering_layer_forward; network.py:185-199 folded into the resolver)
    const float* im_gray;  // [B,H,W,1]
    float* net_in;         // [B,H,W,7] = [mask*im | pncc x3 | normalised normal x3]  (network.py:122)
    float* depth_img;      // [B,H,W,1] = max(depth, 1e-6)
    float wm1, hm1;        // (float)(W-1), (float)(H-1)
    uint32_t rows_magic;   // ceil(2^32 / rows): y / rows == umulhi(y, magic) for y, rows < 2^16
    int resolve_opt;       // A/B knob FR_RESOLVE_OPT: 1 = single-trip bins keep records + normals in registers
    int use_filter;        // A/B knob FR_EMIT_FILTER: 0 sends every pixel through the fp64 sequence
    const int4* tri4;      // [nseg*SEG] pre-validated triangles {4*p1, 4*p2, 4*p3, valid} by id, [1] header, [nseg*SEG] the same in
                           // the emit kernel's lane order, .w = valid | local index << 1 (pack_tri_kernel)
    uint32_t nseg_magic;   // ceil(2^32 / nseg): lid / nseg == umulhi(lid, magic) (launcher checks the range)
};

constexpr int SEG = 504;         // triangles (and record capacity) per segment: 504 * 40 B of queue + counters <= 20 KiB,
                                 // so eight emit workgroups share a CU's 160 KiB of LDS (8 waves per SIMD)
constexpr int OFF_STRIDE = 64;   // u16 offsets per segment (strips + 1 <= 64)
constexpr int MAX_STRIPS = OFF_STRIDE / 2;  // buckets: 0 = big, 1+2s = strip s, 2+2s = windows straddling strips s / s+1
constexpr int SMALL_W = 8, SMALL_H = 4;  // hit-mask window: bit = dy*8 + dx

// XCD-aware block remap (bijective for any grid): blocks that share blockIdx%8 share an XCD/L2, so give each
// XCD a contiguous run of bins -- all work on a face then meets that face's vertices (and records) in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// three consecutive floats as ONE 12-byte store (global_store_dwordx3): a scattered per-pixel normal costs one
// cache-line transaction per lane instead of three, and a lane-contiguous plane write one instruction instead of three
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void store3(float* p, float x, float y, float z) {
    *reinterpret_cast<f32x3u*>(p) = (f32x3u){x, y, z};
}


// One triangle against one strip [r0, r1): the reference's per-triangle body (render_depth_op.cc:201-219, 263-316)
// with the depth test replaced by the packed-key LDS max.
// WINNER = false: resolve (LDS max).  WINNER = true: second pass -- the pixels whose resolved key is this triangle's
// get `nval` stored at `nplane + nstride * pixel` (the strip's normal plane, or the normal channels of the fused
// 7-channel output).
template <bool WINNER = false>
__device__ __forceinline__ void raster_triangle_into_strip(int t, const float* __restrict__ tri,
                                                           const float* __restrict__ vx, const float* __restrict__ vy,
                                                           const float* __restrict__ vz, int nver, int ntri, int H,
                                                           int W, int r0, int r1, unsigned long long* keys,
                                                           float* nplane = nullptr, float4 nval = float4(),
                                                           int nstride = 3) {
    // vertex ids: (int) truncation of float-stored indices, render_depth_op.cc:204-206
    int p1 = f2i_x86(tri[t]);
    int p2 = f2i_x86(tri[(size_t)ntri + t]);
    int p3 = f2i_x86(tri[2 * (size_t)ntri + t]);
    if ((unsigned)p1 >= (unsigned)nver || (unsigned)p2 >= (unsigned)nver || (unsigned)p3 >= (unsigned)nver)
        return;  // deviation 3: the reference would read out of bounds
    float y1 = vy[p1], y2 = vy[p2], y3 = vy[p3];
    int y_min = f2i_x86(ceilf(mn(mn(y1, y2), y3)));   // render_depth_op.cc:279-280
    int y_max = f2i_x86(floorf(mx(mx(y1, y2), y3)));
    if (y_max < y_min || y_max > H - 1 || y_min < 0) return;  // part of the whole-triangle reject, :282
    int ya = max(y_min, r0), yb = min(y_max, r1 - 1);
    if (ya > yb) return;
    float x1 = vx[p1], x2 = vx[p2], x3 = vx[p3];
    int x_min = f2i_x86(ceilf(mn(mn(x1, x2), x3)));   // :276-277
    int x_max = f2i_x86(floorf(mx(mx(x1, x2), x3)));
    if (x_max < x_min || x_max > W - 1 || x_min < 0) return;  // :282
    float h = ((vz[p1] + vz[p2]) + vz[p3]) / 3.0f;    // centroid depth in fp32, :217
    h = h + 0.0f;                                     // -0 -> +0 (the serial code treats them as equal)
    if (!(h > bg_depth())) return;                    // NaN or <= background can never pass 'depth < h' (:295)
    const unsigned long long key = make_key(h, t);
    const TriSetup ts = tri_setup(x1, y1, x2, y2, x3, y3);
    // one flat, rolled loop over the bbox (not y / x nests): keeps this rarely taken path from setting the resolver's
    // register budget
    const int bw = x_max - x_min + 1;
    const long long npx = (long long)bw * (yb - ya + 1);
    int x = x_min, y = ya;
#pragma clang loop unroll(disable)
    for (long long k = 0; k < npx; k++) {
        if (point_in_tri(ts, x, y)) {
            unsigned long long* kp = keys + (size_t)(y - r0) * W + x;
            if constexpr (!WINNER) {
                atomicMax(kp, key);
            } else if (*kp == key) {
                float* np = nplane + (size_t)nstride * ((size_t)(y - r0) * W + x);
                store3(np, nval.x, nval.y, nval.z);
            }
        }
        if (++x > x_max) { x = x_min; y++; }
    }
}

// Caller-side post-processing of the reference's rendering_layer (nets/network.py:185-199), fused variant.
// normal: flip to n_z >= 0, divide by sqrt(|n|^2) + 1e-6 with |n|^2 <= 1e-6 replaced by 1 (:188-192)
__device__ __forceinline__ float4 post_normal(float4 n) {
    float nx = n.x, ny = n.y, nz = n.z;
    if (nz < 0.0f) { nx = -1.0f * nx; ny = -1.0f * ny; nz = -1.0f * nz; }
    float mag = (nx * nx + ny * ny) + nz * nz;
    mag = (mag > 1e-6f) ? mag : 1.0f;
    const float d = sqrtf(mag) + 1e-6f;
    return make_float4(nx / d, ny / d, nz / d, 0.0f);
}
__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 1e-6f), 1.0f); }  // tf.clip_by_value(v,1e-6,1)

struct PixelOut {
    float depth, tind;
    float tex[3];
    float nrm[3];
};

__device__ __forceinline__ PixelOut resolve_pixel(unsigned long long key, const float* __restrict__ vx,
                                                  const float* __restrict__ vy, const float* __restrict__ vz,
                                                  const float* __restrict__ tri, const float* __restrict__ tex,
                                                  int nver, int ntri) {
    PixelOut o;
    if (key == bg_key()) {
        o.depth = bg_depth();
        o.tind = -1.0f;
        o.tex[0] = o.tex[1] = o.tex[2] = 0.0f;
        o.nrm[0] = o.nrm[1] = o.nrm[2] = 0.0f;
        return o;
    }
    int t = (int)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
    o.depth = f32_unord((uint32_t)(key >> 32));
    o.tind = (float)t;
    int p1 = (int)tri[t], p2 = (int)tri[(size_t)ntri + t], p3 = (int)tri[2 * (size_t)ntri + t];
    float x1 = vx[p1], x2 = vx[p2], x3 = vx[p3];
    float y1 = vy[p1], y2 = vy[p2], y3 = vy[p3];
    float z1 = vz[p1], z2 = vz[p2], z3 = vz[p3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float* tj = tex + (size_t)j * nver;
        o.tex[j] = ((tj[p1] + tj[p2]) + tj[p3]) / 3.0f;  // fp32, render_depth_op.cc:223
    }
    // differences in fp32, cross product in fp64 without FMA, one rounding to fp32 (render_depth_op.cc:227-236,308)
    double ax = (double)(x1 - x2), ay = (double)(y1 - y2), az = (double)(z1 - z2);
    double bx = (double)(x1 - x3), by = (double)(y1 - y3), bz = (double)(z1 - z3);
    o.nrm[0] = (float)(ay * bz - az * by);
    o.nrm[1] = (float)(az * bx - ax * bz);
    o.nrm[2] = (float)(ax * by - ay * bx);
    return o;
}

// Fused rendering-layer output of one strip (binned path only): [mask*im | pncc | normal] + depth image + raw depth +
// tri_ind.  The covered pixels' normalised normals were stored by the resolver's second pass; here the background
// pixels get zeros (the post-processing maps a zero normal to zero, network.py:190-192).
template <int BLOCK>
__device__ __forceinline__ void write_strip_fused(const RenderArgs& a, int b, int r0, int npix,
                                                  const unsigned long long* keys) {
    const int tid = threadIdx.x;
    const int ntri = a.ntri;
    const size_t pix0 = ((size_t)b * a.H + r0) * a.W;
    float* dep = a.depth + pix0;
    float* tin = a.tri_ind + pix0;
    float* dim = a.depth_img + pix0;
    float* nin = a.net_in + pix0 * 7;
    const float* __restrict__ img = a.im_gray + pix0;
    const float4* __restrict__ tws = a.tritex_ws + (a.tex_stride ? (size_t)b * ntri : 0);
    const unsigned long long KBG = bg_key();
    constexpr int UNR = 4;
    for (int i0 = tid; i0 < npix; i0 += BLOCK * UNR) {
        unsigned long long kk[UNR];
        bool cov[UNR];
        int t[UNR];
        float4 tv[UNR];
        float im[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int i = i0 + u * BLOCK;
            kk[u] = (i < npix) ? keys[i] : KBG;
            cov[u] = kk[u] != KBG;
            t[u] = cov[u] ? (int)(0xFFFFFFFFu - (uint32_t)kk[u]) : 0;
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            tv[u] = tws[t[u]];
            im[u] = img[min(i0 + u * BLOCK, npix - 1)];
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int i = i0 + u * BLOCK;
            if (i < npix) {
                const float d = cov[u] ? f32_unord((uint32_t)(kk[u] >> 32)) : bg_depth();
                dep[i] = d;
                tin[i] = cov[u] ? (float)t[u] : -1.0f;
                dim[i] = fmaxf(d, 1e-6f);                                   // depthimg, network.py:199
                float* o = nin + 7 * (size_t)i;
                o[0] = clip01(d) * im[u];                                   // mask * im_gray, :195-196
                store3(o + 1, clip01(cov[u] ? tv[u].x : 0.0f), clip01(cov[u] ? tv[u].y : 0.0f),     // pncc, :185
                       clip01(cov[u] ? tv[u].z : 0.0f));
                if (!cov[u]) store3(o + 4, 0.0f, 0.0f, 0.0f);
            }
        }
    }
}

// resolve + write the strip's four planes from the LDS keys (16-byte stores when the strip is 4-pixel aligned)
// BINNED: the caller is the binned path's resolver (winners' normals already stored, texture means in the table);
// otherwise the fallback kernel, which resolves everything from the vertices.  A compile-time switch so that the
// resolver does not carry the fallback writer's register budget.
template <int BLOCK, bool BINNED, int UNR_BINNED = 8>
__device__ __forceinline__ void write_strip(const RenderArgs& a, int b, int r0, int npix,
                                            const unsigned long long* keys, const float* __restrict__ vx,
                                            const float* __restrict__ vy, const float* __restrict__ vz) {
    const int tid = threadIdx.x;
    const int nver = a.nver, ntri = a.ntri;
    const float* __restrict__ tri = a.tri;
    const float* __restrict__ tex = a.texture + (size_t)b * a.tex_stride;
    const size_t pix0 = ((size_t)b * a.H + r0) * a.W;  // first pixel of the strip in the [B,H,W] planes
    float* dep = a.depth + pix0;
    float* tin = a.tri_ind + pix0;
    float* txi = a.tex_img + pix0 * 3;
    float* nrm = a.normal + pix0 * 3;
    const bool vec_ok = ((npix & 3) == 0) && ((pix0 & 3) == 0) &&
                        ((((uintptr_t)a.depth | (uintptr_t)a.tri_ind | (uintptr_t)a.tex_img | (uintptr_t)a.normal) & 15) == 0);
    if constexpr (BINNED) {
        // binned path: the normals of the covered pixels were already stored by the resolver's second pass; here
        // depth / tri_ind / texture go out for every pixel and zeros for the background pixels' normals.  One pixel per
        // lane: neighbouring lanes hold neighbouring pixels -> neighbouring triangles -> shared table lines.
        const float4* __restrict__ tws = a.tritex_ws + (a.tex_stride ? (size_t)b * ntri : 0);
        const unsigned long long KBG = bg_key();
        constexpr int UNR = UNR_BINNED;   // 8: a 10-row strip of 200 pixels is 7.8 pixels per thread of 256: ONE trip, one gather round trip
        for (int i0 = tid; i0 < npix; i0 += BLOCK * UNR) {
            unsigned long long kk[UNR];
            bool cov[UNR];
            int t[UNR];
            float4 tv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int i = i0 + u * BLOCK;
                kk[u] = (i < npix) ? keys[i] : KBG;
                cov[u] = kk[u] != KBG;
                t[u] = cov[u] ? (int)(0xFFFFFFFFu - (uint32_t)kk[u]) : 0;
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) tv[u] = tws[t[u]];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int i = i0 + u * BLOCK;
                if (i < npix) {
                    dep[i] = cov[u] ? f32_unord((uint32_t)(kk[u] >> 32)) : bg_depth();
                    tin[i] = cov[u] ? (float)t[u] : -1.0f;
                    float* tp = txi + 3 * (size_t)i;
                    store3(tp, cov[u] ? tv[u].x : 0.0f, cov[u] ? tv[u].y : 0.0f, cov[u] ? tv[u].z : 0.0f);
                    if (!cov[u]) {
                        float* np = nrm + 3 * (size_t)i;
                        store3(np, 0.0f, 0.0f, 0.0f);
                    }
                }
            }
        }
        return;
    }
    if (vec_ok) {
        // Four consecutive pixels per lane (16-byte stores).  The winner's data sits two dependent gathers away
        // (key -> vertex ids -> positions / texture); all loads of the four pixels are issued together and
        // unconditionally (background pixels read triangle 0 with clamped ids and discard it), so a pass costs two
        // memory round trips instead of eight.
        const bool can_gather = ntri > 0 && nver > 0;
        const unsigned long long KBG = bg_key();
        for (int g = tid; g < (npix >> 2); g += BLOCK) {
            unsigned long long kk[4];
            bool cov[4];
            int t[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                kk[k] = keys[4 * g + k];
                cov[k] = can_gather && kk[k] != KBG;
                t[k] = cov[k] ? (int)(0xFFFFFFFFu - (uint32_t)kk[k]) : 0;
            }
            PixelOut o[4];
            if (can_gather) {
                int p[4][3];
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        int q = (int)tri[(size_t)c * ntri + t[k]];
                        p[k][c] = min(max(q, 0), nver - 1);  // no-op for winners (their ids passed the bounds test)
                    }
                float P[4][3][3], T[4][3][3];  // [pixel][coord / channel][vertex]
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float* vc = vx + (size_t)c * a.vpitch;
                        const float* tc = tex + (size_t)c * nver;
#pragma unroll
                        for (int v = 0; v < 3; v++) {
                            P[k][c][v] = vc[p[k][v]];
                            T[k][c][v] = tc[p[k][v]];
                        }
                    }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    o[k].depth = cov[k] ? f32_unord((uint32_t)(kk[k] >> 32)) : bg_depth();
                    o[k].tind = cov[k] ? (float)t[k] : -1.0f;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        float m = ((T[k][c][0] + T[k][c][1]) + T[k][c][2]) / 3.0f;  // fp32, render_depth_op.cc:223
                        o[k].tex[c] = cov[k] ? m : 0.0f;
                    }
                    // differences in fp32, cross product in fp64 without FMA (render_depth_op.cc:227-236, 308)
                    double ax = (double)(P[k][0][0] - P[k][0][1]), ay = (double)(P[k][1][0] - P[k][1][1]),
                           az = (double)(P[k][2][0] - P[k][2][1]);
                    double bx = (double)(P[k][0][0] - P[k][0][2]), by = (double)(P[k][1][0] - P[k][1][2]),
                           bz = (double)(P[k][2][0] - P[k][2][2]);
                    o[k].nrm[0] = cov[k] ? (float)(ay * bz - az * by) : 0.0f;
                    o[k].nrm[1] = cov[k] ? (float)(az * bx - ax * bz) : 0.0f;
                    o[k].nrm[2] = cov[k] ? (float)(ax * by - ay * bx) : 0.0f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = resolve_pixel(KBG, vx, vy, vz, tri, tex, nver, ntri);
            }
            reinterpret_cast<float4*>(dep)[g] = make_float4(o[0].depth, o[1].depth, o[2].depth, o[3].depth);
            reinterpret_cast<float4*>(tin)[g] = make_float4(o[0].tind, o[1].tind, o[2].tind, o[3].tind);
            float4* t4 = reinterpret_cast<float4*>(txi) + 3 * (size_t)g;
            t4[0] = make_float4(o[0].tex[0], o[0].tex[1], o[0].tex[2], o[1].tex[0]);
            t4[1] = make_float4(o[1].tex[1], o[1].tex[2], o[2].tex[0], o[2].tex[1]);
            t4[2] = make_float4(o[2].tex[2], o[3].tex[0], o[3].tex[1], o[3].tex[2]);
            float4* n4 = reinterpret_cast<float4*>(nrm) + 3 * (size_t)g;
            n4[0] = make_float4(o[0].nrm[0], o[0].nrm[1], o[0].nrm[2], o[1].nrm[0]);
            n4[1] = make_float4(o[1].nrm[1], o[1].nrm[2], o[2].nrm[0], o[2].nrm[1]);
            n4[2] = make_float4(o[2].nrm[2], o[3].nrm[0], o[3].nrm[1], o[3].nrm[2]);
        }
    } else {
        for (int i = tid; i < npix; i += BLOCK) {
            PixelOut o = resolve_pixel(keys[i], vx, vy, vz, tri, tex, nver, ntri);
            dep[i] = o.depth;
            tin[i] = o.tind;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                txi[3 * (size_t)i + j] = o.tex[j];
                nrm[3 * (size_t)i + j] = o.nrm[j];
            }
        }
    }
}

// ---- first-generation path: every (face, strip) bin scans every triangle ---------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void render_strip_kernel(RenderArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int tid = threadIdx.x;
    const int bin = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bin / a.strips;
    const int s = bin - b * a.strips;
    const int r0 = s * a.rows;
    const int r1 = min(a.H, r0 + a.rows);
    const int npix = (r1 - r0) * a.W;
    const unsigned long long KBG = bg_key();
    for (int i = tid; i < npix; i += BLOCK) keys[i] = KBG;
    __syncthreads();
    const float* __restrict__ vx = a.vertex + (size_t)b * 3 * a.vpitch;
    const float* __restrict__ vy = vx + a.vpitch;
    const float* __restrict__ vz = vy + a.vpitch;
    for (int t = tid; t < a.ntri; t += BLOCK)
        raster_triangle_into_strip(t, a.tri, vx, vy, vz, a.nver, a.ntri, a.H, a.W, r0, r1, keys);
    __syncthreads();
    write_strip<BLOCK, false>(a, b, r0, npix, keys, vx, vy, vz);
}

// ---- binned path, kernel 1: per-triangle setup + hit test, records counting-sorted by strip -----------------
// record = {key.lo, key.hi, x0 | y0 << 16, mask}; mask != 0: hit bits (dy*8+dx) of an 8x4 window at (x0,y0) -- in a
// strip's own bucket when the window lies inside that strip, in the boundary bucket between two strips when it straddles
// them; mask == 0 (bucket 0): "big" record, the resolver rasterises triangle ~key.lo itself.
// bucket order inside a segment: [big | strip 0 | boundary 0/1 | strip 1 | boundary 1/2 | ...], so what strip s needs --
// boundary s-1/s, strip s, boundary s/s+1 -- is one contiguous range; segoff[k] = end of bucket k.
// 32-bit-offset gather: base pointer stays in SGPRs, one VALU shift per address (ids < 2^30 by construction).
__device__ __forceinline__ float ld_off(const float* __restrict__ base, uint32_t idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(idx << 2));
}
// float-stored id -> int with the reference's truncation; false for NaN / negative / >= n (x86 cvttss2si would give
// INT_MIN there, which is out of range too).  (-1,0) truncates to 0 like (int) does.
__device__ __forceinline__ bool id_ok(float f, int n, int& p) {
    p = (int)f;  // saturating, NaN -> 0
    return (f > -1.0f) && ((unsigned)p < (unsigned)n);
}

// The triangle list is a per-call constant shared by every face of the batch (network.py:178 makes it a tf.constant):
// its float-stored ids are converted, range-checked and turned into byte offsets ONCE per call (or once per model through
// fr_render_pack_tri) instead of once per (face, triangle) -- phase A of the emit kernel then costs one 16-byte load per
// triangle.  Entry = {4*p1, 4*p2, 4*p3, valid}; an invalid triangle (an id outside [0,nver): deviation 3, the reference
// would read out of bounds) carries offsets 0 (safe dummy gathers) and valid = 0.
// The slot behind the table's capacity holds a header {magic, nver, ntri, 0}: the emit kernel treats every triangle as
// invalid (the planes come out as pure background) when the table it is handed was not packed for its (nver, ntri) --
// a caller of the phase-by-phase entry point that skipped the pack phase, or reused the workspace for another shape,
// gets a defined result instead of out-of-bounds gathers.
constexpr int TRI4_MAGIC = 0x46525435;  // "FRT5"
constexpr int PACK_TPT = 2;                 // (= TPT of the emit kernel, declared below: two triangles per thread)
constexpr int PACK_ACTIVE = SEG / PACK_TPT;
// Lane order of a segment's triangles in the emit kernel.  Thread `tid` of an emit workgroup takes the table positions
// j = tid and j = PACK_ACTIVE + tid of its segment; WHICH triangle sits at a position is free (records carry the triangle's id,
// the resolve is order-independent), and it decides what the gathers cost: the CU's address path serves a dword gather at
// 16 lanes per clock when neighbouring lanes read neighbouring dwords, and at 4 lanes per clock or worse when they alternate
// between two rows (tools/gather_rate_probe.hip: 2.1 against 7.0 ns per wave instruction).  A triangle list that walks a grid
// cell by cell -- (v00, v10, v01), (v01, v10, v11), next cell -- makes every second lane jump a row under the identity order;
// taking the even triangles in the first pass and the odd ones in the second makes all eighteen streams consecutive.  The
// candidate orders are scored per segment by the number of neighbouring-lane pairs that do NOT read neighbouring dwords.
__device__ __forceinline__ int perm_local(int c, int j) {
    const int u = j >= PACK_ACTIVE ? 1 : 0, tid = j - u * PACK_ACTIVE;
    return c == 1 ? 2 * tid + u : j;
}
constexpr int PACK_NCAND = 2;
// One workgroup per segment.  out[t] (by triangle id, what the resolver and the per-face texture path read) =
// {4*p1, 4*p2, 4*p3, valid}; outp[seg * SEG + j] (by position, what the emit kernel's phase A reads) = the same offsets with
// .w = valid | local index << 1.  Both are defined for all nseg * SEG slots (invalid beyond ntri).
__global__ __launch_bounds__(256) void pack_tri_kernel(const float* __restrict__ tri, int nver, int ntri, int4* __restrict__ out,
                                                       int4* __restrict__ hdr, int4* __restrict__ outp, int force) {
    __shared__ int po[3][SEG];
    __shared__ unsigned char okf[SEG];
    __shared__ uint32_t cost[PACK_NCAND];
    const int seg = blockIdx.x, tid = threadIdx.x;
    if (seg == 0 && tid == 0) *hdr = make_int4(TRI4_MAGIC, nver, ntri, 0);
    if (tid < PACK_NCAND) cost[tid] = 0;
    for (int i = tid; i < SEG; i += 256) {
        const int t = seg * SEG + i;
        int p1 = 0, p2 = 0, p3 = 0;
        bool ok = false;
        if (t < ntri)
            ok = id_ok(tri[t], nver, p1) & id_ok(tri[(size_t)ntri + t], nver, p2) & id_ok(tri[2 * (size_t)ntri + t], nver, p3);
        const int4 e = ok ? make_int4(p1 << 2, p2 << 2, p3 << 2, 1) : make_int4(0, 0, 0, 0);
        out[t] = e;
        po[0][i] = e.x; po[1][i] = e.y; po[2][i] = e.z;
        okf[i] = ok ? 1 : 0;
    }
    __syncthreads();
    for (int j = tid; j < SEG; j += 256) {
        const int tl = j >= PACK_ACTIVE ? j - PACK_ACTIVE : j;
        if (tl & 15) {   // neighbours inside a 16-lane group
#pragma unroll
            for (int c = 0; c < PACK_NCAND; c++) {
                const int ia = perm_local(c, j), ib = perm_local(c, j - 1);
                uint32_t br = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int d = po[k][ia] - po[k][ib];
                    br += (d > 4 || d < -4) ? 1u : 0u;
                }
                if (br) atomicAdd(&cost[c], br);
            }
        }
    }
    __syncthreads();
    int c = 0;
    uint32_t best = cost[0];
#pragma unroll
    for (int k = 1; k < PACK_NCAND; k++)
        if (cost[k] < best) { best = cost[k]; c = k; }   // ties: the lower-numbered order
    if (force >= 0 && force < PACK_NCAND) c = force;      // A/B knob FR_EMIT_ORDER
    for (int j = tid; j < SEG; j += 256) {
        const int i = perm_local(c, j);
        outp[(size_t)seg * SEG + j] = make_int4(po[0][i], po[1][i], po[2][i], (int)okf[i] | (i << 1));
    }
}
__device__ __forceinline__ float ld_boff(const float* __restrict__ base, int boff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)(uint32_t)boff);
}
// x / 3.0f, correctly rounded like the division the reference performs (render_depth_op.cc:217, 223): for |x| in
// [2^-100, 2^100] the quotient comes from q0 = x*(1/3), r = fma(-3, q0, x) (exact), q = fma(r, 1/3, q0) -- three
// instructions instead of the ~12 of the IEEE division sequence (bit-identical to x / 3.0f on every finite x of that
// range: tests/test_render_gpu.py sweeps all 2^32 bit patterns); anything else takes the division.
__device__ __forceinline__ float div3(float x) {
    const float ax = __builtin_fabsf(x);
    if (ax >= 7.888609052e-31f && ax <= 1.267650600e30f) {
        const float y = 0.3333333432674407958984375f;
        const float q0 = x * y;
        const float r = __builtin_fmaf(-3.0f, q0, x);
        return __builtin_fmaf(r, y, q0);
    }
    return x / 3.0f;
}

// Inclusive prefix sum over the 64 lanes of a wave in six v_add_u32_dpp (Hillis-Steele inside each row of 16 lanes, then the
// row totals passed on with row_bcast:15 / row_bcast:31) -- the __shfl_up form is six DEPENDENT trips through the LDS
// crossbar (ds_bpermute), several hundred cycles on the critical path of every workgroup of both kernels.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1 (out-of-row lanes read 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}

// The kernel is VALU-bound (rocprof: SQ_ACTIVE_INST_VALU ~ 94 % of its duration), and about half of all triangles
// are rejected by the bbox rule before any fp64 work.  So it runs in two phases.  Phase A: every thread takes TPT
// triangles of the segment (all their loads in flight together), does the ids, the nine gathers and the bbox reject,
// and survivors are compacted into an LDS queue.  Phase B runs the expensive part (depth, fp64 barycentric test,
// normal, record emission) on DENSE waves -- about half as many wave-instructions -- and parks the records in LDS.
// Phase C counting-sorts them by strip and writes them out.
// Development hooks of the emit kernel (tools/emit_probe.hip supplies a stamping policy; the product instantiates NoEmitProbe,
// whose hooks are empty inlines).
struct NoEmitProbe {
    static constexpr int abl = 0;   // tools/emit_probe.hip: ablation level (the kernel returns at cut point `abl`); 0 = the product
    __device__ __forceinline__ void begin() {}
    template <int ID> __device__ __forceinline__ void stamp() {}
    __device__ __forceinline__ void finish(int) {}
};
constexpr int EMIT_BLOCK = 256;
constexpr int TPT = 2;                 // triangles per thread in phase A
constexpr int EMIT_ACTIVE = SEG / TPT; // threads that own triangles in phase A (252 of 256)

// LDS of one emit workgroup (20,424 B: eight of them share a CU's 160 KiB), carved out of ONE array by offset.
constexpr size_t EMIT_LDS_QA = 0;                                   // float4 qa[SEG]  phase A->B: x1 y1 x2 y2     B->C: the record
constexpr size_t EMIT_LDS_QB = EMIT_LDS_QA + SEG * sizeof(float4);  // float4 qb[SEG]  phase A->B: x3 y3 z1 z2     B->C: its normal
constexpr size_t EMIT_LDS_QD = EMIT_LDS_QB + SEG * sizeof(float4);  // uint2 qd[SEG]   .x: z3 (raw bits) in A->B, then bucket << 16 | pos
                                                                    //                 (~0 = none) in B->C; .y: local index
constexpr size_t EMIT_LDS_CNT = EMIT_LDS_QD + SEG * sizeof(uint2);  // u32 cnt[OFF_STRIDE]  per-bucket record count; start offset in phase C
constexpr size_t EMIT_LDS_QN2 = EMIT_LDS_CNT + OFF_STRIDE * 4;      // u64 qn2  survivors queued: low word = single-pixel ones (from slot 0
                                                                    //          up), high word = multi-pixel ones (from slot SEG-1 down)
constexpr size_t EMIT_LDS_BYTES = EMIT_LDS_QN2 + 8;
static_assert(EMIT_LDS_BYTES == 20424 && 8 * EMIT_LDS_BYTES <= 160 * 1024, "eight emit workgroups per CU");

// One emit workgroup: segment `lid % nseg` of face `lid / nseg`.  `lds` = EMIT_LDS_BYTES of the block's LDS (16-byte aligned).
template <class PR>
__device__ __forceinline__ void emit_body(const RenderArgs& a, const int lid, unsigned char* lds, PR& pr) {
    float4* const qa = reinterpret_cast<float4*>(lds + EMIT_LDS_QA);
    float4* const qb = reinterpret_cast<float4*>(lds + EMIT_LDS_QB);
    uint2* const qd = reinterpret_cast<uint2*>(lds + EMIT_LDS_QD);
    uint32_t* const cnt = reinterpret_cast<uint32_t*>(lds + EMIT_LDS_CNT);
    unsigned long long& qn2 = *reinterpret_cast<unsigned long long*>(lds + EMIT_LDS_QN2);
    const int tid = threadIdx.x;
    if constexpr (PR::abl == 1) return;   // (ablation: empty workgroups)
    const int b = a.nseg_magic ? (int)__umulhi((uint32_t)lid, a.nseg_magic) : lid / a.nseg;  // lid / nseg
    const int seg = lid - b * a.nseg;
    const int S = a.strips;
    if (tid < 2 * S) cnt[tid] = 0;
    if (tid == 0) qn2 = 0ull;
    // (the barrier that publishes these zeros sits below, behind the issue of phase A's gathers: the waves wait for memory
    // there anyway, and an LDS-only barrier -- no vmcnt(0) fence -- leaves the gathers in flight)

    const int nver = a.nver, ntri = a.ntri;
    const float* __restrict__ vx = a.vertex + (size_t)b * 3 * a.vpitch;
    const float* __restrict__ vy = vx + a.vpitch;
    const float* __restrict__ vz = vy + a.vpitch;

    // ---------------- phase A: pre-validated ids, gathers, bbox reject ----------------
    {
        const int4 hdr = a.tri4[(size_t)a.nseg * SEG];  // uniform: which (nver, ntri) the table was packed for
        const int table_ok = (int)(hdr.x == TRI4_MAGIC) & (int)(hdr.y == nver) & (int)(hdr.z == ntri);
        bool surv[TPT], single[TPT];
        int4 e[TPT];
        float x1[TPT], x2[TPT], x3[TPT], y1[TPT], y2[TPT], y3[TPT], z1[TPT], z2[TPT], z3[TPT];
        bool valid[TPT];
        // the segment's triangles in the lane order pack_tri_kernel chose: position -> {offsets, valid | local index << 1}
        const int4* __restrict__ tri4p = a.tri4 + ((size_t)a.nseg * SEG + 1) + (size_t)seg * SEG;
        int tl[TPT];
#pragma unroll
        for (int u = 0; u < TPT; u++) {
            // one unconditional 16-byte load (saddr + 32-bit offset form); no short-circuit on .w, or the compiler
            // splits the load and serialises the two halves
            e[u] = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(tri4p) + (size_t)((uint32_t)min(u * EMIT_ACTIVE + tid, SEG - 1) << 4));
            valid[u] = ((int)(tid < EMIT_ACTIVE) & e[u].w & table_ok) != 0;   // (a valid entry has t < ntri: pack_tri_kernel)
            tl[u] = e[u].w >> 1;
        }
        if constexpr (PR::abl == 2) {   // (ablation: the table loads only)
            if (e[0].x + e[1].y == 0x7fffffff) qd[tid].x = 1;
            return;
        }
#pragma unroll
        for (int u = 0; u < TPT; u++) {
            x1[u] = ld_boff(vx, e[u].x); x2[u] = ld_boff(vx, e[u].y); x3[u] = ld_boff(vx, e[u].z);
            y1[u] = ld_boff(vy, e[u].x); y2[u] = ld_boff(vy, e[u].y); y3[u] = ld_boff(vy, e[u].z);
            z1[u] = ld_boff(vz, e[u].x); z2[u] = ld_boff(vz, e[u].y); z3[u] = ld_boff(vz, e[u].z);
        }
        if constexpr (PR::abl == 3) {   // (ablation: table + the eighteen gathers, nothing computed)
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < TPT; u++) acc += ((x1[u] + x2[u]) + x3[u]) + ((y1[u] + y2[u]) + y3[u]) + ((z1[u] + z2[u]) + z3[u]);
            if (acc == 1.2345e30f) qd[tid].x = 1;
            return;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // counters zeroed (first used by the compaction below)
        pr.template stamp<0>();
#pragma unroll
        for (int u = 0; u < TPT; u++) {
            // bbox = ceil(min) .. floor(max) per axis and the whole-triangle reject of render_depth_op.cc:276-283, in the
            // float domain.  A NaN coordinate makes PointInTri false for every pixel whatever the bbox (all dot products
            // turn NaN), so the NaN-ignoring v_min3/v_max3 are equivalent to the reference's macros here; the
            // comparisons reject NaN / +-inf / out-of-int-range bounds exactly where (int) gives INT_MIN on x86.
            const float fx0 = ceilf(fminf(fminf(x1[u], x2[u]), x3[u])), fx1 = floorf(fmaxf(fmaxf(x1[u], x2[u]), x3[u]));
            const float fy0 = ceilf(fminf(fminf(y1[u], y2[u]), y3[u])), fy1 = floorf(fmaxf(fmaxf(y1[u], y2[u]), y3[u]));
            surv[u] = valid[u] && (fx0 >= 0.0f) && (fy0 >= 0.0f) && (fx1 <= a.wm1) && (fy1 <= a.hm1) && !(fx1 < fx0) &&
                      !(fy1 < fy0);
            single[u] = (fx0 == fx1) && (fy0 == fy1);
            // A bbox with a single pixel centre (78 % of the survivors on a sub-pixel mesh) misses that centre 70 % of the
            // time.  The certified fp32 inside test of phase B (same expressions, same pixel, see there) is cheap
            // enough to run here, on the sparse lanes: a triangle it certifies as a MISS can emit nothing and is dropped
            // before the compaction, so that phase B runs on less than half the lanes; anything it cannot certify stays.  (Flagging
            // the certified HITS so that phase B skips their test measured no further gain.)
            if ((a.use_filter & 2) && surv[u] && single[u]) {
                const float v0x = x3[u] - x1[u], v0y = y3[u] - y1[u], v1x = x2[u] - x1[u], v1y = y2[u] - y1[u];
                const float D = __builtin_fmaf(v0x, v1y, -(v0y * v1x));   // (fused: one rounding fewer than the bound assumes)
                const float M0 = fmaxf(fmaxf(fabsf(v0x), fabsf(v0y)), fmaxf(fabsf(v1x), fabsf(v1y)));
                const float S = M0 * M0;
                const float v2x = fx0 - x1[u], v2y = fy0 - y1[u];
                const float A = __builtin_fmaf(v2x, v1y, -(v2y * v1x)), Bq = __builtin_fmaf(v0x, v2y, -(v0y * v2x)), C = (D - A) - Bq;
                const bool certain = (M0 < 1073741824.0f) && (M0 > 9.094947017729282e-13f) && (fabsf(D) >= 0.00390625f * S) &&
                                     fminf(fminf(fabsf(A), fabsf(Bq)), fabsf(C)) >= 1.52587890625e-05f * S;
                const bool in = D > 0.0f ? fminf(fminf(A, Bq), C) > 0.0f : fmaxf(fmaxf(A, Bq), C) < 0.0f;
                if (certain && !in) surv[u] = false;
            }
            // per-triangle texture mean ((t1+t2)+t3)/3 in fp32 (render_depth_op.cc:223) when the texture is shared by the
            // batch: computed once, by face 0's workgroups, for every valid triangle
            if (a.tex_stride == 0 && b == 0 && valid[u]) {
                float tm[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float* tj = a.texture + (size_t)j * nver;
                    tm[j] = div3((ld_boff(tj, e[u].x) + ld_boff(tj, e[u].y)) + ld_boff(tj, e[u].z));
                }
                a.tritex_ws[seg * SEG + tl[u]] = make_float4(tm[0], tm[1], tm[2], 0.0f);
            }
        }
        pr.template stamp<1>();   // gathers back, bbox + pre-cull done
        // Survivors of both triangles are compacted with ONE LDS atomic per wave, into a two-ended queue: bboxes holding a
        // single pixel centre (~78 % on the BFM-scale mesh) fill it from the front, the others from the back.  Phase B's
        // waves are then (nearly) homogeneous: the pixel loop of a wave runs as long as its LONGEST lane, and mixing one
        // 2- or 4-pixel triangle into a wave of 1-pixel ones doubles that wave's fp64 work.
        // (Round 3, measured and not kept: per-wave queue regions filled by wave-local ranks -- no allocation atomic, no
        // zero-init barrier at kernel entry: 12.4 k vs 12.2 k cycles of wave life in the stamped build, and 45-46 vs 42.5-44 us
        // for the kernel in three bench sessions each.)
        static_assert(TPT == 2, "compaction below is written for two triangles per thread");
        const unsigned long long ms0 = __ballot(surv[0] && single[0]), ms1 = __ballot(surv[1] && single[1]);
        const unsigned long long mm0 = __ballot(surv[0] && !single[0]), mm1 = __ballot(surv[1] && !single[1]);
        const uint32_t cs0 = (uint32_t)__popcll(ms0), cs1 = (uint32_t)__popcll(ms1);
        const uint32_t cm0 = (uint32_t)__popcll(mm0), cm1 = (uint32_t)__popcll(mm1);
        unsigned long long wb = 0;
        if ((tid & 63) == 0 && (cs0 + cs1 + cm0 + cm1))
            wb = atomicAdd(&qn2, ((unsigned long long)(cm0 + cm1) << 32) | (unsigned long long)(cs0 + cs1));
        const uint32_t fbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wb);
        const uint32_t bbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wb >> 32));
#pragma unroll
        for (int u = 0; u < TPT; u++) {
            if (surv[u]) {
                const unsigned long long m = single[u] ? (u ? ms1 : ms0) : (u ? mm1 : mm0);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t slot = single[u] ? fbase + (u ? cs0 : 0u) + rank
                                                : (uint32_t)(SEG - 1) - (bbase + (u ? cm0 : 0u) + rank);
                qa[slot] = make_float4(x1[u], y1[u], x2[u], y2[u]);
                qb[slot] = make_float4(x3[u], y3[u], z1[u], z2[u]);
                qd[slot] = make_uint2(__float_as_uint(z3[u]), (uint32_t)tl[u]);
            }
        }
    }
    pr.template stamp<2>();   // compaction done
    if constexpr (PR::abl == 5) return;   // (ablation: phase A complete)
    __syncthreads();
    pr.template stamp<3>();

    // ---------------- phase B: dense lanes, one surviving triangle each ----------------
    const unsigned long long q2 = qn2;
    const int nqf = (int)(uint32_t)q2, nq = nqf + (int)(uint32_t)(q2 >> 32);
    for (int qi = tid; qi < nq; qi += EMIT_BLOCK) {
        const int sl = qi < nqf ? qi : SEG - 1 - (qi - nqf);  // front part, then the back part (densely packed lanes)
        const float4 A4 = qa[sl], B4 = qb[sl];
        const uint2 D2 = qd[sl];
        const float x1 = A4.x, y1 = A4.y, x2 = A4.z, y2 = A4.w, x3 = B4.x, y3 = B4.y;
        const float z1 = B4.z, z2 = B4.w, z3 = __uint_as_float(D2.x);
        bool emit = false;
        int bucket = 0;
        uint4 rec = make_uint4(0, 0, 0, 0);
        float4 nrm4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float h = div3((z1 + z2) + z3);     // fp32 centroid depth ((z1+z2)+z3)/3.0f, :217
        h = h + 0.0f;                        // -0 -> +0
        const int t = seg * SEG + (int)D2.y;
        if (h > bg_depth()) {                // NaN or <= background never passes 'depth < h' (:295)
            const int x_min = (int)ceilf(fminf(fminf(x1, x2), x3)), x_max = (int)floorf(fmaxf(fmaxf(x1, x2), x3));
            const int y_min = (int)ceilf(fminf(fminf(y1, y2), y3)), y_max = (int)floorf(fmaxf(fmaxf(y1, y2), y3));
            const unsigned long long key = make_key(h, t);
            rec.x = (uint32_t)key;
            rec.y = (uint32_t)(key >> 32);
            rec.z = (uint32_t)x_min | ((uint32_t)y_min << 16);
            // strip of the first / last row: y / rows through the exact 2^32 reciprocal (y, rows < 2^16)
            const int s0 = a.rows_magic ? (int)__umulhi((uint32_t)y_min, a.rows_magic) : y_min;
            const int s1 = a.rows_magic ? (int)__umulhi((uint32_t)y_max, a.rows_magic) : y_max;
            if (x_max - x_min < SMALL_W && y_max - y_min < SMALL_H) {
                // Inside test of the window's pixel centres.  The reference's test is an fp64 computation
                // (render_depth_op.cc:90-121); its DECISION equals the exact-arithmetic one -- with D = v0 x v1,
                // A = v2 x v1, B = v0 x v2 (2-D cross products): inside <=> A/D >= 0, B/D >= 0, (A+B)/D < 1 -- whenever the
                // point is not within rounding distance of an edge and the triangle is not a sliver.  So the decision is
                // first taken in fp32 WITH error bounds, and only pixels it cannot certify go through the fp64 operation
                // sequence (about one lane in 10^4 on a sub-pixel mesh).  Bounds, with S = (largest |component| of v0, v1 --
                // and hence of v2, see below)^2: every fp32 cross product is within 2^-21 S of its exact value (C = D - A - B within 2^-19 S); a
                // pixel is certified when |D| >= 2^-8 S and |A|, |B|, |C| >= 2^-16 S.  Then the signs of the exact A, B, C
                // are the computed ones, and the reference's fp64 u = A/D, v = B/D carry a relative error below 2^-22
                // (numerators >= 2^-24 S^2 against 80 * 2^-53 S^2 of rounding), so its comparisons u < 0, v < 0, u > 1,
                // v > 1, u + v < 1 -- whose exact margins are >= 2^-17 -- fall the same way.  NaN / Inf / huge / tiny
                // (fp32 underflow) inputs fail the certificate and take the fp64 path.
                const float v0x = x3 - x1, v0y = y3 - y1, v1x = x2 - x1, v1y = y2 - y1;
                const float D = __builtin_fmaf(v0x, v1y, -(v0y * v1x));   // (fused: one rounding fewer than the bound assumes)
                // every tested pixel centre Q lies inside the triangle's bbox, so |Q - P1| <= max_i |P_i - P1| per axis:
                // the largest component of v0, v1 bounds v2's as well and S is a per-triangle constant
                const float M0 = fmaxf(fmaxf(fabsf(v0x), fabsf(v0y)), fmaxf(fabsf(v1x), fabsf(v1y)));
                const float S = M0 * M0;
                const float tm = 1.52587890625e-05f * S;  // 2^-16 S
                // 2^-40 < M0 < 2^30: S and the thresholds are normal fp32 numbers, the error bounds hold; not a sliver
                const bool tri_ok = (a.use_filter & 1) && (M0 < 1073741824.0f) && (M0 > 9.094947017729282e-13f) &&
                                    (fabsf(D) >= 0.00390625f * S);
                const bool dpos = D > 0.0f;
                // one flat loop over the window's pixels (not y / x nests): the compiler keeps it rolled, which holds the
                // kernel's VGPR count at the 8-waves-per-SIMD budget
                const int bw = x_max - x_min + 1, npx = bw * (y_max - y_min + 1);
                int dx = 0, bit = 0, yy = y_min;
                uint32_t m = 0, unsure = 0;
#pragma clang loop unroll(disable)
                for (int k = 0; k < npx; k++) {
                    const float v2x = (float)(x_min + dx) - x1, v2y = (float)yy - y1;
                    const float A = __builtin_fmaf(v2x, v1y, -(v2y * v1x)), Bq = __builtin_fmaf(v0x, v2y, -(v0y * v2x)), C = (D - A) - Bq;
                    const bool certain = tri_ok && fminf(fminf(fabsf(A), fabsf(Bq)), fabsf(C)) >= tm;
                    const bool in = dpos ? fminf(fminf(A, Bq), C) > 0.0f : fmaxf(fmaxf(A, Bq), C) < 0.0f;
                    const uint32_t bm = 1u << (bit + dx);
                    if (!certain) unsure |= bm;
                    else if (in) m |= bm;
                    if (++dx == bw) { dx = 0; bit += SMALL_W; yy++; }
                }
                if (unsure) {  // the reference's own operation sequence for the pixels the certificate left open
                    const TriSetup ts = tri_setup(x1, y1, x2, y2, x3, y3);
                    while (unsure) {
                        const int bi = __ffs((int)unsure) - 1;
                        unsure &= unsure - 1;
                        if (point_in_tri(ts, x_min + (bi & 7), y_min + (bi >> 3))) m |= 1u << bi;
                    }
                }
                rec.w = m;
                emit = (m != 0);
                // a window that straddles strips s0 / s0+1 goes to the bucket between the two strips' own: a strip's resolver
                // reads [boundary above | own | boundary below] as one contiguous range and applies the rows that are its own
                bucket = (s0 == s1) ? 1 + 2 * s0 : 2 + 2 * s0;
            } else {
                rec.w = 0;
                emit = true;
                bucket = 0;
            }
            if (emit) {
                // un-normalised normal (p1-p2) x (p1-p3): fp32 differences, fp64 products without FMA, one rounding
                // (render_depth_op.cc:227-236, 308) -- computed here, where the vertices are at hand
                double ax = (double)(x1 - x2), ay = (double)(y1 - y2), az = (double)(z1 - z2);
                double bx = (double)(x1 - x3), by = (double)(y1 - y3), bz = (double)(z1 - z3);
                nrm4 = make_float4((float)(ay * bz - az * by), (float)(az * bx - ax * bz), (float)(ax * by - ay * bx), 0.0f);
                if (a.tex_stride) {  // every face has its own texture: mean per emitting (face, triangle)
                    const float* __restrict__ tex = a.texture + (size_t)b * a.tex_stride;
                    const int4 e = a.tri4[t];  // valid: checked in phase A
                    float tm[3];
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const float* tj = tex + (size_t)j * nver;
                        tm[j] = div3((ld_boff(tj, e.x) + ld_boff(tj, e.y)) + ld_boff(tj, e.z));
                    }
                    a.tritex_ws[(size_t)b * ntri + t] = make_float4(tm[0], tm[1], tm[2], 0.0f);
                }
            }
        }
        uint32_t tag = 0xFFFFFFFFu;
        // (one LDS atomic per record: aggregating a wave's records per distinct bucket -- they fall into one to three --
        // into one atomic each measured 3 us SLOWER per launch; same-address LDS atomics are cheap on gfx950)
        uint32_t pos = 0;
        if (emit) pos = atomicAdd(&cnt[bucket], 1u);
        if (emit) {
            tag = ((uint32_t)bucket << 16) | pos;
            qa[sl] = make_float4(__uint_as_float(rec.x), __uint_as_float(rec.y), __uint_as_float(rec.z),
                                 __uint_as_float(rec.w));
            qb[sl] = nrm4;
        }
        qd[sl].x = tag;
    }
    pr.template stamp<4>();   // phase B done
    if constexpr (PR::abl == 6) return;   // (ablation: phases A + B)
    __syncthreads();
    pr.template stamp<5>();
    // ---------------- phase C: bucket offsets, records out ----------------
    // (one wave scans, the other three wait at the barrier: a variant in which every wave scans for itself and reads the
    // bucket base with a cross-lane shuffle -- no third barrier -- measured 2 us SLOWER per launch, A/B in one process)
    uint16_t* off = a.segoff + (size_t)b * OFF_STRIDE * a.nseg + seg;   // bucket k of this segment at off[k * nseg]
    if (tid < 64) {  // one wave (all 64 lanes active, as the DPP scan needs) scans the (at most 64) bucket counts
        const uint32_t c = (tid < 2 * S) ? cnt[tid] : 0u;
        const uint32_t inc = wave_inclusive_scan(c);
        if (tid < 2 * S) {
            cnt[tid] = inc - c;          // start of bucket k
            off[(size_t)tid * a.nseg] = (uint16_t)inc;    // end of bucket k (bucket 0: #big)
        }
    }
    __syncthreads();
    pr.template stamp<6>();
    if constexpr (PR::abl == 7) return;   // (ablation: everything but the record stores)
    uint4* R = a.recs + ((size_t)b * a.nseg + seg) * (2 * SEG);
    for (int qi = tid; qi < nq; qi += EMIT_BLOCK) {
        const int sl = qi < nqf ? qi : SEG - 1 - (qi - nqf);
        const uint32_t tag = qd[sl].x;
        if (tag != 0xFFFFFFFFu) {
            const float4 r = qa[sl];
            const uint32_t slot = cnt[tag >> 16] + (tag & 0xFFFFu);
            R[2 * slot] = make_uint4(__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w));
            *reinterpret_cast<float4*>(R + 2 * slot + 1) = qb[sl];
        }
    }
    pr.finish(nq);
}

template <class PR = NoEmitProbe>
__global__ __launch_bounds__(EMIT_BLOCK) void raster_emit_kernel(RenderArgs a) {
    PR pr;
    pr.begin();
    __shared__ __attribute__((aligned(16))) unsigned char lds[EMIT_LDS_BYTES];
    emit_body<PR>(a, xcd_remap(blockIdx.x, gridDim.x), lds, pr);
}

// ---- binned path, kernel 2: per (face, strip) LDS resolve + output ----------------------------------------
// Block-wide exclusive scan of one value per thread (wave scan + one LDS hop).
template <int BLOCK>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wtot, uint32_t& total) {
    constexpr int NW = BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint32_t x = wtot[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// The same for TWO values per thread at once (one pass of shuffles over a packed 64-bit word, ONE LDS hop, one pair of
// barriers instead of two): exclusive prefixes and totals of va and vb, each total < 2^32.
template <int BLOCK>
__device__ __forceinline__ void block_exclusive_scan2(uint32_t va, uint32_t vb, unsigned long long* wtot2, uint32_t& exa,
                                                      uint32_t& exb, uint32_t& tota, uint32_t& totb) {
    constexpr int NW = BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (two wave scans, then ONE packed LDS hop: the halves cannot carry into each other, the totals are < 2^32)
    const unsigned long long inc = ((unsigned long long)wave_inclusive_scan(vb) << 32) | wave_inclusive_scan(va);
    if (lane == 63) wtot2[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const unsigned long long x = wtot2[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    const unsigned long long ex = base + inc - (((unsigned long long)vb << 32) | va);
    exa = (uint32_t)ex;
    exb = (uint32_t)(ex >> 32);
    tota = (uint32_t)tot;
    totb = (uint32_t)(tot >> 32);
}

// bits of the 8x4 window mask (bit = dy*8 + dx) whose row dy is < n
__device__ __forceinline__ uint32_t window_rows_below(int n) {
    return n >= SMALL_H ? 0xFFFFFFFFu : (n <= 0 ? 0u : (1u << (8 * n)) - 1u);
}

constexpr int SLOT_CAP = 3072;  // flattened record list of a bin kept in LDS (12 KiB); longer lists fall back to a search
constexpr size_t resolve_scratch_bytes(int block) {
    return 2 * (size_t)(block + 1) * 4 + (size_t)block * 2 + 64 * 4 + 16 + (size_t)SLOT_CAP * 4;
}
// The bin's records are spread over the face's segments (a few per segment).  Walking segments one after the other
// would serialise ~4 dependent memory round trips per segment; instead the per-segment counts are prefix-summed in
// LDS and the threads take records from the flattened list, so all record loads of a bin are in flight together.
template <int BLOCK, bool FUSED, class PR>
__device__ __forceinline__ void resolve_body(const RenderArgs& a, const int bin, unsigned long long* keys, PR& pr) {
    constexpr int CAP = SLOT_CAP;
    const int tid = threadIdx.x;
    const int b = bin / a.strips;
    const int s = bin - b * a.strips;
    const int r0 = s * a.rows;
    const int r1 = min(a.H, r0 + a.rows);
    const int W = a.W;
    const int npix = (r1 - r0) * W;
    // scratch behind the keys of a full strip (the launcher sizes the dynamic LDS for it)
    uint32_t* const pref = reinterpret_cast<uint32_t*>(keys + (size_t)a.rows * W);  // [BLOCK+1] small-record list offsets
    uint32_t* const prefb = pref + BLOCK + 1;                                         // [BLOCK+1] big-record list offsets
    uint16_t* const lo16 = reinterpret_cast<uint16_t*>(prefb + BLOCK + 1);           // [BLOCK]
    uint32_t* const wtot = reinterpret_cast<uint32_t*>(lo16 + BLOCK);                 // [64] (BLOCK even: 4-byte aligned)
    uint32_t* const slotlist = wtot + 64;                                             // [SLOT_CAP] record slot of list entry j
    const unsigned long long KBG = bg_key();
    for (int i = tid; i < npix; i += BLOCK) keys[i] = KBG;
    const float* __restrict__ vx = a.vertex + (size_t)b * 3 * a.vpitch;
    const float* __restrict__ vy = vx + a.vpitch;
    const float* __restrict__ vz = vy + a.vpitch;
    // where the winners' normals go: the strip's slice of the normal plane, or (fused) channels 4..6 of the 7-channel
    // CoarseNet input, post-processed
    constexpr int NSTRIDE = FUSED ? 7 : 3;
    float* nplane = FUSED ? a.net_in + (((size_t)b * a.H + r0) * W) * 7 + 4
                          : a.normal + (((size_t)b * a.H + r0) * W) * 3;
    constexpr int RU = 4;  // records per lane per trip: all loads in flight before the first LDS operation

    // pass 0: z-resolve -- every hit becomes one ds_max_u64.  pass 1: the winners are known; each record looks its hit
    // pixels up again and, where its key won, stores its normal (kept in the record's companion slot) to the normal
    // plane.  The second read of the records is an L1/L2 hit.
    const bool one_chunk = a.nseg <= BLOCK;  // the usual case: the offsets and both scans are done once, not per pass
    // ---- wave-local front (FR_RESOLVE_OPT=2, 256-thread bins) -------------------------------------------------------------
    // The general front below flattens the bin's records into ONE list for the workgroup: a block scan (two barriers), the
    // list (a third), then the loads.  Here every wave flattens the records of ITS segments (dealt round-robin, so the four
    // waves own equal shares) with a DPP scan into its own quarter of the slot list -- no barrier, LDS is in order inside
    // a wave -- and issues its loads at once; the ONE barrier that follows (LDS-only: the loads stay in flight) publishes
    // the initialised keys and each wave's verdict.  The first six records per lane stay in registers (with their normals)
    // for the second pass; what a busy wave has beyond that goes through a short reload loop.  A wave whose share does not
    // fit its quarter of the list, or that saw a big record, says so and the whole bin takes the general path (the
    // speculative loads are dropped).
    if constexpr (BLOCK == 256) {
        if (one_chunk && a.resolve_opt == 2) {
            constexpr int NW = BLOCK / 64, RF = 6, WCAP = CAP / NW;
            static_assert(64 * RF <= WCAP, "a wave's share of the slot list");
            const int lane = tid & 63, wave = tid >> 6;
            const int seg = lane * NW + wave;
            const uint4* Rbase = a.recs + (size_t)b * a.nseg * (2 * SEG);
            const float4* Nbase = reinterpret_cast<const float4*>(Rbase) + 1;
            uint32_t nbig = 0, lo = 0, hi = 0;
            if (seg < a.nseg) {
                const uint16_t* off = a.segoff + (size_t)b * OFF_STRIDE * a.nseg + seg;
                nbig = off[0];
                lo = off[(size_t)(s > 0 ? 2 * s - 1 : 0) * a.nseg];
                hi = off[(size_t)min(2 * s + 2, 2 * a.strips - 1) * a.nseg];
            }
            pr.template stamp<0>();   // offsets back
            const uint32_t cntw = hi - lo, inc = wave_inclusive_scan(cntw), ex = inc - cntw;
            const uint32_t total_w = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            const bool ok_w = total_w <= (uint32_t)WCAP && __ballot(nbig != 0) == 0ull;
            uint32_t* wl = slotlist + wave * WCAP;
            uint32_t* wstat = wtot + 48;
            if (lane == 0) wstat[wave] = ok_w ? 1u : 0u;
            uint4 r[RF];
            float4 nv[RF];
            if (ok_w) {
                for (uint32_t i = 0; i < cntw; i++) wl[ex + i] = (uint32_t)seg * SEG + lo + i;
                pr.template stamp<1>();   // wave list done
#pragma unroll
                for (int u = 0; u < RF; u++) {
                    const uint32_t j = lane + u * 64;
                    r[u] = make_uint4(0, 0, 0, 0);
                    nv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < total_w) {
                        const uint32_t slot = wl[j];
                        r[u] = Rbase[2 * slot];
                        nv[u] = Nbase[2 * slot];
                    }
                }
                pr.template stamp<2>();   // (issue of the record + normal loads)
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // keys initialised, verdicts published
            bool all_ok = true;
#pragma unroll
            for (int w = 0; w < NW; w++) all_ok = all_ok && wstat[w] != 0;
            if (all_ok) {
                uint32_t msk[RF];
                int p0s[RF];
#pragma unroll
                for (int u = 0; u < RF; u++) {
                    const unsigned long long key = ((unsigned long long)r[u].y << 32) | r[u].x;
                    const int x0 = (int)(r[u].z & 0xFFFFu), y0 = (int)(r[u].z >> 16);
                    p0s[u] = (y0 - r0) * W + x0;
                    uint32_t m = r[u].w;
                    m &= window_rows_below(r1 - y0) & ~window_rows_below(r0 - y0);
                    msk[u] = m;
                    while (m) {
                        const int bit = __ffs((int)m) - 1;
                        m &= m - 1;
                        atomicMax(keys + (p0s[u] + (bit >> 3) * W + (bit & 7)), key);
                    }
                }
                for (uint32_t j = 64 * RF + lane; j < total_w; j += 64) {   // the wave's records beyond six per lane
                    const uint4 rr = Rbase[2 * wl[j]];
                    const unsigned long long key = ((unsigned long long)rr.y << 32) | rr.x;
                    const int x0 = (int)(rr.z & 0xFFFFu), y0 = (int)(rr.z >> 16);
                    const int p0 = (y0 - r0) * W + x0;
                    uint32_t m = rr.w & window_rows_below(r1 - y0) & ~window_rows_below(r0 - y0);
                    while (m) {
                        const int bit = __ffs((int)m) - 1;
                        m &= m - 1;
                        atomicMax(keys + (p0 + (bit >> 3) * W + (bit & 7)), key);
                    }
                }
                pr.template stamp<3>();   // records back, LDS max done
                __syncthreads();
                pr.template stamp<4>();
#pragma unroll
                for (int u = 0; u < RF; u++) {
                    const unsigned long long key = ((unsigned long long)r[u].y << 32) | r[u].x;
                    uint32_t m = msk[u];
                    if (m) {
                        const float4 nq = FUSED ? post_normal(nv[u]) : nv[u];
                        while (m) {
                            const int bit = __ffs((int)m) - 1;
                            m &= m - 1;
                            const int px = p0s[u] + (bit >> 3) * W + (bit & 7);
                            if (keys[px] == key) store3(nplane + NSTRIDE * (ptrdiff_t)px, nq.x, nq.y, nq.z);
                        }
                    }
                }
                for (uint32_t j = 64 * RF + lane; j < total_w; j += 64) {
                    const uint32_t slot = wl[j];
                    const uint4 rr = Rbase[2 * slot];   // (an L1 / L2 hit: read a moment ago)
                    const unsigned long long key = ((unsigned long long)rr.y << 32) | rr.x;
                    const int x0 = (int)(rr.z & 0xFFFFu), y0 = (int)(rr.z >> 16);
                    const int p0 = (y0 - r0) * W + x0;
                    uint32_t m = rr.w & window_rows_below(r1 - y0) & ~window_rows_below(r0 - y0);
                    uint32_t won = 0;
                    while (m) {
                        const int bit = __ffs((int)m) - 1;
                        m &= m - 1;
                        if (keys[p0 + (bit >> 3) * W + (bit & 7)] == key) won |= 1u << bit;
                    }
                    if (won) {
                        float4 nq = Nbase[2 * slot];
                        if (FUSED) nq = post_normal(nq);
                        while (won) {
                            const int bit = __ffs((int)won) - 1;
                            won &= won - 1;
                            store3(nplane + NSTRIDE * (ptrdiff_t)(p0 + (bit >> 3) * W + (bit & 7)), nq.x, nq.y, nq.z);
                        }
                    }
                }
                pr.template stamp<5>();   // winners' normals stored
                if (FUSED)
                    write_strip_fused<BLOCK>(a, b, r0, npix, keys);
                else
                    write_strip<BLOCK, true, 8>(a, b, r0, npix, keys, vx, vy, vz);
                pr.finish(0);
                return;
            }
            __syncthreads();   // (general path: everyone has read the verdicts before wtot / the slot list are reused)
        }
    }
    for (int pass = 0; pass < 2; pass++) {
        for (int c0 = 0; c0 < a.nseg; c0 += BLOCK) {
            const uint4* Rbase = a.recs + ((size_t)b * a.nseg + c0) * (2 * SEG);   // record k at [2k], its normal at [2k + 1]
            const float4* Nbase = reinterpret_cast<const float4*>(Rbase) + 1;
            if (pass == 0 || !one_chunk) {
                const int seg = c0 + tid;
                uint32_t nbig = 0, lo = 0, hi = 0;
                if (seg < a.nseg) {
                    const uint16_t* off = a.segoff + (size_t)b * OFF_STRIDE * a.nseg + seg;
                    nbig = off[0];
                    lo = off[(size_t)(s > 0 ? 2 * s - 1 : 0) * a.nseg];               // start of bucket 2s (s = 0: of bucket 1, past the big ones)
                    hi = off[(size_t)min(2 * s + 2, 2 * a.strips - 1) * a.nseg];  // end of the boundary bucket below (last strip: of its own)
                }
                if (pass == 0 && c0 == 0) pr.template stamp<0>();   // offsets back
                uint32_t tot_s, tot_b, ex_s, ex_b;
                block_exclusive_scan2<BLOCK>(hi - lo, nbig, reinterpret_cast<unsigned long long*>(wtot), ex_s, ex_b, tot_s, tot_b);
                pref[tid] = ex_s;
                prefb[tid] = ex_b;
                lo16[tid] = (uint16_t)lo;
                if (tid == 0) {
                    pref[BLOCK] = tot_s;
                    prefb[BLOCK] = tot_b;
                }
                // the flattened list itself: segment `seg` owns entries [ex_s, ex_s + hi - lo) -- a handful each -- and
                // writes their record slots, so that both passes find record j with ONE LDS read instead of a
                // log2(BLOCK)-step search through the prefix sums
                // (Round 3, measured and reverted: the lanes of a wave filling the ranges of the wave's non-empty segments
                // together -- ballot, readlane, strided fill -- instead of every segment's own thread: this phase 6.3 k ->
                // 12.2 k cycles.  A strip's records are spread over MANY segments with a few entries each, not a few
                // segments with many.)
                if (tot_s <= (uint32_t)SLOT_CAP)
                    for (uint32_t i = 0; i < hi - lo; i++) slotlist[ex_s + i] = (uint32_t)tid * SEG + lo + i;
                __syncthreads();  // also orders the key initialisation before the first atomics
                if (pass == 0 && c0 == 0) pr.template stamp<1>();   // scans + slot list done
            }
            // ---- this strip's small records (own bucket + the two boundary buckets) ----
            const uint32_t total = pref[BLOCK];
            const bool listed = total <= (uint32_t)SLOT_CAP;
            // The usual bin -- every record fits one trip (six per lane), no big records -- keeps its records AND their
            // normals in registers across the barrier between the two passes: the second pass then issues no loads at all
            // (it used to re-read the records and only then gather the winners' normals: two dependent round trips of a
            // kernel whose front part is a chain of them).  Measured: 28.4 -> 27.5 us per launch with six records per
            // lane; four cover too few bins (no gain), eight cost the kernel its occupancy (28.8 us, and 31.6 without
            // the path) -- FR_RESOLVE_OPT=0 turns it off.
            if (pass == 0 && one_chunk && a.resolve_opt && total <= (uint32_t)(BLOCK * 6) && prefb[BLOCK] == 0 && listed) {
                constexpr int RF = 6;
                uint4 r[RF];
                float4 nv[RF];
#pragma unroll
                for (int u = 0; u < RF; u++) {
                    const uint32_t j = tid + u * BLOCK;
                    r[u] = make_uint4(0, 0, 0, 0);
                    nv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < total) {
                        const uint32_t slot = slotlist[j];
                        r[u] = Rbase[2 * slot];
                        nv[u] = Nbase[2 * slot];
                    }
                }
                pr.template stamp<2>();   // (issue of the record + normal loads)
                uint32_t msk[RF];
                int p0s[RF];
#pragma unroll
                for (int u = 0; u < RF; u++) {
                    const unsigned long long key = ((unsigned long long)r[u].y << 32) | r[u].x;
                    const int x0 = (int)(r[u].z & 0xFFFFu), y0 = (int)(r[u].z >> 16);
                    p0s[u] = (y0 - r0) * W + x0;
                    uint32_t m = r[u].w;
                    m &= window_rows_below(r1 - y0) & ~window_rows_below(r0 - y0);
                    msk[u] = m;
                    while (m) {
                        const int bit = __ffs((int)m) - 1;
                        m &= m - 1;
                        atomicMax(keys + (p0s[u] + (bit >> 3) * W + (bit & 7)), key);
                    }
                }
                pr.template stamp<3>();   // records back, LDS max done
                __syncthreads();
                pr.template stamp<4>();
#pragma unroll
                for (int u = 0; u < RF; u++) {
                    const unsigned long long key = ((unsigned long long)r[u].y << 32) | r[u].x;
                    uint32_t m = msk[u];
                    if (m) {
                        const float4 nq = FUSED ? post_normal(nv[u]) : nv[u];
                        while (m) {
                            const int bit = __ffs((int)m) - 1;
                            m &= m - 1;
                            const int px = p0s[u] + (bit >> 3) * W + (bit & 7);
                            if (keys[px] == key) store3(nplane + NSTRIDE * (ptrdiff_t)px, nq.x, nq.y, nq.z);
                        }
                    }
                }
                pass = 2;   // both passes done
                break;
            }
            for (uint32_t j0 = tid; j0 < total; j0 += BLOCK * RU) {
                uint4 r[RU];
                uint32_t slot[RU];
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const uint32_t j = j0 + u * BLOCK;
                    r[u] = make_uint4(0, 0, 0, 0);
                    slot[u] = 0;
                    if (j < total) {
                        if (listed) {
                            slot[u] = slotlist[j];
                        } else {
                            int k = 0;
#pragma unroll
                            for (int step = BLOCK >> 1; step > 0; step >>= 1)
                                if (pref[k + step] <= j) k += step;  // largest k with pref[k] <= j
                            slot[u] = (uint32_t)k * SEG + lo16[k] + (j - pref[k]);
                        }
                        r[u] = Rbase[2 * slot[u]];
                    }
                }
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const unsigned long long key = ((unsigned long long)r[u].y << 32) | r[u].x;
                    const int x0 = (int)(r[u].z & 0xFFFFu), y0 = (int)(r[u].z >> 16);
                    const int p0 = (y0 - r0) * W + x0;  // may be negative for a window that starts in the strip above
                    uint32_t m = r[u].w;  // 0 for the slots past the end
                    // rows of the 8x4 window that belong to this strip (all four unless the window straddles a boundary)
                    m &= window_rows_below(r1 - y0) & ~window_rows_below(r0 - y0);
                    if (pass == 0) {
                        while (m) {
                            const int bit = __ffs((int)m) - 1;
                            m &= m - 1;
                            atomicMax(keys + (p0 + (bit >> 3) * W + (bit & 7)), key);
                        }
                    } else {
                        uint32_t won = 0;
                        while (m) {
                            const int bit = __ffs((int)m) - 1;
                            m &= m - 1;
                            if (keys[p0 + (bit >> 3) * W + (bit & 7)] == key) won |= 1u << bit;
                        }
                        if (won) {
                            float4 nv = Nbase[2 * slot[u]];
                            if (FUSED) nv = post_normal(nv);
                            while (won) {
                                const int bit = __ffs((int)won) - 1;
                                won &= won - 1;
                                float* np = nplane + NSTRIDE * (ptrdiff_t)(p0 + (bit >> 3) * W + (bit & 7));
                                store3(np, nv.x, nv.y, nv.z);
                            }
                        }
                    }
                }
            }
            // ---- the face's big records (bucket 0 of every segment; none on a mesh of sub-pixel triangles) ----
            const uint32_t totalb = prefb[BLOCK];
            for (uint32_t j = tid; j < totalb; j += BLOCK) {
                int k = 0;
#pragma unroll
                for (int step = BLOCK >> 1; step > 0; step >>= 1)
                    if (prefb[k + step] <= j) k += step;
                const uint32_t slot = (uint32_t)k * SEG + (j - prefb[k]);
                const uint4 r = Rbase[2 * slot];
                const int t = (int)(0xFFFFFFFFu - r.x);
                if (pass == 0)
                    raster_triangle_into_strip<false>(t, a.tri, vx, vy, vz, a.nver, a.ntri, a.H, W, r0, r1, keys);
                else
                    raster_triangle_into_strip<true>(t, a.tri, vx, vy, vz, a.nver, a.ntri, a.H, W, r0, r1, keys, nplane,
                                                     FUSED ? post_normal(Nbase[2 * slot]) : Nbase[2 * slot], NSTRIDE);
            }
            __syncthreads();
        }
    }
    pr.template stamp<5>();   // winners' normals stored
    if (FUSED)
        write_strip_fused<BLOCK>(a, b, r0, npix, keys);
    else
        write_strip<BLOCK, true, 8>(a, b, r0, npix, keys, vx, vy, vz);
    pr.finish(0);
}

template <int BLOCK, bool FUSED = false, class PR = NoEmitProbe>
__global__ __launch_bounds__(BLOCK) void resolve_write_kernel(RenderArgs a) {
    PR pr;
    pr.begin();
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    resolve_body<BLOCK, FUSED, PR>(a, xcd_remap(blockIdx.x, gridDim.x), keys, pr);
}

// ---- backward: zeros + scatter-add of g/3 to the z row (render_depth_op.cc:345-363) -------------------------
// The reference is a serial loop (one fixed summation order); float atomics would make the per-vertex order depend on
// the schedule.  Here the sum is made ORDER-INDEPENDENT instead: every contribution c = (g * 1.0f) / 3.0f (fp32, as
// :361 computes it) is converted EXACTLY to a 64-bit fixed-point integer (c * 2^k is exact in double; one llrint), the
// integers are added with LDS integer atomics (associative => bit-reproducible whatever the order), and the total is
// rounded to fp32 once.  k is chosen per face from max|g| so that the largest term has at least 38 significant bits below
// the int64 headroom the H*W*3 possible terms need: every term is represented to 2^-39 of the face's largest term, i.e.
// the result is the exactly rounded real sum up to  n_terms * 2^-39 * max|c|  -- at least as close to the real-number sum
// as the reference's sequential fp32 order (whose error grows with the partial sums), and identical run to run.
// One workgroup owns one (face, vertex range) pair: it scans ALL the face's pixels (L2-resident planes) and keeps only
// the contributions that land in its range, so no two workgroups ever add to the same vertex and nothing needs zeroing:
// the owner writes its range of all three rows (x and y rows: zeros, render_depth_op.cc:359-363).
// A face whose gradients contain Inf / NaN cannot be scaled: it takes fp32 LDS atomics (the class of the result --
// NaN / +-Inf -- does not depend on the order).
constexpr int BWD_BLOCK = 1024;
constexpr int BWD_RANGE_MAX = 16 * 1024;  // vertices per owner workgroup (8 B each: 128 KiB of LDS)

struct BwdRenderArgs {
    const float* depth_grad;  // [B,H,W,1]
    const int4* rec;          // [B,H,W] per-pixel records {p1,p2,p3,g bits} (bwd_records_kernel), or null: float ids
    const uint2* partial;     // [B,chunks] {largest |g| bits, Inf/NaN flag} of each 1,024-pixel chunk
    int B, chunks;
    const float* tri;         // [3,ntri]
    const float* tri_ind;     // [B,H,W,1]
    float* vertex_grad;       // [B,3,nver]
    int nver, ntri, npix;     // npix = H*W
    int splits, range;        // owner workgroups per face, vertices per owner
    int shift;                // headroom bits given up by images above 2^20 pixels: ceil(log2 npix) - 20, else 0
};

// the three vertex ids of pixel value `tv` (a float-stored triangle index, -1 on the background): false when the pixel
// contributes nothing (deviation 2: tri_ind < 0; deviation 3: an id outside [0,nver))
__device__ __forceinline__ int bwd_tri_of(float tv, int ntri) {
    const int t = f2i_x86(tv);
    return (t >= 0 && t < ntri) ? t : -1;
}

// Per-face scan: largest |g| over the covered pixels + an Inf/NaN flag, by one 1,024-thread
// workgroup, no atomics (nothing to zero).  Used by the plain (no-workspace) variant.
__device__ __forceinline__ void bwd_face_max(const BwdRenderArgs& a, int b, uint32_t* red /*[2 * BWD_BLOCK / 64]*/, uint2* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npix = a.npix;
    const float* __restrict__ g = a.depth_grad + (size_t)b * npix;
    const float* __restrict__ ti = a.tri_ind + (size_t)b * npix;
    uint32_t m = 0, bad = 0;
    constexpr int PU = 8;  // pixels per lane per trip, all loads issued before the first use
    for (int i0 = tid; i0 < npix; i0 += PU * BWD_BLOCK) {
        float gq[PU], tq[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int i = min(i0 + u * BWD_BLOCK, npix - 1);
            gq[u] = g[i];
            tq[u] = ti[i];
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            if (i0 + u * BWD_BLOCK < npix && bwd_tri_of(tq[u], a.ntri) >= 0) {
                const uint32_t v = __float_as_uint(gq[u]) & 0x7FFFFFFFu;
                if (v >= 0x7F800000u) bad = 1; else m = max(m, v);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        m = max(m, (uint32_t)__shfl_xor((int)m, d));
        bad |= (uint32_t)__shfl_xor((int)bad, d);
    }
    if (lane == 0) { red[wave] = m; red[BWD_BLOCK / 64 + wave] = bad; }
    __syncthreads();
    if (tid == 0) {
        m = 0; bad = 0;
        for (int w = 0; w < BWD_BLOCK / 64; w++) { m = max(m, red[w]); bad |= red[BWD_BLOCK / 64 + w]; }
        *out = make_uint2(m, bad);
    }
}

// Pre-kernel of the workspace variant: ONE pass over all pixels of the batch resolves each pixel's triangle to its three
// vertex ids (the scattered gathers, done once instead of once per owner workgroup) and writes a 16-byte record
// {p1, p2, p3, g bits} per pixel -- p1 = -1 for pixels that contribute nothing (background, bad ids).  The owners then
// STREAM the records.  Lane-consecutive pixels: a gather instruction's 64 lanes hold neighbouring triangles.
constexpr int REC_PX = 1024;  // pixels per records-kernel workgroup (256 threads x 4)
__global__ __launch_bounds__(256) void bwd_records_kernel(BwdRenderArgs a, int4* rec, uint2* partial, int chunks) {
    __shared__ uint32_t red[8];
    const int b = (int)blockIdx.x / chunks, ch = (int)blockIdx.x - b * chunks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ tri0 = a.tri;
    const float* __restrict__ tri1 = a.tri + a.ntri;
    const float* __restrict__ tri2 = a.tri + 2 * (size_t)a.ntri;
    const float* __restrict__ g = a.depth_grad + (size_t)b * a.npix;
    const float* __restrict__ ti = a.tri_ind + (size_t)b * a.npix;
    int4* __restrict__ out = rec + (size_t)b * a.npix;
    constexpr int PU = REC_PX / 256;
    const int i0 = ch * REC_PX + tid;
    float gq[PU], tq[PU];
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const int i = min(i0 + u * 256, a.npix - 1);
        gq[u] = g[i];
        tq[u] = ti[i];
    }
    int t[PU];
    float f[PU][3];
#pragma unroll
    for (int u = 0; u < PU; u++) {
        t[u] = bwd_tri_of(tq[u], a.ntri);
        const int tt = max(t[u], 0);
        f[u][0] = tri0[tt]; f[u][1] = tri1[tt]; f[u][2] = tri2[tt];
    }
    uint32_t m = 0, bad = 0;
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const int i = i0 + u * 256;
        if (i < a.npix) {
            const int p1 = f2i_x86(f[u][0]), p2 = f2i_x86(f[u][1]), p3 = f2i_x86(f[u][2]);
            const bool ok = t[u] >= 0 && (unsigned)p1 < (unsigned)a.nver && (unsigned)p2 < (unsigned)a.nver &&
                            (unsigned)p3 < (unsigned)a.nver;
            out[i] = make_int4(ok ? p1 : -1, p2, p3, (int)__float_as_uint(gq[u]));
            if (t[u] >= 0) {  // the face's largest |g| over the covered pixels (the predicate of bwd_face_max), in parts
                const uint32_t v = __float_as_uint(gq[u]) & 0x7FFFFFFFu;
                if (v >= 0x7F800000u) bad = 1; else m = max(m, v);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        m = max(m, (uint32_t)__shfl_xor((int)m, d));
        bad |= (uint32_t)__shfl_xor((int)bad, d);
    }
    if (lane == 0) { red[wave] = m; red[4 + wave] = bad; }
    __syncthreads();
    if (tid == 0)
        partial[(size_t)b * chunks + ch] = make_uint2(max(max(red[0], red[1]), max(red[2], red[3])),
                                                      red[4] | red[5] | red[6] | red[7]);
}

// PACKED (workspace variant): the owners stream the per-pixel records of bwd_records_kernel (the id gathers -- repeated by
// every owner of the face, they are what the plain variant spends its time on -- were done once); otherwise float ids
// gathered in-kernel.
template <bool PACKED>
__global__ __launch_bounds__(BWD_BLOCK) void render_backward_kernel(BwdRenderArgs a) {
    // all LDS is dynamic (the launcher raises the dynamic limit to the CU's full 160 KiB, which leaves no room for
    // static objects): [range] accumulators, then two small per-wave reduction arrays
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];  // [range]
    uint32_t* red = reinterpret_cast<uint32_t*>(acc + a.range);               // [2 * BWD_BLOCK / 64]
    const int tid = threadIdx.x;
    // block -> (face, owner): blocks that share blockIdx % 8 share an XCD / L2; when the batch is a multiple of 8 the
    // owners of a face are given ids of one residue class, so the face's planes / records are fetched into ONE L2 and
    // re-read there, instead of once per owner
    int b, sp;
    if ((a.B & 7) == 0) {
        const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
        b = (q / a.splits) * 8 + xcd;
        sp = q % a.splits;
    } else {
        b = (int)blockIdx.x / a.splits;
        sp = (int)blockIdx.x - b * a.splits;
    }
    const int v0 = sp * a.range;
    const int v1 = min(a.nver, v0 + a.range);
    const int npix = a.npix, ntri = a.ntri, nver = a.nver;
    const float* __restrict__ g = a.depth_grad + (size_t)b * npix;
    const float* __restrict__ ti = a.tri_ind + (size_t)b * npix;
    const float* __restrict__ tri0 = a.tri;
    const float* __restrict__ tri1 = a.tri + ntri;
    const float* __restrict__ tri2 = a.tri + 2 * (size_t)ntri;
    for (int i = tid; i < v1 - v0; i += BWD_BLOCK) acc[i] = 0ull;

    // the face's largest |g| over the covered pixels (max is order independent); c = g/3 is at most two binades below,
    // which the scale accounts for -- so the scan needs no division
    uint32_t m, bad;
    const int4* __restrict__ rec = PACKED ? a.rec + (size_t)b * npix : nullptr;
    if constexpr (PACKED) {
        // the records kernel left the face's largest |g| in parts: one per 1,024-pixel chunk
        m = 0; bad = 0;
        for (int c = tid; c < a.chunks; c += BWD_BLOCK) {
            const uint2 pm = a.partial[(size_t)b * a.chunks + c];
            m = max(m, pm.x); bad |= pm.y;
        }
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            m = max(m, (uint32_t)__shfl_xor((int)m, d));
            bad |= (uint32_t)__shfl_xor((int)bad, d);
        }
        if (lane == 0) { red[wave] = m; red[BWD_BLOCK / 64 + wave] = bad; }
        __syncthreads();
        m = 0; bad = 0;
#pragma unroll
        for (int w = 0; w < BWD_BLOCK / 64; w++) { m = max(m, red[w]); bad |= red[BWD_BLOCK / 64 + w]; }
    } else {
        uint2* slot = reinterpret_cast<uint2*>(red + 2 * (BWD_BLOCK / 64));
        bwd_face_max(a, b, red, slot);
        __syncthreads();
        m = slot->x; bad = slot->y;
    }
    float* gx = a.vertex_grad + (size_t)b * 3 * nver;
    float* gy = gx + nver;
    float* gz = gy + nver;
    // scale 2^k from e = floor(log2 max|g|): the largest term c = g/3 lands in [2^38, 2^40); up to 2^21 terms (3 per
    // pixel) stay below 2^62
    const int e = (int)(m >> 23) - 127;  // floor(log2 max|g|) for a normal float; -127 for subnormals / zero
    const double scale = ldexp(1.0, 40 - a.shift - e);
    const double inv_scale = ldexp(1.0, e - 40 + a.shift);
    float* facc = reinterpret_cast<float*>(acc);  // Inf / NaN gradients: fp32 LDS atomics in the same buffer
    if (bad) {
        __syncthreads();
        for (int i = tid; i < v1 - v0; i += BWD_BLOCK) facc[i] = 0.0f;
    }
    __syncthreads();
    // one contribution: c = (g * 1.0f) / 3.0f to the three vertices of triangle t that this workgroup owns
    auto add = [&](float gv, int p1, int p2, int p3, bool ids_ok) {
        if (!ids_ok) return;
        // ownership first: every owner of the face sees every pixel, but only ~1 / splits of them land in its range --
        // the division and the fixed-point conversion are done for those only
        const bool in1 = p1 >= v0 && p1 < v1, in2 = p2 >= v0 && p2 < v1, in3 = p3 >= v0 && p3 < v1;
        if (!(in1 || in2 || in3)) return;
        const float c = gv * 1.0f / 3.0f;
        if (bad) {
            if (in1) atomicAdd(&facc[p1 - v0], c);
            if (in2) atomicAdd(&facc[p2 - v0], c);
            if (in3) atomicAdd(&facc[p3 - v0], c);
        } else {
            const unsigned long long q = (unsigned long long)__double2ll_rn((double)c * scale);  // exact product, one rounding
            if (q == 0ull) return;
            if (in1) atomicAdd(&acc[p1 - v0], q);
            if (in2) atomicAdd(&acc[p2 - v0], q);
            if (in3) atomicAdd(&acc[p3 - v0], q);
        }
    };
    if ((m != 0 || bad) && PACKED) {
        // the owners stream the face's records: no gathers, no dependent loads
        constexpr int QU = 8;
        for (int i0 = tid; i0 < npix; i0 += QU * BWD_BLOCK) {
            int4 rq[QU];
#pragma unroll
            for (int u = 0; u < QU; u++) rq[u] = rec[min(i0 + u * BWD_BLOCK, npix - 1)];
#pragma unroll
            for (int u = 0; u < QU; u++)
                if (i0 + u * BWD_BLOCK < npix) add(__uint_as_float((uint32_t)rq[u].w), rq[u].x, rq[u].y, rq[u].z, rq[u].x >= 0);
        }
    }
    if ((m != 0 || bad) && !PACKED) {
        // The scan.  Lane l of a trip's u-th slice takes pixel i0 + u * BLOCK: the 64 lanes of a gather instruction hold
        // 64 CONSECUTIVE pixels -> neighbouring triangles -> a few cache lines of the id table per instruction (four
        // pixels per lane, the obvious 16-byte-load mapping, puts every lane of a gather on its own line and runs the
        // texture addresser at one lane per cycle).  Software pipelined: the (g, tri_ind) values of the NEXT trip are
        // requested before the id gathers of the current one are consumed.
        constexpr int QU = 8;
        float gv[QU], tv[QU];
#pragma unroll
        for (int u = 0; u < QU; u++) {
            const int i = tid + u * BWD_BLOCK;
            gv[u] = 0.f; tv[u] = -1.f;
            if (i < npix) { gv[u] = g[i]; tv[u] = ti[i]; }
        }
        for (int i0 = tid; i0 < npix; i0 += QU * BWD_BLOCK) {
            int t[QU], id[QU][3];
            bool ok[QU];
            float gc[QU];
#pragma unroll
            for (int u = 0; u < QU; u++) {
                t[u] = bwd_tri_of(tv[u], ntri);
                gc[u] = gv[u];
                const int tt = max(t[u], 0);
                id[u][0] = f2i_x86(tri0[tt]); id[u][1] = f2i_x86(tri1[tt]); id[u][2] = f2i_x86(tri2[tt]);
                ok[u] = (unsigned)id[u][0] < (unsigned)nver && (unsigned)id[u][1] < (unsigned)nver &&
                        (unsigned)id[u][2] < (unsigned)nver;
            }
#pragma unroll
            for (int u = 0; u < QU; u++) {
                const int in = i0 + (QU + u) * BWD_BLOCK;
                tv[u] = -1.f;
                if (in < npix) { gv[u] = g[in]; tv[u] = ti[in]; }
            }
#pragma unroll
            for (int u = 0; u < QU; u++)
                if (t[u] >= 0) add(gc[u], id[u][0], id[u][1], id[u][2], ok[u]);
        }
    }
    __syncthreads();
    for (int i = tid; i < v1 - v0; i += BWD_BLOCK) {
        gx[v0 + i] = 0.0f;
        gy[v0 + i] = 0.0f;
        // fixed point: one rounding to 24 bits (int64 -> fp32), then an exact power-of-two scaling in double
        gz[v0 + i] = bad ? facc[i] : (float)((double)(float)(long long)acc[i] * inv_scale);
    }
}

}  // namespace fr

namespace {
struct RenderGeom {
    int rows, strips, nseg;
    size_t lds, recs_bytes, segoff_bytes, nrm_bytes, tri4_bytes;
    bool binned_ok;
};
constexpr size_t kLdsMax = 160 * 1024;

// bins: enough workgroups to cover the 256 CUs a few times over, never more rows than fit in LDS
RenderGeom render_geom(int B, int ntri, int H, int W, int rows_override) {
    RenderGeom g{};
    const size_t row_bytes = (size_t)W * sizeof(unsigned long long);
    int rows_max = row_bytes ? (int)((kLdsMax - fr::resolve_scratch_bytes(1024)) / row_bytes) : H;
    if (rows_max < 1) rows_max = 0;  // a row does not fit: unsupported
    int want_strips = B > 0 ? (1280 + B - 1) / B : 1;  // ~5 resolver workgroups of 256 threads per CU (10-row strips at B = 64)
    if (want_strips > fr::MAX_STRIPS) want_strips = fr::MAX_STRIPS;  // two buckets per strip must fit the offset table
    int rows = H > 0 ? (H + want_strips - 1) / want_strips : 1;
    if (rows < fr::SMALL_H) rows = fr::SMALL_H;
    if (rows_override > 0) rows = rows_override;  // tuning override (FR_RENDER_ROWS)
    if (rows > rows_max) rows = rows_max;
    if (rows > H) rows = H;
    if (rows < 1) rows = 1;
    g.rows = rows;
    g.strips = H > 0 ? (H + rows - 1) / rows : 0;
    g.nseg = (ntri + fr::SEG - 1) / fr::SEG;
    g.lds = (size_t)rows * row_bytes;
    g.recs_bytes = (size_t)B * g.nseg * fr::SEG * sizeof(uint4);
    g.segoff_bytes = (size_t)B * g.nseg * fr::OFF_STRIDE * sizeof(uint16_t);
    g.nrm_bytes = (size_t)B * g.nseg * fr::SEG * sizeof(float4);  // per-record normals; also bounds the tritex table
    g.tri4_bytes = (2 * (size_t)g.nseg * fr::SEG + 1) * sizeof(int4);  // triangle table by id + header slot + the table in lane order
    // An 8x4 hit window may touch at most TWO strips (its own bucket or the boundary bucket between them): with strips
    // shorter than the window (a very wide image, or the override) it could span three and the emit kernel's bucket
    // choice would drop hits -- such shapes take the scan path instead.
    const bool window_fits = rows >= fr::SMALL_H || g.strips <= 1;
    g.binned_ok = rows_max >= 1 && window_fits && g.strips <= fr::MAX_STRIPS && H <= 0xFFFF && W <= 0xFFFF &&
                  (long long)B * g.nseg <= 0x7FFFFFFFll;
    return g;
}
RenderGeom render_geom(int B, int ntri, int H, int W) { return render_geom(B, ntri, H, W, fr::opt(fr::OPT_RENDER_ROWS)); }
}  // namespace

// test hook (tests/test_render_gpu.py): div3(x) against x / 3.0f on the bit patterns [first, first + count)
namespace fr {
__global__ __launch_bounds__(256) void div3_sweep_kernel(unsigned long long first, unsigned long long count,
                                                         unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * 256) {
        const float x = __uint_as_float((uint32_t)(first + i));
        const float q = div3(x), w = x / 3.0f;
        const bool same = __float_as_uint(q) == __float_as_uint(w) || (q != q && w != w);
        bad += same ? 0 : 1;
    }
    if (bad) atomicAdd(mismatches, bad);
}
}  // namespace fr
extern "C" int fr_debug_div3_sweep(unsigned long long first, unsigned long long count, unsigned long long* mismatches,
                                   void* hip_stream) {
    hipStream_t st = (hipStream_t)hip_stream;
    if (hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), st) != hipSuccess) return FR_ERR_LAUNCH;
    hipLaunchKernelGGL(fr::div3_sweep_kernel, dim3(4096), dim3(256), 0, st, first, count, mismatches);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

// test hook (tests/test_capi_cpu.py): the strip geometry the launcher would choose, without a GPU.
// out = {rows, strips, nseg, binned_ok}
extern "C" void fr_debug_render_geom(int B, int ntri, int H, int W, int rows_override, int* out) {
    RenderGeom g = render_geom(B, ntri, H, W, rows_override);
    out[0] = g.rows; out[1] = g.strips; out[2] = g.nseg; out[3] = g.binned_ok ? 1 : 0;
}

extern "C" int fr_render_depth_strip_rows(int B, int ntri, int H, int W) {
    if (B <= 0 || ntri <= 0 || H <= 0 || W <= 0) return 0;
    const RenderGeom g = render_geom(B, ntri, H, W, 0);   // (the library's own choice: no FR_RENDER_ROWS override)
    return g.binned_ok ? g.rows : 0;
}

size_t fr_render_workspace_bytes_impl(int B, int ntri, int H, int W) {
    if ((size_t)B * H * W == 0 || ntri == 0) return 0;
    RenderGeom g = render_geom(B, ntri, H, W);
    if (!g.binned_ok) return 0;
    return g.recs_bytes + g.segoff_bytes + 2 * g.nrm_bytes + g.tri4_bytes;
}

template <int BLK, bool FUSED>
static int launch_resolve(const fr::RenderArgs& a, long long nbins, size_t lds, hipStream_t stream) {
    static fr_lds_flags_t ok[64];
    if (fr_allow_full_lds(reinterpret_cast<const void*>(&fr::resolve_write_kernel<BLK, FUSED>), ok) != hipSuccess)
        return FR_ERR_LAUNCH;
    hipLaunchKernelGGL((fr::resolve_write_kernel<BLK, FUSED>), dim3((unsigned)nbins), dim3(BLK),
                       lds + fr::resolve_scratch_bytes(BLK), stream, a);
    return FR_OK;
}

static int launch_render_impl(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri, int H,
                              int W, int tex_batch, float* depth, float* tex_img, float* normal, float* tri_ind,
                              const float* im_gray, float* net_in, float* depth_img, void* workspace, size_t ws_bytes,
                              hipStream_t stream, int phases = 7, long long vpitch = 0, int rows_hint = 0);

int fr_launch_render_forward(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                             int H, int W, int tex_batch, float* depth, float* tex_img, float* normal,
                             float* tri_ind, void* workspace, size_t ws_bytes, hipStream_t stream) {
    return launch_render_impl(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal, tri_ind, nullptr,
                              nullptr, nullptr, workspace, ws_bytes, stream);
}

// the forward op phase by phase: 4 = pack the triangle list into the workspace, 1 = emit, 2 = resolve (7 = the whole op)
int fr_launch_render_forward_phases(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                                    int H, int W, int tex_batch, float* depth, float* tex_img, float* normal,
                                    float* tri_ind, void* workspace, size_t ws_bytes, hipStream_t stream, int phases,
                                    long long vpitch, int rows_hint) {
    return launch_render_impl(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal, tri_ind, nullptr,
                              nullptr, nullptr, workspace, ws_bytes, stream, phases, vpitch, rows_hint);
}

int fr_launch_rendering_layer(const float* vertex, const float* tri, const float* texture, const float* im_gray, int B,
                              int nver, int ntri, int H, int W, int tex_batch, float* net_in, float* depth_img,
                              float* depth, float* tri_ind, void* workspace, size_t ws_bytes, hipStream_t stream, int phases) {
    return launch_render_impl(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, nullptr, nullptr, tri_ind, im_gray,
                              net_in, depth_img, workspace, ws_bytes, stream, phases);
}

// Argument block + geometry of one forward call (everything the kernels read); *binned = the binned rasteriser serves it.
static int prepare_render(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri, int H, int W,
                          int tex_batch, float* depth, float* tex_img, float* normal, float* tri_ind, const float* im_gray,
                          float* net_in, float* depth_img, void* workspace, size_t ws_bytes, long long vpitch,
                          fr::RenderArgs& a, RenderGeom& g, bool* binned_out, int rows_override = -1) {
    using namespace fr;
    const bool fused = net_in != nullptr;
    if ((size_t)W * sizeof(unsigned long long) > kLdsMax) return FR_ERR_UNSUPPORTED;
    g = rows_override > 0 ? render_geom(B, ntri, H, W, rows_override) : render_geom(B, ntri, H, W);
    if ((long long)B * g.strips > 0x7FFFFFFFll) return FR_ERR_UNSUPPORTED;
    a.vertex = vertex; a.tri = tri; a.texture = texture;
    a.vpitch = vpitch > 0 ? vpitch : nver;
    a.depth = depth; a.tex_img = tex_img; a.normal = normal; a.tri_ind = tri_ind;
    a.B = B; a.nver = nver; a.ntri = ntri; a.H = H; a.W = W;
    a.rows = g.rows; a.strips = g.strips;
    a.tex_stride = (tex_batch == 1) ? 0 : 3ll * nver;
    a.recs = nullptr; a.segoff = nullptr; a.nseg = g.nseg;
    a.tritex_ws = nullptr;
    a.im_gray = im_gray; a.net_in = net_in; a.depth_img = depth_img;
    a.wm1 = (float)(W - 1);
    a.hm1 = (float)(H - 1);
    a.resolve_opt = opt(OPT_RESOLVE_OPT);
    a.use_filter = opt(OPT_EMIT_FILTER);  // bit 0: certified fp32 inside test, bit 1: single-pixel pre-cull in phase A
    a.rows_magic = g.rows > 1 ? (uint32_t)((0x100000000ull + (unsigned)g.rows - 1) / (unsigned)g.rows) : 0u;
    a.tri4 = nullptr; a.nseg_magic = 0;
    const bool binned = g.binned_ok && ntri > 0 && opt(OPT_RENDER_IMPL) != 1;
    *binned_out = binned;
    if (fused && !binned) return FR_ERR_UNSUPPORTED;  // the caller falls back to the unfused op + elementwise post-processing
    if (!binned) return FR_OK;
    if (ws_bytes < g.recs_bytes + g.segoff_bytes + 2 * g.nrm_bytes + g.tri4_bytes || !workspace ||
        ((uintptr_t)workspace & 15))
        return FR_ERR_WORKSPACE;
    char* wsp = reinterpret_cast<char*>(workspace);
    a.recs = reinterpret_cast<uint4*>(wsp);   // records interleaved with their normals: recs_bytes + nrm_bytes
    a.segoff = reinterpret_cast<uint16_t*>(wsp + g.recs_bytes + g.nrm_bytes);
    a.tritex_ws = reinterpret_cast<float4*>(wsp + g.recs_bytes + g.segoff_bytes + g.nrm_bytes);
    a.tri4 = reinterpret_cast<int4*>(wsp + g.recs_bytes + g.segoff_bytes + 2 * g.nrm_bytes);
    // lid / nseg through the 2^32 reciprocal is exact while lid * (magic * nseg - 2^32) < 2^32, i.e. B * nseg^2 < 2^32
    a.nseg_magic = ((unsigned long long)B * g.nseg * g.nseg < 0x100000000ull && g.nseg > 1)
                       ? (uint32_t)((0x100000000ull + (unsigned)g.nseg - 1) / (unsigned)g.nseg) : 0u;
    return FR_OK;
}

static int launch_render_impl(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri, int H,
                              int W, int tex_batch, float* depth, float* tex_img, float* normal, float* tri_ind,
                              const float* im_gray, float* net_in, float* depth_img, void* workspace, size_t ws_bytes,
                              hipStream_t stream, int phases, long long vpitch, int rows_hint) {
    using namespace fr;
    const bool fused = net_in != nullptr;
    constexpr int BLOCK = 1024;
    if (nver == 0) ntri = 0;  // no vertex can be valid: every triangle is skipped, the planes are pure background
    RenderArgs a;
    RenderGeom g;
    bool binned = false;
    // The caller's strip-height hint (fr_decode_render_forward, phase bits 8-15) is taken only where the binned rasteriser serves
    // the resulting geometry; anything else keeps the library's own choice.  No workspace size depends on the strip height and
    // no result bit does (tests hold every height to the oracle): the hint only changes how many resolver workgroups there are.
    // (... and only where the library's own geometry is binned too: that is what the caller's workspace was sized for)
    if (rows_hint > 0 && !(render_geom(B, ntri, H, W, rows_hint).binned_ok && render_geom(B, ntri, H, W).binned_ok)) rows_hint = 0;
    const int prc = prepare_render(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal, tri_ind, im_gray,
                                   net_in, depth_img, workspace, ws_bytes, vpitch, a, g, &binned, rows_hint > 0 ? rows_hint : -1);
    if (prc != FR_OK) return prc;
    const long long nbins = (long long)B * g.strips;
    if (!binned) {
        if (!(phases & 2)) return FR_OK;  // the fallback is a single kernel: it counts as the resolve phase
        static fr_lds_flags_t lds_ok[64];
        if (fr_allow_full_lds(reinterpret_cast<const void*>(&render_strip_kernel<BLOCK>), lds_ok) != hipSuccess)
            return FR_ERR_LAUNCH;
        hipLaunchKernelGGL(render_strip_kernel<BLOCK>, dim3((unsigned)nbins), dim3(BLOCK), g.lds, stream, a);
        return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
    }
    if (phases & 4) {
        int4* tri4 = const_cast<int4*>(a.tri4);
        hipLaunchKernelGGL(pack_tri_kernel, dim3((unsigned)g.nseg), dim3(256), 0, stream, tri, nver, ntri, tri4,
                           tri4 + (size_t)g.nseg * SEG, tri4 + (size_t)g.nseg * SEG + 1, opt(OPT_EMIT_ORDER));
    }
    if (phases & 1)
        hipLaunchKernelGGL(raster_emit_kernel<NoEmitProbe>, dim3((unsigned)((long long)B * g.nseg)), dim3(EMIT_BLOCK), 0, stream, a);
    if (!(phases & 2)) return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
    // 256-thread resolvers when several of them fit a CU's LDS side by side, 512 threads for wide strips
    const int rblk = opt(OPT_RESOLVE_BLOCK) > 0 ? opt(OPT_RESOLVE_BLOCK) : (g.lds <= 32 * 1024 ? 256 : 512);
    int rc;
    if (rblk == 1024)
        rc = fused ? launch_resolve<1024, true>(a, nbins, g.lds, stream) : launch_resolve<1024, false>(a, nbins, g.lds, stream);
    else if (rblk == 512)
        rc = fused ? launch_resolve<512, true>(a, nbins, g.lds, stream) : launch_resolve<512, false>(a, nbins, g.lds, stream);
    else
        rc = fused ? launch_resolve<256, true>(a, nbins, g.lds, stream) : launch_resolve<256, false>(a, nbins, g.lds, stream);
    if (rc != FR_OK) return rc;
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

// workspace of the ws variant: one 16-byte record per pixel of the batch + one {max, flag} pair per 1,024-pixel chunk
size_t fr_render_backward_workspace_bytes_impl(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t npix = (size_t)H * W, chunks = (npix + fr::REC_PX - 1) / fr::REC_PX;
    return (size_t)B * npix * sizeof(int4) + (size_t)B * chunks * sizeof(uint2);
}

int fr_launch_render_backward(const float* depth_grad, const float* tri, const float* tri_ind, float* vertex_grad,
                              int B, int nver, int ntri, int H, int W, void* workspace, size_t ws_bytes,
                              hipStream_t stream) {
    using namespace fr;
    const size_t bytes = (size_t)B * 3 * nver * sizeof(float);
    const long long npix = (long long)H * W;
    if (npix * B == 0 || ntri == 0 || nver == 0)
        return (!bytes || hipMemsetAsync(vertex_grad, 0, bytes, stream) == hipSuccess) ? FR_OK : FR_ERR_LAUNCH;
    if (npix > 0x7FFFFFFFll) return FR_ERR_UNSUPPORTED;
    // the int64 headroom covers 3 * 2^20 terms per vertex at the full 38-bit resolution; larger images give up one bit of
    // resolution per doubling (the forward renders them through the scan fallback, so the backward must take them too)
    int shift = 0;
    while ((1ll << (20 + shift)) < npix) shift++;
    // owners per face: enough for the LDS budget, and for ~one workgroup per CU on small batches
    int splits = (nver + BWD_RANGE_MAX - 1) / BWD_RANGE_MAX;
    const int want = (256 + B - 1) / B;
    if (splits < want) splits = want;
    if (splits > nver) splits = nver;
    const int range = (nver + splits - 1) / splits;
    splits = (nver + range - 1) / range;
    if ((long long)B * splits > 0x7FFFFFFFll) return FR_ERR_UNSUPPORTED;
    BwdRenderArgs a;
    a.depth_grad = depth_grad; a.tri = tri; a.tri_ind = tri_ind; a.vertex_grad = vertex_grad;
    a.nver = nver; a.ntri = ntri; a.npix = (int)npix; a.splits = splits; a.range = range; a.shift = shift;
    // with a workspace one pre-kernel resolves every pixel to its vertex ids once (instead of once per owner workgroup)
    // and the owners stream 16-byte records
    const bool packed = workspace && ws_bytes >= fr_render_backward_workspace_bytes_impl(B, H, W) &&
                        (((uintptr_t)workspace) & 15) == 0;
    int4* rec = reinterpret_cast<int4*>(workspace);
    const int chunks = (int)((npix + REC_PX - 1) / REC_PX);
    uint2* partial = reinterpret_cast<uint2*>(rec + (size_t)B * npix);
    a.rec = packed ? rec : nullptr;
    a.partial = packed ? partial : nullptr;
    a.B = B; a.chunks = chunks;
    const size_t lds = (size_t)range * sizeof(unsigned long long) + 2 * (BWD_BLOCK / 64) * sizeof(uint32_t) + 16;
    static fr_lds_flags_t lds_ok[2][64];
    if (packed) {
        if ((long long)B * chunks > 0x7FFFFFFFll) return FR_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(bwd_records_kernel, dim3((unsigned)(B * chunks)), dim3(256), 0, stream, a, rec, partial, chunks);
        if (fr_allow_full_lds(reinterpret_cast<const void*>(&render_backward_kernel<true>), lds_ok[1]) != hipSuccess)
            return FR_ERR_LAUNCH;
        hipLaunchKernelGGL(render_backward_kernel<true>, dim3((unsigned)(B * splits)), dim3(BWD_BLOCK), lds, stream, a);
    } else {
        if (fr_allow_full_lds(reinterpret_cast<const void*>(&render_backward_kernel<false>), lds_ok[0]) != hipSuccess)
            return FR_ERR_LAUNCH;
        hipLaunchKernelGGL(render_backward_kernel<false>, dim3((unsigned)(B * splits)), dim3(BWD_BLOCK), lds, stream, a);
    }
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}
