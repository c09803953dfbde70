// render_depth forward / backward for gfx950 (MI355X).
//
// What it computes: the CPU functor of the reference, rendering_layer/ops_src/render_depth_op.cc:132-322
// (forward) and :325-368 (backward) -- NOT the reference's CUDA kernels (render_depth_op.cu.cc:176-237 race
// on the z-test and are not the numerical spec).
//
// How (MI355X-first, nothing like the reference's three-kernel + 705 MB fp64 scratch pipeline):
//   * one workgroup owns one (face, strip-of-rows) screen bin; the bin's per-pixel 64-bit keys live in LDS
//     (rows*W*8 bytes, up to the whole 160 KiB of a CU);
//   * every lane takes a triangle, gathers its vertices from the face's [3,nver] planes (L2-resident: 638 KB
//     per face), rejects on the strip / bbox with the reference's exact integer rules, runs the fp64
//     barycentric test for the few pixel centres in its bbox and resolves depth with one LDS ds_max_u64 per
//     hit: max over (orderable(h) << 32 | ~tri) == "largest h, ties to the lowest index" -- deterministic
//     and order independent, so no global atomics and no races;
//   * the same workgroup then unpacks the winners and streams the four output planes of its strip with
//     16-byte stores; per-triangle texture means and normals are recomputed from the winner's vertices
//     instead of being materialised per triangle.
// Bound: HBM (1.28 MB of output per face is the dominant algorithmic traffic); see DESIGN.md.
#include "fr_common.h"

namespace fr {

struct RenderArgs {
    const float* vertex;   // [B,3,nver]
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
};

// XCD-aware block remap (bijective for any grid): blocks that share blockIdx%8 share an XCD/L2, so give each
// XCD a contiguous run of (face, strip) bins -- all strips of a face then hit one L2 with that face's vertices.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

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

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void render_strip_kernel(RenderArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int tid = threadIdx.x;
    const int bin = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bin / a.strips;
    const int s = bin - b * a.strips;
    const int r0 = s * a.rows;
    const int r1 = min(a.H, r0 + a.rows);
    const int W = a.W, H = a.H, nver = a.nver, ntri = a.ntri;
    const int npix = (r1 - r0) * W;

    const unsigned long long KBG = bg_key();
    for (int i = tid; i < npix; i += BLOCK) keys[i] = KBG;
    __syncthreads();

    const float* __restrict__ vx = a.vertex + (size_t)b * 3 * nver;
    const float* __restrict__ vy = vx + nver;
    const float* __restrict__ vz = vy + nver;
    const float* __restrict__ tri = a.tri;

    for (int t = tid; t < ntri; t += BLOCK) {
        // vertex ids: (int) truncation of float-stored indices, render_depth_op.cc:204-206
        int p1 = f2i_x86(tri[t]);
        int p2 = f2i_x86(tri[(size_t)ntri + t]);
        int p3 = f2i_x86(tri[2 * (size_t)ntri + t]);
        if ((unsigned)p1 >= (unsigned)nver || (unsigned)p2 >= (unsigned)nver || (unsigned)p3 >= (unsigned)nver)
            continue;  // deviation 3: the reference would read out of bounds
        // rows first: most triangles miss this strip
        float y1 = vy[p1], y2 = vy[p2], y3 = vy[p3];
        int y_min = f2i_x86(ceilf(mn(mn(y1, y2), y3)));   // render_depth_op.cc:279-280
        int y_max = f2i_x86(floorf(mx(mx(y1, y2), y3)));
        if (y_max < y_min || y_max > H - 1 || y_min < 0) continue;  // part of the whole-triangle reject, :282
        int ya = max(y_min, r0), yb = min(y_max, r1 - 1);
        if (ya > yb) continue;
        float x1 = vx[p1], x2 = vx[p2], x3 = vx[p3];
        int x_min = f2i_x86(ceilf(mn(mn(x1, x2), x3)));   // :276-277
        int x_max = f2i_x86(floorf(mx(mx(x1, x2), x3)));
        if (x_max < x_min || x_max > W - 1 || x_min < 0) continue;  // :282
        // centroid depth in fp32, :217
        float h = ((vz[p1] + vz[p2]) + vz[p3]) / 3.0f;
        h = h + 0.0f;                   // -0 -> +0 (serial code treats them as equal)
        if (!(h > bg_depth())) continue;  // NaN or <= background can never pass 'depth < h' (:295)
        const unsigned long long key = make_key(h, t);
        const TriSetup ts = tri_setup(x1, y1, x2, y2, x3, y3);
        for (int y = ya; y <= yb; y++) {
            unsigned long long* row = keys + (size_t)(y - r0) * W;
            for (int x = x_min; x <= x_max; x++) {
                if (point_in_tri(ts, x, y)) atomicMax(&row[x], key);
            }
        }
    }
    __syncthreads();

    // ---- resolve + write the strip's four planes -------------------------------------------------------
    const float* __restrict__ tex = a.texture + (size_t)b * a.tex_stride;
    const size_t pix0 = ((size_t)b * H + r0) * W;  // first pixel of the strip in the [B,H,W] planes
    float* dep = a.depth + pix0;
    float* tin = a.tri_ind + pix0;
    float* txi = a.tex_img + pix0 * 3;
    float* nrm = a.normal + pix0 * 3;
    const bool vec_ok = ((npix & 3) == 0) && ((pix0 & 3) == 0) &&
                        ((((uintptr_t)a.depth | (uintptr_t)a.tri_ind | (uintptr_t)a.tex_img | (uintptr_t)a.normal) & 15) == 0);
    if (vec_ok) {
        for (int g = tid; g < (npix >> 2); g += BLOCK) {
            PixelOut o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = resolve_pixel(keys[4 * g + k], vx, vy, vz, tri, tex, nver, ntri);
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

// ---- backward: zeros + scatter-add of g/3 to the z row (render_depth_op.cc:345-363) -------------------------
__global__ __launch_bounds__(256) void render_backward_kernel(const float* __restrict__ depth_grad,
                                                              const float* __restrict__ tri,
                                                              const float* __restrict__ tri_ind,
                                                              float* __restrict__ vertex_grad, int nver, int ntri,
                                                              long long npix_face, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        int t = f2i_x86(tri_ind[i]);
        if (t < 0 || t >= ntri) continue;  // deviation 2: background pixels carry tri_ind = -1
        int p1 = f2i_x86(tri[t]), p2 = f2i_x86(tri[(size_t)ntri + t]), p3 = f2i_x86(tri[2 * (size_t)ntri + t]);
        if ((unsigned)p1 >= (unsigned)nver || (unsigned)p2 >= (unsigned)nver || (unsigned)p3 >= (unsigned)nver)
            continue;
        float g = depth_grad[i] * 1.0f / 3.0f;  // (g*1.0f)/3.0f, :361
        long long b = i / npix_face;
        float* gz = vertex_grad + ((size_t)b * 3 + 2) * nver;
        atomicAdd(gz + p1, g);
        atomicAdd(gz + p2, g);
        atomicAdd(gz + p3, g);
    }
}

}  // namespace fr

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

int fr_launch_render_forward(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                             int H, int W, int tex_batch, float* depth, float* tex_img, float* normal,
                             float* tri_ind, void* /*workspace*/, size_t /*ws_bytes*/, hipStream_t stream) {
    using namespace fr;
    constexpr int BLOCK = 1024;
    const size_t lds_max = 160 * 1024;
    const size_t row_bytes = (size_t)W * sizeof(unsigned long long);
    if (row_bytes > lds_max) return FR_ERR_UNSUPPORTED;
    int rows_max = (int)(lds_max / row_bytes);
    // bins: enough workgroups to cover the 256 CUs, never more rows than fit in LDS
    int want_strips = (256 + B - 1) / B;
    int rows = (H + want_strips - 1) / want_strips;
    if (rows < 4) rows = H < 4 ? H : 4;
    if (rows > rows_max) rows = rows_max;
    int ov = env_int("FR_RENDER_ROWS", 0);  // tuning override
    if (ov > 0) rows = ov > rows_max ? rows_max : ov;
    if (rows > H) rows = H;
    int strips = (H + rows - 1) / rows;
    long long nbins = (long long)B * strips;
    if (nbins > 0x7FFFFFFFll) return FR_ERR_UNSUPPORTED;

    RenderArgs a;
    a.vertex = vertex; a.tri = tri; a.texture = texture;
    a.depth = depth; a.tex_img = tex_img; a.normal = normal; a.tri_ind = tri_ind;
    a.B = B; a.nver = nver; a.ntri = ntri; a.H = H; a.W = W;
    a.rows = rows; a.strips = strips;
    a.tex_stride = (tex_batch == 1) ? 0 : 3ll * nver;
    size_t lds = (size_t)rows * row_bytes;
    static unsigned char lds_ok[64];
    if (fr_allow_full_lds(reinterpret_cast<const void*>(&render_strip_kernel<BLOCK>), lds_ok) != hipSuccess)
        return FR_ERR_LAUNCH;
    hipLaunchKernelGGL(render_strip_kernel<BLOCK>, dim3((unsigned)nbins), dim3(BLOCK), lds, stream, a);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}

int fr_launch_render_backward(const float* depth_grad, const float* tri, const float* tri_ind, float* vertex_grad,
                              int B, int nver, int ntri, int H, int W, hipStream_t stream) {
    size_t bytes = (size_t)B * 3 * nver * sizeof(float);
    if (bytes && hipMemsetAsync(vertex_grad, 0, bytes, stream) != hipSuccess) return FR_ERR_LAUNCH;
    long long npix = (long long)H * W, total = npix * B;
    if (total == 0 || ntri == 0 || nver == 0) return FR_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fr::render_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, depth_grad, tri,
                       tri_ind, vertex_grad, nver, ntri, npix, total);
    return hipGetLastError() == hipSuccess ? FR_OK : FR_ERR_LAUNCH;
}
