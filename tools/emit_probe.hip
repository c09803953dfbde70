// Development probe for raster_emit_kernel (csrc/fr_render.hip) -- NOT part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -shared \
//         -o tools/libemit_probe.so tools/emit_probe.hip
// Exports fr_probe_emit_stamps: launches a stamped build of the emit kernel on the caller's (real) inputs -- every wave
// records s_memtime at the phase boundaries -- and leaves [workgroups][4 waves][10] stamps in a caller buffer.  Driven by
// tools/emit_probe.py, which reduces the stamps to the phase shares of a workgroup's life.
#include "../3dfacerecon_amd/csrc/fr_render.hip"

namespace fr { int opt(Opt o) { return o == OPT_RESOLVE_OPT ? 2 : o == OPT_EMIT_FILTER ? 3 : 0; } }   // the product's defaults

__device__ unsigned long long* g_emit_stamps;

__device__ __forceinline__ unsigned long long estamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
struct EmitStampProbe {
    static constexpr int abl = 0;
    unsigned long long t[8];
    __device__ __forceinline__ void begin() {
#pragma unroll
        for (int i = 1; i < 8; i++) t[i] = 0;
        t[0] = estamp();
    }
    template <int ID> __device__ __forceinline__ void stamp() { t[ID + 1] = estamp(); }
    __device__ __forceinline__ void finish(int nq) {
        const unsigned long long te = estamp();
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* o = g_emit_stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10;
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = t[i];
            o[8] = te;
            o[9] = (unsigned long long)nq;
        }
    }
};

// ablated builds of the emit kernel: the kernel returns at cut point LEVEL (1 = empty workgroups, 2 = table loads, 3 = + the
// eighteen gathers, 5 = + bbox / pre-cull + compaction (phase A complete), 6 = + phase B, 7 = + bucket scan, 0 = all)
template <int LEVEL>
struct EmitAblate {
    static constexpr int abl = LEVEL;
    __device__ __forceinline__ void begin() {}
    template <int ID> __device__ __forceinline__ void stamp() {}
    __device__ __forceinline__ void finish(int) {}
};
template <int LEVEL>
static void launch_ablate(const fr::RenderArgs& a, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL((fr::raster_emit_kernel<EmitAblate<LEVEL>>), dim3(grid), dim3(fr::EMIT_BLOCK), 0, st, a);
}
extern "C" int fr_probe_emit_ablate(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri, int H,
                                    int W, int tex_batch, float* depth, float* tex_img, float* normal, float* tri_ind,
                                    void* workspace, size_t ws_bytes, long long vpitch, int level, void* hip_stream) {
    using namespace fr;
    RenderArgs a;
    RenderGeom g;
    bool binned = false;
    int rc = prepare_render(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal, tri_ind, nullptr, nullptr,
                            nullptr, workspace, ws_bytes, vpitch, a, g, &binned);
    if (rc != FR_OK || !binned) return rc ? rc : -100;
    const unsigned grid = (unsigned)((long long)B * g.nseg);
    hipStream_t st = (hipStream_t)hip_stream;
    switch (level) {
        case 0: hipLaunchKernelGGL(raster_emit_kernel<NoEmitProbe>, dim3(grid), dim3(EMIT_BLOCK), 0, st, a); break;
        case 1: launch_ablate<1>(a, grid, st); break;
        case 2: launch_ablate<2>(a, grid, st); break;
        case 3: launch_ablate<3>(a, grid, st); break;
        case 5: launch_ablate<5>(a, grid, st); break;
        case 6: launch_ablate<6>(a, grid, st); break;
        case 7: launch_ablate<7>(a, grid, st); break;
        default: return -101;
    }
    return hipGetLastError() == hipSuccess ? 0 : FR_ERR_LAUNCH;
}

// the PRODUCT emit kernel launched with `extra_lds` bytes of (unused) dynamic LDS: limits the workgroups resident per CU
// (160 KiB / (20.4 KiB + extra)) -- what a kernel fused with the resolver's larger footprint would run at
extern "C" int fr_probe_emit_with_extra_lds(const float* vertex, const float* tri, const float* texture, int B, int nver,
                                            int ntri, int H, int W, int tex_batch, float* depth, float* tex_img, float* normal,
                                            float* tri_ind, void* workspace, size_t ws_bytes, long long vpitch, int extra_lds,
                                            void* hip_stream) {
    using namespace fr;
    RenderArgs a;
    RenderGeom g;
    bool binned = false;
    int rc = prepare_render(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal, tri_ind, nullptr, nullptr,
                            nullptr, workspace, ws_bytes, vpitch, a, g, &binned);
    if (rc != FR_OK || !binned) return rc ? rc : -100;
    hipLaunchKernelGGL(raster_emit_kernel<NoEmitProbe>, dim3((unsigned)((long long)B * g.nseg)), dim3(EMIT_BLOCK), (size_t)extra_lds,
                       (hipStream_t)hip_stream, a);
    return hipGetLastError() == hipSuccess ? 0 : FR_ERR_LAUNCH;
}

// the same for resolve_write_kernel<256>: stamps [bins][4 waves][10]
extern "C" int fr_probe_resolve_stamps(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                                       int H, int W, int tex_batch, float* depth, float* tex_img, float* normal, float* tri_ind,
                                       void* workspace, size_t ws_bytes, long long vpitch, unsigned long long* stamps,
                                       void* hip_stream) {
    using namespace fr;
    RenderArgs a;
    RenderGeom g;
    bool binned = false;
    int rc = prepare_render(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal, tri_ind, nullptr, nullptr,
                            nullptr, workspace, ws_bytes, vpitch, a, g, &binned);
    if (rc != FR_OK || !binned || g.lds > 32 * 1024) return rc ? rc : -100;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_emit_stamps), &stamps, sizeof(stamps));
    const long long nbins = (long long)B * g.strips;
    hipLaunchKernelGGL((resolve_write_kernel<256, false, EmitStampProbe>), dim3((unsigned)nbins), dim3(256),
                       g.lds + resolve_scratch_bytes(256), (hipStream_t)hip_stream, a);
    return hipGetLastError() == hipSuccess ? (int)nbins : FR_ERR_LAUNCH;
}

extern "C" int fr_probe_emit_stamps(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                                    int H, int W, int tex_batch, float* depth, float* tex_img, float* normal, float* tri_ind,
                                    void* workspace, size_t ws_bytes, long long vpitch, unsigned long long* stamps,
                                    void* hip_stream) {
    using namespace fr;
    RenderArgs a;
    RenderGeom g;
    bool binned = false;
    int rc = prepare_render(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal, tri_ind, nullptr, nullptr,
                            nullptr, workspace, ws_bytes, vpitch, a, g, &binned);
    if (rc != FR_OK || !binned) return rc ? rc : -100;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_emit_stamps), &stamps, sizeof(stamps));
    hipLaunchKernelGGL(raster_emit_kernel<EmitStampProbe>, dim3((unsigned)((long long)B * g.nseg)), dim3(EMIT_BLOCK), 0,
                       (hipStream_t)hip_stream, a);
    return hipGetLastError() == hipSuccess ? B * g.nseg : FR_ERR_LAUNCH;
}
