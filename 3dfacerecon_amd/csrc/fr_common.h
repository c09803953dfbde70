// Shared device helpers for the gfx950 hot-path kernels.  This TU set is compiled with -ffp-contract=off:
// the reference CPU functor is built without FMA (g++ -O2 on baseline x86-64, rendering_layer/ops.py:51), so
// every fp64/fp32 product and sum below must round individually; fused operations appear only where written
// explicitly (__builtin_fmaf, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fr_hotpath.h"

namespace fr {

// (float)(-99999999999999) == -100000000376832.0f  (render_depth_op.cc:186)
__device__ __forceinline__ float bg_depth() { return -99999999999999.0f; }

// x86 cvttss2si semantics of the reference's (int) casts: NaN / out-of-range -> INT_MIN.
__device__ __forceinline__ int f2i_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// render_depth_op.h:15-16 min/max macros (NaN-sensitive ordering of the comparisons matters).
__device__ __forceinline__ float mn(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float mx(float a, float b) { return a > b ? a : b; }

// Monotone map fp32 -> u32 (a < b  <=>  ord(a) < ord(b) for non-NaN a, b with -0 canonicalised to +0).
__device__ __forceinline__ uint32_t f32_ord(float h) {
    uint32_t u = __float_as_uint(h);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unord(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

// Packed per-pixel key: max over keys == "largest h wins, ties -> lowest triangle index" (the serial
// semantics of render_depth_op.cc:263-316 with its strict '<' at :295).
__device__ __forceinline__ unsigned long long make_key(float h, int t) {
    return ((unsigned long long)f32_ord(h) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)t);
}
__device__ __forceinline__ unsigned long long bg_key() {
    return ((unsigned long long)f32_ord(bg_depth()) << 32) | 0xFFFFFFFFull;
}

// Barycentric inside test, op flavour (u+v < 1), fp64, operation order of render_depth_op.cc:90-121.
struct TriSetup {
    double x1, y1;
    double v0x, v0y, v1x, v1y;
    double dot00, dot01, dot11, inv;
};
__device__ __forceinline__ TriSetup tri_setup(float fx1, float fy1, float fx2, float fy2, float fx3, float fy3) {
    TriSetup s;
    s.x1 = (double)fx1;
    s.y1 = (double)fy1;
    s.v0x = (double)fx3 - s.x1;
    s.v0y = (double)fy3 - s.y1;
    s.v1x = (double)fx2 - s.x1;
    s.v1y = (double)fy2 - s.y1;
    s.dot00 = s.v0x * s.v0x + s.v0y * s.v0y;
    s.dot01 = s.v0x * s.v1x + s.v0y * s.v1y;
    s.dot11 = s.v1x * s.v1x + s.v1y * s.v1y;
    double den = s.dot00 * s.dot11 - s.dot01 * s.dot01;
    s.inv = (den == 0) ? 0.0 : 1 / den;
    return s;
}
__device__ __forceinline__ bool point_in_tri(const TriSetup& s, int x, int y) {
    double v2x = (double)x - s.x1;
    double v2y = (double)y - s.y1;
    double dot02 = s.v0x * v2x + s.v0y * v2y;
    double dot12 = s.v1x * v2x + s.v1y * v2y;
    double u = (s.dot11 * dot02 - s.dot01 * dot12) * s.inv;
    if (u < 0 || u > 1) return false;
    double v = (s.dot00 * dot12 - s.dot01 * dot02) * s.inv;
    if (v < 0 || v > 1) return false;
    return u + v < 1;
}

}  // namespace fr

// Raise a kernel's dynamic-LDS limit to the CU's full 160 KiB, once per (kernel, device): not a stream operation, so
// it is kept out of the steady-state launch path (and out of hipGraph captures).  The per-device "done" flags are atomics:
// the entry points may be called from several host threads (the header says so); two threads racing here both set the same
// attribute to the same value and both store 1 -- idempotent, and now also free of a data race in the C++ sense.
typedef std::atomic<unsigned char> fr_lds_flags_t;
inline hipError_t fr_allow_full_lds(const void* kernel, fr_lds_flags_t* done /*[64], zero-initialised (static storage)*/) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev].store(1, std::memory_order_release);
    return e;
}

// Tuning / A-B knobs of the launchers.  Each has an initial value taken ONCE per process from its environment variable
// (first use of any knob) and can be changed afterwards through fr_set_option() only: the launch path never calls getenv.
namespace fr {
enum Opt {
    OPT_DECODE_IMPL,    // FR_DECODE_IMPL    0 = ring schedule for the model's basis shape (default), 1 ("loop") = generic kernel
    OPT_DECODE_WIDE,    // FR_DECODE_WIDE    1 = 128-column passes for batches above 64 (default), 0 = off
    OPT_DECODE_NBW,     // FR_DECODE_NBW     0 = auto, 1 / 4 = column blocks per work item
    OPT_DECODE_WAVES,   // FR_DECODE_WAVES   16 (default) or 8 waves per decode workgroup
    OPT_DECODE_NT,      // FR_DECODE_NT      -1 = by batch (default: non-temporal basis stream for passes of 64 faces, default cache policy
                        //                   below), 1 = always non-temporal, 0 = always the default cache policy
    OPT_RESOLVE_OPT,    // FR_RESOLVE_OPT    2 = wave-local front for 256-thread bins (default), 1 = single-trip bins keep records in
                        //                   registers behind the block-wide list, 0 = two-pass resolver
    OPT_EMIT_FILTER,    // FR_EMIT_FILTER    bit 0: certified fp32 inside test, bit 1: single-pixel pre-cull (default 3)
    OPT_RENDER_IMPL,    // FR_RENDER_IMPL    0 = binned rasteriser (default), 1 ("scan") = strip-scan fallback
    OPT_RESOLVE_BLOCK,  // FR_RESOLVE_BLOCK  0 = auto, 256 / 512 / 1024 threads per resolver workgroup
    OPT_RENDER_ROWS,    // FR_RENDER_ROWS    0 = auto, > 0 = rows per screen strip
    OPT_DECODE_STORE,   // FR_DECODE_STORE   0 = default epilogue, 1 = transposed accumulators + one dword per lane (A/B knob)
    OPT_BWD_CHUNKS,     // FR_BWD_CHUNKS     row chunks (workgroups, partial slabs) of the packed decode-backward GEMM: 256 (default: one workgroup per CU), 1 .. 512
    OPT_BWD_CB,         // FR_BWD_CB         16-coefficient blocks per wave of the fused decode backward: 0 = by batch (default) | 2 | 4
    OPT_EMIT_ORDER,     // FR_EMIT_ORDER     lane order of a segment's triangles: -1 scored per segment (default), 0 identity, 1 even / odd passes
    OPT_Q30_SCHED,      // FR_Q30_SCHED      Q30 streaming schedule: 0 = 8 waves x whole tiles, 16-deep ring (default) | 1 = 16 waves, 32-column
                        //                   halves on neighbouring waves, 8-deep ring
    OPT_COUNT
};
int opt(Opt o);
}  // namespace fr

// host-side launchers implemented in the .hip files
int fr_launch_render_forward(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                             int H, int W, int tex_batch, float* depth, float* tex_img, float* normal,
                             float* tri_ind, void* workspace, size_t ws_bytes, hipStream_t stream);
int fr_launch_render_forward_phases(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                                    int H, int W, int tex_batch, float* depth, float* tex_img, float* normal,
                                    float* tri_ind, void* workspace, size_t ws_bytes, hipStream_t stream, int phases,
                                    long long vpitch = 0, int rows_hint = 0);
int fr_launch_rendering_layer(const float* vertex, const float* tri, const float* texture, const float* im_gray, int B,
                              int nver, int ntri, int H, int W, int tex_batch, float* net_in, float* depth_img,
                              float* depth, float* tri_ind, void* workspace, size_t ws_bytes, hipStream_t stream, int phases = 7);
int fr_launch_render_backward(const float* depth_grad, const float* tri, const float* tri_ind, float* vertex_grad,
                              int B, int nver, int ntri, int H, int W, void* workspace, size_t ws_bytes,
                              hipStream_t stream);
size_t fr_render_backward_workspace_bytes_impl(int B, int H, int W);
size_t fr_render_workspace_bytes_impl(int B, int ntri, int H, int W);
size_t fr_packed_basis_bytes(int N, int n_shape, int n_exp);
int fr_launch_pack_basis(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp,
                         void* packed, hipStream_t stream);
int fr_launch_decode(const float* params, const void* packed, const float* R_override, int B, int N, int n_shape,
                     int n_exp, float im_size, float* vertex_proj, int pitch, hipStream_t stream);
int fr_launch_clock_probe(unsigned long long* out, int blocks, int iters, hipStream_t stream);
size_t fr_packed_q_bytes(int N, int n_shape, int n_exp);
bool fr_decode_q_supported(int n_shape, int n_exp);
int fr_launch_pack_q(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp,
                     void* qimage, hipStream_t stream);
int fr_launch_decode_q(const float* params, const void* qimage, const float* R_override, int B, int N, int n_shape,
                       int n_exp, float im_size, float* vertex_proj, int pitch, int levels, void* workspace, size_t ws_bytes,
                       hipStream_t stream);
bool fr_decode_q_levels_ok(int levels);
size_t fr_decode_q_workspace_bytes_impl(int n_shape, int n_exp);
int fr_device_cu_count();
size_t fr_decode_backward_workspace_impl(int N, int ns, int ne);
int fr_launch_decode_backward(const float* grad_vertex_proj, const float* params, const float* vertex_proj,
                              const float* pc_shape, const float* pc_exp, const float* R_override, int B, int N, int ns,
                              int ne, float im_size, float* grad_params, void* workspace, hipStream_t stream,
                              const void* packed_t = nullptr, const float* mu = nullptr);
size_t fr_decode_backward_basis_bytes_impl(int N, int ns, int ne);
int fr_launch_decode_backward_pack(const float* pc_shape, const float* pc_exp, int N, int ns, int ne, void* packed_t,
                                   hipStream_t stream);
