// extern "C" entry points declared in include/fr_hotpath.h: argument validation (the OP_REQUIRES checks of
// render_depth_op.cc:408-418, 498-503 re-stated) and dispatch to the gfx950 launchers.
#include <atomic>
#include <mutex>

#include "fr_common.h"

// ---- option table: environment read once, fr_set_option afterwards ------------------------------------------------------
namespace {
struct OptDesc {
    const char* name;
    int dflt;
    const char* word;  // a non-numeric spelling of value 1 ("loop", "scan"), or null
};
const OptDesc kOpts[fr::OPT_COUNT] = {
    {"FR_DECODE_IMPL", 0, "loop"}, {"FR_DECODE_WIDE", 1, nullptr}, {"FR_DECODE_NBW", 0, nullptr},
    {"FR_DECODE_WAVES", 16, nullptr}, {"FR_DECODE_NT", -1, nullptr}, {"FR_RESOLVE_OPT", 2, nullptr},
    {"FR_EMIT_FILTER", 3, nullptr}, {"FR_RENDER_IMPL", 0, "scan"}, {"FR_RESOLVE_BLOCK", 0, nullptr},
    {"FR_RENDER_ROWS", 0, nullptr}, {"FR_DECODE_STORE", 0, nullptr},
    {"FR_BWD_CHUNKS", 256, nullptr}, {"FR_BWD_CB", 0, nullptr}, {"FR_EMIT_ORDER", -1, nullptr},
    {"FR_Q30_SCHED", 0, nullptr},
};
std::atomic<int> g_opt[fr::OPT_COUNT];
std::once_flag g_opt_once;
void opts_init() {
    for (int i = 0; i < fr::OPT_COUNT; i++) {
        int v = kOpts[i].dflt;
        const char* e = getenv(kOpts[i].name);
        if (e && *e) v = (kOpts[i].word && !strcmp(e, kOpts[i].word)) ? 1 : atoi(e);
        g_opt[i].store(v, std::memory_order_relaxed);
    }
}
int opt_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < fr::OPT_COUNT; i++)
        if (!strcmp(name, kOpts[i].name)) return i;
    return -1;
}
}  // namespace

int fr::opt(fr::Opt o) {
    std::call_once(g_opt_once, opts_init);
    return g_opt[o].load(std::memory_order_relaxed);
}

extern "C" {

int fr_set_option(const char* name, int value) {
    const int i = opt_index(name);
    if (i < 0) return FR_ERR_INVALID_ARG;
    std::call_once(g_opt_once, opts_init);
    g_opt[i].store(value, std::memory_order_relaxed);
    return FR_OK;
}

int fr_get_option(const char* name, int* value) {
    const int i = opt_index(name);
    if (i < 0 || !value) return FR_ERR_INVALID_ARG;
    *value = fr::opt((fr::Opt)i);
    return FR_OK;
}

#ifndef FR_SRC_HASH
#define FR_SRC_HASH "unhashed"
#endif
// the build identity: _lib.py refuses a library whose source hash differs from the tree's
const char* fr_version(void) { return "fr_hotpath 0.3 (gfx950) src=" FR_SRC_HASH; }

const char* fr_strerror(int code) {
    switch (code) {
        case FR_OK: return "ok";
        case FR_ERR_INVALID_ARG: return "invalid argument";
        case FR_ERR_WORKSPACE: return "workspace / packed buffer too small";
        case FR_ERR_LAUNCH: return "HIP launch or runtime error";
        case FR_ERR_UNSUPPORTED: return "size not supported by the gfx950 kernels";
        default: return "unknown error";
    }
}

size_t fr_render_depth_workspace_bytes(int B, int nver, int ntri, int H, int W) {
    (void)nver;
    if (B < 0 || ntri < 0 || H < 0 || W < 0) return 0;
    return fr_render_workspace_bytes_impl(B, ntri, H, W);  // per-segment hit records + bucket offsets
}

int fr_render_depth_forward(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                            int H, int W, int C, int tex_batch, float* depth, float* tex_img, float* normal,
                            float* tri_ind, void* workspace, size_t ws_bytes, void* hip_stream) {
    if (B < 0 || nver < 0 || ntri < 0 || H < 0 || W < 0) return FR_ERR_INVALID_ARG;
    if (C != 3) return FR_ERR_INVALID_ARG;                         // render_depth_op.cc:418
    if (tex_batch != 1 && tex_batch != B) return FR_ERR_INVALID_ARG;
    if ((size_t)B * H * W == 0) return FR_OK;                      // empty batch / image: nothing to write
    if (!depth || !tex_img || !normal || !tri_ind) return FR_ERR_INVALID_ARG;
    if (ntri > 0 && (!tri || (nver > 0 && (!vertex || !texture)))) return FR_ERR_INVALID_ARG;
    if (ntri >= (1 << 24)) return FR_ERR_UNSUPPORTED;              // float-stored ids stop being exact
    if (ws_bytes < fr_render_depth_workspace_bytes(B, nver, ntri, H, W)) return FR_ERR_WORKSPACE;
    return fr_launch_render_forward(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal,
                                    tri_ind, workspace, ws_bytes, (hipStream_t)hip_stream);
}

int fr_render_depth_forward_phases(const float* vertex, const float* tri, const float* texture, int B, int nver, int ntri,
                                   int H, int W, int C, int tex_batch, float* depth, float* tex_img, float* normal,
                                   float* tri_ind, void* workspace, size_t ws_bytes, void* hip_stream, int phases) {
    if (phases < 1 || phases > 7) return FR_ERR_INVALID_ARG;
    if (B < 0 || nver < 0 || ntri < 0 || H < 0 || W < 0 || C != 3) return FR_ERR_INVALID_ARG;
    if (tex_batch != 1 && tex_batch != B) return FR_ERR_INVALID_ARG;
    if ((size_t)B * H * W == 0) return FR_OK;
    if (!depth || !tex_img || !normal || !tri_ind) return FR_ERR_INVALID_ARG;
    if (ntri > 0 && (!tri || (nver > 0 && (!vertex || !texture)))) return FR_ERR_INVALID_ARG;
    if (ntri >= (1 << 24)) return FR_ERR_UNSUPPORTED;
    if (ws_bytes < fr_render_depth_workspace_bytes(B, nver, ntri, H, W)) return FR_ERR_WORKSPACE;
    return fr_launch_render_forward_phases(vertex, tri, texture, B, nver, ntri, H, W, tex_batch, depth, tex_img, normal,
                                           tri_ind, workspace, ws_bytes, (hipStream_t)hip_stream, phases);
}

static int rendering_layer_checked(const float* vertex, const float* tri, const float* texture, const float* im_gray, int B,
                                   int nver, int ntri, int H, int W, int tex_batch, float* net_input, float* depth_img,
                                   float* depth, float* tri_ind, void* workspace, size_t ws_bytes, void* hip_stream, int phases) {
    if (phases < 1 || phases > 7) return FR_ERR_INVALID_ARG;
    if (B < 0 || nver < 0 || ntri < 0 || H < 0 || W < 0) return FR_ERR_INVALID_ARG;
    if (tex_batch != 1 && tex_batch != B) return FR_ERR_INVALID_ARG;
    if ((size_t)B * H * W == 0) return FR_OK;
    if (!net_input || !depth_img || !depth || !tri_ind || !im_gray) return FR_ERR_INVALID_ARG;
    if (ntri > 0 && (!tri || (nver > 0 && (!vertex || !texture)))) return FR_ERR_INVALID_ARG;
    if (ntri >= (1 << 24)) return FR_ERR_UNSUPPORTED;
    if (ntri == 0 || nver == 0) return FR_ERR_UNSUPPORTED;  // nothing to fuse: use the plain op
    if (ws_bytes < fr_render_depth_workspace_bytes(B, nver, ntri, H, W)) return FR_ERR_WORKSPACE;
    return fr_launch_rendering_layer(vertex, tri, texture, im_gray, B, nver, ntri, H, W, tex_batch, net_input, depth_img,
                                     depth, tri_ind, workspace, ws_bytes, (hipStream_t)hip_stream, phases);
}

int fr_rendering_layer_forward(const float* vertex, const float* tri, const float* texture, const float* im_gray, int B,
                               int nver, int ntri, int H, int W, int tex_batch, float* net_input, float* depth_img,
                               float* depth, float* tri_ind, void* workspace, size_t ws_bytes, void* hip_stream) {
    return rendering_layer_checked(vertex, tri, texture, im_gray, B, nver, ntri, H, W, tex_batch, net_input, depth_img, depth,
                                   tri_ind, workspace, ws_bytes, hip_stream, 7);
}

int fr_rendering_layer_forward_phases(const float* vertex, const float* tri, const float* texture, const float* im_gray, int B,
                                      int nver, int ntri, int H, int W, int tex_batch, float* net_input, float* depth_img,
                                      float* depth, float* tri_ind, void* workspace, size_t ws_bytes, void* hip_stream,
                                      int phases) {
    return rendering_layer_checked(vertex, tri, texture, im_gray, B, nver, ntri, H, W, tex_batch, net_input, depth_img, depth,
                                   tri_ind, workspace, ws_bytes, hip_stream, phases);
}

static int render_backward_checked(const float* depth_grad, const float* tri, const float* tri_ind, float* vertex_grad,
                                   int B, int nver, int ntri, int H, int W, void* workspace, size_t ws_bytes,
                                   void* hip_stream) {
    if (B < 0 || nver < 0 || ntri < 0 || H < 0 || W < 0) return FR_ERR_INVALID_ARG;
    if ((size_t)B * nver == 0) return FR_OK;
    if (!vertex_grad) return FR_ERR_INVALID_ARG;
    if ((size_t)B * H * W > 0 && ntri > 0 && (!depth_grad || !tri || !tri_ind)) return FR_ERR_INVALID_ARG;
    return fr_launch_render_backward(depth_grad, tri, tri_ind, vertex_grad, B, nver, ntri, H, W, workspace, ws_bytes,
                                     (hipStream_t)hip_stream);
}

int fr_render_depth_backward(const float* depth_grad, const float* tri, const float* tri_ind, float* vertex_grad,
                             int B, int nver, int ntri, int H, int W, void* hip_stream) {
    return render_backward_checked(depth_grad, tri, tri_ind, vertex_grad, B, nver, ntri, H, W, nullptr, 0, hip_stream);
}

size_t fr_render_depth_backward_workspace_bytes(int B, int H, int W) {
    return (B > 0 && H > 0 && W > 0) ? fr_render_backward_workspace_bytes_impl(B, H, W) : 0;
}

int fr_render_depth_backward_ws(const float* depth_grad, const float* tri, const float* tri_ind, float* vertex_grad,
                                int B, int nver, int ntri, int H, int W, void* workspace, size_t ws_bytes,
                                void* hip_stream) {
    if (workspace && (ws_bytes < fr_render_depth_backward_workspace_bytes(B, H, W) || ((uintptr_t)workspace & 15)))
        return FR_ERR_WORKSPACE;
    return render_backward_checked(depth_grad, tri, tri_ind, vertex_grad, B, nver, ntri, H, W, workspace, ws_bytes,
                                   hip_stream);
}

size_t fr_decode_packed_basis_bytes(int N, int n_shape, int n_exp) {
    if (N < 0 || n_shape < 0 || n_exp < 0) return 0;
    return fr_packed_basis_bytes(N, n_shape, n_exp);
}

int fr_decode_pack_basis(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp,
                         void* packed, size_t packed_bytes, void* hip_stream) {
    if (N < 0 || n_shape < 0 || n_exp < 0) return FR_ERR_INVALID_ARG;
    if (packed_bytes < fr_packed_basis_bytes(N, n_shape, n_exp)) return FR_ERR_WORKSPACE;
    if (N == 0) return FR_OK;
    if (!mu || !packed || (n_shape > 0 && !pc_shape) || (n_exp > 0 && !pc_exp)) return FR_ERR_INVALID_ARG;
    if (((uintptr_t)packed & 15) != 0) return FR_ERR_INVALID_ARG;
    return fr_launch_pack_basis(mu, pc_shape, pc_exp, N, n_shape, n_exp, packed, (hipStream_t)hip_stream);
}

int fr_decode_3dmm(const float* params, const void* packed_basis, const float* R_override, int B, int N, int n_shape,
                   int n_exp, float im_size, float* vertex_proj, void* hip_stream) {
    if (B < 0 || N < 0 || n_shape < 0 || n_exp < 0) return FR_ERR_INVALID_ARG;
    if ((size_t)B * N == 0) return FR_OK;
    if (!params || !packed_basis || !vertex_proj) return FR_ERR_INVALID_ARG;
    if (((uintptr_t)packed_basis & 15) != 0) return FR_ERR_INVALID_ARG;
    return fr_launch_decode(params, packed_basis, R_override, B, N, n_shape, n_exp, im_size, vertex_proj, N,
                            (hipStream_t)hip_stream);
}

// ---- fused decode -> render step ---------------------------------------------------------------------------------------
int fr_decode_render_vertex_pitch(int N) { return N <= 0 ? 0 : (N + 31) & ~31; }

size_t fr_decode_render_vertex_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * 3 * (size_t)fr_decode_render_vertex_pitch(N) * sizeof(float);
}

int fr_decode_render_forward(const float* params, const void* packed_basis, const float* R_override, const float* tri,
                             const float* texture, int B, int N, int n_shape, int n_exp, int ntri, int H, int W,
                             int tex_batch, float im_size, float* vertex_handoff, size_t vertex_bytes, float* depth,
                             float* tex_img, float* normal, float* tri_ind, void* workspace, size_t ws_bytes,
                             void* hip_stream, int phases) {
    if ((phases & 15) < 1 || (phases & ~0xFF0F)) return FR_ERR_INVALID_ARG;   // bits 0-3: phases; bits 8-15: strip-height hint
    if (B < 0 || N < 0 || n_shape < 0 || n_exp < 0 || ntri < 0 || H < 0 || W < 0) return FR_ERR_INVALID_ARG;
    if (tex_batch != 1 && tex_batch != B) return FR_ERR_INVALID_ARG;
    if (B == 0) return FR_OK;
    const int pitch = fr_decode_render_vertex_pitch(N);
    if (N > 0 && (!vertex_handoff || vertex_bytes < fr_decode_render_vertex_bytes(B, N) || ((uintptr_t)vertex_handoff & 127)))
        return FR_ERR_WORKSPACE;
    if ((phases & 8) && N > 0) {
        if (!params || !packed_basis) return FR_ERR_INVALID_ARG;
        if (((uintptr_t)packed_basis & 15) != 0) return FR_ERR_INVALID_ARG;
        const int rc = fr_launch_decode(params, packed_basis, R_override, B, N, n_shape, n_exp, im_size, vertex_handoff, pitch,
                                        (hipStream_t)hip_stream);
        if (rc != FR_OK) return rc;
    }
    if (!(phases & 7) || (size_t)H * W == 0) return FR_OK;
    if (!depth || !tex_img || !normal || !tri_ind) return FR_ERR_INVALID_ARG;
    if (ntri > 0 && (!tri || (N > 0 && !texture))) return FR_ERR_INVALID_ARG;
    if (ntri >= (1 << 24)) return FR_ERR_UNSUPPORTED;
    if (ws_bytes < fr_render_depth_workspace_bytes(B, N, ntri, H, W)) return FR_ERR_WORKSPACE;
    return fr_launch_render_forward_phases(vertex_handoff, tri, texture, B, N, ntri, H, W, tex_batch, depth, tex_img, normal,
                                           tri_ind, workspace, ws_bytes, (hipStream_t)hip_stream, phases & 7, pitch, (phases >> 8) & 0xFF);
}

// ---- opt-in Q30 arithmetic: its own image, its own entry point, caller-owned staging workspace ----------------------------
size_t fr_decode_q30_image_bytes(int N, int n_shape, int n_exp) {
    if (N < 0 || n_shape < 0 || n_exp < 0 || !fr_decode_q_supported(n_shape, n_exp)) return 0;
    return fr_packed_q_bytes(N, n_shape, n_exp);
}

int fr_decode_q30_pack(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp,
                       void* qimage, size_t qimage_bytes, void* hip_stream) {
    if (N < 0 || n_shape < 0 || n_exp < 0) return FR_ERR_INVALID_ARG;
    if (!fr_decode_q_supported(n_shape, n_exp)) return FR_ERR_UNSUPPORTED;
    if (qimage_bytes < fr_packed_q_bytes(N, n_shape, n_exp)) return FR_ERR_WORKSPACE;
    if (N == 0) return FR_OK;
    if (!mu || !qimage || (n_shape > 0 && !pc_shape) || (n_exp > 0 && !pc_exp)) return FR_ERR_INVALID_ARG;
    if (((uintptr_t)qimage & 255) != 0) return FR_ERR_INVALID_ARG;
    return fr_launch_pack_q(mu, pc_shape, pc_exp, N, n_shape, n_exp, qimage, (hipStream_t)hip_stream);
}

size_t fr_decode_q30_workspace_bytes(int n_shape, int n_exp) {
    if (n_shape < 0 || n_exp < 0) return 0;
    return fr_decode_q_workspace_bytes_impl(n_shape, n_exp);
}

static int decode_q30_checked(const float* params, const void* qimage, const float* R_override, int B, int N, int n_shape,
                              int n_exp, float im_size, float* vertex_proj, int pitch, int levels, void* workspace,
                              size_t ws_bytes, void* hip_stream) {
    if (B < 0 || N < 0 || n_shape < 0 || n_exp < 0 || !fr_decode_q_levels_ok(levels)) return FR_ERR_INVALID_ARG;
    if (!fr_decode_q_supported(n_shape, n_exp)) return FR_ERR_UNSUPPORTED;
    if ((size_t)B * N == 0) return FR_OK;
    if (!params || !qimage || !vertex_proj || pitch < N) return FR_ERR_INVALID_ARG;
    if (((uintptr_t)qimage & 255) != 0) return FR_ERR_INVALID_ARG;
    if (!workspace || ws_bytes < fr_decode_q_workspace_bytes_impl(n_shape, n_exp) || ((uintptr_t)workspace & 15))
        return FR_ERR_WORKSPACE;
    return fr_launch_decode_q(params, qimage, R_override, B, N, n_shape, n_exp, im_size, vertex_proj, pitch, levels, workspace,
                              ws_bytes, (hipStream_t)hip_stream);
}

int fr_decode_3dmm_q30(const float* params, const void* qimage, const float* R_override, int B, int N, int n_shape,
                       int n_exp, float im_size, float* vertex_proj, void* workspace, size_t ws_bytes, void* hip_stream) {
    return decode_q30_checked(params, qimage, R_override, B, N, n_shape, n_exp, im_size, vertex_proj, N, 7, workspace, ws_bytes,
                              hip_stream);
}

int fr_decode_3dmm_q30_lv(const float* params, const void* qimage, const float* R_override, int B, int N, int n_shape,
                          int n_exp, float im_size, int levels, float* vertex_proj, void* workspace, size_t ws_bytes,
                          void* hip_stream) {
    return decode_q30_checked(params, qimage, R_override, B, N, n_shape, n_exp, im_size, vertex_proj, N, levels, workspace,
                              ws_bytes, hip_stream);
}

int fr_decode_render_forward_q30(const float* params, const void* qimage, const float* R_override, const float* tri,
                                 const float* texture, int B, int N, int n_shape, int n_exp, int ntri, int H, int W,
                                 int tex_batch, float im_size, int levels, float* vertex_handoff, size_t vertex_bytes,
                                 float* depth, float* tex_img, float* normal, float* tri_ind, void* workspace, size_t ws_bytes,
                                 void* q_workspace, size_t q_ws_bytes, void* hip_stream, int phases) {
    if ((phases & 15) < 1 || (phases & ~0xFF0F)) return FR_ERR_INVALID_ARG;   // bits 0-3: phases; bits 8-15: strip-height hint
    if (B < 0 || N < 0 || n_shape < 0 || n_exp < 0 || ntri < 0 || H < 0 || W < 0) return FR_ERR_INVALID_ARG;
    if (tex_batch != 1 && tex_batch != B) return FR_ERR_INVALID_ARG;
    if (!fr_decode_q_levels_ok(levels)) return FR_ERR_INVALID_ARG;
    if (B == 0) return FR_OK;
    const int pitch = fr_decode_render_vertex_pitch(N);
    if (N > 0 && (!vertex_handoff || vertex_bytes < fr_decode_render_vertex_bytes(B, N) || ((uintptr_t)vertex_handoff & 127)))
        return FR_ERR_WORKSPACE;
    if ((phases & 8) && N > 0) {
        const int rc = decode_q30_checked(params, qimage, R_override, B, N, n_shape, n_exp, im_size, vertex_handoff, pitch,
                                          levels, q_workspace, q_ws_bytes, hip_stream);
        if (rc != FR_OK) return rc;
    }
    if (!(phases & 7) || (size_t)H * W == 0) return FR_OK;
    if (!depth || !tex_img || !normal || !tri_ind) return FR_ERR_INVALID_ARG;
    if (ntri > 0 && (!tri || (N > 0 && !texture))) return FR_ERR_INVALID_ARG;
    if (ntri >= (1 << 24)) return FR_ERR_UNSUPPORTED;
    if (ws_bytes < fr_render_depth_workspace_bytes(B, N, ntri, H, W)) return FR_ERR_WORKSPACE;
    return fr_launch_render_forward_phases(vertex_handoff, tri, texture, B, N, ntri, H, W, tex_batch, depth, tex_img, normal,
                                           tri_ind, workspace, ws_bytes, (hipStream_t)hip_stream, phases & 7, pitch, (phases >> 8) & 0xFF);
}

int fr_debug_clock_probe(unsigned long long* ticks, int blocks, int iters, void* hip_stream) {
    if (!ticks || blocks < 1 || blocks > 65535 || iters < 1) return FR_ERR_INVALID_ARG;
    return fr_launch_clock_probe(ticks, blocks, iters, (hipStream_t)hip_stream);
}

size_t fr_decode_backward_workspace_bytes(int B, int N, int n_shape, int n_exp) {
    if (B <= 0 || N <= 0 || n_shape < 0 || n_exp < 0) return 0;
    return fr_decode_backward_workspace_impl(N, n_shape, n_exp);
}

int fr_decode_3dmm_backward(const float* grad_vertex_proj, const float* params, const float* vertex_proj,
                            const float* pc_shape, const float* pc_exp, const float* R_override, int B, int N, int n_shape,
                            int n_exp, float im_size, float* grad_params, void* workspace, size_t ws_bytes,
                            void* hip_stream) {
    if (B < 0 || N < 0 || n_shape < 0 || n_exp < 0) return FR_ERR_INVALID_ARG;
    if (B == 0) return FR_OK;
    if (!grad_params || !params) return FR_ERR_INVALID_ARG;
    if (N > 0 && (!grad_vertex_proj || !vertex_proj || (n_shape > 0 && !pc_shape) || (n_exp > 0 && !pc_exp)))
        return FR_ERR_INVALID_ARG;
    if (ws_bytes < fr_decode_backward_workspace_bytes(B, N, n_shape, n_exp)) return FR_ERR_WORKSPACE;
    if (N > 0 && (!workspace || ((uintptr_t)workspace & 15))) return FR_ERR_WORKSPACE;
    return fr_launch_decode_backward(grad_vertex_proj, params, vertex_proj, pc_shape, pc_exp, R_override, B, N, n_shape,
                                     n_exp, im_size, grad_params, workspace, (hipStream_t)hip_stream);
}

size_t fr_decode_backward_basis_bytes(int N, int n_shape, int n_exp) {
    if (N <= 0 || n_shape < 0 || n_exp < 0) return 0;
    return fr_decode_backward_basis_bytes_impl(N, n_shape, n_exp);
}

int fr_decode_backward_pack_basis(const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp, void* packed_t,
                                  size_t packed_bytes, void* hip_stream) {
    if (N < 0 || n_shape < 0 || n_exp < 0) return FR_ERR_INVALID_ARG;
    if (packed_bytes < fr_decode_backward_basis_bytes(N, n_shape, n_exp)) return FR_ERR_WORKSPACE;
    if (N == 0 || n_shape + n_exp == 0) return FR_OK;
    if (!packed_t || (n_shape > 0 && !pc_shape) || (n_exp > 0 && !pc_exp)) return FR_ERR_INVALID_ARG;
    if (((uintptr_t)packed_t & 15) != 0) return FR_ERR_INVALID_ARG;
    return fr_launch_decode_backward_pack(pc_shape, pc_exp, N, n_shape, n_exp, packed_t, (hipStream_t)hip_stream);
}

int fr_decode_3dmm_backward_packed(const float* grad_vertex_proj, const float* params, const float* vertex_proj,
                                   const void* packed_t, const float* R_override, int B, int N, int n_shape, int n_exp,
                                   float im_size, float* grad_params, void* workspace, size_t ws_bytes, void* hip_stream) {
    if (B < 0 || N < 0 || n_shape < 0 || n_exp < 0) return FR_ERR_INVALID_ARG;
    if (B == 0) return FR_OK;
    if (!grad_params || !params) return FR_ERR_INVALID_ARG;
    if (N > 0 && (!grad_vertex_proj || !vertex_proj || (n_shape + n_exp > 0 && !packed_t))) return FR_ERR_INVALID_ARG;
    if (((uintptr_t)packed_t & 15) != 0) return FR_ERR_INVALID_ARG;
    if (ws_bytes < fr_decode_backward_workspace_bytes(B, N, n_shape, n_exp)) return FR_ERR_WORKSPACE;
    if (N > 0 && (!workspace || ((uintptr_t)workspace & 15))) return FR_ERR_WORKSPACE;
    return fr_launch_decode_backward(grad_vertex_proj, params, vertex_proj, nullptr, nullptr, R_override, B, N, n_shape, n_exp,
                                     im_size, grad_params, workspace, (hipStream_t)hip_stream, n_shape + n_exp > 0 ? packed_t : nullptr);
}

int fr_decode_3dmm_backward_packed_mu(const float* grad_vertex_proj, const float* params, const float* mu, const void* packed_t,
                                      const float* R_override, int B, int N, int n_shape, int n_exp, float im_size,
                                      float* grad_params, void* workspace, size_t ws_bytes, void* hip_stream) {
    if (B < 0 || N < 0 || n_shape < 0 || n_exp < 0) return FR_ERR_INVALID_ARG;
    if (fr_decode_backward_basis_bytes(N, n_shape, n_exp) == 0) return FR_ERR_UNSUPPORTED;   // what the fused kernel does not serve
    if (B == 0) return FR_OK;
    if (!grad_params || !params || !grad_vertex_proj || !mu || !packed_t) return FR_ERR_INVALID_ARG;
    if (((uintptr_t)packed_t & 15) != 0 || ((uintptr_t)mu & 15) != 0) return FR_ERR_INVALID_ARG;
    if (ws_bytes < fr_decode_backward_workspace_bytes(B, N, n_shape, n_exp)) return FR_ERR_WORKSPACE;
    if (!workspace || ((uintptr_t)workspace & 15)) return FR_ERR_WORKSPACE;
    return fr_launch_decode_backward(grad_vertex_proj, params, nullptr, nullptr, nullptr, R_override, B, N, n_shape, n_exp,
                                     im_size, grad_params, workspace, (hipStream_t)hip_stream, packed_t, mu);
}

}  // extern "C"
