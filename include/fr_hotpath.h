/*
 * fr_hotpath.h -- C ABI of the MI355X (gfx950) render_depth + 3DMM-decode hot path.
 *
 * This is the drop-in boundary.  The reference has no C ABI of its own for this path: its only native
 * boundary is the TensorFlow OpKernel C++ ABI (rendering_layer/ops_src/render_depth_op.cc:371-604, loaded by
 * rendering_layer/ops.py:68 through tf.load_op_library).  Each entry point below cites the reference
 * interface it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds to rendering_layer/ops.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (outputs and workspaces included); the library
 *     allocates nothing and never synchronises with the host;
 *   - `hip_stream` is a hipStream_t (NULL = the default stream); all work is enqueued on it, in order;
 *   - return value: FR_OK (0) or a negative FR_ERR_* code (never swallowed, unlike the reference's
 *     printf-and-return at render_depth_op.cu.cc:290-295); fr_strerror() names it;
 *   - reentrant and thread-safe (no static scratch, unlike render_depth_op.cc:125-131).
 *   - calls on DIFFERENT streams with disjoint output / workspace / vertex buffers may run beside each other (the model
 *     constants -- packed basis, triangle list, texture -- are only read): two independent batches in flight on two streams
 *     measure ~100 us per 64-face batch against ~111 us one batch at a time on an MI355X (pipeline.BatchesInFlight,
 *     DESIGN.md 4.7);
 *   - tensors are dense, row-major, fp32; triangle indices and tri_ind stay float-typed at the surface as in
 *     the reference op schema (render_depth_op.cc:535-589).
 */
#ifndef FR_HOTPATH_H_
#define FR_HOTPATH_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_OK 0
#define FR_ERR_INVALID_ARG (-1) /* OP_REQUIRES failures, render_depth_op.cc:408-418, 498-503 */
#define FR_ERR_WORKSPACE (-2)   /* workspace / packed-basis buffer too small */
#define FR_ERR_LAUNCH (-3)      /* HIP launch or runtime error */
#define FR_ERR_UNSUPPORTED (-4) /* size outside what the kernels cover (e.g. image row does not fit in LDS) */

#define FR_N_POSE 7 /* [phi,gamma,theta,tx,ty,tz,f], reference README.md:43-46, utils/parser_3dmm.py:49 */

/* Library identification / error text. */
const char* fr_version(void);
const char* fr_strerror(int code);

/* Tuning / A-B knobs of the launchers (no reference counterpart).  Each knob is named like the environment variable
 * that gives it its initial value -- read ONCE per process, at the first use of any knob; the launch path never calls
 * getenv -- and can be changed afterwards only through fr_set_option:
 *   FR_DECODE_IMPL (0 | 1 = "loop": generic decode kernel)   FR_DECODE_WIDE (1 | 0)   FR_DECODE_NBW (0 = auto | 1 | 4)
 *   FR_DECODE_WAVES (16 | 8)   FR_DECODE_NT (-1 = by batch: non-temporal basis stream for passes of 64 faces, default cache
 *   policy below -- default | 1 = always non-temporal | 0 = always the default cache policy)
 *   FR_RESOLVE_OPT (2 = default: wave-local front for 256-thread bins | 1 = block-wide list, single-trip bins keep their
 *   records in registers | 0 = two-pass resolver)   FR_EMIT_FILTER (bits 0-1, default 3)   FR_RENDER_IMPL (0 | 1 = "scan": strip-scan fallback)
 *   FR_RESOLVE_BLOCK (0 = auto | 256 | 512 | 1024)   FR_RENDER_ROWS (0 = auto | rows per screen strip)
 *   FR_DECODE_STORE (0 | 1 = transposed accumulators, one dword per lane per store: measured +1.1 us, A/B only)
 *   FR_BWD_CHUNKS (row chunks of the packed decode-backward GEMM: 256 = default | 1 .. 512; changes the association of the
 *   partial sums, i.e. the gradient's last bits -- every other knob leaves every result bit unchanged)
 *   FR_BWD_CB (16-coefficient blocks per wave of the fused decode backward: 0 = by batch | 2 | 4)
 *   FR_EMIT_ORDER (lane order of a segment's triangles in the emit kernel, fixed by the pack phase: -1 = scored per segment
 *   (default) | 0 = list order | 1 = even triangles, then odd ones; read when the triangle list is packed)
 * None of them changes a result bit (tests/test_render_gpu.py, tests/test_decode_gpu.py hold every setting to the oracle).
 * Returns FR_OK or FR_ERR_INVALID_ARG (unknown name). */
int fr_set_option(const char* name, int value);
int fr_get_option(const char* name, int* value);

/* ---- render_depth forward ------------------------------------------------------------------------------
 * Replaces RenderDepthOp<Device>::Compute + functor RenderDepth (render_depth_op.cc:378-458, 132-322;
 * CUDA launchers render_depth_op.cu.cc:239-341) reached from rendering_layer/ops.py:78-81.
 *   vertex  [B,3,nver]  projected vertices (x = column, y = row, z = depth)
 *   tri     [3,ntri]    float-stored 0-based vertex ids (truncated with (int), render_depth_op.cc:204-206)
 *   texture [tex_batch,3,nver], tex_batch == B, or 1 to share one texture across the batch
 *   depth [B,H,W,1], tex_img [B,H,W,3], normal [B,H,W,3], tri_ind [B,H,W,1]   (render_depth_op.cc:437-440)
 * C must be 3 (render_depth_op.cc:418).  Semantics are those of the CPU functor: per pixel the triangle with
 * the largest fp32 centroid depth wins, ties go to the lowest triangle index; background depth is
 * (float)(-99999999999999), tri_ind -1, texture/normal 0.  Triangles with a vertex id outside [0,nver) are
 * skipped.  `workspace` must hold fr_render_depth_workspace_bytes() bytes (may be NULL when that is 0). */
size_t fr_render_depth_workspace_bytes(int B, int nver, int ntri, int H, int W);

int fr_render_depth_forward(const float* vertex, const float* tri, const float* texture, int B, int nver,
                            int ntri, int H, int W, int C, int tex_batch, float* depth, float* tex_img,
                            float* normal, float* tri_ind, void* workspace, size_t ws_bytes, void* hip_stream);

/* The forward op phase by phase.  It is three launches: pack_tri_kernel (phase bit 4) converts and range-checks the
 * float-stored triangle list once into a table in the workspace (twice: by triangle id, and per 504-triangle segment in the
 * lane order that makes the emit kernel's gathers cheapest for this list); raster_emit_kernel (bit 1) writes per-strip hit
 * records into the workspace; resolve_write_kernel (bit 2) turns them into the four planes.  phases = 7 is
 * fr_render_depth_forward.  A caller whose triangle list is a constant of the model (the reference makes it a
 * tf.constant, nets/network.py:178) and whose workspace persists may pack once (phases = 4) and then run phases = 3
 * per batch.  The table carries a header naming the (nver, ntri) it was packed for: an emit phase handed a table that
 * was not packed for its arguments (no pack phase yet, or the workspace reused for another shape) treats every triangle
 * as invalid and the planes come out as pure background -- defined, never an out-of-bounds gather.  The table does NOT
 * record the ids themselves: repack after changing `tri` in place.  bench.py also uses single phases to bracket each
 * kernel with HIP events inside the timed region. */
int fr_render_depth_forward_phases(const float* vertex, const float* tri, const float* texture, int B, int nver,
                                   int ntri, int H, int W, int C, int tex_batch, float* depth, float* tex_img,
                                   float* normal, float* tri_ind, void* workspace, size_t ws_bytes, void* hip_stream,
                                   int phases);

/* ---- fused rendering layer ------------------------------------------------------------
 * render_depth + the caller-side post-processing of FaceRecNet.rendering_layer (nets/network.py:185-199) in one
 * pass, emitting CoarseNet's 7-channel input directly (network.py:122):
 *   net_input [B,H,W,7] = [ clip(depth,1e-6,1) * im_gray | clip(texture_image,1e-6,1) x3 | n / (sqrt(|n|^2)+1e-6) x3 ]
 *                         with n flipped to n_z >= 0 and |n|^2 <= 1e-6 replaced by 1
 *   depth_img [B,H,W,1] = max(depth, 1e-6);  depth [B,H,W,1] and tri_ind [B,H,W,1] as in fr_render_depth_forward
 *   im_gray   [B,H,W,1].  Same workspace as fr_render_depth_forward.
 * Returns FR_ERR_UNSUPPORTED for shapes only the fallback rasteriser covers (use the plain op + elementwise ops). */
int fr_rendering_layer_forward(const float* vertex, const float* tri, const float* texture, const float* im_gray,
                               int B, int nver, int ntri, int H, int W, int tex_batch, float* net_input,
                               float* depth_img, float* depth, float* tri_ind, void* workspace, size_t ws_bytes,
                               void* hip_stream);

/* The same phase by phase (bits as in fr_render_depth_forward_phases: 4 = pack the triangle list, 1 = emit, 2 = the fused
 * resolve): a caller whose triangle list is a model constant (nets/network.py:178) packs once and runs phases = 3 per call
 * (rendering_layer/ops.py does, under the same tensor-identity rule as render_depth). */
int fr_rendering_layer_forward_phases(const float* vertex, const float* tri, const float* texture, const float* im_gray,
                                      int B, int nver, int ntri, int H, int W, int tex_batch, float* net_input,
                                      float* depth_img, float* depth, float* tri_ind, void* workspace, size_t ws_bytes,
                                      void* hip_stream, int phases);

/* ---- render_depth backward -----------------------------------------------------------------------------
 * Replaces RenderDepthOpGrad::Compute + functor RenderDepthGrad (render_depth_op.cc:470-528, 325-368;
 * render_depth_op.cu.cc:345-423) reached from the gradient registration at rendering_layer/ops.py:86-95.
 *   depth_grad [B,H,W,1], tri [3,ntri], tri_ind [B,H,W,1] (forward output) -> vertex_grad [B,3,nver]
 * Every pixel with tri_ind >= 0 adds (depth_grad * 1.0f) / 3.0f to the z row of its triangle's three vertices; the x and
 * y rows are 0 (render_depth_op.cc:359-363); all of vertex_grad is written.  The per-vertex sums are formed as exact
 * 64-bit fixed-point integers and rounded to fp32 once (a subnormal result is rounded a second time by the final power-of-
 * two scaling): the result is the correctly rounded real sum up to n * 2^-39 * max|term| per face and is bit-identical
 * from run to run (the reference's serial loop has one fixed fp32 order; its CUDA twin uses order-dependent float
 * atomics).  max|term| is taken over the pixels with 0 <= tri_ind < ntri.  Images above 2^20 pixels give up one bit of
 * that resolution per doubling of the pixel count (2^-38 at 2^21 pixels, ...). */
int fr_render_depth_backward(const float* depth_grad, const float* tri, const float* tri_ind,
                             float* vertex_grad, int B, int nver, int ntri, int H, int W, void* hip_stream);

/* The same with a caller-owned workspace (fr_render_depth_backward_workspace_bytes(B, H, W): 16 bytes per pixel of the
 * batch plus 8 per 1,024 pixels, 16-byte aligned): one pre-kernel resolves every pixel to its triangle's three vertex ids ONCE and writes a
 * record per pixel; the accumulating workgroups (several per face) then stream the records instead of each repeating the
 * scattered id gathers.  Results are bit-identical to fr_render_depth_backward (which is this function with
 * workspace = NULL). */
size_t fr_render_depth_backward_workspace_bytes(int B, int H, int W);

int fr_render_depth_backward_ws(const float* depth_grad, const float* tri, const float* tri_ind, float* vertex_grad,
                                int B, int nver, int ntri, int H, int W, void* workspace, size_t ws_bytes,
                                void* hip_stream);

/* ---- 3DMM decode ---------------------------------------------------------------------------------------
 * Replaces FaceRecNet.vertices_transform + parse_pose_params + rotation_matrix_batch
 * (nets/network.py:140-171, 253-263, 266-297): 235-d parameters -> projected, y-flipped vertices [B,3,N].
 *
 * The basis is a constant of the model (tf.constant at network.py:41-43), so it is re-laid-out ONCE into an
 * MFMA-fragment-ordered image in HBM by fr_decode_pack_basis and then streamed by every decode call:
 *   mu [3N] (blocked: element r = coordinate r/N of vertex r%N, network.py:157),
 *   pc_shape [3N,n_shape], pc_exp [3N,n_exp] row-major (network.py:42-43)
 *   -> packed, fr_decode_packed_basis_bytes(N,n_shape,n_exp) bytes, 16-byte aligned. */
size_t fr_decode_packed_basis_bytes(int N, int n_shape, int n_exp);

int fr_decode_pack_basis(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape,
                         int n_exp, void* packed, size_t packed_bytes, void* hip_stream);

/*   params [B, 7+n_shape+n_exp] = [phi,gamma,theta,tx,ty,tz,f | alpha | beta]   (network.py:142-147)
 *   R_override: NULL, or a caller-computed [B,3,3] rotation (what network.py:150 obtains through tf.py_func);
 *               with NULL the rotation is evaluated in-kernel in float64 exactly as network.py:276-290 does.
 *   vertex_proj [B,3,N]; y row is (im_size - y) - 1 (network.py:167-169).
 * Numerical definition: the reference evaluates S = pc_shape.alpha and E = pc_exp.beta with two fp32
 * tf.matmuls whose summation order is unspecified (network.py:153-156).  fr_decode_3dmm's written definition:
 *   S, E = k-ordered fmaf chains from +0, v = (mu + S) + E  (the f32-input MFMA; the reference's own arithmetic type),
 * restated on the CPU in oracle/fr_oracle.c; the kernel is held to it bit for bit. */
int fr_decode_3dmm(const float* params, const void* packed_basis, const float* R_override, int B, int N,
                   int n_shape, int n_exp, float im_size, float* vertex_proj, void* hip_stream);

/* ---- fused decode -> render step (reference hand-off nets/network.py:140-171 -> :174-182) ---------
 * One call launches fr_decode_3dmm and then fr_render_depth_forward_phases on its result.  The projected vertices never
 * take the op surface's dense [B,3,N] form: they are handed from the decode to the rasteriser in a buffer the CALLER owns
 * but whose layout is the library's -- [B,3,pitch] rows with pitch = fr_decode_render_vertex_pitch(N) (N rounded up to a
 * multiple of 32 floats), fr_decode_render_vertex_bytes(B, N) bytes, 128-byte aligned: every 16-vertex piece a decode wave
 * stores is then an aligned half of a 128-byte line (N = 53,215 is odd: in the dense tensor every row starts at another
 * 4-byte phase and every 64-byte piece straddles two lines; measured -2.7 us per 64-face decode).  Element (b, c, p) sits at
 * (b * 3 + c) * pitch + p, so a strided view of the buffer IS the [B,3,N] tensor (the pad floats are never written or read).
 *   phases: bit 8 = decode, bit 4 = pack the triangle list, bit 1 = emit, bit 2 = resolve (15 = everything; a caller whose
 *   triangle list is a model constant runs 4 once and 11 per batch).  Results are bit-identical to the two separate calls.
 *   FR_PHASES_STRIP_ROWS(n) (bits 8-15 of `phases`, 0 = the library's choice): rows per screen strip of the resolver, i.e. how many
 *   resolver workgroups the screen is cut into -- a scheduling hint with no effect on any result bit or workspace size, to be
 *   passed unchanged with every phase of the same workspace.  The library's own choice (10 rows at 64 faces of 200 x 200) is the
 *   fastest one batch at a time; a caller that keeps two batches in flight on two streams does better with 8 (-0.9 us per
 *   batch: smaller resolver workgroups fill the other batch's gaps; +1.3 us one at a time).  A hint the binned rasteriser does not
 *   serve is ignored. */
#define FR_PHASES_STRIP_ROWS(n) (((n) & 0xFF) << 8)
/* The strip height the library itself picks for a shape (what FR_PHASES_STRIP_ROWS(0) means; no GPU needed), 0 when the binned
 * rasteriser does not serve the shape: what a caller scales its hint from. */
int fr_render_depth_strip_rows(int B, int ntri, int H, int W);
int fr_decode_render_vertex_pitch(int N);
size_t fr_decode_render_vertex_bytes(int B, int N);
int fr_decode_render_forward(const float* params, const void* packed_basis, const float* R_override, const float* tri,
                             const float* texture, int B, int N, int n_shape, int n_exp, int ntri, int H, int W,
                             int tex_batch, float im_size, float* vertex_handoff, size_t vertex_bytes, float* depth,
                             float* tex_img, float* normal, float* tri_ind, void* workspace, size_t ws_bytes,
                             void* hip_stream, int phases);

/* (Rounds 4-5 also exported fr_decode_render_pipelined: the emit of batch k and the resolve of batch k-1 as two roles of ONE
 * launch.  Bit-identical, measured 5-8 % slower than this entry point in both rounds and removed in round 6; a caller
 * that wants batches to overlap calls fr_decode_render_forward on two streams with two sets of buffers -- pipeline.BatchesInFlight.) */

/* Second definition of the same decode (nothing of it is built, allocated or launched unless these entry
 * points are called):
 *   Q30: v = fl32(mu + S + E) with S + E a fixed-point dot product of the operands quantised to 31 bits against power-of-two
 *   row / column scales, evaluated on the int8 matrix cores as products of base-256 digits.  `levels` = how many of the seven
 *   digit-product levels are kept: 7 = all sixteen products (the EXACT product of the quantised operands: the correctly
 *   rounded fp32 value of the real-number blend in > 99 % of the cases, half the f32 chain's mean error), 5 = the thirteen
 *   products of weight >= 2^-32 of full scale (what is dropped is below 2^-38 of a term's scale: the same fp32 result in
 *   > 99.98 % of the cases), 4 = the ten products of weight >= 2^-24 (dropped: below 2^-30; mean error 0.29 ulp against 0.26
 *   for the exact product and 0.34 - 0.5 for the f32 chain).  Integer arithmetic: order-independent, restated bit for bit in
 *   oracle/fr_oracle.c ("Q30 decode", the same `levels`).  A non-finite parameter makes the face's vertices NaN.
 * It has its own basis image (fr_decode_q30_image_bytes, 256-byte aligned; 0 = shape not covered: n_shape + n_exp > 512,
 * for which fr_decode_q30_pack / fr_decode_3dmm_q30 return FR_ERR_UNSUPPORTED; the image does not depend on `levels`) and
 * needs a caller-owned staging workspace (fr_decode_q30_workspace_bytes: 68 KiB for the model's shape, 16-byte aligned) that
 * must not be shared by launches in flight on different streams.  fr_decode_3dmm_q30 is levels = 7 on dense [B,3,N] rows.
 * FR_Q30_SCHED (fr_set_option): 0 = 8 waves per CU, a wave owns a tile's four column blocks, 16-deep ring (default) | 1 = 16
 * waves per CU, a tile's two 32-column halves on neighbouring waves; no result bit depends on it (profiles/round5_probes/r5b:
 * what each measured, and the launch-free staging forms that were built and not kept). */
size_t fr_decode_q30_image_bytes(int N, int n_shape, int n_exp);
int fr_decode_q30_pack(const float* mu, const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp,
                       void* qimage, size_t qimage_bytes, void* hip_stream);
size_t fr_decode_q30_workspace_bytes(int n_shape, int n_exp);
int fr_decode_3dmm_q30(const float* params, const void* qimage, const float* R_override, int B, int N, int n_shape,
                       int n_exp, float im_size, float* vertex_proj, void* workspace, size_t ws_bytes, void* hip_stream);
int fr_decode_3dmm_q30_lv(const float* params, const void* qimage, const float* R_override, int B, int N, int n_shape,
                          int n_exp, float im_size, int levels, float* vertex_proj, void* workspace, size_t ws_bytes,
                          void* hip_stream);
/* fr_decode_render_forward with the Q30 decode (same phases, same pitched vertex hand-off, same render workspace; q_workspace =
 * the decode's staging buffer, one per stream in flight). */
int fr_decode_render_forward_q30(const float* params, const void* qimage, const float* R_override, const float* tri,
                                 const float* texture, int B, int N, int n_shape, int n_exp, int ntri, int H, int W,
                                 int tex_batch, float im_size, int levels, float* vertex_handoff, size_t vertex_bytes,
                                 float* depth, float* tex_img, float* normal, float* tri_ind, void* workspace, size_t ws_bytes,
                                 void* q_workspace, size_t q_ws_bytes, void* hip_stream, int phases);

/* ---- 3DMM decode backward (the gradient TF autodiff derives from nets/network.py:140-171) ------------
 *   grad_vertex_proj [B,3,N] = dL/d vertex_proj;  vertex_proj [B,3,N] = the forward output (used for d f);
 *   mu / pc_shape / pc_exp in the reference layouts (not the packed image);  grad_params [B, 7+n_shape+n_exp].
 * d alpha = pc_shape^T dv, d beta = pc_exp^T dv with dv = (f R)^T dq, dq = (g_x, -g_y, g_z); d t3d = sum_p dq;
 * d f = sum_p (R v_p) . dq evaluated as sum_p (q - t3d) . dq / f, which needs no second pass over the basis; the three
 * angles get 0: in the reference R passes through tf.py_func (network.py:150), which has no gradient.
 * f == 0 (only reachable when set_constraints' sigmoid underflows, raw input < -103): every vertex projects onto t3d, the
 * quotient form is 0/0 and d f is DEFINED as 0 here (the true value sum_p (R v_p) . dq would need the un-projected
 * vertices, i.e. another basis pass); d alpha = d beta = 0 and d t3d are exact in that case.  Stated, tested
 * (tests/test_decode_backward_gpu.py::test_zero_focal_column), not silent.
 * Deterministic (fixed-order partial sums, no float atomics). */
size_t fr_decode_backward_workspace_bytes(int B, int N, int n_shape, int n_exp);

int fr_decode_3dmm_backward(const float* grad_vertex_proj, const float* params, const float* vertex_proj,
                            const float* pc_shape, const float* pc_exp, const float* R_override, int B, int N,
                            int n_shape, int n_exp, float im_size, float* grad_params, void* workspace, size_t ws_bytes,
                            void* hip_stream);

/* The same gradient from a packed basis image, in ONE fused kernel + the fixed-order reduction (round 4).
 * fr_decode_3dmm_backward above reads pc_shape / pc_exp in their reference layouts (no extra memory) and runs three
 * launches: a prepass that writes the 41 MB of dv rows, the reduction over the vertices, the slab sum (70 us of GEMM at 64
 * faces against a 33 us matrix-pipe floor).  A caller that takes gradients every step packs the basis ONCE --
 * fr_decode_backward_basis_bytes(N, n_shape, n_exp) bytes (the size of the forward image), 16-byte aligned: MFMA A-fragment
 * order, [vertex group of 16][x / y / z rows][16-coefficient block][lane] float4 -- and calls the packed entry point: one
 * workgroup per CU turns a 16-vertex tile of the incoming gradient into the three dv row blocks in LDS (and the d t3d / d f
 * partial sums in registers) and multiplies them against the group's basis fragments (counted-wait register ring); dv never
 * reaches global memory.  rocprofv3 at 64 faces: 27.6 + 59.4 + 16.7 us (round 3) -> 70.1 + 6.9 us.  Same workspace
 * (fr_decode_backward_workspace_bytes), same definition of every output, deterministic (bit-reproducible for a given
 * FR_BWD_CHUNKS); the two entry points sum their partial results in different (each fixed) orders, so they agree to rounding,
 * not bit for bit.
 * fr_decode_backward_basis_bytes answers 0 -- and the pack / packed entry points FR_ERR_UNSUPPORTED -- for what the fused kernel
 * does not serve: more than 256 coefficients, or a mesh of fewer than 16 vertices (its tile loads are sixteen floats wide); use
 * fr_decode_3dmm_backward there. */
size_t fr_decode_backward_basis_bytes(int N, int n_shape, int n_exp);
int fr_decode_backward_pack_basis(const float* pc_shape, const float* pc_exp, int N, int n_shape, int n_exp, void* packed_t,
                                  size_t packed_bytes, void* hip_stream);
int fr_decode_3dmm_backward_packed(const float* grad_vertex_proj, const float* params, const float* vertex_proj,
                                   const void* packed_t, const float* R_override, int B, int N, int n_shape, int n_exp,
                                   float im_size, float* grad_params, void* workspace, size_t ws_bytes, void* hip_stream);

/* The same fused backward WITHOUT the forward output (round 5).  d f = sum_p (R v_p) . dq needs the un-projected vertices; the
 * entry points above recover R v_p from vertex_proj ((q - t3d) / f: 41 MB more to read per 64 faces, and the caller must keep
 * the forward's output alive for the backward).  With v = mu + S alpha + E beta and dv = (f R)^T dq,
 *     sum_p (R v_p) . dq  =  sum_p v_p . dv_p / f  =  ( sum_p mu_p . dv_p  +  alpha . d alpha  +  beta . d beta ) / f,
 * whose first term the fused kernel forms from the 0.64 MB of mu beside the dv rows it builds anyway and whose other two are dot
 * products of the parameters with the coefficient gradients: linear, so every workgroup adds the product with ITS partial
 * gradients to its partial of d f and the fixed-order reduction sums them like every other output -- two launches, as before.
 * What it buys is MEMORY, not time: measured in one process (tools/bwd_ab_probe.py) 71.3 against 71.1 us per backward at 64 faces
 * and 56.3 against 53.2 at 32 -- the 41 MB of vertex_proj stream in for free beside the 153 MB of basis.
 * mu [3N] in the reference layout (16-byte aligned); every other output as fr_decode_3dmm_backward_packed; d f := 0 at f == 0 as
 * there; deterministic.  The three entry points agree to rounding.  FR_ERR_UNSUPPORTED where fr_decode_backward_basis_bytes is 0. */
int fr_decode_3dmm_backward_packed_mu(const float* grad_vertex_proj, const float* params, const float* mu, const void* packed_t,
                                      const float* R_override, int B, int N, int n_shape, int n_exp, float im_size,
                                      float* grad_params, void* workspace, size_t ws_bytes, void* hip_stream);

/* ---- test hook ---------------------------------------------------------------------------------------------
 * The screen-bin geometry the forward launcher chooses for a shape (no GPU needed): out = {rows per strip, strips,
 * triangle segments, 1 if the binned path covers the shape else 0 (the strip-scan fallback runs)}.  rows_override > 0
 * plays the FR_RENDER_ROWS tuning knob.  Used by tests/test_capi_cpu.py. */
void fr_debug_render_geom(int B, int ntri, int H, int W, int rows_override, int* out);

/* The kernels divide by 3.0f (render_depth_op.cc:217, 223, 361) through a 3-instruction exact sequence: this hook
 * compares it with x / 3.0f on the fp32 bit patterns [first, first + count) and writes the number of differing results
 * to the device word `mismatches`.  Used by tests/test_render_gpu.py (all 2^32 patterns). */
int fr_debug_div3_sweep(unsigned long long first, unsigned long long count, unsigned long long* mismatches,
                        void* hip_stream);

/* Measurement hook (bench.py `clock_GHz_held`; no reference counterpart): `blocks` 1,024-thread workgroups each issue
 * `iters` x 6 v_mfma_f32_16x16x4_f32 per wave (the decode's matrix instruction at the decode's occupancy) and write
 * ticks[2 b] = shader-clock ticks and ticks[2 b + 1] = 100 MHz ticks their loop took (device buffer of 2 * blocks 64-bit
 * words): clock held = 0.1 GHz * ticks[2 b] / ticks[2 b + 1]. */
int fr_debug_clock_probe(unsigned long long* ticks, int blocks, int iters, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* FR_HOTPATH_H_ */
