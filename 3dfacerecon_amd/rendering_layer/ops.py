"""Operator surface of the rendering layer -- the MI355X drop-in for the reference's rendering_layer/ops.py.

Same names and argument meaning as the reference module (rendering_layer/ops.py:12,23,78-95):

    OP_NAMES                       ['render_depth']
    compile(op=None)               build the native library (hipcc --offload-arch=gfx950, not nvcc + g++)
    render_depth(ver, tri, texture, image, **kwargs)
                                   -> (depth [B,H,W,1], texture_image [B,H,W,3], normal [B,H,W,3], tri_ind [B,H,W,1])
    gradient                       flows to `ver` only (d depth / d vertex z); tri, texture, image get None

Tensors are torch.Tensors on an MI355X instead of tf.Tensors; the op is a torch.autograd.Function calling the
C ABI of include/fr_hotpath.h through ctypes on torch's current HIP stream.  Like the reference, importing the
module loads the native library and builds it first if the .so is missing (reference ops.py:63-72) -- but there
is no fallback path: no library or no GPU tensor => an exception.
"""
import collections
import importlib.util
import os
import sys
import threading
import weakref

import torch

# Register ops for compilation here (reference ops.py:12)
OP_NAMES = ['render_depth']

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host():
    """The ctypes host module (3dfacerecon_amd/_lib.py), loaded by path so that this file works both as
    `3dfacerecon_amd.rendering_layer.ops` and as the reference-style top-level `rendering_layer.ops`."""
    name = "_fr_hotpath_host"
    mod = sys.modules.get(name)
    if mod is None:
        spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG_DIR, "_lib.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return mod


def compile(op=None):
    """Build the native op library.  `op` is accepted for signature compatibility (reference ops.py:23-27);
    all ops of OP_NAMES live in one shared library."""
    if op is not None and op not in OP_NAMES:
        raise ValueError("unknown op %r (known: %s)" % (op, OP_NAMES))
    return _host().compile(force=True, verbose=True)


# build-on-import fallback, then load (reference ops.py:63-72).  A missing hipcc or a failing build raises here.
_host().lib()


# Render workspace (hit records, bucket offsets, tables: ~330 MB at 64 faces) reused across calls: one buffer per
# (device, stream), grown on demand.  The forward writes every part it later reads and nothing of it is needed by the
# backward, so calls on one stream can share it; the reference cudaMallocs / cudaFrees six buffers per call
# (render_depth_op.cu.cc:272-277, 335-340).
#   * bounded: at most WS_CACHE_MAX entries, least recently used evicted first (a caller that cycles streams no longer
#     accumulates one 330 MB buffer per stream handle it ever used); clear_workspace_cache() drops them all;
#   * never used while the current stream is being captured into a hipGraph: a captured launch must not point at a buffer
#     the cache may later replace, so the capture gets a buffer of its own from the graph's memory pool;
#   * guarded by a lock (autograd worker threads call the forward too).
# Each entry also remembers which triangle list its pre-validated triangle table (pack_tri_kernel) was built from -- the
# tensor OBJECT (held weakly), its torch version counter and the geometry -- so a loop that renders with the same `tri`
# tensor every call (the reference makes it a tf.constant, network.py:178) packs it once, not once per call.  A new tensor
# is never mistaken for an old one whose memory the caching allocator handed out again (object identity, not data_ptr);
# the version counter sees in-place torch writes; a caller that rewrites the tensor's memory behind torch's back must call
# clear_workspace_cache().
WS_CACHE_MAX = 4
_WS_CACHE = collections.OrderedDict()
_WS_LOCK = threading.Lock()


class _WsEntry:
    __slots__ = ("buf", "tri_ref", "tri_key")

    def __init__(self, buf):
        self.buf = buf
        self.tri_ref = None   # weakref to the tensor the table was packed from
        self.tri_key = None   # (its version counter, its data_ptr) + geometry


def clear_workspace_cache():
    """Releases every cached render workspace (their memory returns to torch's caching allocator)."""
    with _WS_LOCK:
        _WS_CACHE.clear()


def _workspace(dev, nbytes):
    """-> (entry, cached): the workspace entry for torch's current stream on `dev`."""
    if torch.cuda.is_current_stream_capturing():
        return _WsEntry(torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev)), False
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    with _WS_LOCK:
        ent = _WS_CACHE.get(key)
        if ent is None or ent.buf.numel() < nbytes:
            ent = _WsEntry(torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev))
            _WS_CACHE[key] = ent
        _WS_CACHE.move_to_end(key)
        while len(_WS_CACHE) > WS_CACHE_MAX:
            _WS_CACHE.popitem(last=False)
    return ent, True


def _render_phases(ent, cached, tri_c, geom):
    """-> (phases, pending): phases = 7 (pack + emit + resolve), or 3 when the entry's triangle table was packed from this very
    list for this geometry under the launcher options in force now; `pending` is the record _table_packed() commits once the
    call that packs the table has RETURNED 0 (a failed or unsupported call leaves no claim about the table behind)."""
    # an inference tensor (created under torch.inference_mode()) has no version counter: in-place writes to it cannot be
    # seen, so its table is never reused -- packed every call, like a caller that passes a new tensor each time
    if tri_c.is_inference():
        ent.tri_ref = ent.tri_key = None
        return 7, None
    # (the option epoch: FR_RENDER_IMPL / FR_RENDER_ROWS / FR_EMIT_ORDER decide whether and how the table is written, and
    # can only change through _lib.set_option, which bumps it)
    key = (tri_c._version, tri_c.data_ptr(), _host().option_epoch()) + geom
    if cached and ent.tri_ref is not None and ent.tri_ref() is tri_c and ent.tri_key == key:
        return 3, None
    ent.tri_ref = ent.tri_key = None   # whatever this call does, the old record no longer describes the table
    return 7, ((weakref.ref(tri_c), key) if cached else None)


def _table_packed(ent, pending):
    if pending is not None:
        ent.tri_ref, ent.tri_key = pending


def _check_forward_shapes(ver, tri, texture, image):
    # the OP_REQUIRES checks of RenderDepthOp::Compute (render_depth_op.cc:397-418), same messages
    if image.dim() != 4 or ver.dim() != 3 or tri.dim() != 2 or texture.dim() not in (2, 3):
        raise ValueError("render_depth expects ver [B,3,nver], tri [3,ntri], texture [B,3,nver], image [B,H,W,C]")
    B = image.shape[0]
    if ver.shape[0] != B:
        raise ValueError("The vertex's batch is not the same as image batch")
    if ver.shape[1] != 3:
        raise ValueError("The vertex is not Batch x 3 x nver")
    if tri.shape[0] != 3:
        raise ValueError("The tri is not 3 x ntri")
    if texture.shape[-2] != 3:
        raise ValueError("The texture channel must be equal to image channel namely 3")
    if texture.shape[-1] != ver.shape[2]:
        raise ValueError("The texture is not Batch x 3 x nver")
    if texture.dim() == 3 and texture.shape[0] not in (1, B):
        raise ValueError("The texture's batch is neither 1 nor the image batch")


def _backward_call(h, g, tri_c, tri_ind, vertex_grad, B, nver, ntri, H, W, dev):
    """fr_render_depth_backward_ws with its workspace (one 16-byte record per pixel)."""
    L = h.lib()
    nws = L.fr_render_depth_backward_workspace_bytes(B, H, W)
    ws = torch.empty((max(nws, 16),), dtype=torch.uint8, device=dev)
    rc = L.fr_render_depth_backward_ws(h.ptr(g), h.ptr(tri_c), h.ptr(tri_ind), h.ptr(vertex_grad), B, nver, ntri, H, W,
                                       h.ptr(ws), nws, h.stream_ptr(dev))
    h.check(rc, "fr_render_depth_backward")


class _RenderDepth(torch.autograd.Function):
    """RenderDepth / RenderDepthGrad (render_depth_op.cc:535-589) as one autograd node."""

    @staticmethod
    def forward(ctx, ver, tri, texture, image):
        h = _host()
        _check_forward_shapes(ver, tri, texture, image)
        ver_c = h.require_gpu_f32(ver, "ver")
        tri_c = h.require_gpu_f32(tri, "tri")
        tex_c = h.require_gpu_f32(texture, "texture")
        if not image.is_cuda:
            raise RuntimeError("image is on %s: the fr_hotpath kernels run on an MI355X only" % image.device)
        B, H, W = int(image.shape[0]), int(image.shape[1]), int(image.shape[2])
        nver, ntri = int(ver_c.shape[2]), int(tri_c.shape[1])
        tex_batch = 1 if tex_c.dim() == 2 else int(tex_c.shape[0])
        if tex_batch not in (1, B):
            raise ValueError("The texture's batch is neither 1 nor the image batch")
        dev = ver_c.device
        opts = dict(dtype=torch.float32, device=dev)
        depth = torch.empty((B, H, W, 1), **opts)
        tex_img = torch.empty((B, H, W, 3), **opts)
        normal = torch.empty((B, H, W, 3), **opts)
        tri_ind = torch.empty((B, H, W, 1), **opts)
        L = h.lib()
        with torch.cuda.device(dev):
            ws_bytes = L.fr_render_depth_workspace_bytes(B, nver, ntri, H, W)
            if ws_bytes:
                ent, cached = _workspace(dev, ws_bytes)
                phases, pending = _render_phases(ent, cached, tri_c, (B, nver, ntri, H, W))
                rc = L.fr_render_depth_forward_phases(h.ptr(ver_c), h.ptr(tri_c), h.ptr(tex_c), B, nver, ntri, H, W, 3,
                                                      tex_batch, h.ptr(depth), h.ptr(tex_img), h.ptr(normal), h.ptr(tri_ind),
                                                      h.ptr(ent.buf), ws_bytes, h.stream_ptr(dev), phases)
                if rc == 0:
                    _table_packed(ent, pending)
            else:
                rc = L.fr_render_depth_forward(h.ptr(ver_c), h.ptr(tri_c), h.ptr(tex_c), B, nver, ntri, H, W, 3, tex_batch,
                                               h.ptr(depth), h.ptr(tex_img), h.ptr(normal), h.ptr(tri_ind), None, 0,
                                               h.stream_ptr(dev))
        h.check(rc, "fr_render_depth_forward")
        ctx.save_for_backward(tri_c, tri_ind)
        ctx.dims = (B, nver, ntri, H, W)
        ctx.set_materialize_grads(False)  # an unused depth output (the SfS renders, network.py:423, 454) costs no backward
        return depth, tex_img, normal, tri_ind

    @staticmethod
    def backward(ctx, depth_grad, texture_image_grad, normal_grad, tri_ind_grad):
        # only depth_grad is used; vertex has gradients, tri / texture / image do not (reference ops.py:86-95)
        h = _host()
        tri_c, tri_ind = ctx.saved_tensors
        B, nver, ntri, H, W = ctx.dims
        dev = tri_c.device
        if depth_grad is None:   # nothing downstream used `depth`: the vertices get no gradient (reference ops.py:95)
            return None, None, None, None
        vertex_grad = torch.empty((B, 3, nver), dtype=torch.float32, device=dev)
        g = h.require_gpu_f32(depth_grad, "depth_grad")
        with torch.cuda.device(dev):
            _backward_call(h, g, tri_c, tri_ind, vertex_grad, B, nver, ntri, H, W, dev)
        return vertex_grad, None, None, None


class _RenderingLayerFused(torch.autograd.Function):
    """render_depth + the post-processing of FaceRecNet.rendering_layer (nets/network.py:185-199) as one kernel pass:
    (ver, tri, texture, im_gray) -> (net_input [B,H,W,7], depth_img [B,H,W,1], depth, tri_ind)."""

    @staticmethod
    def forward(ctx, ver, tri, texture, im_gray):
        h = _host()
        image = im_gray
        _check_forward_shapes(ver, tri, texture, image)
        ver_c = h.require_gpu_f32(ver, "ver")
        tri_c = h.require_gpu_f32(tri, "tri")
        tex_c = h.require_gpu_f32(texture, "texture")
        img_c = h.require_gpu_f32(im_gray, "im_gray")
        if img_c.shape[-1] != 1:
            raise ValueError("im_gray must be [B,H,W,1]")
        B, H, W = int(img_c.shape[0]), int(img_c.shape[1]), int(img_c.shape[2])
        nver, ntri = int(ver_c.shape[2]), int(tri_c.shape[1])
        tex_batch = 1 if tex_c.dim() == 2 else int(tex_c.shape[0])
        dev = ver_c.device
        opts = dict(dtype=torch.float32, device=dev)
        net_in = torch.empty((B, H, W, 7), **opts)
        depth_img = torch.empty((B, H, W, 1), **opts)
        depth = torch.empty((B, H, W, 1), **opts)
        tri_ind = torch.empty((B, H, W, 1), **opts)
        L = h.lib()
        with torch.cuda.device(dev):
            ws_bytes = L.fr_render_depth_workspace_bytes(B, nver, ntri, H, W)
            ent, cached = _workspace(dev, ws_bytes)
            # the same "pack once while the same `tri` tensor is passed" rule as render_depth (the table is the same table)
            phases, pending = _render_phases(ent, cached, tri_c, (B, nver, ntri, H, W))
            rc = L.fr_rendering_layer_forward_phases(h.ptr(ver_c), h.ptr(tri_c), h.ptr(tex_c), h.ptr(img_c), B, nver, ntri, H, W,
                                                     tex_batch, h.ptr(net_in), h.ptr(depth_img), h.ptr(depth), h.ptr(tri_ind),
                                                     h.ptr(ent.buf), ws_bytes, h.stream_ptr(dev), phases)
        if rc == -4:
            raise NotImplementedError("fused rendering layer: shape only covered by the fallback rasteriser")
        h.check(rc, "fr_rendering_layer_forward")
        _table_packed(ent, pending)
        ctx.save_for_backward(tri_c, tri_ind, depth, img_c)
        ctx.dims = (B, nver, ntri, H, W)
        ctx.mark_non_differentiable(tri_ind)
        return net_in, depth_img, depth, tri_ind

    @staticmethod
    def backward(ctx, g_net_in, g_depth_img, g_depth, g_tri_ind):
        # only the mask channel and the depth image depend on the vertices (through depth, hence vertex z):
        #   mask = clip(depth, 1e-6, 1) * im  ->  g * im where 1e-6 <= depth <= 1;   depth_img = max(depth, 1e-6)
        h = _host()
        tri_c, tri_ind, depth, img = ctx.saved_tensors
        B, nver, ntri, H, W = ctx.dims
        dg = torch.zeros_like(depth)
        if g_net_in is not None:
            dg = dg + g_net_in[..., 0:1] * img * ((depth >= 1e-6) & (depth <= 1.0)).to(depth.dtype)
        if g_depth_img is not None:
            dg = dg + g_depth_img * (depth >= 1e-6).to(depth.dtype)
        if g_depth is not None:
            dg = dg + g_depth
        dg = dg.contiguous()
        vertex_grad = torch.empty((B, 3, nver), dtype=torch.float32, device=depth.device)
        with torch.cuda.device(depth.device):
            _backward_call(h, dg, tri_c, tri_ind, vertex_grad, B, nver, ntri, H, W, depth.device)
        return vertex_grad, None, None, None


def rendering_layer_fused(ver, tri, texture, im_gray):
    """One-pass rendering layer: returns (net_input [B,H,W,7] = [mask*im | pncc | normal],
    depth_img, raw depth, tri_ind).  Raises NotImplementedError for shapes only the fallback rasteriser covers."""
    return _RenderingLayerFused.apply(ver, tri, texture, im_gray)


def render_depth(ver, tri, texture, image, **kwargs):
    """Forward function of RenderDepth (reference ops.py:78-81).

    The first output is the rendered depth, the fourth the triangle index each depth pixel corresponds to.
    `image` only donates the batch / height / width (its values are never read, render_depth_op.cc:397-403).
    `**kwargs` is accepted for call compatibility (TF passed `name=`); unknown keys are ignored.
    """
    return _RenderDepth.apply(ver, tri, texture, image)


def render_depth_grad(depth_grad, ver, tri, depth, tri_ind, image):
    """The RenderDepthGrad op called directly with the reference's argument order
    (depth_grad, vertex, tri, depth, tri_ind, image; render_depth_op.cc:473-478) -> vertex_grad [B,3,nver]."""
    h = _host()
    g = h.require_gpu_f32(depth_grad, "depth_grad")
    tri_c = h.require_gpu_f32(tri, "tri")
    ti = h.require_gpu_f32(tri_ind, "tri_ind")
    if ver.dim() != 3 or ver.shape[1] != 3:
        raise ValueError("The vertex is not Batch x 3 x nver")
    if ver.shape[0] != image.shape[0]:
        raise ValueError("The vertex's batch is not the same as image batch")
    if tri_c.dim() != 2 or tri_c.shape[0] != 3:
        raise ValueError("The tri is not 3 x ntri")
    B, H, W = int(image.shape[0]), int(image.shape[1]), int(image.shape[2])
    nver, ntri = int(ver.shape[2]), int(tri_c.shape[1])
    dev = g.device
    vertex_grad = torch.empty((B, 3, nver), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _backward_call(h, g, tri_c, ti, vertex_grad, B, nver, ntri, H, W, dev)
    return vertex_grad
