"""Launch plan for the CoarseNet -> render loop's hot path: 235-d parameters -> decode -> render_depth.

`FaceRecNet.vertices_transform` + `rendering_layer.ops.render_depth` allocate their outputs per call, like the
reference op does (render_depth_op.cc:442-445).  A serving / training loop that runs the same shapes every
iteration should not pay that: a `DecodeRenderPlan` owns the vertex buffer and the four output planes once
(HBM is 288 GB; one 64-face plan is 123 MB), keeps the ctypes argument lists prebuilt, launches both kernels on
torch's current HIP stream, and can be captured into a hipGraph (`capture()` / `replay()`), so one batch costs
one graph launch instead of a Python call chain.  Outputs are views of the plan's buffers: they are overwritten
by the next `step()`.  Forward only (no autograd); use rendering_layer.ops.render_depth when gradients are needed.
"""
import ctypes
import importlib.util
import os
import sys

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def _host():
    name = "_fr_hotpath_host"
    mod = sys.modules.get(name)
    if mod is None:
        spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG_DIR, "_lib.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return mod


class DecodeRenderPlan:
    def __init__(self, net, batch, height=None, width=None, texture=None):
        """net: nets.network.FaceRecNet (holds the packed basis, tri, vertex_code); texture: (3,N) or (B,3,N)
        tensor, default net.vertex_code (the PNCC colour code, reference network.py:116)."""
        h = _host()
        self._h = h
        self._L = h.lib()
        self.net = net
        self.device = net.device
        self.B = int(batch)
        self.H = int(net.im_size if height is None else height)
        self.W = int(net.im_size if width is None else width)
        self.N = net.nvert
        self.T = int(net.tri.shape[1])
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros((self.B, net.ndim), **f32)
        self.vertex_proj = torch.empty((self.B, 3, self.N), **f32)
        self.depth = torch.empty((self.B, self.H, self.W, 1), **f32)
        self.texture_image = torch.empty((self.B, self.H, self.W, 3), **f32)
        self.normal = torch.empty((self.B, self.H, self.W, 3), **f32)
        self.tri_ind = torch.empty((self.B, self.H, self.W, 1), **f32)
        tex = net.vertex_code if texture is None else h.require_gpu_f32(texture, "texture")
        self.texture = tex.contiguous()
        self.tex_batch = 1 if self.texture.dim() == 2 else int(self.texture.shape[0])
        if self.tex_batch not in (1, self.B) or self.texture.shape[-2] != 3 or self.texture.shape[-1] != self.N:
            raise ValueError("texture must be (3,N), (1,3,N) or (B,3,N)")
        ws_bytes = self._L.fr_render_depth_workspace_bytes(self.B, self.N, self.T, self.H, self.W)
        self._ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=self.device)
        self._ws_bytes = ws_bytes
        p = h.ptr
        # the decode entry point is fixed when the plan is built: the f32 chain, or -- when the opt-in Q30 arithmetic is
        # selected at that moment -- fr_decode_3dmm_q30 with a staging workspace the PLAN owns (nothing is allocated at
        # launch time, so the plan can be captured on any stream)
        basis = net._basis
        self.q30 = basis.use_q30()
        if self.q30:
            self._q_ws = torch.empty((basis.q30_ws_bytes,), dtype=torch.uint8, device=self.device)
            self._dec_fn = self._L.fr_decode_3dmm_q30
            self._dec_args = (p(self.params), p(basis.qimage()), None, self.B, self.N, net.ndim_shape, net.ndim_exp,
                              ctypes.c_float(float(net.im_size)), p(self.vertex_proj), p(self._q_ws), basis.q30_ws_bytes)
        else:
            self._dec_fn = self._L.fr_decode_3dmm
            self._dec_args = (p(self.params), p(basis.image), None, self.B, self.N, net.ndim_shape, net.ndim_exp,
                              ctypes.c_float(float(net.im_size)), p(self.vertex_proj))
        self._ren_args = (p(self.vertex_proj), p(net.tri), p(self.texture), self.B, self.N, self.T, self.H, self.W, 3,
                          self.tex_batch, p(self.depth), p(self.texture_image), p(self.normal), p(self.tri_ind),
                          p(self._ws), ws_bytes)
        self._graph = None
        # the triangle list is a constant of the model (reference network.py:178): convert + range-check it ONCE into the
        # workspace's table; every step then runs the emit and resolve phases only
        self._tri_packed = False
        self.pack_tri()

    def pack_tri(self):
        """(Re)builds the pre-validated triangle table in the workspace; call again after changing net.tri in place."""
        with torch.cuda.device(self.device):
            rc = self._L.fr_render_depth_forward_phases(*self._ren_args, self._stream(), 4)
        if rc:
            self._h.check(rc, "fr_render_depth_forward_phases(pack)")
        self._tri_packed = True

    # -- eager launches on the current stream ---------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def decode(self):
        rc = self._dec_fn(*self._dec_args, self._stream())
        if rc:
            self._h.check(rc, "fr_decode_3dmm_q30" if self.q30 else "fr_decode_3dmm")

    def render(self):
        rc = self._L.fr_render_depth_forward_phases(*self._ren_args, self._stream(), 3)
        if rc:
            self._h.check(rc, "fr_render_depth_forward_phases")

    def render_phase(self, phases):
        """phases = 1 launches only raster_emit_kernel, 2 only resolve_write_kernel, 4 only pack_tri_kernel."""
        rc = self._L.fr_render_depth_forward_phases(*self._ren_args, self._stream(), int(phases))
        if rc:
            self._h.check(rc, "fr_render_depth_forward_phases")

    def outputs(self):
        return self.depth, self.texture_image, self.normal, self.tri_ind

    def step(self, params=None):
        """decode + render of one batch.  `params` (B,d) is copied into the plan's buffer when given."""
        if params is not None:
            self.params.copy_(params.reshape(self.B, -1), non_blocking=True)
        self.decode()
        self.render()
        return self.outputs()

    # -- hipGraph ------------------------------------------------------------------------------------------------
    def capture(self):
        """Captures decode + render (reading self.params, writing the plan's outputs) into a hipGraph."""
        with torch.cuda.device(self.device):
            self.step()  # warm-up launch outside the capture (function attributes, lazy module load)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.decode()
                self.render()
        self._graph = g
        return g

    def replay(self, params=None):
        if self._graph is None:
            self.capture()
        if params is not None:
            self.params.copy_(params.reshape(self.B, -1), non_blocking=True)
        self._graph.replay()
        return self.outputs()
