"""Hot-path half of the reference's nets/network.py `FaceRecNet`, on PyTorch-ROCm + the gfx950 C ABI.

Only the 3DMM decoder and the rendering-layer wrapper are here -- the methods on the CoarseNet -> render loop
that BASELINE.json's north_star names:

    vertices_transform(pred_params)            reference nets/network.py:140-171   (HIP: fr_decode_3dmm)
    rendering_layer(vertex_proj, tri, colors)  reference nets/network.py:174-201   (HIP: fr_render_depth_forward)
    set_constraints / parse_pose_params / rotation_matrix / rotation_matrix_batch  (:204-218, :253-297)

CoarseNet / FineNet / losses / summaries / checkpoints (network.py:103-136, 311-603) are out of scope.
The 235-d layout is unchanged: [phi, gamma, theta, tx, ty, tz, f | shape x199 | exp x29].
"""
import importlib.util
import os
import sys
from math import cos, sin

import numpy as np
import torch

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, relpath):
    mod = sys.modules.get(name)
    if mod is None:
        spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG_DIR, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return mod


def _host():
    return _load("_fr_hotpath_host", "_lib.py")


def _ops():
    return _load("_fr_hotpath_ops", os.path.join("rendering_layer", "ops.py"))


def _publish(device):
    """A lazily built constant image is packed on whichever stream first needs it (possibly autograd's backward stream) and
    then cached with no event: wait for the pack kernel ONCE, here, so that a later consumer on any other stream finds the
    image complete.  (Not under graph capture, where a synchronize is illegal: the capture's own stream order covers it.)"""
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).synchronize()


class PackedBasis:
    """The constant basis (a tf.constant of the reference model, network.py:41-43) in the layouts the decode kernels
    stream: the f32 MFMA-fragment image, built at once (fr_decode_pack_basis), and -- only if the opt-in Q30 arithmetic
    is ever selected (_lib.set_decode_arith / FR_DECODE_ARITH=q30) -- its digit image, built on first use.  A default user
    holds 153 MB per basis and never launches a q_* kernel."""

    def __init__(self, mu, pc_shape, pc_exp, nvert, ndim_shape, ndim_exp, device):
        h = _host()
        L = h.lib()
        self.mu, self.pc_shape, self.pc_exp = mu, pc_shape, pc_exp
        self.nvert, self.ndim_shape, self.ndim_exp, self.device = nvert, ndim_shape, ndim_exp, device
        nbytes = L.fr_decode_packed_basis_bytes(nvert, ndim_shape, ndim_exp)
        self.image = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            rc = L.fr_decode_pack_basis(h.ptr(mu), h.ptr(pc_shape), h.ptr(pc_exp), nvert, ndim_shape, ndim_exp,
                                        h.ptr(self.image), nbytes, h.stream_ptr(device))
        h.check(rc, "fr_decode_pack_basis")
        _publish(device)
        self._qimage = None
        self._image_t = None   # K-major image for the backward's reduction over the vertices, built at the first backward
        self._t_owner = None   # another PackedBasis of the SAME pc_shape / pc_exp whose image_t this one shares (the K-major
                               # image does not contain mu, so a mu = 0 twin needs no second 153 MB copy)
        self.q30_ws_bytes = L.fr_decode_q30_workspace_bytes(ndim_shape, ndim_exp)
        self._packed_ok = None
        self.backward_from_mu = False   # True: the autograd node's backward does not keep the forward's output (see _Decode3DMM)
        self._bwd_ws = {}      # decode-backward workspaces by (device, stream, bytes): backward_workspace()

    def image_t(self):
        """The basis packed for the decode backward (fr_decode_backward_pack_basis), built on first use: callers that never
        take a gradient never hold it."""
        if self._t_owner is not None:
            return self._t_owner.image_t()
        if self._image_t is None:
            h = _host()
            L = h.lib()
            nbytes = L.fr_decode_backward_basis_bytes(self.nvert, self.ndim_shape, self.ndim_exp)
            if nbytes == 0:   # not served by the packed kernel (more than 256 coefficients, or fewer than 16 vertices): reference-layout entry point
                return None
            buf = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                rc = L.fr_decode_backward_pack_basis(h.ptr(self.pc_shape), h.ptr(self.pc_exp), self.nvert, self.ndim_shape,
                                                     self.ndim_exp, h.ptr(buf), nbytes, h.stream_ptr(self.device))
            h.check(rc, "fr_decode_backward_pack_basis")
            _publish(self.device)
            self._image_t = buf
        return self._image_t

    def backward_packed_ok(self):
        """True when the fused backward (packed image, d f from mu) serves this basis / mesh."""
        if self._t_owner is not None:
            return self._t_owner.backward_packed_ok()
        if self._packed_ok is None:
            self._packed_ok = _host().lib().fr_decode_backward_basis_bytes(self.nvert, self.ndim_shape, self.ndim_exp) > 0
        return self._packed_ok

    def backward_workspace(self, B, dev):
        """(bytes, buffer) of the decode backward's workspace for B faces on torch's current stream of `dev`.  The buffer is kept
        per (stream, size) -- launches on one stream are ordered, so consecutive backwards may share it; at most four are held
        (least recently used dropped); under graph capture a fresh one is taken (a captured launch must not point at a buffer
        the cache may later replace).  The size query and the allocation used to be paid per call: with the kernels at ~70 us
        the autograd route was host-bound below 64 faces (profiles/round4_bwd_probe.log)."""
        L = _host().lib()
        nws = L.fr_decode_backward_workspace_bytes(B, self.nvert, self.ndim_shape, self.ndim_exp)
        if torch.cuda.is_current_stream_capturing():
            return nws, torch.empty((max(nws, 16),), dtype=torch.uint8, device=dev)
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, nws)
        buf = self._bwd_ws.pop(key, None)
        if buf is None:
            buf = torch.empty((max(nws, 16),), dtype=torch.uint8, device=dev)
        self._bwd_ws[key] = buf               # (re-inserted last: most recently used)
        while len(self._bwd_ws) > 4:
            self._bwd_ws.pop(next(iter(self._bwd_ws)))
        return nws, buf

    def use_q30(self):
        """True when the Q30 entry point serves this call: selected AND the shape is covered (else the f32 chain)."""
        h = _host()
        return h.decode_arith() == h.DECODE_ARITH_Q30 and self.q30_ws_bytes > 0

    def qimage(self):
        if self._qimage is None:
            h = _host()
            L = h.lib()
            nbytes = L.fr_decode_q30_image_bytes(self.nvert, self.ndim_shape, self.ndim_exp)
            buf = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                rc = L.fr_decode_q30_pack(h.ptr(self.mu), h.ptr(self.pc_shape), h.ptr(self.pc_exp), self.nvert,
                                          self.ndim_shape, self.ndim_exp, h.ptr(buf), nbytes, h.stream_ptr(self.device))
            h.check(rc, "fr_decode_q30_pack")
            _publish(self.device)
            self._qimage = buf
        return self._qimage

    def decode(self, params, R, B, im_size, out, workspace=None):
        """One decode launch on torch's current stream.  `workspace` (Q30 only): a caller-owned staging buffer of
        q30_ws_bytes; by default one is taken from torch's caching allocator for this call (stream-safe)."""
        h = _host()
        L = h.lib()
        dev = params.device
        with torch.cuda.device(dev):
            if self.use_q30():
                ws = workspace if workspace is not None else torch.empty((self.q30_ws_bytes,), dtype=torch.uint8, device=dev)
                rc = L.fr_decode_3dmm_q30_lv(h.ptr(params), h.ptr(self.qimage()), h.ptr(R), B, self.nvert, self.ndim_shape,
                                             self.ndim_exp, float(im_size), h.q30_levels(), h.ptr(out), h.ptr(ws),
                                             self.q30_ws_bytes, h.stream_ptr(dev))
                h.check(rc, "fr_decode_3dmm_q30_lv")
            else:
                rc = L.fr_decode_3dmm(h.ptr(params), h.ptr(self.image), h.ptr(R), B, self.nvert, self.ndim_shape,
                                      self.ndim_exp, float(im_size), h.ptr(out), h.stream_ptr(dev))
                h.check(rc, "fr_decode_3dmm")


class _Decode3DMM(torch.autograd.Function):
    """vertices_transform as one autograd node: fr_decode_3dmm forward, fr_decode_3dmm_backward for the gradient TF
    autodiff derives from network.py:140-171 (d alpha, d beta, d t3d, d f; the three angles get zero because the
    reference's rotation goes through tf.py_func, network.py:150, which has no gradient)."""

    @staticmethod
    def forward(ctx, params, net, R, basis=None, im_size=None):
        B = int(params.shape[0])
        out = torch.empty((B, 3, net.nvert), dtype=torch.float32, device=params.device)
        basis = net._basis if basis is None else basis
        im_size = float(net.im_size if im_size is None else im_size)
        basis.decode(params, R, B, im_size, out)
        ctx.net = net
        ctx.basis = basis
        ctx.im_size = im_size
        # The fused backward has two forms of d f.  Default: from the forward's output (fr_decode_3dmm_backward_packed; `out` stays
        # alive for the backward -- 41 MB per 64-face decode).  `basis.backward_from_mu = True` selects the form that needs no
        # forward output (fr_decode_3dmm_backward_packed_mu: d f from mu and the coefficient gradients) -- a MEMORY saving, not a
        # faster kernel (same time at 64 faces, 3 us more at 32: profiles/round5_probes/r5d).  Bases / meshes the fused kernel
        # does not serve take the reference-layout entry point, which needs `out`.
        ctx.packed = basis.backward_packed_ok()
        ctx.from_mu = ctx.packed and bool(getattr(basis, "backward_from_mu", False))
        if ctx.from_mu:
            ctx.save_for_backward(params, R if R is not None else params.new_empty(0))
        else:
            ctx.save_for_backward(params, R if R is not None else params.new_empty(0), out)
        ctx.has_R = R is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h = _host()
        net = ctx.net
        basis = ctx.basis
        params, R = ctx.saved_tensors[0], ctx.saved_tensors[1]
        B = int(params.shape[0])
        g = h.require_gpu_f32(grad_out, "grad_vertex_proj")
        gp = torch.empty_like(params)
        L = h.lib()
        dev = params.device
        with torch.cuda.device(dev):
            nws, ws = basis.backward_workspace(B, dev)
            Rp = h.ptr(R) if ctx.has_R else None
            if ctx.from_mu:
                rc = L.fr_decode_3dmm_backward_packed_mu(h.ptr(g), h.ptr(params), h.ptr(basis.mu), h.ptr(basis.image_t()), Rp, B,
                                                         net.nvert, net.ndim_shape, net.ndim_exp, ctx.im_size, h.ptr(gp),
                                                         h.ptr(ws), nws, h.stream_ptr(dev))
            elif ctx.packed:
                # (the fused kernel streams the packed image: built once per basis, at the first backward)
                rc = L.fr_decode_3dmm_backward_packed(h.ptr(g), h.ptr(params), h.ptr(ctx.saved_tensors[2]), h.ptr(basis.image_t()),
                                                      Rp, B, net.nvert, net.ndim_shape, net.ndim_exp, ctx.im_size, h.ptr(gp),
                                                      h.ptr(ws), nws, h.stream_ptr(dev))
            else:   # more than 256 coefficients or fewer than 16 vertices (not the model's): reference-layout entry point
                out = ctx.saved_tensors[2]
                rc = L.fr_decode_3dmm_backward(h.ptr(g), h.ptr(params), h.ptr(out), h.ptr(basis.pc_shape),
                                               h.ptr(basis.pc_exp), Rp, B, net.nvert, net.ndim_shape, net.ndim_exp,
                                               ctx.im_size, h.ptr(gp), h.ptr(ws), nws, h.stream_ptr(dev))
        h.check(rc, "fr_decode_3dmm_backward")
        return gp, None, None, None, None


class FaceRecNet:
    def __init__(self, im_gray=None, params_label=None, mesh_data=None, nIter=4, batch_size=64, im_size=200,
                 weight_decay=1e-4, device="cuda"):
        self.im_gray = im_gray
        self.params_label = params_label
        self.nIter = nIter
        self.batch_size = batch_size
        self.im_size = im_size
        self.weight_decay = weight_decay
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FaceRecNet hot path needs an MI355X device (got %s); there is no CPU fallback"
                               % self.device)

        # mesh data (same dict keys as utils/parser_3dmm.py:50-60)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.vertex_code = torch.as_tensor(np.asarray(mesh_data['vertex'], np.float32), **f32)    # (3, N)
        self.tri = torch.as_tensor(np.asarray(mesh_data['tri'], np.float32), **f32)               # (3, T) float ids
        self.mu = torch.as_tensor(np.asarray(mesh_data['mu'], np.float32).reshape(-1), **f32)      # (3*N,)
        self.pc_shape = torch.as_tensor(np.asarray(mesh_data['pc_shape'], np.float32), **f32)     # (3*N, ndim_shape)
        self.pc_exp = torch.as_tensor(np.asarray(mesh_data['pc_exp'], np.float32), **f32)         # (3*N, ndim_exp)

        # albedo model of the shape-from-shading loss (network.py:26, 45-47): mean texture (3,N), the first ndim_tex
        # texture components (3N, ndim_tex) and their coefficients (ndim_tex, 1); optional in the asset dict
        self.ndim_tex = 10
        self.mu_tex = self.pc_tex = self.param_tex = None
        if mesh_data.get('mu_tex') is not None:
            self.mu_tex = torch.as_tensor(np.asarray(mesh_data['mu_tex'], np.float32), **f32)
        if mesh_data.get('pc_tex') is not None and mesh_data.get('param_tex') is not None:
            self.pc_tex = torch.as_tensor(np.asarray(mesh_data['pc_tex'], np.float32)[:, 0:self.ndim_tex], **f32)
            self.param_tex = torch.as_tensor(np.asarray(mesh_data['param_tex'], np.float32)[0:self.ndim_tex, :], **f32)

        # mesh info
        self.ndim_shape = int(mesh_data['ndim_shape'])
        self.ndim_exp = int(mesh_data['ndim_exp'])
        self.ndim_pose = int(mesh_data['ndim_pose'])
        self.ndim = self.ndim_pose + self.ndim_shape + self.ndim_exp
        self.nvert = int(self.mu.numel() // 3)
        if self.ndim_pose != 7:
            raise ValueError("ndim_pose must be 7")
        if tuple(self.pc_shape.shape) != (3 * self.nvert, self.ndim_shape) or \
                tuple(self.pc_exp.shape) != (3 * self.nvert, self.ndim_exp):
            raise ValueError("pc_shape / pc_exp must be (3*Nvert, ndim)")

        # one-time re-layout of the constant basis into MFMA-fragment order (include/fr_hotpath.h)
        self._basis = PackedBasis(self.mu, self.pc_shape, self.pc_exp, self.nvert, self.ndim_shape, self.ndim_exp,
                                  self.device)
        self._basis_nomu = None  # second image with mu = 0, built on first use by geometry_product()

        # initial parameters (network.py:57-62)
        geo = torch.zeros((batch_size, self.ndim_shape + self.ndim_exp), **f32)
        init_pose = torch.tensor([0, 0, 0, im_size / 2.0, im_size / 2.0, 0, 0.001], **f32)
        self.pred_params = torch.cat([init_pose[None].repeat(batch_size, 1), geo], 1)[:, None, None, :]  # (B,1,1,d)
        self.init_pred_params = self.pred_params.clone()  # eager callers restart every forward from this constant

    # ---- 3DMM decode ----------------------------------------------------------------------------------
    def vertices_transform(self, pred_params, R=None):
        """(B,1,1,d) or (B,d) parameters -> projected vertices (B,3,N)  (network.py:140-171).

        R: optional host-computed (B,3,3) rotation (what the reference gets from tf.py_func at :150); by default
        the rotation is evaluated inside the kernel in float64."""
        h = _host()
        p = pred_params
        if p.dim() == 4:
            p = p.reshape(p.shape[0], p.shape[-1])
        if p.dim() != 2 or p.shape[1] != self.ndim:
            raise ValueError("pred_params must be (B,1,1,%d) or (B,%d)" % (self.ndim, self.ndim))
        p = h.require_gpu_f32(p, "pred_params")
        B = int(p.shape[0])
        Rc = None
        if R is not None:
            Rc = h.require_gpu_f32(torch.as_tensor(R, dtype=torch.float32, device=p.device), "R")
            if tuple(Rc.shape) != (B, 3, 3):
                raise ValueError("R must be (B,3,3)")
        return _Decode3DMM.apply(p, self, Rc)

    def geometry_product(self, geometry_params):
        """(B, ndim_shape + ndim_exp) coefficients -> basis product [pc_shape | pc_exp] . coeff as (B,3,N) (blocked rows),
        the `tf.matmul(geometry_basis, geometry, transpose_b=True)` of the geometry loss (network.py:347-353), on the
        MFMA decode kernel: a second packed image with mu = 0 is decoded with the identity pose (R = I, f = 1, t = 0,
        im_size = 1).  The y row comes out as (1 - y) - 1, i.e. -y to within half an ulp of max(1, |y|) -- the loss
        squares it.  Differentiable (fr_decode_3dmm_backward)."""
        h = _host()
        g = h.require_gpu_f32(geometry_params, "geometry_params")
        B = int(g.shape[0])
        if g.dim() != 2 or g.shape[1] != self.ndim_shape + self.ndim_exp:
            raise ValueError("geometry_params must be (B,%d)" % (self.ndim_shape + self.ndim_exp))
        if self._basis_nomu is None:
            self._basis_nomu = PackedBasis(torch.zeros_like(self.mu), self.pc_shape, self.pc_exp, self.nvert,
                                           self.ndim_shape, self.ndim_exp, self.device)
            self._basis_nomu._t_owner = self._basis   # same pc_shape / pc_exp: one K-major backward image for both
        pose = torch.zeros((B, self.ndim_pose), dtype=torch.float32, device=g.device)
        pose[:, 6] = 1.0
        eye = torch.eye(3, dtype=torch.float32, device=g.device)[None].repeat(B, 1, 1).contiguous()
        return _Decode3DMM.apply(torch.cat([pose, g], 1), self, eye, self._basis_nomu, 1.0)

    # ---- rendering layer wrapper --------------------------------------------------------------------------
    def rendering_layer(self, vertex_proj, triangles, colors, im_gray=None):
        """(network.py:174-201) -> pncc_batch, normalimg_batch, maskimg_batch, depthimg_batch."""
        im_gray = self.im_gray if im_gray is None else im_gray
        ver = vertex_proj.float()
        tri = torch.as_tensor(triangles, dtype=torch.float32, device=ver.device)
        tex = torch.as_tensor(colors, dtype=torch.float32, device=ver.device)  # (3,N) shared across the batch
        B = ver.shape[0]
        if im_gray is None:
            im_gray = torch.ones((B, self.im_size, self.im_size, 1), dtype=torch.float32, device=ver.device)
        image = im_gray.expand(-1, -1, -1, 3)
        depth, tex_img, normal, _ = _ops().render_depth(ver=ver, tri=tri, texture=tex, image=image)
        # 1. pncc result
        pncc_batch = torch.clamp(tex_img, 1e-6, 1.0)
        # 2. normal map: flip normals with negative z, normalise by magnitude
        flip = normal[..., 2:3] < 0
        normal = torch.where(flip, -1.0 * normal, normal)
        mag = (normal * normal).sum(-1)
        mag = torch.where(mag > 1e-6, mag, torch.ones_like(mag))
        normalimg_batch = normal / (torch.sqrt(mag) + 1e-6)[..., None]
        # 3. masked image
        mask = torch.clamp(depth, 1e-6, 1.0)
        maskimg_batch = mask * im_gray
        # 4. depth image
        depthimg_batch = torch.clamp_min(depth, 1e-6)
        return pncc_batch, normalimg_batch, maskimg_batch, depthimg_batch

    def coarse_net_input(self, vertex_proj, triangles=None, colors=None, im_gray=None):
        """The 7-channel CoarseNet input [maskimg | pncc | normal] (network.py:122) and the depth image, produced by
        the fused kernel pass (falls back to rendering_layer + concat for shapes it does not cover)."""
        im_gray = self.im_gray if im_gray is None else im_gray
        ver = vertex_proj.float()
        tri = self.tri if triangles is None else torch.as_tensor(triangles, dtype=torch.float32, device=ver.device)
        tex = self.vertex_code if colors is None else torch.as_tensor(colors, dtype=torch.float32, device=ver.device)
        if im_gray is None:
            im_gray = torch.ones((ver.shape[0], self.im_size, self.im_size, 1), dtype=torch.float32, device=ver.device)
        try:
            net_in, depth_img, _, _ = _ops().rendering_layer_fused(ver, tri, tex, im_gray)
        except NotImplementedError:
            pncc, normal, mask, depth_img = self.rendering_layer(ver, tri, tex, im_gray=im_gray)
            net_in = torch.cat([mask, pncc, normal], dim=3)
        return net_in, depth_img

    def compute_abedo_image(self, vertices, triangles, abedos, im_gray=None):
        """Albedo (3,N) -> albedo image + normalised normal map through a second render (network.py:394-417)."""
        ver = vertices.float()
        tri = torch.as_tensor(triangles, dtype=torch.float32, device=ver.device)
        tex = torch.as_tensor(abedos, dtype=torch.float32, device=ver.device)
        B = ver.shape[0]
        image = torch.zeros((B, self.im_size, self.im_size, 3), dtype=torch.float32, device=ver.device)
        _, tf_abedo, normal, _ = _ops().render_depth(ver=ver, tri=tri, texture=tex, image=image)
        abedos_image = torch.clamp_min(tf_abedo, 1e-6).mean(dim=-1, keepdim=True)  # (B, H, W, 1)
        flip = normal[..., 2:3] < 0
        normal = torch.where(flip, -1.0 * normal, normal)
        mag = (normal * normal).sum(-1)
        mag = torch.where(mag > 1e-6, mag, torch.ones_like(mag))
        normal_map = normal / (torch.sqrt(mag) + 1e-6)[..., None]
        return abedos_image, normal_map

    def depth_rendering_layer(self):
        """(network.py:300-309)"""
        self.vertices_proj = self.vertices_transform(self.pred_params)
        self.pncc_batch, self.normal_batch, self.maskimg_batch, self.coarse_depth_map = \
            self.rendering_layer(self.vertices_proj, self.tri, self.vertex_code)
        return self.coarse_depth_map

    # ---- parameter helpers ----------------------------------------------------------------------------------
    def set_constraints(self, pred_params):
        """sigmoid -> value ranges of the 235-d vector (network.py:204-218)."""
        s = torch.sigmoid(pred_params)
        nps, ns = self.ndim_pose, self.ndim_shape
        self.pred_params = torch.cat([s[..., 0:3] * 3.0 - 1.5,
                                      s[..., 3:5] * self.im_size,
                                      s[..., 5:6] * 0.0,
                                      s[..., 6:7] * 1e-3,
                                      s[..., nps:nps + ns] * 1e4,
                                      s[..., nps + ns:self.ndim] * 3.0 - 1.5], dim=-1)
        return self.pred_params

    def parse_pose_params(self, pose_params):
        """(B,7) -> phi, gamma, theta (B,1), t3d (B,3), f (B,1)  (network.py:253-263)."""
        return (pose_params[:, 0:1], pose_params[:, 1:2], pose_params[:, 2:3], pose_params[:, 3:6],
                pose_params[:, 6:7])

    def rotation_matrix(self, angles):
        """Host numpy rotation, float64 trig and products, one rounding to fp32 (network.py:266-291)."""
        phi, gamma, theta = angles
        R_pitch = np.array([[1, 0, 0], [0, cos(phi), sin(phi)], [0, -sin(phi), cos(phi)]])
        R_yaw = np.array([[cos(gamma), 0, -sin(gamma)], [0, 1, 0], [sin(gamma), 0, cos(gamma)]])
        R_roll = np.array([[cos(theta), sin(theta), 0], [-sin(theta), cos(theta), 0], [0, 0, 1]])
        return np.dot(np.dot(R_pitch, R_yaw), R_roll).astype(np.float32)

    def rotation_matrix_batch(self, angles_batch):
        """(network.py:292-297)"""
        angles_batch = np.asarray(angles_batch)
        R_batch = np.zeros([angles_batch.shape[0], 3, 3], dtype=np.float32)
        for i in range(angles_batch.shape[0]):
            R_batch[i] = self.rotation_matrix(angles_batch[i])
        return R_batch
