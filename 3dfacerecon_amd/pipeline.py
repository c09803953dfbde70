"""Launch plan for the CoarseNet -> render loop's hot path: 235-d parameters -> decode -> render_depth.

`FaceRecNet.vertices_transform` + `rendering_layer.ops.render_depth` allocate their outputs per call, like the
reference op does (render_depth_op.cc:442-445).  A serving / training loop that runs the same shapes every
iteration should not pay that: a `DecodeRenderPlan` owns the vertex hand-off buffer and the four output planes once
(HBM is 288 GB; one 64-face plan is 123 MB), keeps the ctypes argument list prebuilt, and runs the fused
decode -> render entry point (fr_decode_render_forward: ONE C call, three kernel launches on torch's current HIP
stream); it can be captured into a hipGraph (`capture()` / `replay()`).  The projected vertices are handed from the
decode to the rasteriser in the library's pitched row layout (rows padded to 128-byte multiples, so every decode
store is an aligned half line); `plan.vertex_proj` is the strided [B,3,N] VIEW of that buffer -- same values, same
indexing, not contiguous.  Outputs are views of the plan's buffers: they are overwritten by the next `step()`.
Forward only (no autograd); use rendering_layer.ops.render_depth when gradients are needed.
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
    def __init__(self, net, batch, height=None, width=None, texture=None, stream=None, strip_rows=0):
        """net: nets.network.FaceRecNet (holds the packed basis, tri, vertex_code); texture: (3,N) or (B,3,N)
        tensor, default net.vertex_code (the PNCC colour code, reference network.py:116).  stream: a torch.cuda.Stream
        every launch of this plan goes to (default: whatever torch's current stream is at the call).  strip_rows: rows per
        screen strip of the resolver (FR_PHASES_STRIP_ROWS in include/fr_hotpath.h; 0 = the library's choice)."""
        h = _host()
        self.stream = stream
        self.strip_rows = int(strip_rows) & 0xFF
        self._h = h
        self._L = L = h.lib()
        self.net = net
        self.device = net.device
        self.B = int(batch)
        self.H = int(net.im_size if height is None else height)
        self.W = int(net.im_size if width is None else width)
        self.N = net.nvert
        self.T = int(net.tri.shape[1])
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros((self.B, net.ndim), **f32)
        self.depth = torch.empty((self.B, self.H, self.W, 1), **f32)
        self.texture_image = torch.empty((self.B, self.H, self.W, 3), **f32)
        self.normal = torch.empty((self.B, self.H, self.W, 3), **f32)
        self.tri_ind = torch.empty((self.B, self.H, self.W, 1), **f32)
        tex = net.vertex_code if texture is None else h.require_gpu_f32(texture, "texture")
        self.texture = tex.contiguous()
        self.tex_batch = 1 if self.texture.dim() == 2 else int(self.texture.shape[0])
        if self.tex_batch not in (1, self.B) or self.texture.shape[-2] != 3 or self.texture.shape[-1] != self.N:
            raise ValueError("texture must be (3,N), (1,3,N) or (B,3,N)")
        ws_bytes = L.fr_render_depth_workspace_bytes(self.B, self.N, self.T, self.H, self.W)
        self._ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=self.device)
        self._ws_bytes = ws_bytes
        p = h.ptr
        # The decode arithmetic is fixed when the plan is built: the f32 chain (fr_decode_render_forward), or -- when the Q30
        # arithmetic is selected at that moment -- fr_decode_render_forward_q30 with the level count of that moment and a
        # staging workspace the PLAN owns.  Same phases, same pitched hand-off.  Nothing is allocated at launch time either
        # way, so the plan can be captured on any stream.  `plan.q30` = 0 (f32 chain) or the Q30 level count (7 / 5 / 4).
        basis = net._basis
        self.q30 = int(h.q30_levels()) if basis.use_q30() else 0
        self.pitch = int(L.fr_decode_render_vertex_pitch(self.N))
        self._vertex = torch.empty((self.B, 3, self.pitch), **f32)     # (torch allocations are >= 256-byte aligned)
        self._vertex_bytes = self._vertex.numel() * 4
        self.vertex_proj = self._vertex[:, :, :self.N]                   # [B,3,N]; a strided view when pitch > N
        if self.q30:
            self._q_ws = torch.empty((basis.q30_ws_bytes,), dtype=torch.uint8, device=self.device)
            self._fused_args = (p(self.params), p(basis.qimage()), None, p(net.tri), p(self.texture), self.B, self.N,
                                net.ndim_shape, net.ndim_exp, self.T, self.H, self.W, self.tex_batch,
                                ctypes.c_float(float(net.im_size)), self.q30, p(self._vertex), self._vertex_bytes, p(self.depth),
                                p(self.texture_image), p(self.normal), p(self.tri_ind), p(self._ws), ws_bytes,
                                p(self._q_ws), basis.q30_ws_bytes)
            self._fused_fn, self._fused_name = L.fr_decode_render_forward_q30, "fr_decode_render_forward_q30"
        else:
            self._fused_args = (p(self.params), p(basis.image), None, p(net.tri), p(self.texture), self.B, self.N,
                                net.ndim_shape, net.ndim_exp, self.T, self.H, self.W, self.tex_batch,
                                ctypes.c_float(float(net.im_size)), p(self._vertex), self._vertex_bytes, p(self.depth),
                                p(self.texture_image), p(self.normal), p(self.tri_ind), p(self._ws), ws_bytes)
            self._fused_fn, self._fused_name = L.fr_decode_render_forward, "fr_decode_render_forward"
        self._graph = None
        # the triangle list is a constant of the model (reference network.py:178): convert + range-check it ONCE into the
        # workspace's table; every step then runs the decode, emit and resolve phases only
        self._tri_packed = False
        self.pack_tri()

    def _stream(self):
        st = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(st.cuda_stream)

    def _run(self, phases):
        """Phase bits of fr_decode_render_forward: 8 = decode, 4 = pack the triangle list, 1 = emit, 2 = resolve."""
        rc = self._fused_fn(*self._fused_args, self._stream(), phases | (self.strip_rows << 8))
        if rc:
            self._h.check(rc, self._fused_name)

    def pack_tri(self):
        """(Re)builds the pre-validated triangle table in the workspace; call again after changing net.tri in place."""
        with torch.cuda.device(self.device):
            self._run(4)
        self._tri_packed = True

    # -- eager launches on the current stream ---------------------------------------------------------------
    def decode(self):
        self._run(8)

    def render(self):
        self._run(3)

    def render_phase(self, phases):
        """phases = 1 launches only raster_emit_kernel, 2 only resolve_write_kernel, 4 only pack_tri_kernel."""
        self._run(int(phases) & 7)

    def outputs(self):
        return self.depth, self.texture_image, self.normal, self.tri_ind

    def _take_params(self, params):
        """Copies `params` (B,d) into the plan's buffer in the order the launches need.  An unbound plan copies on torch's
        current stream, where its launches go too.  A plan BOUND to a stream launches there, so the copy must run there as
        well, behind whatever produced `params` on the current stream: the bound stream waits for the current one, the copy is
        enqueued on the bound stream, and `params` is marked as used by it (the caller may drop it at once)."""
        if params is None:
            return
        src = params.reshape(self.B, -1)
        if self.stream is None:
            self.params.copy_(src, non_blocking=True)
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.params.copy_(src, non_blocking=True)
        if src.is_cuda:
            src.record_stream(self.stream)

    def step(self, params=None):
        """decode + render of one batch (one C call).  `params` (B,d) is copied into the plan's buffer when given (on the
        plan's own stream when it is bound to one: see _take_params).  A bound plan's outputs are complete when ITS stream
        has run them (`plan.stream.synchronize()`, or `torch.cuda.current_stream().wait_stream(plan.stream)`)."""
        self._take_params(params)
        self._run(11)
        return self.outputs()

    # -- hipGraph ------------------------------------------------------------------------------------------------
    def capture(self):
        """Captures decode + render (reading self.params, writing the plan's outputs) into a hipGraph."""
        if self.stream is not None:
            raise RuntimeError("a plan bound to a stream launches eagerly on it; build an unbound plan to capture a hipGraph")
        with torch.cuda.device(self.device):
            self.step()  # warm-up launch outside the capture (function attributes, lazy module load)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._run(11)
        self._graph = g
        return g

    def replay(self, params=None):
        if self._graph is None:
            self.capture()
        self._take_params(params)   # (capture() refuses bound plans, so this is the current-stream copy)
        self._graph.replay()
        return self.outputs()

    def __del__(self):   # a bound plan's buffers go back to torch's allocator: not while its stream still runs on them
        try:
            if self.stream is not None:
                self.stream.synchronize()
        except Exception:
            pass


class GraphedSteps:
    """R consecutive batches captured into ONE hipGraph: R DecodeRenderPlans (each its own parameters, vertex buffer, workspace
    and output planes) whose 3 R launches are replayed with a single hipGraphLaunch.  Why: a replay is followed by a ~9 us
    bubble before the next replay's first kernel starts (hipGraphLaunch is not pipelined with the previous graph's tail on this
    runtime: profiles/round4_probes/r4h), which makes a one-step graph slower than three eager launches; with R steps per graph
    the bubble is paid once per R batches.  `replay(params_list)` copies up to R parameter sets and replays; `plans[i].outputs()`
    are batch i's planes."""

    def __init__(self, net, batch, steps, height=None, width=None, texture=None):
        if int(steps) < 1:
            raise ValueError("steps must be >= 1")
        self.device = net.device
        self.plans = [DecodeRenderPlan(net, batch, height, width, texture) for _ in range(int(steps))]
        self._graph = None

    def capture(self):
        with torch.cuda.device(self.device):
            for p in self.plans:
                p.step()     # warm-up launches outside the capture
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for p in self.plans:
                    p._run(11)
        self._graph = g
        return g

    def replay(self, params_list=None):
        if self._graph is None:
            self.capture()
        if params_list is not None:
            for p, prm in zip(self.plans, params_list):
                p._take_params(prm)
        self._graph.replay()
        return [p.outputs() for p in self.plans]


class BatchesInFlight:
    """S (default two) independent batches in flight: S DecodeRenderPlans -- each with its own parameters, vertex buffer,
    workspace and output planes -- each bound to its own HIP stream, with NO edge between the streams.  `submit()` sends the
    next batch down the next slot's stream (round robin) and returns that slot.  Consecutive batches of a serving loop are
    independent (render_depth_op.cc:180 loops over them one after the other only because it is a serial program), so the
    hardware may run the decode of one beside the render of the other: one batch's kernels fill the tails, launch gaps and
    idle units of the other's.  Measured on an MI355X at 64 faces: 100-103 us per batch against 110 us for one plan stepping
    on one stream (tools/multistream_probe.py; three in flight: slower again -- the batches' working sets evict each other
    from the 256 MiB Infinity Cache).  The first slot's stream has high priority and the others normal priority: at equal priority the
    streams sometimes lock into step (both decodes, then both emits ... -- 100 to 107 us from one process to the next); with
    one batch entitled to run ahead and the other filling in, 100.5-101.6 us (profiles/round4_probes/r4p).

    Rules of use: a slot's outputs are complete when ITS stream has run dry (`slot.wait()`, or
    `slot.make_current_stream_wait()` to order a consumer on torch's current stream behind it) and are overwritten when the
    slot comes round again (S submits later).  `submit(params)` copies `params` into the slot's buffer on the slot's stream
    after making that stream wait for torch's current stream (the producer of `params`)."""

    def __init__(self, net, batch, height=None, width=None, texture=None, slots=2, strip_rows=None):
        if int(slots) < 1:
            raise ValueError("slots must be >= 1")
        self.device = net.device
        if strip_rows is None:
            strip_rows = self.strip_rows_in_flight(net, batch, height, width) if int(slots) > 1 else 0
        self.strip_rows = int(strip_rows)
        with torch.cuda.device(self.device):
            self.slots = []
            for i in range(int(slots)):
                # torch stream priorities on ROCm: negative = high, and anything >= 0 is clamped to 0 = normal (there is no "low"
                # through this API): slot 0 runs at HIGH priority, the others at NORMAL -- one batch entitled to run ahead, the
                # other filling in (profiles/round4_probes/r4p: equal priorities sometimes lock the streams into step)
                st = torch.cuda.Stream(device=self.device, priority=(-1 if i == 0 else 0))
                self.slots.append(_Slot(net, batch, height, width, texture, stream=st, strip_rows=self.strip_rows))
            torch.cuda.synchronize(self.device)   # every slot's triangle table is packed before anything else touches the slots
        self._next = 0
        self.B = self.slots[0].B

    @staticmethod
    def strip_rows_in_flight(net, batch, height=None, width=None):
        """The resolver's strip height for plans that run BESIDE another batch: four fifths of the library's own choice in the regime
        it was measured in -- strips of ten rows or more and at least four resolver workgroups per CU (200 x 200 at 64 faces: 8
        rows instead of 10, 1,600 workgroups instead of 1,280, which fill the other batch's gaps better: -0.8 us per batch with two in
        flight, +1.3 us one batch at a time; profiles/round4_probes/r4q, round6_probes/r6f) -- otherwise 0 = no hint.  A scheduling
        hint: no result bit depends on it."""
        h = _host()
        H = int(net.im_size if height is None else height)
        W = int(net.im_size if width is None else width)
        rows = int(h.lib().fr_render_depth_strip_rows(int(batch), int(net.tri.shape[1]), H, W))
        if rows < 10 or int(batch) * ((H + rows - 1) // rows) < 1024:
            return 0
        return (4 * rows) // 5

    def submit(self, params=None, marks=None):
        """Launches decode + render of one batch on the next slot's stream; returns the slot (a DecodeRenderPlan).
        marks: four torch.cuda.Events to record on the slot's stream before the decode, between the three launches and
        after the resolve (bench.py's per-kernel timing; the step then goes out as three C calls instead of one)."""
        sl = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        sl._take_params(params)   # (on the slot's stream, behind torch's current stream: DecodeRenderPlan._take_params)
        if marks is None:
            sl._run(11)
        else:
            marks[0].record(sl.stream)
            sl._run(8)
            marks[1].record(sl.stream)
            sl._run(1)
            marks[2].record(sl.stream)
            sl._run(2)
            marks[3].record(sl.stream)
        return sl

    def synchronize(self):
        for sl in self.slots:
            sl.stream.synchronize()

    def __del__(self):   # the slots' buffers go back to torch's allocator: not while their streams still run
        try:
            self.synchronize()
        except Exception:
            pass


class _Slot(DecodeRenderPlan):
    def wait(self):
        self.stream.synchronize()
        return self.outputs()

    def make_current_stream_wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
