"""helpers for the -m gpu parity tests: everything goes through the operator surface -> C ABI -> HIP."""
import numpy as np
import torch

from conftest import pkg


def ops():
    return pkg("rendering_layer.ops")


def net_mod():
    return pkg("nets.network")


def render_gpu(ver, tri, tex, H, W, B=None):
    o = ops()
    dev = torch.device("cuda:0")
    ver_t = torch.as_tensor(np.ascontiguousarray(ver, np.float32), device=dev)
    tri_t = torch.as_tensor(np.ascontiguousarray(tri, np.float32), device=dev)
    tex_t = torch.as_tensor(np.ascontiguousarray(tex, np.float32), device=dev)
    B = ver_t.shape[0] if B is None else B
    image = torch.zeros((B, H, W, 3), device=dev)
    outs = o.render_depth(ver_t, tri_t, tex_t, image)
    torch.cuda.synchronize()
    return tuple(t.cpu().numpy() for t in outs)


def assert_render_equal(got, want, what=""):
    names = ("depth", "texture_image", "normal", "tri_ind")
    for g, w, n in zip(got, want, names):
        assert g.shape == w.shape, (what, n, g.shape, w.shape)
        # bit-exact, except that a NaN equals a NaN whatever its sign / payload (an Inf - Inf in the normal of a triangle
        # with an infinite z is 0xFFC00000 on x86 and 0x7FC00000 on gfx950: the IEEE default NaN differs by platform)
        if not np.array_equal(g, w, equal_nan=True):
            bad = np.argwhere((g != w) & ~(np.isnan(g) & np.isnan(w)))
            raise AssertionError("%s %s: %d mismatches, first at %s: got %r want %r" %
                                 (what, n, len(bad), bad[0], g[tuple(bad[0])], w[tuple(bad[0])]))


def render_pipelined_gpu(ver_a, ver_b, tri, tex, H, W):
    """Two batches of the same shape through fr_decode_render_pipelined's render phases (no decode: the vertex hand-off
    buffers are filled here): emit(a); [emit(b) || resolve(a)] as ONE launch; resolve(b).  -> (planes of a, planes of b),
    or None when the entry point does not serve the shape."""
    import ctypes
    h = pkg("_lib")
    L = h.lib()
    dev = torch.device("cuda:0")
    B, _, nver = ver_a.shape
    ntri = tri.shape[1]
    if not L.fr_decode_render_pipelined_supported(B, nver, ntri, H, W):
        return None
    pitch = L.fr_decode_render_vertex_pitch(nver)
    vbuf = []
    for v in (ver_a, ver_b):
        t = torch.full((B, 3, pitch), float("nan"), dtype=torch.float32, device=dev)
        t[:, :, :nver] = torch.as_tensor(np.ascontiguousarray(v, np.float32), device=dev)
        vbuf.append(t)
    tri_t = torch.as_tensor(np.ascontiguousarray(tri, np.float32), device=dev)
    tex_t = torch.as_tensor(np.ascontiguousarray(tex, np.float32), device=dev)
    tex_batch = 1 if tex_t.dim() == 2 else int(tex_t.shape[0])
    nws = L.fr_render_depth_workspace_bytes(B, nver, ntri, H, W)
    ws = [torch.empty((max(nws, 16),), dtype=torch.uint8, device=dev) for _ in range(2)]
    planes = [torch.full((B, H, W, c), 7.0, dtype=torch.float32, device=dev) for c in (1, 3, 3, 1)]
    p = h.ptr

    def run(phases, new, prev):
        rc = L.fr_decode_render_pipelined(None, None, None, p(tri_t), p(tex_t), B, nver, 0, 0, ntri, H, W, tex_batch,
                                          ctypes.c_float(float(H)), p(vbuf[new]), p(vbuf[prev]), vbuf[0].numel() * 4,
                                          p(planes[0]), p(planes[1]), p(planes[2]), p(planes[3]), p(ws[new]), p(ws[prev]),
                                          nws, h.stream_ptr(dev), phases)
        h.check(rc, "fr_decode_render_pipelined")

    run(4, 0, 1)
    run(4, 1, 0)
    run(1, 0, 1)          # emit(a)
    run(3, 1, 0)          # emit(b) || resolve(a)
    torch.cuda.synchronize()
    got_a = tuple(t.cpu().numpy() for t in planes)
    for t in planes:
        t.fill_(7.0)
    run(2, 0, 1)          # resolve(b)
    torch.cuda.synchronize()
    got_b = tuple(t.cpu().numpy() for t in planes)
    return got_a, got_b
