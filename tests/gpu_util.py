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
