"""Loader / builder of the C-ABI shared library (include/fr_hotpath.h) and thin ctypes call helpers.

There is deliberately NO fallback: if libfr_hotpath.so cannot be built or loaded, or a tensor is not on a GPU,
the hot path raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import hashlib
import os
import shutil
import subprocess
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_PATH = os.path.join(_PKG_DIR, "libfr_hotpath.so")
SOURCES = ["fr_capi.hip", "fr_render.hip", "fr_decode.hip", "fr_decode_q.hip", "fr_decode_bwd.hip"]
HEADERS = [os.path.join(_CSRC, "fr_common.h"), os.path.join(_CSRC, "fr_decode_shared.h"), os.path.join(_PKG_DIR, "..", "include", "fr_hotpath.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]  # single-lane LDS atomics stay single instructions

FR_OK = 0
_ERR_NAMES = {-1: "invalid argument", -2: "workspace / packed buffer too small", -3: "HIP launch or runtime error",
              -4: "size not supported by the gfx950 kernels"}

_lock = threading.Lock()
_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def src_hash():
    """sha256 over the kernel sources, the headers and the compile flags (16 hex digits): the identity of a build."""
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for d in [os.path.join(_CSRC, s) for s in SOURCES] + HEADERS:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _built_hash():
    try:
        with open(LIB_PATH + ".srchash") as f:
            return f.read().strip()
    except OSError:
        return None


def is_stale():
    """True when libfr_hotpath.so is missing or was built from other sources than the ones in the tree (content hash,
    not mtimes: the .so travels to the GPU box in a snapshot whose timestamps mean nothing)."""
    return not os.path.exists(LIB_PATH) or _built_hash() != src_hash()


def compile(force=False, verbose=False):
    """Cross-compiles the HIP kernels + C ABI for gfx950 with hipcc (works without a GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        raise RuntimeError("fr_hotpath: %s is %s and hipcc is not available to rebuild it -- there is no CPU fallback"
                           % (LIB_PATH, "stale" if os.path.exists(LIB_PATH) else "missing"))
    want = src_hash()
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    cmd = [hipcc] + HIPCC_FLAGS + ['-DFR_SRC_HASH="%s"' % want, "-o", tmp] + [os.path.join(_CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    os.replace(tmp, LIB_PATH)
    with open(LIB_PATH + ".srchash", "w") as f:
        f.write(want + "\n")
    return LIB_PATH


def _bind(L):
    L.fr_version.restype = ctypes.c_char_p
    L.fr_strerror.argtypes = [_i]
    L.fr_strerror.restype = ctypes.c_char_p
    L.fr_render_depth_workspace_bytes.argtypes = [_i] * 5
    L.fr_render_depth_workspace_bytes.restype = ctypes.c_size_t
    L.fr_render_depth_forward.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                          ctypes.c_size_t, _vp]
    L.fr_render_depth_forward.restype = _i
    L.fr_render_depth_forward_phases.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                                 ctypes.c_size_t, _vp, _i]
    L.fr_render_depth_forward_phases.restype = _i
    L.fr_rendering_layer_forward.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                             ctypes.c_size_t, _vp]
    L.fr_rendering_layer_forward.restype = _i
    L.fr_render_depth_backward.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]
    L.fr_render_depth_backward.restype = _i
    L.fr_render_depth_backward_workspace_bytes.argtypes = [_i, _i, _i]
    L.fr_render_depth_backward_workspace_bytes.restype = ctypes.c_size_t
    L.fr_render_depth_backward_ws.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp]
    L.fr_render_depth_backward_ws.restype = _i
    L.fr_decode_packed_basis_bytes.argtypes = [_i, _i, _i]
    L.fr_decode_packed_basis_bytes.restype = ctypes.c_size_t
    L.fr_decode_pack_basis.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]
    L.fr_decode_pack_basis.restype = _i
    L.fr_decode_3dmm.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp]
    L.fr_decode_3dmm.restype = _i
    L.fr_decode_set_arith.argtypes = [_i]
    L.fr_decode_set_arith.restype = _i
    L.fr_decode_get_arith.argtypes = []
    L.fr_decode_get_arith.restype = _i
    L.fr_decode_backward_workspace_bytes.argtypes = [_i, _i, _i, _i]
    L.fr_decode_backward_workspace_bytes.restype = ctypes.c_size_t
    L.fr_decode_3dmm_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp,
                                          ctypes.c_size_t, _vp]
    L.fr_decode_3dmm_backward.restype = _i
    L.fr_debug_render_geom.argtypes = [_i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int)]
    L.fr_debug_render_geom.restype = None
    L.fr_debug_div3_sweep.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _vp]
    L.fr_debug_div3_sweep.restype = _i
    return L


EXPORTS = ["fr_version", "fr_strerror", "fr_render_depth_workspace_bytes", "fr_render_depth_forward",
           "fr_render_depth_backward", "fr_decode_packed_basis_bytes", "fr_decode_pack_basis", "fr_decode_3dmm",
           "fr_decode_backward_workspace_bytes", "fr_decode_3dmm_backward", "fr_rendering_layer_forward",
           "fr_render_depth_forward_phases", "fr_debug_render_geom", "fr_debug_div3_sweep",
           "fr_render_depth_backward_workspace_bytes", "fr_render_depth_backward_ws", "fr_decode_set_arith",
           "fr_decode_get_arith"]


def lib():
    """Returns the bound ctypes library.  The .so is (re)built first when it is missing or stale -- built from other
    sources than the tree holds -- (the reference's build-on-import fallback, rendering_layer/ops.py:63-72); a stale
    binary that cannot be rebuilt is refused, never loaded.  Raises if it can be neither built nor loaded."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                import torch  # noqa: F401  (loads the ROCm runtime this library binds to, by SONAME)
                if is_stale():
                    compile()
                try:
                    L = _bind(ctypes.CDLL(LIB_PATH))
                except OSError as e:
                    raise RuntimeError("fr_hotpath: cannot load %s (%s); run compile() -- there is no CPU fallback"
                                       % (LIB_PATH, e))
                want = src_hash()
                if want.encode() not in L.fr_version():
                    raise RuntimeError("fr_hotpath: %s reports %r but the sources hash to %s: stale binary refused"
                                       % (LIB_PATH, L.fr_version(), want))
                _lib = L
    return _lib


def check(rc, what):
    if rc != FR_OK:
        msg = _ERR_NAMES.get(rc, "unknown error %d" % rc)
        if rc == -1:
            raise ValueError("%s: %s" % (what, msg))
        raise RuntimeError("%s: %s" % (what, msg))


def require_gpu_f32(t, name):
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the fr_hotpath kernels run on an MI355X only (no CPU fallback)"
                           % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t.contiguous()


def stream_ptr(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None
