"""Loader / builder of the C-ABI shared library (include/fr_hotpath.h) and thin ctypes call helpers.

There is deliberately NO fallback: if libfr_hotpath.so cannot be built or loaded, or a tensor is not on a GPU,
the hot path raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import hashlib
import os
import shutil
import subprocess
import sys
import threading
import types

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


def sources_present():
    """False for a binary-only deployment (the .so shipped without csrc/ or include/)."""
    return all(os.path.exists(d) for d in [os.path.join(_CSRC, s) for s in SOURCES] + HEADERS)


def src_hash():
    """sha256 over the kernel sources, the headers and the compile flags (16 hex digits): the identity of a build.
    None when the sources are not in the tree (binary-only deployment: nothing to compare the library with)."""
    if not sources_present():
        return None
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for d in [os.path.join(_CSRC, s) for s in SOURCES] + HEADERS:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _built_hash():
    """The source hash the existing .so was built from: its sidecar file, or -- when the sidecar did not travel with the
    binary -- the hash embedded in the binary itself (the `src=` field of fr_version(), found by scanning the file: the
    library is not loaded to ask it)."""
    try:
        with open(LIB_PATH + ".srchash") as f:
            return f.read().strip()
    except OSError:
        pass
    try:
        with open(LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    tag = b"(gfx950) src="
    i = blob.find(tag)
    if i < 0:
        return None
    return blob[i + len(tag):i + len(tag) + 16].decode("ascii", "replace")


def is_stale():
    """True when libfr_hotpath.so is missing or was built from other sources than the ones in the tree (content hash,
    not mtimes: the .so travels to the GPU box in a snapshot whose timestamps mean nothing).  Without sources in the
    tree an existing library is taken as it is."""
    if not os.path.exists(LIB_PATH):
        return True
    want = src_hash()
    return want is not None and _built_hash() != want


def compile(force=False, verbose=False):
    """Cross-compiles the HIP kernels + C ABI for gfx950 with hipcc (works without a GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        raise RuntimeError("fr_hotpath: %s is %s and hipcc is not available to rebuild it -- there is no CPU fallback"
                           % (LIB_PATH, "stale" if os.path.exists(LIB_PATH) else "missing"))
    want = src_hash()
    if want is None:
        raise RuntimeError("fr_hotpath: %s is missing and the kernel sources are not in the tree" % LIB_PATH)
    tmp = LIB_PATH + ".tmp%d" % os.getpid()
    cmd = [hipcc] + HIPCC_FLAGS + ['-DFR_SRC_HASH="%s"' % want, "-o", tmp] + [os.path.join(_CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    with open(LIB_PATH + ".srchash.tmp%d" % os.getpid(), "w") as f:
        f.write(want + "\n")
    os.replace(tmp, LIB_PATH)
    os.replace(LIB_PATH + ".srchash.tmp%d" % os.getpid(), LIB_PATH + ".srchash")
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
    L.fr_rendering_layer_forward_phases.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                                    ctypes.c_size_t, _vp, _i]
    L.fr_rendering_layer_forward_phases.restype = _i
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
    L.fr_decode_q30_image_bytes.argtypes = [_i, _i, _i]
    L.fr_decode_q30_image_bytes.restype = ctypes.c_size_t
    L.fr_decode_q30_pack.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]
    L.fr_decode_q30_pack.restype = _i
    L.fr_decode_q30_workspace_bytes.argtypes = [_i, _i]
    L.fr_decode_q30_workspace_bytes.restype = ctypes.c_size_t
    L.fr_decode_3dmm_q30.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, ctypes.c_size_t, _vp]
    L.fr_decode_3dmm_q30.restype = _i
    L.fr_decode_3dmm_q30_lv.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _i, _vp, _vp, ctypes.c_size_t, _vp]
    L.fr_decode_3dmm_q30_lv.restype = _i
    L.fr_decode_render_forward_q30.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, _i, _vp,
                                               ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t,
                                               _vp, _i]
    L.fr_decode_render_forward_q30.restype = _i
    L.fr_decode_render_vertex_pitch.argtypes = [_i]
    L.fr_decode_render_vertex_pitch.restype = _i
    L.fr_decode_render_vertex_bytes.argtypes = [_i, _i]
    L.fr_decode_render_vertex_bytes.restype = ctypes.c_size_t
    L.fr_decode_render_forward.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.c_float, _vp,
                                           ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp, _i]
    L.fr_decode_render_forward.restype = _i
    L.fr_set_option.argtypes = [ctypes.c_char_p, _i]
    L.fr_set_option.restype = _i
    L.fr_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.fr_get_option.restype = _i
    L.fr_decode_backward_workspace_bytes.argtypes = [_i, _i, _i, _i]
    L.fr_decode_backward_workspace_bytes.restype = ctypes.c_size_t
    L.fr_decode_3dmm_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp,
                                          ctypes.c_size_t, _vp]
    L.fr_decode_3dmm_backward.restype = _i
    L.fr_decode_backward_basis_bytes.argtypes = [_i, _i, _i]
    L.fr_decode_backward_basis_bytes.restype = ctypes.c_size_t
    L.fr_decode_backward_pack_basis.argtypes = [_vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]
    L.fr_decode_backward_pack_basis.restype = _i
    L.fr_decode_3dmm_backward_packed.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp,
                                                 ctypes.c_size_t, _vp]
    L.fr_decode_3dmm_backward_packed.restype = _i
    L.fr_decode_3dmm_backward_packed_mu.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp,
                                                    ctypes.c_size_t, _vp]
    L.fr_decode_3dmm_backward_packed_mu.restype = _i
    L.fr_render_depth_strip_rows.argtypes = [_i] * 4
    L.fr_render_depth_strip_rows.restype = _i
    L.fr_debug_render_geom.argtypes = [_i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int)]
    L.fr_debug_render_geom.restype = None
    L.fr_debug_div3_sweep.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _vp]
    L.fr_debug_div3_sweep.restype = _i
    L.fr_debug_clock_probe.argtypes = [_vp, _i, _i, _vp]
    L.fr_debug_clock_probe.restype = _i
    return L


EXPORTS = ["fr_version", "fr_strerror", "fr_render_depth_workspace_bytes", "fr_render_depth_forward",
           "fr_render_depth_backward", "fr_decode_packed_basis_bytes", "fr_decode_pack_basis", "fr_decode_3dmm",
           "fr_decode_backward_workspace_bytes", "fr_decode_3dmm_backward", "fr_rendering_layer_forward",
           "fr_render_depth_forward_phases", "fr_debug_render_geom", "fr_debug_div3_sweep",
           "fr_render_depth_backward_workspace_bytes", "fr_render_depth_backward_ws", "fr_set_option", "fr_get_option",
           "fr_decode_q30_image_bytes", "fr_decode_q30_pack", "fr_decode_q30_workspace_bytes", "fr_decode_3dmm_q30",
           "fr_decode_3dmm_q30_lv", "fr_decode_render_forward_q30",
           "fr_decode_render_vertex_pitch", "fr_decode_render_vertex_bytes", "fr_decode_render_forward",
           "fr_decode_backward_basis_bytes", "fr_decode_backward_pack_basis", "fr_decode_3dmm_backward_packed",
           "fr_debug_clock_probe", "fr_rendering_layer_forward_phases", "fr_decode_3dmm_backward_packed_mu",
           "fr_render_depth_strip_rows"]


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
                if want is not None and want.encode() not in L.fr_version():
                    raise RuntimeError("fr_hotpath: %s reports %r but the sources hash to %s: stale binary refused"
                                       % (LIB_PATH, L.fr_version(), want))
                _lib = L
    return _lib


# ---- host-side choices --------------------------------------------------------------------------------------------
DECODE_ARITH_Q30, DECODE_ARITH_F32 = 0, 1
# This file is loaded under two module names (the package's `3dfacerecon_amd._lib`, and by path from the reference-style
# flat modules rendering_layer/ops.py, nets/network.py, pipeline.py): process-wide choices live in ONE shared namespace.
_STATE = sys.modules.setdefault("_fr_hotpath_state", types.SimpleNamespace(decode_arith=None, q30_levels=7, option_epoch=0))
_ARITH_ENV = {"q30": 7, "q30l7": 7, "q30l5": 5, "q30l4": 4}


def decode_arith():
    """Which written definition of the basis blend the Python callers (FaceRecNet.vertices_transform, DecodeRenderPlan)
    use: DECODE_ARITH_F32 (default: fr_decode_3dmm, the k-ordered fmaf chain) or DECODE_ARITH_Q30 (fr_decode_3dmm_q30_lv,
    the fixed-point blend on the int8 matrix cores with q30_levels() digit-product levels; its image and workspace are built
    on first use).  Process-wide; the initial value comes from the environment variable FR_DECODE_ARITH = "f32" | "q30"
    (all seven levels) | "q30l5" | "q30l4", read once."""
    if _STATE.decode_arith is None:
        lv = _ARITH_ENV.get(os.environ.get("FR_DECODE_ARITH", ""))
        _STATE.decode_arith = DECODE_ARITH_Q30 if lv else DECODE_ARITH_F32
        if lv:
            _STATE.q30_levels = lv
    return _STATE.decode_arith


def q30_levels():
    """Digit-product levels of the Q30 blend (include/fr_hotpath.h): 7 = the exact product, 5, or 4."""
    decode_arith()
    return _STATE.q30_levels


def set_decode_arith(mode, levels=None):
    if mode not in (DECODE_ARITH_Q30, DECODE_ARITH_F32):
        raise ValueError("decode arithmetic must be DECODE_ARITH_F32 or DECODE_ARITH_Q30")
    if levels is not None and int(levels) not in (4, 5, 7):
        raise ValueError("Q30 levels must be 7, 5 or 4")
    decode_arith()
    _STATE.decode_arith = mode
    if levels is not None:
        _STATE.q30_levels = int(levels)


def set_option(name, value):
    """fr_set_option: a launcher knob by its FR_* name (the environment is only read once, at the first launch).  Every
    change bumps option_epoch(): host-side caches of what a launch left in a workspace (rendering_layer/ops.py: the packed
    triangle table) are keyed on it, since some knobs decide whether and how that state is written."""
    check(lib().fr_set_option(name.encode(), int(value)), "fr_set_option(%s)" % name)
    _STATE.option_epoch += 1


def option_epoch():
    return _STATE.option_epoch


def get_option(name):
    v = ctypes.c_int(0)
    check(lib().fr_get_option(name.encode(), ctypes.byref(v)), "fr_get_option(%s)" % name)
    return v.value


class options:
    """Context manager for A/B runs and tests: `with options(FR_EMIT_FILTER=0): ...` sets the knobs and restores them."""

    def __init__(self, **kv):
        self.kv = kv
        self.old = {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


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
