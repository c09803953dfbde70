import importlib
import os
import sys

import numpy as np
import pytest

# (nets/coarse_net.py, apply_miopen_workaround: a MIOpen solver that faults on gfx950 while convolutions are being benchmarked.  A
# process setting, so it is the ENTRY POINT's to make -- for the test session that is here, before anything runs a convolution)
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name):
    """import a module of the digit-named package: pkg('rendering_layer.ops')"""
    return importlib.import_module("3dfacerecon_amd." + name)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def synth():
    return pkg("utils.synth")


@pytest.fixture(scope="session")
def small_assets(synth):
    return synth.make_small_assets()


@pytest.fixture(scope="session")
def full_assets(synth):
    return synth.make_assets()


def kat_inputs(case, W, H):
    """numpy inputs (vertex [1,3,n], tri [3,t], texture [1,3,n]) of a SURVEY 8(a) known-answer case."""
    v = np.array(case["vertices"], np.float32)
    n = v.shape[0]
    ver = np.zeros((1, 3, n), np.float32)
    ver[0, 0] = v[:, 0]
    ver[0, 1] = v[:, 1]
    ver[0, 2] = np.array(case.get("z", [5] * n), np.float32)
    tex = np.zeros((1, 3, n), np.float32)
    for p in range(n):
        tex[0, :, p] = np.array([0.1, 0.2, 0.3], np.float32) * np.float32(p + 1)
    tri = np.array(case["tris"], np.float32).T.copy()
    return ver, tri, tex


def grid_from_rows(rows):
    return np.array([[-1 if ch == "." else int(ch) for ch in r] for r in rows], np.float32)
