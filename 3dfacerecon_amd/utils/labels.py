"""235-d parameter label files, as written by the reference's dataset generator (prepare_data/
script_generate_dataset.m:115-125: one '%.6f' value per line) and read back with np.loadtxt
(utils/data_process.py:53).  Layout: [phi, gamma, theta, tx, ty, tz, f | shape x199 | exp x29]."""
import numpy as np


def read_label_file(path, ndim=None):
    v = np.loadtxt(path, dtype=np.float32).reshape(-1)
    if ndim is not None and v.shape[0] != ndim:
        raise ValueError("%s holds %d values, expected %d" % (path, v.shape[0], ndim))
    return v


def write_label_file(path, params):
    v = np.asarray(params, np.float64).reshape(-1)
    with open(path, "w") as f:
        for x in v:
            f.write("%.6f\n" % x)


def read_label_batch(paths, ndim=235):
    """[len(paths), ndim] float32 -- the `params_label` batch of the reference's generators (data_process.py:63-101)."""
    return np.stack([read_label_file(p, ndim) for p in paths]) if len(paths) else np.zeros((0, ndim), np.float32)
