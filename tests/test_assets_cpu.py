"""CPU: asset I/O with the reference's contracts (utils/parser_3dmm.py dict keys, 235-line label files)."""
import numpy as np

from conftest import pkg


def test_parser_3dmm_roundtrip(tmp_path, small_assets):
    parser = pkg("utils.parser_3dmm")
    A = dict(small_assets)
    parser.write_3dmm_model(str(tmp_path), A, tri_base=1)          # MATLAB-style 1-based triangle list
    M = parser.read_3dmm_model(str(tmp_path))                       # reference behaviour: passed through unshifted
    assert sorted(M.keys()) == sorted(['vertex', 'tri', 'mu', 'mu_tex', 'pc_tex', 'param_tex', 'pc_shape', 'pc_exp',
                                       'ndim_shape', 'ndim_exp', 'ndim_pose'])
    assert M['ndim_pose'] == 7 and M['ndim_shape'] == A['ndim_shape'] and M['ndim_exp'] == A['ndim_exp']
    np.testing.assert_array_equal(M['tri'], A['tri'].astype(np.float64) + 1)
    M0 = parser.read_3dmm_model(str(tmp_path), tri_base=1)
    np.testing.assert_array_equal(M0['tri'], A['tri'].astype(np.float64))
    for k in ('mu', 'pc_shape', 'pc_exp', 'vertex', 'mu_tex'):
        np.testing.assert_array_equal(np.asarray(M0[k], np.float32).reshape(np.shape(A[k])), A[k])
    assert M0['mu'].shape == (A['mu'].shape[0], 1)                  # mu = mu_shape + mu_exp, (3N, 1)


def test_label_files_roundtrip(tmp_path, synth):
    labels = pkg("utils.labels")
    P = synth.sample_params_batch(3, beta=0.7, seed=1)
    paths = []
    for i in range(3):
        p = str(tmp_path / ("%d.txt" % i))
        labels.write_label_file(p, P[i])
        paths.append(p)
    assert len(open(paths[0]).read().split()) == 235
    L = labels.read_label_batch(paths)
    assert L.shape == (3, 235) and L.dtype == np.float32
    np.testing.assert_allclose(L, P, atol=5e-7 + 1e-6 * 0, rtol=0)   # '%.6f' text precision
    assert np.abs(L - P).max() <= 5.1e-7 + np.abs(P).max() * 6e-8
