"""3DMM asset loader with the reference's dict contract (utils/parser_3dmm.py:6-61).

`read_3dmm_model(model_path)` reads Model_Shape.mat {mu_shape, w, tri, tex, w_tex, alpha_tex}, Model_Expression.mat
{mu_exp, w_exp} and vertex_code.mat {vertex_code} (produced by prepare_data/script_ModelGenerate.m:1-16) and returns
    {'vertex', 'tri', 'mu' (= mu_shape + mu_exp), 'mu_tex', 'pc_tex', 'param_tex', 'pc_shape', 'pc_exp',
     'ndim_shape', 'ndim_exp', 'ndim_pose' (= 7)}
-- the `mesh_data` argument of nets.network.FaceRecNet.  The licensed BFM files are not shipped (reference
3dmm/.gitignore:1-4); `write_3dmm_model` stores any such dict (e.g. utils.synth.make_assets()) in the same three files.

`tri_base`: MATLAB triangle lists are 1-based and the reference hands them to the op unshifted (network.py:178), so the
last vertex id is one past the end.  The default (None) keeps the file's values, like the reference; pass tri_base=1
to subtract 1 when loading real BFM data (this implementation skips out-of-range triangles instead of reading out of
bounds, so the two choices render differently exactly where the reference would be reading garbage).
"""
import os

import numpy as np
import scipy.io as sio


def parse_3dmm_files(model_shapefile, model_expfile, vertex_codefile):
    '''Parse params and data from 3DMM model files (reference parser_3dmm.py:6-33).'''
    assert os.path.exists(model_shapefile), 'File %s does not exist!' % (model_shapefile)
    data = sio.loadmat(model_shapefile)
    mu_shape, pc_shape, tri = data['mu_shape'], data['w'], data['tri']
    mu_tex, pc_tex, param_tex = data['tex'], data['w_tex'], data['alpha_tex']

    assert os.path.exists(model_expfile), 'File %s does not exist!' % (model_expfile)
    data = sio.loadmat(model_expfile)
    mu_exp, pc_exp = data['mu_exp'], data['w_exp']

    assert os.path.exists(vertex_codefile), 'File %s does not exist!' % (vertex_codefile)
    vertex_code = sio.loadmat(vertex_codefile)['vertex_code']

    mu = mu_shape + mu_exp
    return vertex_code, tri, mu, pc_shape, pc_exp, mu_tex, pc_tex, param_tex


def read_3dmm_model(model_path, tri_base=None):
    ''' Parse the 3dmm parameters data (reference parser_3dmm.py:36-61).  :return: model_params '''
    vertex_code, tri, mu, pc_shape, pc_exp, mu_tex, pc_tex, param_tex = parse_3dmm_files(
        os.path.join(model_path, 'Model_Shape.mat'), os.path.join(model_path, 'Model_Expression.mat'),
        os.path.join(model_path, 'vertex_code.mat'))
    if tri_base:
        tri = np.asarray(tri, np.float64) - float(tri_base)
    return {'vertex': vertex_code,
            'tri': tri,
            'mu': mu,
            'mu_tex': mu_tex,
            'pc_tex': pc_tex,
            'param_tex': param_tex,
            'pc_shape': pc_shape,
            'pc_exp': pc_exp,
            'ndim_shape': np.shape(pc_shape)[1],
            'ndim_exp': np.shape(pc_exp)[1],
            'ndim_pose': 7}


def write_3dmm_model(model_path, model_params, tri_base=0):
    """Stores a model dict as the three .mat files read_3dmm_model expects.  `tri_base` is added to the triangle ids
    (use 1 to write MATLAB-style lists).  mu is stored as mu_shape with a zero mu_exp."""
    os.makedirs(model_path, exist_ok=True)
    mu = np.asarray(model_params['mu'], np.float64).reshape(-1, 1)
    n3 = mu.shape[0]
    pc_tex = np.asarray(model_params.get('pc_tex', np.zeros((n3, 1))), np.float64)
    param_tex = np.asarray(model_params.get('param_tex', np.zeros((pc_tex.shape[1], 1))), np.float64)
    sio.savemat(os.path.join(model_path, 'Model_Shape.mat'),
                {'mu_shape': mu, 'w': np.asarray(model_params['pc_shape'], np.float64),
                 'tri': np.asarray(model_params['tri'], np.float64) + float(tri_base),
                 'tex': np.asarray(model_params['mu_tex'], np.float64), 'w_tex': pc_tex, 'alpha_tex': param_tex},
                do_compression=True)
    sio.savemat(os.path.join(model_path, 'Model_Expression.mat'),
                {'mu_exp': np.zeros_like(mu), 'w_exp': np.asarray(model_params['pc_exp'], np.float64)},
                do_compression=True)
    sio.savemat(os.path.join(model_path, 'vertex_code.mat'),
                {'vertex_code': np.asarray(model_params['vertex'], np.float64)}, do_compression=True)
