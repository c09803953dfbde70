"""BASELINE.json configs[0]: single 200x200 face, fixed 235-d params (sample_test.py equivalent).
CPU leg: the oracle's op functor and MEX z-buffer restatements against the committed hashes / samples.
GPU leg: the HIP path against the oracle on the same inputs, and the harness script end to end."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, pkg

CFG = json.load(open(os.path.join(GOLDEN, "config1_oracle.json")))


def _inputs(synth, full_assets):
    rs = np.random.RandomState(3456)
    pose, shp, exp = synth.get_random_params(200, 199, 29, beta=1.0, rand=rs.rand)
    P = np.concatenate([pose[:, 0], shp[:, 0], exp[:, 0]]).astype(np.float32)[None]
    return full_assets, P


def test_config1_cpu_oracle_matches_golden(oracle, synth, full_assets):
    A, P = _inputs(synth, full_assets)
    np.testing.assert_array_equal(P[0, :7], np.array(CFG["pose"], np.float32))        # beta = 1.0: the fixed pose
    V = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0)
    d, t, n, ti = oracle.render_depth(V, A["tri"], A["vertex"][None], 200, 200)
    for k, v in (("vertex_proj", V), ("depth", d), ("texture_image", t), ("normal", n), ("tri_ind", ti)):
        assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() == CFG["sha256"][k], k
    for y, x, tri, dep in CFG["samples"]:
        assert ti[0, y, x, 0] == tri and d[0, y, x, 0] == np.float32(dep)
    assert abs(float((ti >= 0).mean()) - CFG["coverage"]) < 1e-12
    # the MEX z-buffer (all double, column-major, u+v<=1) sees the same winners on this input
    _, ti_mex = oracle.zbuffer_mex(V[0].astype(np.float64), A["tri"].astype(np.float64),
                                   A["vertex"].astype(np.float64), np.zeros((200, 200, 3)))
    assert int((ti_mex != ti[0, :, :, 0]).sum()) == CFG["mex_vs_op_tri_ind_mismatches"] == 0


@pytest.mark.gpu
def test_config1_gpu_vs_oracle(oracle, synth, full_assets):
    import torch
    from gpu_util import assert_render_equal, net_mod, ops
    A, P = _inputs(synth, full_assets)
    R = oracle.rotation_matrix_batch(P[:, :3])
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=1, im_size=200)
    V = net.vertices_transform(torch.as_tensor(P, device="cuda:0"), R=torch.as_tensor(R, device="cuda:0"))
    Vo = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=R)
    np.testing.assert_array_equal(V.cpu().numpy(), Vo)
    outs = ops().render_depth(V, net.tri, net.vertex_code, torch.zeros((1, 200, 200, 3), device="cuda:0"))
    got = tuple(o.cpu().numpy() for o in outs)
    assert_render_equal(got, oracle.render_depth(Vo, A["tri"], A["vertex"][None], 200, 200), "config 1")
    for y, x, tri, dep in CFG["samples"]:
        assert got[3][0, y, x, 0] == tri and got[0][0, y, x, 0] == np.float32(dep)


@pytest.mark.gpu
def test_sample_test_harness_runs(tmp_path):
    st = pkg("rendering_layer.sample_test")
    res = st.main(["GPU", "--out", str(tmp_path)])
    assert abs(res["coverage"] - CFG["coverage"]) < 0.02
    assert res["grad_abs_max"] == 0.0      # texture output carries no gradient to the vertices (reference ops.py:95)
    assert len(list(tmp_path.iterdir())) == 5


def test_sample_test_refuses_cpu():
    st = pkg("rendering_layer.sample_test")
    with pytest.raises(SystemExit):
        st.main(["CPU"])
