"""Config-1 harness: the MI355X counterpart of the reference's rendering_layer/sample_test.py (main(), 75-186).

    python 3dfacerecon_amd/rendering_layer/sample_test.py GPU [--out DIR] [--model-dir 3dmm]

One 200x200 face from a fixed-pose random 235-d vector (get_random_params(beta=1.0), sample_test.py:93), decoded and
rendered twice (PNCC code texture, then albedo texture) with one Adam step on MSE(pncc, image) in between, exactly the
sequence of the reference script (129-150), then the same post-processing into mask / depth / normal / PNCC / albedo
images (161-184) and the "Op time / Running time" print (186).

Differences, on purpose:
  * the decode follows nets/network.py (the CoarseNet -> render loop this repository targets), not the script's own
    numpy decode, whose conventions differ (rotation order, `im_size - y`, interleaved mu);
  * the BFM .mat files are licensed and absent: without --model-dir the synthetic BFM-scale assets of utils/synth.py
    are used and the background image is synthetic noise;
  * argv[1] must be GPU: there is no CPU path in the product (the CPU leg of config 1 is the oracle, in tests/).
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    mod = sys.modules.get(name)
    if mod is None:
        spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG_DIR, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return mod


def _save_png(path, arr):
    try:
        from PIL import Image
        Image.fromarray(np.round(np.clip(arr, 0, 255)).astype(np.uint8).squeeze()).save(path)
    except ImportError:  # keep the data even without an image library
        np.save(os.path.splitext(path)[0] + ".npy", arr)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("device", help="GPU (the reference also accepts CPU; this implementation has no CPU path)")
    ap.add_argument("--out", default=".")
    ap.add_argument("--seed", type=int, default=3456)
    args = ap.parse_args(argv)
    if args.device.upper() != "GPU":
        raise SystemExit("only GPU is supported: the hot path has no CPU fallback (the CPU restatement lives in oracle/)")
    dev = torch.device("cuda:0")
    ops = _load("_fr_hotpath_ops", os.path.join("rendering_layer", "ops.py"))
    synth = _load("_fr_synth", os.path.join("utils", "synth.py"))
    netm = _load("_fr_network", os.path.join("nets", "network.py"))

    modeldata_3dmm = synth.make_assets()
    im_size = 200
    rs = np.random.RandomState(args.seed)
    im = rs.randint(0, 256, (im_size, im_size, 3)).astype(np.float32)

    pose_param, shape_param, exp_param = synth.get_random_params(im_size, modeldata_3dmm['ndim_shape'],
                                                                 modeldata_3dmm['ndim_exp'], beta=1.0, rand=rs.rand)
    params = np.concatenate([pose_param[:, 0], shape_param[:, 0], exp_param[:, 0]]).astype(np.float32)[None]

    net = netm.FaceRecNet(mesh_data=modeldata_3dmm, batch_size=1, im_size=im_size, device=dev)
    vertex_proj = net.vertices_transform(torch.as_tensor(params, device=dev))        # (1, 3, N)
    tf_vertex = vertex_proj.clone().requires_grad_(True)
    tf_triangles = net.tri
    tf_tex_pncc = net.vertex_code[None]
    tf_tex_abedo = torch.as_tensor(np.asarray(modeldata_3dmm['mu_tex'], np.float32), device=dev)[None]
    tf_image = torch.as_tensor(im / 255.0, device=dev)[None]

    t_start = time.time()
    tf_depth, tf_pncc, tf_normal, tf_tri_ind = ops.render_depth(ver=tf_vertex, tri=tf_triangles, texture=tf_tex_pncc,
                                                                image=tf_image)
    torch.cuda.synchronize()
    t_graph = time.time() - t_start

    # a second call, with the albedo texture
    _, tf_abedo, _, _ = ops.render_depth(ver=tf_vertex, tri=tf_triangles, texture=tf_tex_abedo, image=tf_image)

    optimizer = torch.optim.Adam([tf_vertex], lr=0.001)
    t_start = time.time()
    loss = torch.nn.functional.mse_loss(tf_pncc, tf_image)
    optimizer.zero_grad()
    loss.backward()     # the texture output carries no gradient to the vertices (reference ops.py:95): grads are zero
    optimizer.step()
    tf_depth, tf_pncc, tf_normal, tf_tri_ind = ops.render_depth(ver=tf_vertex, tri=tf_triangles, texture=tf_tex_pncc,
                                                                image=tf_image)
    torch.cuda.synchronize()
    t_run = time.time() - t_start

    depth_buffer = tf_depth[0, :, :, 0].detach().cpu().numpy()
    pncc_map = np.clip(tf_pncc[0].detach().cpu().numpy(), 0.0, 1.0) * 255.0
    normal_map = tf_normal[0].detach().cpu().numpy().copy()
    abedo_map = np.maximum(tf_abedo[0].detach().cpu().numpy(), 0.0)

    # binarization for masking
    mask = np.minimum(np.maximum(depth_buffer, 0.0), 1.0)
    maskimg = np.tile(np.expand_dims(mask, axis=2), [1, 1, 3]) * im
    # depth image
    ind = np.where(depth_buffer > 0.0)
    depthimg = np.zeros_like(depth_buffer)
    if ind[0].size:
        lo, hi = np.min(depth_buffer[ind]), np.max(depth_buffer[ind])
        depthimg = np.maximum((depth_buffer - lo) / max(hi - lo, 1e-12), 0.0) * 255.0
    # normal image
    flip_ind = (normal_map[:, :, 2] < 0)
    normal_map[flip_ind] *= -1.0
    mag_map = np.sum(normal_map ** 2, axis=2)
    zero_ind = (mag_map == 0)
    mag_map[zero_ind] = 1.0
    normal_map = normal_map / np.expand_dims(np.sqrt(mag_map), axis=2)
    normalimg = (normal_map + 1) / 2.0 * 255.0
    normalimg[zero_ind] = 0.0

    os.makedirs(args.out, exist_ok=True)
    _save_png(os.path.join(args.out, 'test_pncc_hip_GPU.png'), pncc_map)
    _save_png(os.path.join(args.out, 'test_maskimg_hip_GPU.png'), maskimg)
    _save_png(os.path.join(args.out, 'test_depthimg_hip_GPU.png'), depthimg)
    _save_png(os.path.join(args.out, 'test_normalimg_hip_GPU.png'), normalimg)
    _save_png(os.path.join(args.out, 'test_abedoimg_hip_GPU.png'), abedo_map * 255.0)
    cov = float((tf_tri_ind >= 0).float().mean())
    print('Op time: {} s, Running time: {} s.  coverage {:.3f}'.format(t_graph, t_run, cov))
    # the loss reads the texture output only, which carries no vertex gradient (reference ops.py:95): autograd then
    # hands the vertices no gradient at all
    g = tf_vertex.grad
    return {"coverage": cov, "loss": float(loss.detach()), "grad_abs_max": 0.0 if g is None else float(g.abs().max())}


if __name__ == '__main__':
    main()
