"""CPU: asset I/O with the reference's contracts (utils/parser_3dmm.py dict keys, 235-line label files)."""
import os

import numpy as np
import pytest

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


def _make_dataset(root, synth, names, size=(12, 10), ext=".jpg"):
    """synthetic dataset in the reference's layout: face_images/<name>.jpg (PNG bytes are fine for the decoder, the
    extension is what the list readers build), labels/<name>.txt, *_list.txt"""
    from PIL import Image
    labels = pkg("utils.labels")
    rs = np.random.RandomState(0)
    imgs, P = {}, synth.sample_params_batch(len(names), beta=0.7, seed=9)
    for i, n in enumerate(names):
        os.makedirs(os.path.dirname(os.path.join(root, "face_images", n)), exist_ok=True)
        os.makedirs(os.path.dirname(os.path.join(root, "labels", n)), exist_ok=True)
        rgb = rs.randint(0, 256, (size[0], size[1], 3)).astype(np.uint8)
        Image.fromarray(rgb).save(os.path.join(root, "face_images", n + ext), format="PNG")
        labels.write_label_file(os.path.join(root, "labels", n + ".txt"), P[i])
        imgs[n] = rgb
    return imgs, P


def test_listfile_readers_contract(tmp_path):
    lr = pkg("utils.listfile_reader")
    root = str(tmp_path)
    open(os.path.join(root, "train_list.txt"), "w").write("Ana/000045.txt\nBob/17.txt\ntext/00.txt\n\nignored/1.txt\n")
    im, lab = lr.read_listfile_trainval(root, "train_list.txt")
    # character-set strip, exactly like the reference: 'text/00.txt' loses its leading 't' as well
    assert [os.path.relpath(p, root) for p in im] == ["face_images/Ana/000045.jpg", "face_images/Bob/17.jpg",
                                                      "face_images/ext/00.jpg"]
    assert [os.path.relpath(p, root) for p in lab] == ["labels/Ana/000045.txt", "labels/Bob/17.txt", "labels/ext/00.txt"]
    open(os.path.join(root, "test_list.txt"), "w").write("Ana/000045.txt\n")
    assert lr.read_listfile_test(root, "test_list.txt") == [os.path.join(root, "face_images", "Ana/000045.jpg")]
    open(os.path.join(root, "empty_list.txt"), "w").write("")
    assert lr.read_listfile_trainval(root, "empty_list.txt") == ([], [])


def test_generators_roundtrip(tmp_path, synth):
    dp = pkg("utils.data_process")
    root = str(tmp_path / "vggface_synth")
    names = ["Ana/0001", "Ana/0002", "Bob/0001", "Cid/0009"]
    imgs, P = _make_dataset(root, synth, names)
    for fn, sel in (("train_list.txt", names), ("val_list.txt", names[:2]), ("test_list.txt", names[2:])):
        open(os.path.join(root, fn), "w").write("".join(n + ".txt\n" for n in sel))
    mean = 126.064                                                       # network.py:32 gray_mean
    gen = dp.trainval_generator(2, [12, 10], 235, dataset=root, img_mean=mean, phase="train")
    for rnd in range(3):                                                 # wraps around after two batches
        images, labels = next(gen)
        assert images.shape == (2, 12, 10, 1) and images.dtype == np.float64 and labels.shape == (2, 1, 1, 235)
        for j in range(2):
            n = names[(2 * rnd + j) % 4]
            rgb = imgs[n].astype(np.float64)
            want = 0.3 * rgb[:, :, 0] + 0.59 * rgb[:, :, 1] + 0.11 * rgb[:, :, 2] - mean
            np.testing.assert_allclose(images[j, :, :, 0], want, rtol=0, atol=1e-9)
            np.testing.assert_allclose(labels[j, 0, 0], P[names.index(n)], rtol=0, atol=6e-7 + 1e-3 * 6e-8)
    v_im, v_lab = next(dp.trainval_generator(2, [12, 10], 235, dataset=root, img_mean=mean, phase="val"))
    assert v_im.shape == (2, 12, 10, 1)
    t_im, t_files = next(dp.test_generator(2, [12, 10], dataset=root, img_mean=mean))
    assert t_im.shape == (2, 12, 10, 1) and [os.path.basename(f) for f in t_files] == ["0001.jpg", "0009.jpg"]
    with pytest.raises(NotImplementedError):
        next(dp.trainval_generator(2, [12, 10], 235, dataset=root, phase="test"))
    # error behaviour of the reference: missing file, wrong label length, short tail batch
    with pytest.raises(FileNotFoundError):
        dp.prepare_input_image([os.path.join(root, "face_images", "nope.jpg")], 1, [12, 10])
    with pytest.raises(IOError):
        dp.prepare_input_label([os.path.join(root, "labels", "Ana/0001.txt")], 1, 100)
    # an image with the right pixel COUNT but the wrong height x width must raise, not be folded into the plane (the
    # reference's `input_image[i,:,:,0] = ...` assignment raises a ValueError there)
    with pytest.raises(ValueError):
        dp.prepare_input_image([os.path.join(root, "face_images", "Ana/0001.jpg")], 1, [6, 20])
    g3 = dp.trainval_generator(3, [12, 10], 235, dataset=root, phase="train")
    next(g3)
    with pytest.raises(AssertionError):
        next(g3)                                                         # 4 samples, batch 3: the tail slice is short
