"""Image / label batch generators with the reference's contract (utils/data_process.py:7-101): the host-side feeders of
the CoarseNet -> render loop (`im_gray` [B,H,W,1] float64 minus the dataset mean, `params_label` [B,1,1,235]).

The reference decodes with cv2.imread (BGR, always 3 channels); OpenCV is not in this image, so files are decoded with
Pillow into the same BGR uint8 array (`read_image_bgr`; `.npy` arrays are accepted too, for tests and for callers that
already hold decoded frames).  Behaviour kept from the reference, on purpose:
  * gray = 0.3 R + 0.59 G + 0.11 B in float64, minus `img_mean` (data_process.py:30-31); the image must already have the
    network's size -- 3-channel images are NOT resized (only 2-D arrays are, and those skip the mean subtraction, :26-28);
  * a missing file raises FileNotFoundError, an undecodable one IOError, a label of the wrong length IOError (:17-35, :49-59);
  * the generators walk the list with `counter = (counter + batch_size) % len(files)` (:83, :100): when the list length
    is not a multiple of the batch size the short tail slice fails the `len == batch_size` assertion, as there.
"""
import os

import numpy as np

try:  # package import (3dfacerecon_amd.utils.data_process) or the reference's flat `utils.` layout
    from . import listfile_reader as file_reader
except ImportError:  # pragma: no cover
    import listfile_reader as file_reader

ROOT_PATH = os.path.join(os.path.dirname(__file__), '..', '..')


def read_image_bgr(path):
    """uint8 [H,W,3] in BGR order (what cv2.imread returns), or the array stored in a .npy file as it is."""
    if path.endswith('.npy'):
        return np.load(path)
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'), np.uint8)
    return rgb[:, :, ::-1]


def _resize_nearest(im, size_wh):
    w, h = int(size_wh[0]), int(size_wh[1])  # cv2.resize takes (width, height)
    ys = np.minimum((np.arange(h) + 0.5) * im.shape[0] / h, im.shape[0] - 1).astype(int)
    xs = np.minimum((np.arange(w) + 0.5) * im.shape[1] / w, im.shape[1] - 1).astype(int)
    return im[ys][:, xs]


def prepare_input_image(image_files, batch_size, img_size, img_mean=127.0):
    ''' a batch of image files -> [batchsize, h, w, 1] float64 (reference data_process.py:7-35) '''
    assert len(image_files) == batch_size
    input_image = np.zeros([batch_size, img_size[0], img_size[1], 1])
    for i in range(batch_size):
        if not os.path.exists(image_files[i]):
            raise FileNotFoundError(image_files[i])
        try:
            im = read_image_bgr(image_files[i])
        except Exception:
            raise IOError(image_files[i])
        if im.ndim == 2:      # gray image already: resized, stored as it is (no mean subtraction, as in the reference)
            input_image[i, :, :, 0] = _resize_nearest(im, img_size)
        elif im.ndim == 3:
            # R: 0.3, G: 0.59, B: 0.11 on a BGR array
            im_gray = 0.3 * im[:, :, 2] + 0.59 * im[:, :, 1] + 0.11 * im[:, :, 0]
            input_image[i, :, :, 0] = im_gray - img_mean
        else:
            raise IOError(image_files[i])
    return input_image


def prepare_input_label(label_files, batch_size, label_dim):
    '''a batch of label files -> [batchsize, 1, 1, label_dim] float64 (reference data_process.py:39-60)'''
    assert len(label_files) == batch_size
    input_label = np.zeros([batch_size, 1, 1, label_dim])
    for i in range(batch_size):
        if not os.path.exists(label_files[i]):
            raise FileNotFoundError(label_files[i])
        try:
            labels = np.loadtxt(label_files[i])
        except Exception:
            raise IOError(label_files[i])
        if labels.ndim == 1 and labels.shape[0] == label_dim:
            input_label[i, 0, 0, :] = labels
        else:
            raise IOError(label_files[i])
    return input_label


def trainval_generator(batch_size, img_size, label_dim, dataset=None, img_mean=127.0, phase='train'):
    '''Endless generator of (images, labels) batches from <ROOT>/data/<dataset> (an absolute `dataset` path is used as
    it is); phase 'train' reads train_list.txt, 'val' val_list.txt (reference data_process.py:63-83).'''
    dataset_path = os.path.join(ROOT_PATH, 'data', dataset)
    if phase == 'train':
        image_files, label_files = file_reader.read_listfile_trainval(dataset_path, 'train_list.txt')
    elif phase == 'val':
        image_files, label_files = file_reader.read_listfile_trainval(dataset_path, 'val_list.txt')
    else:
        raise NotImplementedError
    counter = 0
    while True:
        yield prepare_input_image(image_files[counter:counter + batch_size], batch_size, img_size, img_mean), \
            prepare_input_label(label_files[counter:counter + batch_size], batch_size, label_dim)
        counter = (counter + batch_size) % len(image_files)


def test_generator(batch_size, img_size, dataset=None, img_mean=127.0):
    '''Endless generator of (images, image file names) batches from test_list.txt (reference data_process.py:86-101).'''
    dataset_path = os.path.join(ROOT_PATH, 'data', dataset)
    image_files = file_reader.read_listfile_test(dataset_path, 'test_list.txt')
    counter = 0
    while True:
        yield prepare_input_image(image_files[counter:counter + batch_size], batch_size, img_size, img_mean), \
            image_files[counter:counter + batch_size]
        counter = (counter + batch_size) % len(image_files)


test_generator.__test__ = False  # not a pytest test
