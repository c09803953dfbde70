"""Host-side feeders of the CoarseNet -> render loop: image / label batches with the reference's contract
(utils/data_process.py:7-101) -- `im_gray` [B,H,W,1] float64 minus the dataset mean, `params_label` [B,1,1,235].

Design: two pure per-sample decoders (`load_gray`, `load_label`), one batch assembler (`_stack`) and ONE cursor class
(`ListCursor`) that walks a list file in batch-sized windows; the reference's public names (`prepare_input_image`,
`prepare_input_label`, `trainval_generator`, `test_generator`) are thin views of those.  OpenCV is not in this image, so
files are decoded with Pillow into the BGR uint8 array cv2.imread would return (`read_image_bgr`; `.npy` arrays are
accepted too, for tests and for callers that already hold decoded frames).

Behaviour that is the reference's and is held by tests/test_assets_cpu.py (as behaviour, not as copied code):
  * gray = 0.3 R + 0.59 G + 0.11 B in float64, minus `img_mean` (reference :30-31); a 3-channel image must already have
    the network's size -- only 2-D arrays are resized, and those skip the mean subtraction (:26-28);
  * a missing file raises FileNotFoundError, an undecodable one IOError, a label of the wrong length IOError;
  * the window start advances by the batch size modulo the list length (:83, :100), so a list whose length is not a
    multiple of the batch size eventually yields a short window, which fails the `len == batch_size` assertion.
"""
import os

import numpy as np

try:  # package import (3dfacerecon_amd.utils.data_process) or the reference's flat `utils.` layout
    from . import listfile_reader as file_reader
except ImportError:  # pragma: no cover
    import listfile_reader as file_reader

ROOT_PATH = os.path.join(os.path.dirname(__file__), '..', '..')
GRAY_WEIGHTS_BGR = np.array([0.11, 0.59, 0.3])  # B, G, R


def read_image_bgr(path):
    """uint8 [H,W,3] in BGR order (what cv2.imread returns), or the array stored in a .npy file as it is."""
    if path.endswith('.npy'):
        return np.load(path)
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'), np.uint8)[:, :, ::-1]


def _resize_nearest(im, size_wh):
    w, h = int(size_wh[0]), int(size_wh[1])  # cv2.resize takes (width, height)
    rows = np.minimum((np.arange(h) + 0.5) * im.shape[0] / h, im.shape[0] - 1).astype(int)
    cols = np.minimum((np.arange(w) + 0.5) * im.shape[1] / w, im.shape[1] - 1).astype(int)
    return im[np.ix_(rows, cols)]


def _existing(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return path


def load_gray(path, img_size, img_mean):
    """One image file -> [h, w] float64 network input plane."""
    try:
        im = read_image_bgr(_existing(path))
    except FileNotFoundError:
        raise
    except Exception:
        raise IOError(path)
    if im.ndim == 2:   # already gray: resized, no mean subtraction
        return _resize_nearest(im, img_size).astype(np.float64)
    if im.ndim != 3:
        raise IOError(path)
    b, g, r = (im[:, :, k] for k in range(3))
    return (GRAY_WEIGHTS_BGR[2] * r + GRAY_WEIGHTS_BGR[1] * g + GRAY_WEIGHTS_BGR[0] * b) - img_mean


def load_label(path, label_dim):
    """One whitespace-separated label file -> [label_dim] float64."""
    try:
        vec = np.loadtxt(_existing(path))
    except FileNotFoundError:
        raise
    except Exception:
        raise IOError(path)
    if vec.shape != (label_dim,):
        raise IOError(path)
    return vec


def _stack(files, batch_size, sample_shape, load):
    assert len(files) == batch_size
    out = np.zeros((batch_size,) + tuple(sample_shape))
    for row, path in zip(out, files):
        arr = np.asarray(load(path))
        # same element count is not enough: a 100x400 image must not be folded into a 200x200 plane (the reference's
        # `input_image[i,:,:,0] = im_gray - mean` raises on that mismatch)
        if tuple(d for d in arr.shape if d != 1) != tuple(d for d in sample_shape if d != 1):
            raise ValueError("%s: sample of shape %s does not fit %s" % (path, arr.shape, tuple(sample_shape)))
        row[...] = np.reshape(arr, sample_shape)
    return out


def prepare_input_image(image_files, batch_size, img_size, img_mean=127.0):
    """A batch of image files -> [batchsize, h, w, 1] float64."""
    return _stack(image_files, batch_size, (img_size[0], img_size[1], 1), lambda p: load_gray(p, img_size, img_mean))


def prepare_input_label(label_files, batch_size, label_dim):
    """A batch of label files -> [batchsize, 1, 1, label_dim] float64."""
    return _stack(label_files, batch_size, (1, 1, label_dim), lambda p: load_label(p, label_dim))


class ListCursor:
    """Endless iterator over batch-sized windows of one or more parallel file lists; `make_batch(*windows)` turns the
    windows into what the caller wants.  The start of the window moves by `batch_size` modulo the list length."""

    def __init__(self, lists, batch_size, make_batch):
        self.lists = [list(l) for l in lists]
        self.batch_size = int(batch_size)
        self.make_batch = make_batch
        self.start = 0

    def __iter__(self):
        return self

    def __next__(self):
        lo, hi = self.start, self.start + self.batch_size
        batch = self.make_batch(*[l[lo:hi] for l in self.lists])
        self.start = hi % len(self.lists[0])
        return batch


_TRAINVAL_LISTS = {'train': 'train_list.txt', 'val': 'val_list.txt'}


def trainval_generator(batch_size, img_size, label_dim, dataset=None, img_mean=127.0, phase='train'):
    """Endless (images, labels) batches from <ROOT>/data/<dataset> (an absolute `dataset` is used as it is): phase
    'train' reads train_list.txt, 'val' val_list.txt; any other phase raises NotImplementedError at the first batch."""
    def gen():
        if phase not in _TRAINVAL_LISTS:
            raise NotImplementedError(phase)
        images, labels = file_reader.read_listfile_trainval(os.path.join(ROOT_PATH, 'data', dataset),
                                                            _TRAINVAL_LISTS[phase])
        yield from ListCursor([images, labels], batch_size,
                               lambda im, lab: (prepare_input_image(im, batch_size, img_size, img_mean),
                                                prepare_input_label(lab, batch_size, label_dim)))
    return gen()


def test_generator(batch_size, img_size, dataset=None, img_mean=127.0):
    """Endless (images, image file names) batches from test_list.txt."""
    images = file_reader.read_listfile_test(os.path.join(ROOT_PATH, 'data', dataset), 'test_list.txt')
    return ListCursor([images], batch_size, lambda im: (prepare_input_image(im, batch_size, img_size, img_mean), im))


test_generator.__test__ = False  # not a pytest test
