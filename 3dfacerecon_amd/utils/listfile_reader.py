"""List-file readers with the reference's contract (utils/listfile_reader.py:4-49).

A dataset directory holds `train_list.txt` / `val_list.txt` / `test_list.txt` whose lines name samples as
`<identity>/<frame>.txt` (prepare_data/script_split_dataset.py writes them); images live under
`<dataset>/face_images/<name>.jpg`, 235-d labels under `<dataset>/labels/<name>.txt`.

Faithful to the reference in two details worth knowing:
  * every line is cleaned with str.strip('.txt\\n'), i.e. a CHARACTER-SET strip of '.', 't', 'x' and newlines from both
    ends -- not a suffix removal: a sample name that begins or ends with 't' / 'x' / '.' loses those characters
    (`Ana/000045.txt` -> `Ana/000045`, but `text/00.txt` -> `ext/00`);
  * reading stops at the first line that is empty after that cleaning, later lines are ignored.
"""
import os

_STRIP_CHARS = '.txt\n'  # the reference's strip argument (listfile_reader.py:17, 41)


def _sample_names(listfile):
    names = []
    with open(listfile) as f:
        for raw in f:
            name = raw.strip(_STRIP_CHARS)
            if name == '':
                break
            names.append(name)
    return names


def read_listfile_trainval(dataset_path, filename):
    '''File reader for list in train and val -> (image files, label files), absolute paths in list order.'''
    names = _sample_names(os.path.join(dataset_path, filename))
    images = [os.path.join(dataset_path, 'face_images', n + '.jpg') for n in names]
    labels = [os.path.join(dataset_path, 'labels', n + '.txt') for n in names]
    return images, labels


def read_listfile_test(dataset_path, filename):
    '''File reader for list in test -> image files.'''
    names = _sample_names(os.path.join(dataset_path, filename))
    return [os.path.join(dataset_path, 'face_images', n + '.jpg') for n in names]
