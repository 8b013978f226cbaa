"""Helpers of the reference's utils/common.py that the hot path and its callers use:
``iou`` (:451-468, computed on the GPU), ``options`` (:470-471), path/list helpers.

Out of scope (external engines): image cropping / CNN preprocessing
(:141-280), MATLAB and Caffe launchers (:302-396), window files (:48-121), SVM loader (:416-423).
"""
import argparse
import codecs
import os
import pickle as _pickle
import re
import tempfile


class AttrDict(dict):
    """Stand-in for easydict.EasyDict (not installed here): dict with attribute access, nested
    dicts converted recursively.  ``hasattr(opts, 'nms_thres')`` works as vdet/track.py expects."""

    def __init__(self, d=None, **kwargs):
        super(AttrDict, self).__init__()
        merged = dict(d or {})
        merged.update(kwargs)
        for k, v in merged.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super(AttrDict, self).__setitem__(k, AttrDict._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)


def options(option_dict):
    """utils/common.py:470-471 (EasyDict there)."""
    return AttrDict(option_dict)


def iou(boxes1, boxes2):
    """utils/common.py:451-468: float64 IoU matrix [n1,n2], +1 pixel convention.  Runs on the GPU
    (vdet_iou_f64); bit-exact with the numpy expression of the reference."""
    from .. import ops
    return ops.iou(boxes1, boxes2)


def pickle(data, file_path):
    with open(file_path, 'wb') as f:
        _pickle.dump(data, f, _pickle.HIGHEST_PROTOCOL)


def unpickle(file_path):
    with open(file_path, 'rb') as f:
        return _pickle.load(f)


def read_list(file_path, coding=None):
    """One stripped string per line (utils/common.py:28-35)."""
    if coding is None:
        with open(file_path, 'r') as f:
            return [line.strip() for line in f.readlines()]
    with codecs.open(file_path, 'r', coding) as f:
        return [line.strip() for line in f.readlines()]


def write_list(arr, file_path, coding=None):
    """utils/common.py:38-45: items joined by newlines, no trailing newline."""
    if coding is None:
        with open(file_path, 'w') as f:
            f.write('\n'.join('{}'.format(item) for item in arr))
    else:
        with codecs.open(file_path, 'w', coding) as f:
            f.write(u'\n'.join(arr))


def _tryint(s):
    try:
        return int(s)
    except ValueError:
        return s


def alphanum_key(s):
    """"z23a" -> ["z", 23, "a"] (utils/common.py:129-133)."""
    return [_tryint(c) for c in re.split('([0-9]+)', s)]


def sort_nicely(l):
    """In-place human sort (utils/common.py:135-138)."""
    l.sort(key=alphanum_key)


def basename(file_path):
    return os.path.basename(file_path)


def stem(file_path):
    return os.path.splitext(os.path.basename(file_path))[0]


def isimg(name):
    return name.lower().endswith(('.jpeg', '.png', '.jpg'))


def imread(image_path):
    """utils/common.py:375-376 reads with OpenCV.  The CNN side is external to this build; callers
    that need pixels inject their own reader (``video_det.imread = ...``)."""
    try:
        import cv2
    except ImportError:
        raise RuntimeError("imread needs OpenCV (the image/CNN side is external to vdetlib_amd); "
                           "assign your own reader to the module attribute `imread`")
    return cv2.imread(image_path, cv2.IMREAD_COLOR)


def temp_file(suffix=''):
    f, name = tempfile.mkstemp(suffix=suffix)
    os.close(f)
    return name


def quick_args(arglist):
    """utils/common.py:406-413: positional args, optionally (name, type) tuples."""
    parser = argparse.ArgumentParser()
    for arg in arglist:
        if type(arg) == tuple:
            parser.add_argument(arg[0], type=arg[1])
        else:
            parser.add_argument(arg)
    return parser.parse_args()
