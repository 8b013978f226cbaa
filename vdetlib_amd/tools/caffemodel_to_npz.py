#!/usr/bin/env python3
"""Import the weights of a Caffe temporal-convolution net into the build's TCN format.

``score_conv_cls`` (reference vdet/tubelet_cls.py:15-51) runs an external Caffe net whose prototxt /
caffemodel live in the T-CNN model zoo, not in the reference tree.  This tool reads a ``.caffemodel``
WITHOUT caffe or compiled protobuf classes -- a minimal reader of the protobuf wire format for the few
fields needed (caffe.proto: NetParameter.layer = 100 / V1 .layers = 2; LayerParameter.name = 1,
.type = 2, .blobs = 7 (V1: name = 4, type = 5 (enum), blobs = 6); BlobProto.shape = 7 {dim = 1},
.data = 5 (packed float), legacy .num/.channels/.height/.width = 1..4) -- and writes an ``.npz`` with
``w0, b0, w1, b1, ...``: convolution / inner-product layers in file order, weights reshaped to
``[Cout, Cin, K]`` (a temporal convolution is a 1 x K or K x 1 Caffe convolution).

    python -m vdetlib_amd.tools.caffemodel_to_npz tcn.caffemodel tcn.npz
    net = TCNNet.from_npz([('det_scores', 1), ('track_scores', 1), ...], 'tcn.npz')
"""
import struct
import sys

import numpy as np


def _varint(buf, i):
    v, shift = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, i
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) of one message; value = int, bytes or 4/8 raw bytes."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _blob(buf):
    dims, legacy, data = [], {}, []
    for fn, wt, v in _fields(buf):
        if fn == 7 and wt == 2:                       # BlobShape
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:               # packed int64 dims
                    j = 0
                    while j < len(v2):
                        d, j = _varint(v2, j)
                        dims.append(d)
                elif f2 == 1 and w2 == 0:
                    dims.append(v2)
        elif fn in (1, 2, 3, 4) and wt == 0:
            legacy[fn] = v
        elif fn == 5 and wt == 2:                     # packed floats
            data.append(np.frombuffer(v, dtype='<f4'))
        elif fn == 5 and wt == 5:
            data.append(np.frombuffer(v, dtype='<f4'))
    arr = np.concatenate(data) if data else np.zeros(0, np.float32)
    if not dims and legacy:
        dims = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
    return arr.astype(np.float32).reshape(dims) if dims and int(np.prod(dims)) == arr.size else arr.astype(np.float32)


def read_caffemodel(path):
    """[(layer_name, [blob arrays])] for every layer that carries blobs, in file order."""
    buf = memoryview(open(path, 'rb').read())
    out = []
    for fn, wt, v in _fields(buf):
        if wt != 2 or fn not in (100, 2):
            continue
        name_field, blob_field = (1, 7) if fn == 100 else (4, 6)
        name, blobs = '', []
        for f2, w2, v2 in _fields(v):
            if f2 == name_field and w2 == 2:
                name = bytes(v2).decode('utf-8', 'replace')
            elif f2 == blob_field and w2 == 2:
                blobs.append(_blob(v2))
        if blobs:
            out.append((name, blobs))
    return out


def tcn_layers(layers):
    """Caffe conv / inner-product blobs -> [(W [Cout,Cin,K], b [Cout])]."""
    res = []
    for name, blobs in layers:
        w = np.asarray(blobs[0], dtype=np.float32)
        if w.ndim == 4:                                # [Cout, Cin, kh, kw], one of kh/kw is 1
            if 1 not in w.shape[2:]:
                raise ValueError("layer %r: %r is not a temporal (1 x K) convolution" % (name, w.shape))
            w = w.reshape(w.shape[0], w.shape[1], -1)
        elif w.ndim == 2:                              # inner product = K 1 convolution
            w = w[:, :, None]
        elif w.ndim != 3:
            continue                                   # not a weight layer we can map (e.g. batch-norm stats)
        b = np.asarray(blobs[1], dtype=np.float32).reshape(-1) if len(blobs) > 1 else np.zeros(w.shape[0], np.float32)
        res.append((np.ascontiguousarray(w), np.ascontiguousarray(b)))
    return res


def convert(src, dst):
    layers = tcn_layers(read_caffemodel(src))
    if not layers:
        raise ValueError("no convolution / inner-product weights found in %s" % src)
    arrs = {}
    for i, (w, b) in enumerate(layers):
        arrs['w%d' % i] = w
        arrs['b%d' % i] = b
    np.savez(dst, **arrs)
    return layers


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        print(__doc__)
        return 2
    layers = convert(argv[0], argv[1])
    for i, (w, b) in enumerate(layers):
        print("layer %d: W %s b %s" % (i, w.shape, b.shape))
    return 0


if __name__ == '__main__':
    sys.exit(main())
