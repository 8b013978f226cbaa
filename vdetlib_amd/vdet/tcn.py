"""A gfx950 temporal-convolution network with pycaffe's calling convention, to plug into
``score_conv_cls`` (reference vdet/tubelet_cls.py:15-51).

The reference feeds per-tubelet channel sequences to an external Caffe TCN whose prototxt/weights
are NOT part of the reference tree, so the architecture here is the build's own (parity unpinned,
DESIGN.md section 2): a stack of 1-D "same" convolutions over the tubelet length, ReLU between
layers, a 2-way softmax at the end; ``forward()`` returns ``{'probs': [1, 2, L]}`` like the
reference expects (:47-48).  Every layer runs on the GPU (vdet_conv1d_f32).
"""
import ctypes

import numpy as np

from .. import _lib


class Blob(object):
    """Minimal pycaffe blob: .shape, .reshape(*dims), .data (numpy float32)."""

    def __init__(self, channels):
        self.shape = (1, channels, 1, 1)
        self.data = np.zeros(self.shape, dtype=np.float32)

    def reshape(self, *dims):
        self.shape = tuple(int(d) for d in dims)
        self.data = np.zeros(self.shape, dtype=np.float32)


class TCNNet(object):
    """inputs: ordered list of (blob_name, channels) concatenated along the channel axis;
    layers: list of (W [Cout,Cin,K] float32, b [Cout] float32); the last layer must have Cout == 2."""

    def __init__(self, inputs, layers):
        self.inputs = [(n, int(c)) for n, c in inputs]
        self.blobs = {n: Blob(c) for n, c in self.inputs}
        self.layers = [(np.ascontiguousarray(w, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32))
                       for w, b in layers]
        cin = sum(c for _, c in self.inputs)
        for w, b in self.layers:
            if w.ndim != 3 or w.shape[1] != cin or w.shape[2] % 2 != 1 or b.shape != (w.shape[0],):
                raise ValueError("layer shapes do not chain: %r after %d channels" % (w.shape, cin))
            cin = w.shape[0]
        if cin != 2:
            raise ValueError("the last layer must produce 2 channels (probs[:, 1, :] is the score)")

    @staticmethod
    def random(inputs, hidden=(16, 16), kernel=3, seed=0):
        rng = np.random.RandomState(seed)
        cin = sum(c for _, c in inputs)
        layers = []
        for cout in list(hidden) + [2]:
            layers.append((rng.randn(cout, cin, kernel).astype(np.float32) / np.sqrt(cin * kernel),
                           (0.1 * rng.randn(cout)).astype(np.float32)))
            cin = cout
        return TCNNet(inputs, layers)

    @staticmethod
    def from_npz(inputs, path):
        """Weights saved as ``w0, b0, w1, b1, ...`` (e.g. by vdetlib_amd.tools.caffemodel_to_npz)."""
        z = np.load(path)
        layers = []
        i = 0
        while 'w%d' % i in z.files:
            layers.append((z['w%d' % i], z['b%d' % i]))
            i += 1
        return TCNNet(inputs, layers)

    def save_npz(self, path):
        arrs = {}
        for i, (w, b) in enumerate(self.layers):
            arrs['w%d' % i] = w
            arrs['b%d' % i] = b
        np.savez(path, **arrs)

    def forward(self):
        L = self.blobs[self.inputs[0][0]].shape[3]
        x = np.concatenate([np.asarray(self.blobs[n].data, dtype=np.float32).reshape(c, L) for n, c in self.inputs], 0)
        x = np.ascontiguousarray(x)
        ctx = _lib.get_context()
        ctx.reset_stream()
        for li, (w, b) in enumerate(self.layers):
            act = 2 if li == len(self.layers) - 1 else 1
            out = np.empty((w.shape[0], L), dtype=np.float32)
            ctx.check(ctx.lib.vdet_conv1d_f32(ctx.h, x.ctypes.data, x.shape[0], L, w.ctypes.data, b.ctypes.data,
                                              w.shape[0], w.shape[2], act, out.ctypes.data))
            x = out
        return {'probs': x[None]}
