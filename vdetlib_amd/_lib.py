"""ctypes binding of libvdet_hip.so (C-ABI: include/vdet_hip.h).

Fails loudly: no library or no GPU -> RuntimeError.  Nothing here (or anywhere under
vdetlib_amd/) falls back to a CPU implementation.
"""
import ctypes
import importlib.util
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdet_hip.so")

VDET_OK, VDET_EINVAL, VDET_ECAP, VDET_EHIP, VDET_EDIVZERO, VDET_ENOMEM, VDET_EINDEX, VDET_EAGAIN = 0, -1, -2, -3, -4, -5, -6, -7
LAYOUT_FBC, LAYOUT_FCB = 0, 1

# every symbol include/vdet_hip.h declares: (name, restype, argtypes)
_i64, _f64, _f32, _vp, _ci = ctypes.c_int64, ctypes.c_double, ctypes.c_float, ctypes.c_void_p, ctypes.c_int
SYMBOLS = {
    "vdet_create": (_ci, [ctypes.POINTER(_vp), _ci]),
    "vdet_destroy": (_ci, [_vp]),
    "vdet_set_stream": (_ci, [_vp, _vp]),
    "vdet_reset_stream": (_ci, [_vp]),
    "vdet_sync": (_ci, [_vp]),
    "vdet_last_error": (ctypes.c_char_p, [_vp]),
    "vdet_version": (ctypes.c_char_p, []),
    "vdet_last_timing_ms": (_ci, [_vp, _vp]),
    "vdet_last_launches": (_ci, [_vp, _vp]),
    "vdet_set_timing": (_ci, [_vp, _ci]),
    "vdet_query": (_ci, [_vp, _ci]),
    "vdet_set_cache": (_ci, [_vp, _ci]),
    "vdet_invalidate": (_ci, [_vp]),
    "vdet_set_async": (_ci, [_vp, _ci]),
    "vdet_nms_f32": (_ci, [_vp, _vp, _i64, _i64, _ci, _f64, _vp, _vp, ctypes.POINTER(_i64)]),
    "vdet_track_det_nms_f32": (_ci, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _f64, _vp, ctypes.POINTER(_i64)]),
    "vdet_track_det_nms_batch": (_ci, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _f64, _vp, _vp]),
    "vdet_iou_f64": (_ci, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "vdet_svm_scores_f64": (_ci, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "vdet_svm_scores_f32": (_ci, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "vdet_spatial_maxpool_f64": (_ci, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _f64, _vp, _vp]),
    "vdet_series_completion_f64": (_ci, [_vp, _vp, _vp, _i64]),
    "vdet_series_maxpool_f64": (_ci, [_vp, _vp, _vp, _vp, _i64, _ci, _f64]),
    "vdet_series_interp_f64": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _ci, _vp]),
    "vdet_threshold_topk": (_ci, [_vp, _vp, _ci, _i64, _i64, _ci, _ci, _f64, _ci, _vp, _vp]),
    "vdet_conv1d_f32": (_ci, [_vp, _vp, _ci, _ci, _vp, _vp, _ci, _ci, _ci, _vp]),
    "vdet_nms_volume": (_ci, [_vp, _vp, _vp, _ci, _i64, _i64, _i64, _f64, _ci, _f32, _vp, _vp, _i64]),
    "vdet_nms_volume_topk": (_ci, [_vp, _vp, _vp, _ci, _i64, _i64, _i64, _f64, _ci, _f32, _ci, _vp, _vp, _i64]),
    "vdet_det_nms_volume": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _ci, _ci, _f32, _ci, _f64, _vp, _vp, _vp, _vp, _vp]),
    "vdet_nms_volume_ordered": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f64, _vp, _vp, _i64]),
    "vdet_argsort_volume": (_ci, [_vp, _vp, _ci, _i64, _i64, _i64, _ci, _f32, _vp, _vp]),
    "vdet_video_batch": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f64, _f64, _ci, _f64, _ci, _vp, _vp, _vp, _i64, _vp, _vp,
                               _f64, _ci, _vp, _vp, _vp]),
    "vdet_volume_pass_batch": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _ci, _f32, _vp, _f32, _f32, _vp, _vp, _ci, _f32]),
    "vdet_volume_pass": (_ci, [_vp, _vp, _i64, _i64, _i64, _ci, _f32, _vp, _f32, _f32, _vp, _vp, _ci, _f32]),
    "vdet_track_volume": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _f64, _f64, _ci, _f64, _ci, _vp, _vp, _vp]),
    "vdet_nms_track_volume": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _f64, _f64, _ci, _f64, _ci, _vp, _vp, _vp,
                                    _i64, _vp, _vp]),
    "vdet_rescore_tracks": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _f64, _ci, _vp, _vp, _vp]),
    "vdet_temporal_maxpool_f32": (_ci, [_vp, _vp, _vp, _i64, _i64, _ci, _f32]),
    "vdet_temporal_conv_f32": (_ci, [_vp, _vp, _vp, _i64, _i64, _vp, _ci, _f32, _f32]),
    "vdet_temporal_maxpool_conv_f32": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _ci, _f32, _vp, _f32, _f32]),
}



class RetryError(RuntimeError):
    """vdet_sync in asynchronous mode: a graph build outgrew its scratch; the scratch has been enlarged,
    enqueue the same calls again (include/vdet_hip.h: VDET_EAGAIN)."""


_lib = None
_lock = threading.Lock()
_ctxs = {}


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME
    libamdhip64.so.7 as /opt/rocm's).  If libvdet_hip.so pulled in the system copy first, a later
    `import torch` would map a SECOND runtime (torch then sees no device, and streams / pointers
    could not be shared).  Loading torch's copy first makes the dynamic linker hand that same
    runtime to libvdet_hip.so (SONAME match) and to torch."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.isfile(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


def load_library():
    """dlopen libvdet_hip.so and declare the prototypes (no GPU needed for this step)."""
    global _lib
    if _lib is None:
        _preload_hip_runtime()
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "libvdet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or vdetlib_amd/csrc/build.sh. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)     # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class Context(object):
    """One vdet_ctx (one GPU, one stream). Not thread-safe, like the reference (GIL held)."""

    def __init__(self, device=-1):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.vdet_create(ctypes.byref(h), int(device))
        if rc != VDET_OK or not h.value:
            raise RuntimeError("vdet_create failed (rc=%d): no usable MI355X/HIP device. "
                               "vdetlib_amd has no CPU fallback." % rc)
        self.h = h

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.vdet_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        m = self.lib.vdet_last_error(self.h)
        return m.decode() if m else ""

    def check(self, rc):
        if rc == VDET_OK:
            return
        msg = self.error()
        if rc == VDET_EDIVZERO:
            raise ZeroDivisionError("float division")
        if rc in (VDET_EINVAL, VDET_ECAP):
            raise ValueError(msg or "invalid argument")
        if rc == VDET_ENOMEM:
            raise MemoryError(msg)
        if rc == VDET_EINDEX:
            raise IndexError(msg or "list index out of range")
        if rc == VDET_EAGAIN:
            raise RetryError(msg)
        raise RuntimeError("libvdet_hip: %s (rc=%d)" % (msg, rc))

    def set_stream(self, stream_ptr):
        self.check(self.lib.vdet_set_stream(self.h, _vp(stream_ptr or 0)))

    def reset_stream(self):
        self.check(self.lib.vdet_reset_stream(self.h))

    def sync(self):
        self.check(self.lib.vdet_sync(self.h))

    def set_cache(self, on):
        self.check(self.lib.vdet_set_cache(self.h, 1 if on else 0))

    def invalidate(self):
        self.check(self.lib.vdet_invalidate(self.h))

    def set_async(self, on):
        """No host synchronisation inside the volume entry points (include/vdet_hip.h: vdet_set_async)."""
        self.check(self.lib.vdet_set_async(self.h, 1 if on else 0))

    def query(self, what):
        return int(self.lib.vdet_query(self.h, int(what)))

    def set_timing(self, on):
        """False/0 off, True/1 on, 2 = accumulate over calls until last_timing() reads."""
        self.check(self.lib.vdet_set_timing(self.h, int(on)))

    def last_timing(self):
        ms = (ctypes.c_float * 16)()
        n = (ctypes.c_int * 16)()
        self.check(self.lib.vdet_last_timing_ms(self.h, ms))
        self.check(self.lib.vdet_last_launches(self.h, n))
        names = ["iou_bits", "adj_build", "sort", "walk", "temporal", "merge_sort", "iou_bits_general", "other",
                 "transpose_keys", "track_pick", "track_link", "track_suppress", "rescore_spatial", "rescore_series",
                 "sort_fallback", "track_loop"]
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}


def get_context(device=-1):
    """Process-wide context per device (created on first use)."""
    with _lock:
        c = _ctxs.get(device)
        if c is None:
            c = Context(device)
            _ctxs[device] = c
        return c
