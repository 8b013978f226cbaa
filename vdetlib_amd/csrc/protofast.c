/* protofast.c -- host-side marshalling of protocol dicts into the float32 rows the C-ABI takes (CPython extension
 * `vdetlib_amd._protofast`; plain C, no device code, no numerics beyond double -> float casts).
 *
 * The reference builds the [N,6] rows of `apply_vid_nms` with one python-level pass per detection and a `det_score` scan per
 * detection (vdet/video_det.py:54-56, utils/protocol.py:323-327): at BASELINE configs[0] that is 9 000 detections x 30 calls,
 * 20x the time of the GPU work behind it.  This module does the same dict reads with the C API -- same keys, same order, same
 * first-match rule, same exceptions (KeyError for a missing key, TypeError / ValueError from the float conversion) -- and
 * writes straight into the caller's float32 buffer.  `vdetlib_amd/vdet/video_det.py` falls back to its itertools form of the
 * same loop when this module is not built (identical rows: tests/test_protocol_cpu.py checks both against each other).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <math.h>

static PyObject *s_frame, *s_bbox, *s_scores, *s_score, *s_class_index;

static PyObject *dict_get(PyObject *d, PyObject *key)
{   /* d[key] with python's semantics for any mapping; borrowed reference for real dicts, new reference otherwise -> always new */
    if (PyDict_CheckExact(d)) {
        PyObject *v = PyDict_GetItemWithError(d, key);
        if (!v) {
            if (!PyErr_Occurred()) PyErr_SetObject(PyExc_KeyError, key);
            return NULL;
        }
        Py_INCREF(v);
        return v;
    }
    return PyObject_GetItem(d, key);
}

static int as_float(PyObject *o, float *out)
{
    double v = PyFloat_AsDouble(o);      /* ints, floats, numpy scalars (__float__ / __index__), like numpy's float64 conversion */
    if (v == -1.0 && PyErr_Occurred()) return -1;
    *out = (float)v;
    return 0;
}

/* vid_nms_rows(detections, class_index, out) -- out: writable C-contiguous float32 buffer of >= 6 * len(detections) items.
 * Row i = (frame, x1, y1, x2, y2, score of the FIRST entry of detections[i]['scores'] whose 'class_index' == class_index,
 * -inf when there is none). */
static PyObject *vid_nms_rows(PyObject *self, PyObject *args)
{
    PyObject *dets, *cls, *outobj;
    Py_buffer view;
    (void)self;
    if (!PyArg_ParseTuple(args, "OOO", &dets, &cls, &outobj)) return NULL;
    PyObject *seq = PySequence_Fast(dets, "detections must be a sequence");
    if (!seq) return NULL;
    if (PyObject_GetBuffer(outobj, &view, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) < 0) {
        Py_DECREF(seq);
        return NULL;
    }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    int ok = 0;
    if (view.itemsize != 4 || !view.format || view.format[0] != 'f' || view.len < (Py_ssize_t)(n * 6 * 4)) {
        PyErr_SetString(PyExc_ValueError, "out must be a C-contiguous float32 buffer of at least 6 * len(detections) items");
        goto done;
    }
    float *rows = (float *)view.buf;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *d = PySequence_Fast_GET_ITEM(seq, i);      /* borrowed */
        float *row = rows + 6 * i;
        PyObject *v = dict_get(d, s_frame);
        if (!v) goto done;
        int rc = as_float(v, &row[0]);
        Py_DECREF(v);
        if (rc) goto done;
        v = dict_get(d, s_bbox);
        if (!v) goto done;
        PyObject *bb = PySequence_Fast(v, "bbox must be a sequence of four numbers");
        Py_DECREF(v);
        if (!bb) goto done;
        if (PySequence_Fast_GET_SIZE(bb) != 4) {
            Py_DECREF(bb);
            PyErr_SetString(PyExc_ValueError, "bbox must hold four numbers");
            goto done;
        }
        for (int k = 0; k < 4; ++k)
            if (as_float(PySequence_Fast_GET_ITEM(bb, k), &row[1 + k])) { Py_DECREF(bb); goto done; }
        Py_DECREF(bb);
        v = dict_get(d, s_scores);
        if (!v) goto done;
        PyObject *sc = PySequence_Fast(v, "scores must be a sequence");
        Py_DECREF(v);
        if (!sc) goto done;
        row[5] = -INFINITY;
        const Py_ssize_t m = PySequence_Fast_GET_SIZE(sc);
        PyObject **ent = PySequence_Fast_ITEMS(sc);
        for (Py_ssize_t j = 0; j < m; ++j) {
            PyObject *e = ent[j];
            /* the entry dicts of a det_proto are scattered over the heap (280 000 of them at configs[0]): the scan is a chain of
             * cache misses -- request the dict object 16 entries ahead and the key table of the one 8 ahead */
            if (j + 16 < m) __builtin_prefetch(ent[j + 16]);
            if (j + 8 < m && PyDict_CheckExact(ent[j + 8])) {
                const char *kt = (const char *)((PyDictObject *)ent[j + 8])->ma_keys;
                __builtin_prefetch(kt);
                __builtin_prefetch(kt + 64);
                __builtin_prefetch(kt + 128);
            }
            PyObject *ci = dict_get(e, s_class_index);
            if (!ci) { Py_DECREF(sc); goto done; }
            const int eq = (ci == cls) ? 1 : PyObject_RichCompareBool(ci, cls, Py_EQ);
            Py_DECREF(ci);
            if (eq < 0) { Py_DECREF(sc); goto done; }
            if (eq) {
                PyObject *val = dict_get(e, s_score);
                if (!val) { Py_DECREF(sc); goto done; }
                rc = as_float(val, &row[5]);
                Py_DECREF(val);
                if (rc) { Py_DECREF(sc); goto done; }
                break;
            }
        }
        Py_DECREF(sc);
    }
    ok = 1;
done:
    PyBuffer_Release(&view);
    Py_DECREF(seq);
    if (!ok) return NULL;
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"vid_nms_rows", vid_nms_rows, METH_VARARGS,
     "vid_nms_rows(detections, class_index, out): fill float32 [N,6] rows (frame, bbox, det_score of class_index)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_protofast", "protocol dicts -> float32 rows (host marshalling)", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__protofast(void)
{
    s_frame = PyUnicode_InternFromString("frame");
    s_bbox = PyUnicode_InternFromString("bbox");
    s_scores = PyUnicode_InternFromString("scores");
    s_score = PyUnicode_InternFromString("score");
    s_class_index = PyUnicode_InternFromString("class_index");
    if (!s_frame || !s_bbox || !s_scores || !s_score || !s_class_index) return NULL;
    return PyModule_Create(&module);
}
