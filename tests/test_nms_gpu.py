"""-m gpu parity tests: the HIP path (through the C-ABI / the cython_nms drop-in) against the
golden vectors recorded from the reference and against the CPU oracle on seeded inputs.
Bar: bit-exact keep indices."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnms1():
    from vdetlib_amd.utils import cython_nms
    return cython_nms


@pytest.fixture(scope="module", params=["single_launch", "general"])
def cnms(request):
    """the drop-in module twice: calls of <= 640 rows as ONE launch (csrc/fused_kernels.hpp, the default) and -- on a
    context created under VDET_NO_FUSED=1 -- through the general kernel chain that larger inputs take"""
    import os
    from vdetlib_amd.utils import cython_nms
    from vdetlib_amd import _lib
    if request.param == "single_launch":
        yield cython_nms
        return
    os.environ["VDET_NO_FUSED"] = "1"
    try:
        cx = _lib.Context(-1)
    finally:
        del os.environ["VDET_NO_FUSED"]
    old = _lib._ctxs.get(-1)
    _lib._ctxs[-1] = cx
    yield cython_nms
    if old is None:
        del _lib._ctxs[-1]
    else:
        _lib._ctxs[-1] = old
    cx.close()


def test_nms_golden(cnms, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['nms']):
        d = synth.dets5(c['seed'], c['n'], c['frac'], c['degenerate'], c['kind'])
        if c['n'] == 0:
            d = np.zeros((0, 5), np.float32)
        got = cnms.nms(d, c['thresh'])
        assert isinstance(got, list) and all(type(k) is int for k in got[:3])
        assert got == z['nms_%d' % i].tolist(), c


def test_vid_nms_golden(cnms, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['vid_nms']):
        d = synth.dets6(c['seed'], c['n'], c['n_frames'], c['frac'])
        if c['n'] == 0:
            d = np.zeros((0, 6), np.float32)
        assert cnms.vid_nms(d, thresh=c['thresh']) == z['vid_nms_%d' % i].tolist(), c


def test_track_det_nms_golden(cnms, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['track_det_nms']):
        d = synth.dets6(c['seed'], c['m'], c['n_frames'])
        if c['m'] == 0:
            d = np.zeros((0, 6), np.float32)
        rng = np.random.RandomState(c['seed'] + 1)
        tb = synth.boxes_1(rng, c['t'])
        tf = rng.randint(1, c['n_frames'] + 1, c['t']).astype(np.float32)
        tr = np.hstack([tf[:, None], tb]).astype(np.float32).reshape(-1, 5)
        assert cnms.track_det_nms(tr, d, c['thresh']) == z['tdn_%d' % i].tolist(), c


def test_ties_injected_order_and_default_rule(cnms, oracle, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['ties']):
        rng = np.random.RandomState(c['seed'])
        b = synth.boxes_1(rng, c['n'])
        s = (rng.randint(0, c['levels'], c['n']) / float(c['levels'])).astype(np.float32)
        d = np.hstack([b, s[:, None]]).astype(np.float32)
        order = z['ties_order_%d' % i].astype(np.int64)
        assert cnms.nms(d, c['thresh'], order=order) == z['ties_keep_%d' % i].tolist()
        assert cnms.nms(d, c['thresh']) == oracle.nms(d, c['thresh'])


@pytest.mark.parametrize("seed", range(12))
def test_nms_random_vs_oracle(cnms, oracle, seed):
    rng = np.random.RandomState(9000 + seed)
    n = int(rng.choice([5, 63, 64, 65, 200, 777, 1500, 4097]))
    d = synth.dets5(seed, n, frac=bool(seed & 1), degenerate=(n // 2 if seed % 3 == 0 else 0),
                    kind='randn' if seed % 2 else 'perm')
    if seed % 4 == 0:      # ties + signed zeros + a NaN score
        d[:, 4] = np.round(d[:, 4] * 8) / 8
        d[0, 4] = -0.0
        d[1, 4] = 0.0
        if n > 10:
            d[7, 4] = np.nan
    for thresh in (0.3, 0.5, 0.75):
        assert cnms.nms(d, thresh) == oracle.nms(d, thresh), (seed, n, thresh)


def test_exotic_inputs_golden(cnms, exotic_golden):
    """The device path on NaN / inf / zero-area inputs and odd thresholds, against what the REFERENCE returned."""
    from test_oracle_golden import check_exotic
    check_exotic(cnms, exotic_golden)


def test_nms_nan_and_inf_coordinates(cnms, oracle):
    d = synth.dets5(31337, 300, degenerate=200)
    d[5, 0] = np.nan
    d[17, 2] = np.nan
    d[40, 1] = np.inf
    d[41, 3] = -np.inf
    d[100:104, :4] = np.nan
    for thresh in (0.3, 0.6):
        assert cnms.nms(d, thresh) == oracle.nms(d, thresh)


def test_zero_division_rule(cnms, oracle):
    """ZeroDivisionError exactly where the reference raises (only for EVALUATED zero-union pairs)."""
    d = np.array([[10, 10, 9, 20, 0.9], [10, 10, 9, 20, 0.8]], np.float32)
    with pytest.raises(ZeroDivisionError):
        cnms.nms(d, 0.3)
    # a zero-union pair that is never evaluated: box 2 (degenerate) is suppressed ... it cannot be,
    # its IoU is 0; so build: degenerate A (score .5), degenerate B (score .4) -> evaluated -> raises
    d2 = np.array([[0, 0, 50, 50, 0.9], [100, 100, 99, 120, 0.5], [100, 100, 99, 120, 0.4]], np.float32)
    for dd in (d2,):
        try:
            want = oracle.nms(dd, 0.3)
            assert cnms.nms(dd, 0.3) == want
        except ZeroDivisionError:
            with pytest.raises(ZeroDivisionError):
                cnms.nms(dd, 0.3)
    # randomized: sprinkle zero-area boxes, compare outcome (list or exception) with the oracle
    for seed in range(20):
        rng = np.random.RandomState(500 + seed)
        dd = synth.dets5(600 + seed, 120, degenerate=60)
        k = rng.randint(0, 4)
        idx = rng.choice(120, k, replace=False)
        dd[idx, 2] = dd[idx, 0] - 1          # zero width (+1 convention)
        try:
            want = oracle.nms(dd, 0.3)
        except ZeroDivisionError:
            with pytest.raises(ZeroDivisionError):
                cnms.nms(dd, 0.3)
        else:
            assert cnms.nms(dd, 0.3) == want


def test_errors_and_views(cnms):
    d = synth.dets5(1, 50)
    with pytest.raises(ValueError):
        cnms.nms(d.astype(np.float64), 0.3)
    with pytest.raises(ValueError):
        cnms.nms(d[0], 0.3)
    with pytest.raises(TypeError):
        cnms.nms(d, None)
    with pytest.raises(TypeError):
        cnms.nms(d.tolist(), 0.3)
    assert cnms.nms(np.zeros((0, 5), np.float32), 0.3) == []
    # non-contiguous views (the Cython buffer interface honours strides)
    big = np.zeros((100, 9), np.float32)
    big[::2, 2:7] = d
    assert cnms.nms(big[::2, 2:7], 0.3) == cnms.nms(d, 0.3)
    wide = np.hstack([d, np.ones((50, 3), np.float32)])
    assert cnms.nms(wide, 0.3) == cnms.nms(d, 0.3)
    assert cnms.nms(d, 1) == cnms.nms(d, 1.0)
    assert cnms.nms(d, np.float32(0.3)) == cnms.nms(d, float(np.float32(0.3)))


def test_vid_nms_many_frames_and_nan_frames(cnms, oracle):
    d = synth.dets6(4321, 2500, 40, frac=True)
    assert cnms.vid_nms(d, 0.3) == oracle.vid_nms(d, 0.3)
    d[3, 0] = np.nan
    d[9, 0] = np.nan
    d[11, 0] = -0.0
    d[12, 0] = 0.0
    d[12, 1:5] = d[11, 1:5]
    assert cnms.vid_nms(d, 0.3) == oracle.vid_nms(d, 0.3)
    # every detection on its own frame: nothing can be suppressed
    d2 = synth.dets6(5, 300, 1)
    d2[:, 0] = np.arange(300)
    assert cnms.vid_nms(d2, 0.3) == oracle.vid_nms(d2, 0.3) == np.argsort(-d2[:, 5], kind='stable').tolist()


def test_track_det_nms_random(cnms, oracle):
    for seed in range(6):
        rng = np.random.RandomState(seed)
        m, t, nf = int(rng.choice([10, 300, 1200])), int(rng.choice([1, 1, 4])), int(rng.choice([1, 3]))
        d = synth.dets6(700 + seed, m, nf)
        tr = np.hstack([rng.randint(1, nf + 1, (t, 1)), d[rng.choice(m, t), 1:5] + rng.randint(-4, 5, (t, 4))]).astype(np.float32)
        assert cnms.track_det_nms(tr, d, 0.3) == oracle.track_det_nms(tr, d, 0.3)
    assert cnms.track_det_nms(np.zeros((0, 5), np.float32), d, 0.3) == oracle.vid_nms(d, 0.3)


def test_batched_graph_build(cnms1, oracle, monkeypatch):
    """Same results when the bit-matrix scratch is split into many batches."""
    import ctypes
    from vdetlib_amd import _lib
    monkeypatch.setenv("VDET_BITS_BUDGET_MB", "1")
    ctx = _lib.Context()
    d = synth.dets6(888, 6000, 25)
    keep = np.empty(len(d), np.int64)
    nk = ctypes.c_int64(0)
    ctx.check(ctx.lib.vdet_nms_f32(ctx.h, d.ctypes.data, len(d), 6, 6, 0.3, None, keep.ctypes.data, ctypes.byref(nk)))
    assert keep[:nk.value].tolist() == oracle.vid_nms(d, 0.3)
    ctx.close()


def test_nms_size_limits(cnms1):
    d = np.zeros((40000, 5), np.float32)
    with pytest.raises(ValueError):
        cnms1.nms(d, 0.3)


def test_iou_f64(oracle, nms_golden):
    from vdetlib_amd import ops
    z, index = nms_golden
    for i, c in enumerate(index['iou']):
        rng = np.random.RandomState(c['seed'])
        b1 = synth.boxes_1(rng, c['n1'], c['frac']).astype(np.float64)
        b2 = synth.boxes_1(rng, c['n2'], c['frac']).astype(np.float64)
        if i == 1:
            b2[:5] = b1[0]
        got = ops.iou(b1, b2)
        assert got.dtype == np.float64 and np.array_equal(got, z['iou_%d' % i])
    rng = np.random.RandomState(1)
    b1 = rng.uniform(0, 500, (37, 4)); b2 = rng.uniform(0, 500, (1000, 4))
    b2[3, 1] = np.nan
    with np.errstate(all='ignore'):
        assert np.array_equal(ops.iou(b1, b2), oracle.iou(b1, b2), equal_nan=True)
    assert np.array_equal(ops.iou([[1, 2, 3, 4]], [[1, 2, 3, 4]]), [[1.0]])


@pytest.mark.parametrize("n", [17000, 20000, 32767])
def test_nms_large_single_problem(cnms1, oracle, n):
    cnms = cnms1
    """Beyond the in-LDS sort (~18k boxes) the host entry points fall back to a global bitonic sort;
    the u16 index limit is 32767."""
    d = synth.dets5(5000 + n, n, degenerate=n // 3, kind='randn')
    assert cnms.nms(d, 0.4) == oracle.nms(d, 0.4)
    if n == 20000:
        d6 = np.hstack([np.ones((n, 1), np.float32), d])
        d6[::3, 0] = 2
        assert cnms.vid_nms(d6, 0.3) == oracle.vid_nms(d6, 0.3)
    with pytest.raises(ValueError):
        cnms.nms(np.zeros((32768, 5), np.float32), 0.3)
