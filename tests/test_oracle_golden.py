"""Pins the CPU oracle (oracle/) against golden vectors recorded from the reference itself
(tests/golden/make_golden.py).  CPU-only."""
import hashlib

import numpy as np
import pytest

import synth


def test_appendix_a_sanity_rows(oracle):
    """reproducible sanity values of the reference's nms."""
    rows = {300: (167, [6, 88, 263, 150, 190], '3bb82fa2e7de'),
            1000: (368, [273, 960, 255, 81, 414], '74069f537403'),
            10000: (1344, [1671, 3511, 1261, 9115, 9385], '2e76f9c2fc25')}
    for n, (ln, head, md5) in rows.items():
        rng = np.random.RandomState(n)
        x1 = rng.uniform(0, 1230, n); y1 = rng.uniform(0, 670, n)
        w = rng.uniform(10, 300, n); h = rng.uniform(10, 300, n)
        b = np.round(np.stack([x1, y1, np.minimum(x1 + w, 1279), np.minimum(y1 + h, 719)], 1))
        s = (rng.permutation(n) + 0.5) / n
        d = np.hstack([b, s[:, None]]).astype(np.float32)
        k = oracle.nms(d, 0.3)
        assert len(k) == ln and k[:5] == head
        assert hashlib.md5(np.asarray(k, dtype=np.int64).tobytes()).hexdigest()[:12] == md5


def test_nms_golden(oracle, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['nms']):
        d = synth.dets5(c['seed'], c['n'], c['frac'], c['degenerate'], c['kind'])
        if c['n'] == 0:
            d = np.zeros((0, 5), np.float32)
        assert oracle.nms(d, c['thresh']) == z['nms_%d' % i].tolist(), c


def test_vid_nms_golden(oracle, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['vid_nms']):
        d = synth.dets6(c['seed'], c['n'], c['n_frames'], c['frac'])
        if c['n'] == 0:
            d = np.zeros((0, 6), np.float32)
        assert oracle.vid_nms(d, c['thresh']) == z['vid_nms_%d' % i].tolist(), c


def test_track_det_nms_golden(oracle, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['track_det_nms']):
        d = synth.dets6(c['seed'], c['m'], c['n_frames'])
        if c['m'] == 0:
            d = np.zeros((0, 6), np.float32)
        rng = np.random.RandomState(c['seed'] + 1)
        tb = synth.boxes_1(rng, c['t'])
        tf = rng.randint(1, c['n_frames'] + 1, c['t']).astype(np.float32)
        tr = np.hstack([tf[:, None], tb]).astype(np.float32).reshape(-1, 5)
        assert oracle.track_det_nms(tr, d, c['thresh']) == z['tdn_%d' % i].tolist(), c


def check_exotic(mod, want_by_name):
    import warnings
    for c in synth.EXOTIC_CASES:
        args = synth.exotic_inputs(c)
        want = want_by_name[c['name']]
        fn = getattr(mod, c['fn'])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if want is ZeroDivisionError:
                with pytest.raises(ZeroDivisionError):
                    fn(*args, c['thresh'])
            else:
                assert fn(*args, c['thresh']) == want, c['name']


def test_exotic_inputs_golden(oracle, exotic_golden):
    """NaN / inf coordinates, a NaN score, zero-area boxes (ZeroDivisionError), NaN and negative thresholds,
    NaN frame ids, NaN track boxes: the oracle does what the reference did."""
    check_exotic(oracle, exotic_golden)


def test_iou_golden(oracle, nms_golden):
    z, index = nms_golden
    for i, c in enumerate(index['iou']):
        rng = np.random.RandomState(c['seed'])
        b1 = synth.boxes_1(rng, c['n1'], c['frac']).astype(np.float64)
        b2 = synth.boxes_1(rng, c['n2'], c['frac']).astype(np.float64)
        if i == 1:
            b2[:5] = b1[0]
        got = oracle.iou(b1, b2)
        assert got.dtype == np.float64 and np.array_equal(got, z['iou_%d' % i]), c


def test_ties_with_injected_order(oracle, nms_golden):
    """numpy's default argsort is unstable: with the reference's recorded order injected the
    oracle reproduces its keep list exactly; without it the build's documented tie rule
    (descending index among equals) applies."""
    z, index = nms_golden
    for i, c in enumerate(index['ties']):
        rng = np.random.RandomState(c['seed'])
        b = synth.boxes_1(rng, c['n'])
        s = (rng.randint(0, c['levels'], c['n']) / float(c['levels'])).astype(np.float32)
        d = np.hstack([b, s[:, None]]).astype(np.float32)
        order = z['ties_order_%d' % i].astype(np.int64)
        assert oracle.nms(d, c['thresh'], order=order) == z['ties_keep_%d' % i].tolist()
        stable = d[:, 4].argsort(kind='stable')[::-1]
        assert np.array_equal(oracle.argsort_desc(d[:, 4]), stable)
        assert oracle.nms(d, c['thresh']) == oracle.nms(d, c['thresh'], order=stable)


def test_zero_division(oracle):
    """utils/nms.pyx raises ZeroDivisionError on a zero union (two degenerate boxes x2 = x1-1)."""
    d = np.array([[10, 10, 9, 20, 0.9], [10, 10, 9, 20, 0.8]], np.float32)
    with pytest.raises(ZeroDivisionError):
        oracle.nms(d, 0.3)


def test_vid_nms_is_per_frame_nms(oracle):
    """The decomposition the GPU path relies on."""
    d = synth.dets6(77, 1500, 9, frac=True)
    ref = oracle.vid_nms(d, 0.3)
    kept = []
    for f in np.unique(d[:, 0]):
        ids = np.where(d[:, 0] == f)[0]
        kept.extend(ids[oracle.nms(d[ids][:, 1:], 0.3)].tolist())
    kept.sort(key=lambda i: -d[i, 5])
    assert kept == ref


def test_completion_golden(oracle, proto_golden):
    for k, c in proto_golden['completion'].items():
        assert oracle.score_completion(c['inp']).tolist() == c['out'], k
    with pytest.raises(IndexError):
        oracle.score_completion([-1e5, -1e5])


def test_temporal_maxpool_golden(oracle, proto_golden):
    g = proto_golden['temporal_maxpool_series']
    x = np.asarray(g['inp'], dtype=np.float64)
    for w in (3, 5, 9):
        got = oracle.temporal_maxpool(x.astype(np.float32)[:, None], w)[:, 0]
        assert np.array_equal(got, np.asarray(g['w%d' % w], dtype=np.float64).astype(np.float32))
    with pytest.raises(ValueError):
        oracle.temporal_maxpool(x[:, None], 4)


def test_interpolation_golden(oracle, proto_golden):
    for tag, c in proto_golden['interpolation'].items():
        tin, tout = c['inp']['tubelets'][0], c['out']['tubelets'][0]
        if len(tin['boxes']) < 2:
            assert tout['boxes'] == tin['boxes']
            continue
        frames = [b['frame'] for b in tin['boxes']]
        fields = [[b['bbox'][0], b['bbox'][1], b['bbox'][2], b['bbox'][3], b['det_score'], b['anchor']]
                  for b in tin['boxes']]
        dense, vals = oracle.tubelet_interpolation(frames, fields, 12)
        assert dense.tolist() == [b['frame'] for b in tout['boxes']], tag
        want = np.asarray([b['bbox'] + [b['det_score'], b['anchor']] for b in tout['boxes']])
        assert np.allclose(vals, want, rtol=0, atol=1e-9), tag
        assert np.array_equal(vals, want), tag
