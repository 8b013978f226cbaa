"""-m gpu: the single-launch drop-in calls (csrc/fused_kernels.hpp: nms / vid_nms / track_det_nms on <= 640 rows as ONE
workgroup) against the CPU oracle -- which the golden vectors recorded from the reference pin (tests/test_oracle_golden.py) --
on every size around the kernel's internal boundaries (32-row groups of the triangular matrix, the 64 / 128 / 256 / 512 sort
sizes, the 128-row block-size switch, the 640-row limit and the first sizes of the general chain behind it), with tied scores,
zero-area / NaN / inf boxes (ZeroDivisionError exactly where the oracle raises), NaN and signed-zero frames, caller orders,
strided inputs, and up to 256 track rows (+ the first count that leaves the single launch)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 511, 512, 513, 639, 640, 641, 700]


@pytest.fixture(scope="module")
def cnms():
    from vdetlib_amd.utils import cython_nms
    return cython_nms


def _same(fn_gpu, fn_cpu):
    """same list, or ZeroDivisionError on both sides"""
    try:
        want = fn_cpu()
    except ZeroDivisionError:
        with pytest.raises(ZeroDivisionError):
            fn_gpu()
        return None
    got = fn_gpu()
    assert got == want
    return got


@pytest.mark.parametrize("n", SIZES)
def test_nms_every_boundary_size(cnms, oracle, n):
    for seed, frac, kind in ((1, False, "perm"), (2, True, "randn")):
        d = synth.dets5(7000 + 13 * n + seed, n, frac=frac, kind=kind)
        for t in (0.3, 0.7):
            _same(lambda: cnms.nms(d, t), lambda: oracle.nms(d, t))
    # ties (quantised scores, signed zeros, a NaN score), dense boxes (long rows of the matrix), degenerate boxes
    rng = np.random.RandomState(n)
    d = synth.dets5(8000 + n, n, degenerate=n // 3)
    d[:, 4] = np.round(d[:, 4] * 6) / 6
    if n > 4:
        d[0, 4], d[1, 4], d[2, 4] = -0.0, 0.0, np.nan
    _same(lambda: cnms.nms(d, 0.3), lambda: oracle.nms(d, 0.3))
    x, y = rng.uniform(0, 60, n), rng.uniform(0, 40, n)
    dense = np.stack([x, y, x + rng.uniform(50, 150, n), y + rng.uniform(50, 150, n), rng.rand(n)], 1).astype(np.float32)
    for t in (0.05, 0.5, 0.95):
        _same(lambda: cnms.nms(dense, t), lambda: oracle.nms(dense, t))


@pytest.mark.parametrize("n", [5, 64, 129, 300, 640])
def test_nms_irregular_boxes_and_zero_unions(cnms, oracle, n):
    for seed in range(6):
        rng = np.random.RandomState(100 * n + seed)
        d = synth.dets5(9000 + 7 * n + seed, n, degenerate=n // 2)
        k = min(n, int(rng.randint(0, 5)))
        idx = rng.choice(n, k, replace=False)
        d[idx, 2] = d[idx, 0] - 1                                   # zero width under the +1 convention
        if seed % 2 and n > 3:
            d[rng.randint(n), rng.randint(4)] = np.nan
            d[rng.randint(n), rng.randint(4)] = np.inf if seed % 4 == 1 else -np.inf
        for t in (0.3, 0.0):
            _same(lambda: cnms.nms(d, t), lambda: oracle.nms(d, t))


@pytest.mark.parametrize("n", [1, 33, 64, 200, 513, 640, 641])
def test_vid_nms_frames(cnms, oracle, n):
    for nf in (1, 3, 40):
        d = synth.dets6(11000 + n + nf, n, nf, frac=bool(nf & 1))
        _same(lambda: cnms.vid_nms(d, 0.3), lambda: oracle.vid_nms(d, 0.3))
    d = synth.dets6(12000 + n, n, 4)
    if n > 12:
        d[3, 0] = np.nan; d[9, 0] = np.nan                         # NaN frames equal nothing, not even themselves
        d[11, 0], d[12, 0] = -0.0, 0.0                              # -0.0 == +0.0
        d[12, 1:5] = d[11, 1:5]
    d[:, 5] = np.round(d[:, 5] * 5) / 5                             # ties across and inside frames
    _same(lambda: cnms.vid_nms(d, 0.3), lambda: oracle.vid_nms(d, 0.3))
    # strided view and a caller's order
    wide = np.zeros((n, 9), np.float32)
    wide[:, 1:7] = d
    _same(lambda: cnms.vid_nms(wide[:, 1:7], 0.3), lambda: oracle.vid_nms(d, 0.3))
    order = np.random.RandomState(n).permutation(n).astype(np.int64)
    _same(lambda: cnms.vid_nms(d, 0.3, order=order), lambda: oracle.vid_nms(d, 0.3, order=order))


@pytest.mark.parametrize("m,t", [(1, 1), (40, 0), (300, 1), (300, 7), (640, 1), (640, 256), (640, 257), (641, 1), (97, 300)])
def test_track_det_nms_tracks_and_sizes(cnms, oracle, m, t):
    for nf in (1, 3):
        rng = np.random.RandomState(13 * m + t + nf)
        d = synth.dets6(14000 + m + t, m, nf)
        tr = np.zeros((t, 5), np.float32)
        if t:
            pick = rng.choice(m, t, replace=t > m)
            tr[:, 0] = rng.randint(1, nf + 1, t)
            tr[:, 1:5] = d[pick, 1:5] + rng.randint(-6, 7, (t, 4))
        _same(lambda: cnms.track_det_nms(tr, d, 0.3), lambda: oracle.track_det_nms(tr, d, 0.3))
    # a zero-area track box on the detections' frame: round 1 divides by a zero union iff a zero-area det meets it
    if t >= 1 and m >= 8:
        d = synth.dets6(15000 + m, m, 1)
        d[5, 1:5] = [10, 10, 9, 30]
        tr = np.array([[1, 10, 10, 9, 30]], np.float32)
        _same(lambda: cnms.track_det_nms(tr, d, 0.3), lambda: oracle.track_det_nms(tr, d, 0.3))


def _batch_case(rng, sizes, seed, nf=1):
    """K problems (the reference's per-tracked-box calls, vdet/track.py:236-250): problem k = one track row on frame k + 1 and
    the m_k detections of that frame"""
    dets, tracks = [], []
    for k, m in enumerate(sizes):
        d = synth.dets6(seed + 31 * k + m, m, 1) if m else np.zeros((0, 6), np.float32)
        d[:, 0] = k + 1
        dets.append(d)
        tr = np.zeros((1, 5), np.float32)
        tr[0, 0] = k + 1
        tr[0, 1:5] = (d[rng.randint(m), 1:5] + rng.randint(-6, 7, 4)) if m else [0, 0, 10, 10]
        tracks.append(tr)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return np.concatenate(tracks), np.concatenate(dets), off


@pytest.mark.parametrize("sizes", [[1], [0, 5, 0], [7, 128, 129, 64, 1, 300], [640, 639, 2], [100] * 37, [641, 10], [200, 0, 200, 33]])
def test_track_det_nms_batch_equals_one_call_per_problem(cnms, oracle, sizes):
    rng = np.random.RandomState(sum(sizes) + len(sizes))
    tracks, dets, off = _batch_case(rng, sizes, 16000)
    keep, counts = cnms.track_det_nms_batch(tracks, dets, off, 0.3)
    assert counts.shape == (len(sizes),)
    for k in range(len(sizes)):
        want = oracle.track_det_nms(tracks[k:k + 1], np.ascontiguousarray(dets[off[k]:off[k + 1]]), 0.3)
        assert keep[off[k]:off[k] + counts[k]].tolist() == want, k
        assert want == cnms.track_det_nms(tracks[k:k + 1], dets[off[k]:off[k + 1]], 0.3)
    # several track rows per problem (track_offsets), incl. a problem without any
    toff = np.asarray([0] + list(np.cumsum([k % 3 for k in range(len(sizes))])), np.int64)
    tr2 = np.zeros((int(toff[-1]), 5), np.float32)
    for k in range(len(sizes)):
        for q in range(int(toff[k]), int(toff[k + 1])):
            m = sizes[k]
            tr2[q, 0] = k + 1
            tr2[q, 1:5] = (dets[off[k] + rng.randint(m), 1:5] + rng.randint(-9, 10, 4)) if m else [1, 1, 5, 5]
    keep, counts = cnms.track_det_nms_batch(tr2, dets, off, 0.3, track_offsets=toff)
    for k in range(len(sizes)):
        want = oracle.track_det_nms(np.ascontiguousarray(tr2[toff[k]:toff[k + 1]]), np.ascontiguousarray(dets[off[k]:off[k + 1]]), 0.3)
        assert keep[off[k]:off[k] + counts[k]].tolist() == want, k


def test_track_det_nms_batch_errors(cnms, oracle):
    rng = np.random.RandomState(3)
    tracks, dets, off = _batch_case(rng, [20, 30, 25], 17000)
    # the reference's ZeroDivisionError from ANY problem of the batch
    dets[off[1] + 5, 1:5] = [10, 10, 9, 30]
    tracks[1, 1:5] = [10, 10, 9, 30]
    with pytest.raises(ZeroDivisionError):
        oracle.track_det_nms(tracks[1:2], np.ascontiguousarray(dets[off[1]:off[2]]), 0.3)
    with pytest.raises(ZeroDivisionError):
        cnms.track_det_nms_batch(tracks, dets, off, 0.3)
    with pytest.raises(ValueError):
        cnms.track_det_nms_batch(tracks, dets, [0, 20, 50], 0.3)              # offsets do not cover the rows
    with pytest.raises(ValueError):
        cnms.track_det_nms_batch(tracks[:2], dets, off, 0.3)                  # one track row per problem
    with pytest.raises(ValueError):
        cnms.track_det_nms_batch(tracks.astype(np.float64), dets, off, 0.3)
    keep, counts = cnms.track_det_nms_batch(np.zeros((0, 5), np.float32), np.zeros((0, 6), np.float32), [0], 0.3)
    assert len(counts) == 0
