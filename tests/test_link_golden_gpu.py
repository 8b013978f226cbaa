"""-m gpu: the benchmarked LINK + re-scoring stage (vdet_track_volume / vdet_nms_track_volume +
vdet_rescore_tracks) against outputs of THE REFERENCE (tests/golden/link_golden.npz, see
tests/test_link_golden_cpu.py): greedily_track_from_raw_dets, raw_dets_spatial_max_pooling,
do_score_completion, score_proto_temporal_maxpool."""
import os

import numpy as np
import pytest

import synth
from conftest import GOLDEN
from test_link_golden_cpu import check_against_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def link_golden():
    return np.load(os.path.join(GOLDEN, 'link_golden.npz'))


@pytest.mark.parametrize("cached", [False, True], ids=["plain", "cached"])
@pytest.mark.parametrize("fused", [False, True], ids=["track_volume", "nms_track_volume"])
@pytest.mark.parametrize("case", synth.LINK_CASES, ids=[c['name'] for c in synth.LINK_CASES])
def test_device_link_chain_vs_reference(link_golden, case, fused, cached):
    """cached: a context with the cache on -- the re-scoring then takes its candidates from the suppression graph
    (the neighbours of the proposal each tubelet box came from) instead of scanning the frame's x-window."""
    import torch
    from vdetlib_amd import ops, _lib
    boxes, scores = synth.link_case_video(case)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    kw = dict(nms_thres=case['nms_thres'], thres=case['thres'], max_tracks=case['max_tracks'], link_thres=case['link'],
              max_frames=case['max_frames'])
    if cached:
        kw['ctx'] = _lib.Context(torch.cuda.current_device())
        kw['ctx'].set_cache(True)
    if fused:
        _, _, tr, an, nt = ops.nms_track_volume(tb, ts, **kw)
    else:
        tr, an, nt = ops.track_volume(tb, ts, **kw)
    det, pooled, ob = ops.rescore_tracks(tr, nt, tb, ts, overlap_thres=case['pool'], window=case['window'], ctx=kw.get('ctx'))
    check_against_golden(link_golden, case, tr.cpu().numpy(), nt.cpu().numpy(), an.cpu().numpy(), det.cpu().numpy(),
                         pooled.cpu().numpy(), ob.cpu().numpy())
