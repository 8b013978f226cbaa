"""-m gpu: BASELINE configs[0] end to end through the reference's import names (`vdetlib.*`, served by the build) against
what the REFERENCE returned for the same seeded inputs (tests/golden/c1_flow_golden.json.gz, recorded by
tests/golden/make_golden.py --c1-only with tests/c1_flow.py -- the one driver both sides run): the kept lists of 900
apply_image_nms calls, 30 apply_vid_nms calls on 9 000 detections, the tubelets of 30 x greedily_track_from_raw_dets (one
track_det_nms per tracked box), raw_dets_spatial_max_pooling + score_proto_temporal_maxpool.  Indices / boxes / hashes
exact, float scores within 1e-5 (north_star)."""
import gzip
import json
import os

import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'c1_flow_golden.json.gz')


def build_modules():
    from vdetlib.vdet import video_det as V, image_det as I, track as K, tubelet_cls as T
    from vdetlib.utils import protocol as P, common as Cm
    return dict(V=V, I=I, K=K, T=T, P=P, Cm=Cm)


def test_c1_flow_matches_the_reference():
    import c1_flow
    with gzip.open(GOLDEN, 'rt') as f:
        want = json.load(f)
    sec, got = c1_flow.run(build_modules())
    assert c1_flow.compare(got, want, tol=1e-5) == []
    assert got['methods'] == want['methods']
