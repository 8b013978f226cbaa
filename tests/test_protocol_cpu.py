"""CPU-only: the protocol-dict host logic (vdetlib_amd.utils.protocol / common) against golden
outputs of the reference (tests/golden/proto_golden.json.gz)."""
import copy
import gzip
import json
import os

import numpy as np
import pytest

import synth
from vdetlib_amd.utils import protocol as P
from vdetlib_amd.utils import common as Cm
from vdetlib_amd.utils.timer import Timer


def test_constructors_and_accessors(proto_golden):
    g = proto_golden['protocol_misc']
    case = synth.proto_case()
    vid, det = case['vid'], case['det']
    assert P.score_proto(synth.CLS5, np.asarray([0.1, 0.2, 0.3, 0.4, 0.5])) == g['score_proto']
    assert P.boxes_proto_from_boxes([1, 2], [[[1, 2, 3, 4], [5, 6, 7, 8]], [[9, 9, 20, 20]]], 'vv') == g['boxes_proto_from_boxes']
    rows = np.asarray([[1, 2, 30.7, 40.2, 0.9], [2, 3, 31, 41, 0.8], [np.nan] * 5, [4, 5, 33, 43, 0.6],
                       [np.nan] * 5, [np.nan] * 5, [7, 8, 36, 46, 0.3]])
    assert P.tracks_proto_from_boxes(rows, 'vv', 5, 3, 2) == g['tracks_proto_from_boxes']
    assert P.track_proto_from_annot_proto(copy.deepcopy(g['annot'])) == g['track_proto_from_annot_proto']
    assert P.sample_vid_proto(synth.make_vid_proto('s', 25), 10) == g['sample_vid_proto']
    assert P.empty_det_from_box(synth.make_box_proto(1101, 'e', 2, 3)) == g['empty_det_from_box']
    assert P.frame_path_at(vid, 3) == g['frame_paths']['at']
    assert P.frame_path_before(vid, 3) == g['frame_paths']['before']
    assert P.frame_path_after(vid, 5) == g['frame_paths']['after']
    assert repr(P.det_score(det['detections'][0], 99)) == g['det_score_missing'] == '-inf'
    assert [d['hash'] for d in P.top_detections(det, 7, 2)['detections']] == g['top_detections']
    assert sorted(d['hash'] for d in P.frame_top_detections(det, 3, 1)['detections']) == g['frame_top_detections']
    with pytest.raises(IndexError):
        P.frame_path_at(vid, 99)
    # hashes of the synthetic protos are the reference's bbox_hash
    b = det['detections'][5]
    assert P.bbox_hash(case['name'], b['frame'], b['bbox']) == b['hash']


def test_tubelets_from_tracks_and_merge(proto_golden):
    g = proto_golden['protocol_misc']
    tracks = proto_golden['greedy_track']['plain_det_c1']['tracks']
    assert P.tubelets_proto_from_tracks_proto(copy.deepcopy(tracks), 1) == g['tubelets_proto_from_tracks_proto']
    a = copy.deepcopy(proto_golden['spatial_maxpool']['dets_c1_0.7'])
    # golden 'w3' was computed on a deep copy of 'dets_c1_0.7'
    b = copy.deepcopy(proto_golden['temporal_maxpool']['w3'])
    assert P.merge_score_protos(copy.deepcopy(a), copy.deepcopy(b), 'max') == g['merge_max']
    a2 = copy.deepcopy(a)
    merged = P.merge_score_protos(a2, copy.deepcopy(b), 'combine')
    assert merged == g['merge_combine']
    assert len(a2['tubelets']) == 2 * len(a['tubelets'])      # 'combine' extends proto_1 in place
    with pytest.raises(AssertionError):
        P.merge_score_protos(a, b, 'sum')


def test_proto_io_roundtrip(tmp_path, proto_golden):
    g = proto_golden['proto_io']
    obj = g['obj']
    p1 = str(tmp_path / 'a.det')
    P.proto_dump(obj, p1)
    text = open(p1).read()
    assert text == json.dumps(obj, indent=2)          # same serialisation as the reference (indent=2)
    assert text.startswith('{\n  "') and g['text_head'].startswith('{\n  "')
    assert P.proto_load(p1) == obj
    p2 = str(tmp_path / 'b.det.gz')
    P.proto_dump(obj, p2)
    with gzip.GzipFile(p2) as f:
        assert json.loads(f.read().decode()) == obj
    # a sibling .gz wins silently (utils/protocol.py:212-213)
    P.proto_dump({'video': 'plain'}, str(tmp_path / 'b.det'))
    assert P.proto_load(str(tmp_path / 'b.det')) == obj
    assert g['load_plain_eq'] and g['load_gz_pref_eq']


def test_vid_proto_from_dir(tmp_path):
    d = tmp_path / 'ILSVRC2015_val_00007'
    d.mkdir()
    for n in ('10.JPEG', '9.JPEG', '000100.jpg', 'notes.txt', '2.png'):
        (d / n).write_text('x')
    v = P.vid_proto_from_dir(str(d))
    assert v['video'] == 'ILSVRC2015_val_00007' and v['root_path'] == str(d)
    assert [f['path'] for f in v['frames']] == ['2.png', '9.JPEG', '10.JPEG', '000100.jpg']
    assert [f['frame'] for f in v['frames']] == [1, 2, 3, 4]
    assert P.vid_proto_from_dir(str(d), 'named')['video'] == 'named'
    assert P.path_to_index(v, '10') == 3 and P.path_to_index(v, 'zzz') is None


def test_common_helpers(tmp_path):
    o = Cm.options({'max_tracks': 5, 'thres': 0.2, 'nested': {'a': 1}})
    assert o.max_tracks == 5 and o['thres'] == 0.2 and o.nested.a == 1
    assert hasattr(o, 'thres') and not hasattr(o, 'nms_thres')
    o.nms_thres = 0.4
    assert o['nms_thres'] == 0.4
    l = ['f10.jpg', 'f9.jpg', 'f100.jpg', 'a1b20', 'a1b3']
    Cm.sort_nicely(l)
    assert l == ['a1b3', 'a1b20', 'f9.jpg', 'f10.jpg', 'f100.jpg']
    assert Cm.stem('/x/y/z.tar.gz') == 'z.tar' and Cm.isimg('A.JPEG') and not Cm.isimg('a.gif')
    p = str(tmp_path / 'l.txt')
    Cm.write_list([1, 'two', 3.5], p)
    assert open(p).read() == '1\ntwo\n3.5' and Cm.read_list(p) == ['1', 'two', '3.5']
    f = Cm.temp_file(suffix='.mat')
    assert f.endswith('.mat') and os.path.isfile(f)
    os.remove(f)
    Cm.pickle({'a': [1, 2]}, str(tmp_path / 'p.pkl'))
    assert Cm.unpickle(str(tmp_path / 'p.pkl')) == {'a': [1, 2]}


def test_timer():
    t = Timer()
    t.tic(); a = t.toc()
    t.tic(); d = t.toc(average=False)
    assert t.calls == 2 and abs(t.average_time - t.total_time / 2) < 1e-12 and a >= 0 and d == t.diff


def test_dataset_tables():
    from vdetlib_amd.vdet import dataset as D
    assert len(D.imagenet_vdet_classes) == 31 and D.imagenet_vdet_classes[0] == '__background__'
    assert len(D.imagenet_det_200_classes) == 201
    assert D.imagenet_vdet_class_idx['airplane'] == 1 and D.index_vdet_to_det[0] == 0
    for v, d in D.index_vdet_to_det.items():
        assert D.imagenet_vdet_classes[v] == D.imagenet_det_200_classes[d]


def test_track_plugins_are_external():
    from vdetlib_amd.vdet import track
    with pytest.raises(RuntimeError):
        track.fcn_tracker(None, 1, [0, 0, 1, 1], None)
    assert track.fcn_tracker.__name__ == 'fcn_tracker'


def test_timer_contract():
    """utils/timer.py:10-32: attributes and return values of tic / toc"""
    import time
    from vdetlib_amd.utils.timer import Timer
    t = Timer()
    assert (t.total_time, t.calls, t.start_time, t.diff, t.average_time) == (0.0, 0, 0.0, 0.0, 0.0)
    t.tic(); time.sleep(0.01)
    a = t.toc()
    assert t.calls == 1 and a == t.average_time == t.total_time == t.diff and a >= 0.009
    t.tic()
    d = t.toc(average=False)
    assert t.calls == 2 and d == t.diff and abs(t.average_time - t.total_time / 2) < 1e-12
