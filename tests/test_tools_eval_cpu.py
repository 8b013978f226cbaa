"""CPU-only: the 'next rows' of SURVEY 8f -- VID annotation tooling, array transport, AP evaluator."""
import json
import os

import numpy as np
import pytest

import synth
from vdetlib_amd import io as vio
from vdetlib_amd import eval as vev
from vdetlib_amd.tools import imagenet_annotation_processor as iap
from vdetlib_amd.utils import protocol as P

XML = """<annotation><folder>v</folder><filename>{fn}</filename><source><database>ILSVRC_2015</database></source>
<size><width>1280</width><height>720</height></size>{objs}</annotation>"""
OBJ = """<object><trackid>{tid}</trackid><name>{wnid}</name><bndbox><xmax>{x2}</xmax><xmin>{x1}</xmin>
<ymax>{y2}</ymax><ymin>{y1}</ymin></bndbox><occluded>{occ}</occluded><generated>0</generated></object>"""


def test_annotation_processor(tmp_path):
    d = tmp_path / 'ILSVRC2015_val_00000001'
    d.mkdir()
    for f in range(3):
        objs = OBJ.format(tid=0, wnid='n02691156', x1=10 + f, y1=20, x2=110 + f, y2=140, occ=0)
        if f >= 1:
            objs += OBJ.format(tid=1, wnid='n02391049', x1=300, y1=200, x2=420, y2=330, occ=1)
        (d / ('%06d.xml' % f)).write_text(XML.format(fn='%06d' % f, objs=objs))
    (d / '000003.xml').write_text(XML.format(fn='000003', objs=''))       # a frame without objects
    a = iap.annot_proto_from_dir(str(d))
    assert a['video'] == 'ILSVRC2015_val_00000001' and [t['id'] for t in a['annotations']] == ['0', '1']
    t0, t1 = a['annotations']
    assert [b['frame'] for b in t0['track']] == [1, 2, 3] and [b['frame'] for b in t1['track']] == [2, 3]
    assert t0['track'][2] == {'frame': 3, 'bbox': [12, 20, 112, 140], 'name': 'n02691156', 'class': 'airplane',
                              'class_index': 1, 'generated': 0, 'occluded': 0, 'frame_size': [720, 1280]}
    assert t1['track'][0]['class'] == 'zebra' and t1['track'][0]['class_index'] == 30 and t1['track'][0]['occluded'] == 1
    assert len(iap.name_map) == 30 and iap.name_map['n02084071'] == (9, 'dog')
    out = tmp_path / 'o' / 'a.annot'
    assert iap.main([str(d), str(out)]) == 0 and json.load(open(out)) == a
    assert iap.main([str(d), str(out)]) == 0                              # exists -> early exit
    # the gt tracks feed the rest of the API
    tp = P.track_proto_from_annot_proto(a)
    assert tp['method'] == 'gt' and len(tp['tracks']) == 2


def test_array_transport_roundtrip(tmp_path):
    case = synth.proto_case()
    vid, f2d = case['vid'], case['frame_to_det']
    boxes, scores, counts = vio.arrays_from_frame_to_det(vid, f2d)
    assert boxes.shape == (6, 40, 4) and scores.shape == (6, 40, 4) and counts.tolist() == [40, 40, 40, 0, 0, 40]
    assert np.isneginf(scores[3]).all() and (boxes[3, :, 0] < -1e5).all()        # padded frames
    back = vio.frame_to_det_from_arrays(vid, boxes, scores, counts)
    assert sorted(back) == [1, 2, 3, 6]
    for f in back:
        assert np.array_equal(back[f][0], f2d[f][0].astype(np.float32)) and np.array_equal(back[f][1], f2d[f][1])
    di = vio.det_info_from_arrays(vid, boxes, scores, counts)
    assert di.shape == (160, 9) and di[0, 0] == 1 and di[-1, 0] == 6
    vio.save_video_npz(str(tmp_path / 'v.npz'), boxes, scores, counts)
    b2, s2, c2 = vio.load_video_npz(str(tmp_path / 'v.npz'))
    assert np.array_equal(b2, boxes) and np.array_equal(s2, scores) and np.array_equal(c2, counts)
    pads = vio.pad_boxes(5)
    assert ((pads[:, 2] - pads[:, 0] + 1) == 1).all() and len(set(pads[:, 0])) == 5


def test_average_precision_evaluator():
    annot = {'video': 'v', 'annotations': [
        {'id': '0', 'track': [{'frame': f, 'bbox': [10, 10, 60, 60], 'class_index': 1} for f in (1, 2)]},
        {'id': '1', 'track': [{'frame': 1, 'bbox': [200, 200, 260, 280], 'class_index': 2}]}]}
    gt = vev.ground_truth_from_annots([annot])
    perfect = [('v', 1, 1, [10, 10, 60, 60], .9), ('v', 2, 1, [11, 10, 60, 60], .8), ('v', 1, 2, [200, 200, 260, 280], .7)]
    aps, m = vev.evaluate(perfect, gt)
    assert aps == {1: 1.0, 2: 1.0} and m == 1.0
    # class 1: TP(.9), FP(.85, duplicate of the same gt), TP(.8): precision envelope 1, 2/3 -> AP = .5*1 + .5*(2/3)
    dets = [('v', 1, 1, [10, 10, 60, 60], .9), ('v', 1, 1, [12, 10, 60, 60], .85), ('v', 2, 1, [11, 10, 60, 60], .8),
            ('v', 1, 2, [0, 0, 5, 5], .99)]
    aps, m = vev.evaluate(dets, gt)
    assert abs(aps[1] - (0.5 + 0.5 * 2 / 3)) < 1e-12 and aps[2] == 0.0 and abs(m - (aps[1] / 2)) < 1e-12
    assert np.isnan(vev.average_precision([], 0))
    sp = {'video': 'v', 'tubelets': [{'class_index': 1, 'boxes': [{'frame': 1, 'bbox': [10, 10, 60, 60], 'det_score': .5}]}]}
    assert vev.detections_from_score_protos([sp]) == [('v', 1, 1, [10, 10, 60, 60], .5)]


def _pb_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_len(fn, payload):
    return _pb_varint((fn << 3) | 2) + _pb_varint(len(payload)) + payload


def _pb_blob(arr, legacy=False):
    arr = np.asarray(arr, dtype='<f4')
    data = _pb_len(5, arr.tobytes())
    if legacy:
        dims = list(arr.shape) + [1] * (4 - arr.ndim)
        head = b''.join(_pb_varint((k << 3) | 0) + _pb_varint(d) for k, d in zip((1, 2, 3, 4), dims))
        return head + data
    shape = _pb_len(1, b''.join(_pb_varint(d) for d in arr.shape))
    return _pb_len(7, shape) + data


def test_caffemodel_to_npz_roundtrip(tmp_path):
    """A hand-encoded NetParameter (new-style `layer` and V1 `layers` entries, packed blobs, legacy
    num/channels/height/width shapes) -> npz -> TCNNet weights."""
    from vdetlib_amd.tools import caffemodel_to_npz as c2n
    rng = np.random.RandomState(0)
    w0 = rng.randn(8, 6, 1, 3).astype(np.float32); b0 = rng.randn(8).astype(np.float32)
    w1 = rng.randn(2, 8, 5, 1).astype(np.float32); b1 = rng.randn(2).astype(np.float32)
    layer0 = _pb_len(1, b'conv1') + _pb_len(2, b'Convolution') + _pb_len(7, _pb_blob(w0)) + _pb_len(7, _pb_blob(b0))
    relu = _pb_len(1, b'relu1') + _pb_len(2, b'ReLU')
    layer1_v1 = _pb_len(4, b'conv2') + _pb_len(6, _pb_blob(w1, legacy=True)) + _pb_len(6, _pb_blob(b1.reshape(1, 1, 1, 2), legacy=True))
    net = _pb_len(1, b'tcn') + _pb_len(100, layer0) + _pb_len(100, relu) + _pb_len(2, layer1_v1)
    src = tmp_path / 'tcn.caffemodel'
    src.write_bytes(net)
    dst = tmp_path / 'tcn.npz'
    layers = c2n.convert(str(src), str(dst))
    assert len(layers) == 2
    assert np.array_equal(layers[0][0], w0.reshape(8, 6, 3)) and np.array_equal(layers[0][1], b0)
    assert np.array_equal(layers[1][0], w1.reshape(2, 8, 5)) and np.array_equal(layers[1][1], b1)
    z = np.load(str(dst))
    assert sorted(z.files) == ['b0', 'b1', 'w0', 'w1']
    assert c2n.main([str(src), str(dst)]) == 0
    from vdetlib_amd.vdet.tcn import TCNNet
    net2 = TCNNet.from_npz([('det_scores', 1), ('track_scores', 1), ('anchors', 1), ('abs_anchors', 1), ('gt_overlaps', 1),
                            ('labels', 1)], str(dst))
    assert [w.shape for w, _ in net2.layers] == [(8, 6, 3), (2, 8, 5)]
    net2.save_npz(str(tmp_path / 'again.npz'))
    assert np.array_equal(np.load(str(tmp_path / 'again.npz'))['w1'], w1.reshape(2, 8, 5))


def test_raw_video_files_are_memory_mapped(tmp_path):
    """save_video_raw / open_video_raw: uncompressed .npy per array, read back as read-only memory maps (no copy)."""
    rng = np.random.RandomState(3)
    boxes = rng.rand(4, 7, 4).astype(np.float32)
    scores = rng.randn(4, 7, 5).astype(np.float32)
    counts = np.array([7, 3, 0, 7], np.int32)
    vio.save_video_raw(str(tmp_path / 'v'), boxes, scores, counts)
    b, s, c = vio.open_video_raw(str(tmp_path / 'v'))
    assert isinstance(b, np.memmap) and isinstance(s, np.memmap) and not s.flags.writeable
    assert np.array_equal(b, boxes) and np.array_equal(s, scores) and np.array_equal(c, counts)


def test_gen_vid_proto_file_cli(tmp_path, capsys):
    """tools/gen_vid_proto_file.py:10-24: frames -> .vid file, parent directory created, and the idempotent early exit
    (an existing out_file is left untouched, exit status 0)."""
    from vdetlib_amd.tools import gen_vid_proto_file as gen
    frames = tmp_path / "frames"
    frames.mkdir()
    for name in ("10.JPEG", "2.JPEG", "1.JPEG", "notes.txt"):
        (frames / name).write_bytes(b"")
    out = tmp_path / "protos" / "deep" / "v.vid"
    assert gen.main(["myvid", str(frames), str(out)]) == 0
    vid = P.proto_load(str(out))
    assert vid["video"] == "myvid" and vid["root_path"] == str(frames)
    assert [(f["frame"], f["path"]) for f in vid["frames"]] == [(1, "1.JPEG"), (2, "2.JPEG"), (3, "10.JPEG")]
    before = out.read_bytes()
    (frames / "3.JPEG").write_bytes(b"")
    assert gen.main(["other", str(frames), str(out)]) == 0          # resume rule: nothing is redone
    assert out.read_bytes() == before
    assert "already exists" in capsys.readouterr().out


def test_bench_reference_stream_generator_is_synth_video():
    """bench.synth_video_reference_stream (host RandomState draws, ranking on the device) == tests/synth.video, the
    BASELINE.md section 3 generator, value for value"""
    import torch
    import bench
    import synth
    b, s = bench.synth_video_reference_stream(torch, 2000, 3, 57, 5, torch.device("cpu"))
    wb, ws = synth.video(2000, 3, 57, 5)
    assert np.array_equal(b.numpy(), wb) and np.array_equal(s.numpy(), ws)


def test_annotation_processor_on_committed_vid_xml(tmp_path):
    """Four ILSVRC2015-VID style XML files (tests/golden/vid_xml: the dataset's own layout -- folder / filename / source /
    size / object{trackid, name, bndbox{xmax, xmin, ymax, ymin}, occluded, generated}, one file without objects, tracks
    that start late and interleave) against ILSVRC2015_val_00007000.reference.annot.
    PINNING STATUS (round 6): that file is the OUTPUT OF THE REFERENCE'S OWN TOOL (tools/imagenet_annotation_processor.py:53-118,
    run by tests/golden/make_golden.py --xml-only on these XML files).  The tool imports the third-party `xmltodict` (:6), which
    this image lacks; the run used an xml.etree stand-in for `xmltodict.parse` placed in the /tmp stub directory (same status as
    the cv2 / matlab / easydict stubs of the golden recipe), so the pin covers the tool's logic -- frame = int(filename) + 1,
    tracks in order of first appearance, single / several / no objects, field set, types, json indent=2 -- not xmltodict's
    parser.  The reference reads the files in glob's (file-system) order; the recorded run and the build use sorted order."""
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vid_xml')
    d = os.path.join(src, 'ILSVRC2015_val_00007000')
    want = json.load(open(os.path.join(src, 'ILSVRC2015_val_00007000.reference.annot')))
    assert iap.annot_proto_from_dir(d) == want
    out = tmp_path / 'a' / 'v.annot'
    assert iap.main([d, str(out)]) == 0
    assert open(out).read() == open(os.path.join(src, 'ILSVRC2015_val_00007000.reference.annot')).read()      # byte for byte (indent=2)
    assert [t['id'] for t in want['annotations']] == ['0', '1', '2'] and want['annotations'][1]['track'][0]['frame'] == 2
