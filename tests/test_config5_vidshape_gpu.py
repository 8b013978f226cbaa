"""-m gpu: BASELINE configs[4] at VID shape (the dataset itself is not available: synthetic annotation
protos of the same shape) -- videos of ~500 frames with RAGGED per-frame proposal counts (<= 300) x 30
classes, moved through the array transport (vdetlib_amd.io: padding boxes, -inf scores), NMS + greedy
tubelets + re-scoring on the GPU and through the oracle, VOC-style AP on both: identical tubelets, hence
identical AP per class and mAP.  Also the GPU test of SURVEY 8(f) rank 1 (array ingest)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def vid_shape_video(seed, F=500, Bmax=300, C=30, n_obj=4):
    """Annotated objects drifting slowly; per frame a ragged set of proposals (jittered copies of the visible
    objects, scored for the object's class, + clutter).  Returns (vid_proto, frame_to_det, annot_proto)."""
    rng = np.random.RandomState(seed)
    name = 'vidshape_%d' % seed
    vid = synth.make_vid_proto(name, F)
    objs = []
    for k in range(n_obj):
        x, y = rng.uniform(50, 800), rng.uniform(50, 400)
        w, h = rng.uniform(60, 250), rng.uniform(60, 220)
        objs.append(dict(cls=int(rng.randint(1, C + 1)), box=np.array([x, y, x + w, y + h]), v=rng.uniform(-0.6, 0.6, 2),
                         first=int(rng.randint(0, F // 3)), last=int(rng.randint(2 * F // 3, F))))
    annot = {'video': name, 'annotations': [{'id': str(k), 'track': []} for k in range(n_obj)]}
    frame_to_det = {}
    for f in range(F):
        n = int(rng.randint(Bmax // 2, Bmax + 1))
        if f % 97 == 13:
            n = 0                                    # a frame without detections (a missing .mat)
        if n == 0:
            continue
        cx, cy = rng.uniform(0, 1100, n), rng.uniform(0, 600, n)
        boxes = np.stack([cx, cy, cx + rng.uniform(20, 200, n), cy + rng.uniform(20, 150, n)], 1)
        zs = (0.05 * rng.rand(n, C)).astype(np.float32)
        slot = 0
        for k, o in enumerate(objs):
            if not (o['first'] <= f <= o['last']):
                continue
            gtb = np.round(o['box'] + np.tile(o['v'], 2) * f)
            annot['annotations'][k]['track'].append({'frame': f + 1, 'bbox': [int(v) for v in gtb], 'class_index': o['cls'],
                                                     'class': 'c%d' % o['cls']})
            for j in range(5):
                boxes[slot] = gtb + rng.randint(-5, 6, 4)
                zs[slot, o['cls'] - 1] = 0.6 + 0.39 * rng.rand()
                slot += 1
        frame_to_det[f + 1] = (np.round(boxes).astype(np.float32), zs)
    return vid, frame_to_det, annot


def test_vid_shape_ragged_map_parity(oracle):
    import torch
    from vdetlib_amd import ops, io as vio
    from vdetlib_amd import eval as vev
    T = 3
    gpu_dets, cpu_dets, annots = [], [], []
    for seed in (9001, 9002):
        vid, frame_to_det, annot = vid_shape_video(seed)
        annots.append(annot)
        boxes, scores, counts = vio.arrays_from_frame_to_det(vid, frame_to_det)
        F, B, C = scores.shape
        assert (F, C) == (500, 30) and 150 <= B <= 300 and counts.min() == 0 and len(set(counts.tolist())) > 50
        # round trip of the transport
        back = vio.frame_to_det_from_arrays(vid, boxes, scores, counts)
        assert sorted(back) == sorted(frame_to_det)
        assert all(np.array_equal(back[k][0], frame_to_det[k][0]) and np.array_equal(back[k][1], frame_to_det[k][1]) for k in back)
        tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        # per-(frame, class) NMS: padding (score -inf) never survives, real boxes as in the oracle
        idx, cnt = ops.nms_volume(tb, ts, 0.3, score_thresh=float('-inf'), cap=B)
        widx, wcnt = oracle.nms_volume(boxes, scores, 0.3, score_thresh=float('-inf'), cap=B, frames=(0, 40))
        assert np.array_equal(cnt[:40].cpu().numpy(), wcnt[:40]) and np.array_equal(idx[:40].cpu().numpy(), widx[:40])
        assert bool((cnt <= torch.from_numpy(counts).cuda()[:, None]).all())
        # tubelets + re-scoring
        pooled_vol, _ = ops.volume_pass(ts, 3)                  # C = 30: the non-fused path
        assert np.array_equal(pooled_vol[:, :8].cpu().numpy(), oracle.temporal_maxpool(scores[:, :8], 3))
        tr, an, nt = ops.track_volume(tb, ts, nms_thres=0.3, thres=0.5, max_tracks=T, link_thres=0.4)
        det, pooled, ob = ops.rescore_tracks(tr, nt, tb, ts, overlap_thres=0.5, window=3)
        wtr, wnt, wsc, wbx = oracle.rescored_tubelets(boxes, scores, 0.3, 0.5, T, 0.4, 0.5, 3)
        assert np.array_equal(nt.cpu().numpy(), wnt)
        assert np.array_equal(tr.cpu().numpy(), wtr, equal_nan=True)
        np.testing.assert_allclose(pooled.cpu().numpy(), wsc, rtol=0, atol=1e-9, equal_nan=True)
        assert np.array_equal(ob.cpu().numpy(), wbx, equal_nan=True)
        assert float(np.nanmin(wbx)) > -1.0e5                   # no tubelet ever sits on a padding box
        gpu_dets += vev.detections_from_tracks(annot['video'], tr.cpu().numpy(), nt.cpu().numpy(), pooled.cpu().numpy(), ob.cpu().numpy())
        cpu_dets += vev.detections_from_tracks(annot['video'], wtr, wnt, wsc, wbx)
    gt = vev.ground_truth_from_annots(annots)
    aps_g, map_g = vev.evaluate(gpu_dets, gt)
    aps_c, map_c = vev.evaluate(cpu_dets, gt)
    assert gpu_dets == cpu_dets
    assert aps_g == aps_c and map_g == map_c
    assert map_g > 0.5


def test_file_to_hbm_pipeline(tmp_path, oracle):
    """SURVEY 8(f) rank 1 end to end: videos on disk as raw .npy -> memory map -> pinned staging (the one host pass)
    -> asynchronous upload on its own stream, double-buffered, while the previous video is processed."""
    import torch
    from vdetlib_amd import ops, io as vio
    vids = []
    for i in range(3):
        vid, f2d, _ = vid_shape_video(9100 + i, F=60, Bmax=200, C=8)
        b, s, n = vio.arrays_from_frame_to_det(vid, f2d)
        vio.save_video_raw(str(tmp_path / ('v%d' % i)), b, s, n)
        vids.append((b, s, n))
    up = vio.VideoUploader('cuda', nbuf=2)
    pending = up.submit(str(tmp_path / 'v0'))
    for i in range(3):
        tb, ts, counts, ev = pending
        if i + 1 < 3:
            pending = up.submit(str(tmp_path / ('v%d' % (i + 1))))       # next video: copy + upload under this one's kernels
        up.acquire(tb, ts, ev)
        idx, cnt = ops.nms_volume(tb, ts, 0.3, score_thresh=float('-inf'))
        b, s, n = vids[i]
        assert np.array_equal(counts, n) and torch.equal(ts.cpu(), torch.from_numpy(s))
        widx, wcnt = oracle.nms_volume(b, s, 0.3, score_thresh=float('-inf'), frames=(0, 10))
        assert np.array_equal(cnt[:10].cpu().numpy(), wcnt[:10]) and np.array_equal(idx[:10].cpu().numpy(), widx[:10])


def test_vid_shape_batch_map_parity(oracle):
    """configs[4] through the BATCHED entry points: four annotated VID-shaped videos of different lengths, ragged proposals
    padded to one B, frames concatenated -> ops.volume_pass(frame_off=...) + ops.video_batch -> detections -> VOC-style
    AP, against the oracle's tubelets of every video on its own: identical detections, AP per class and mAP."""
    import torch
    from vdetlib_amd import ops, io as vio
    from vdetlib_amd import eval as vev
    T, Bmax, C = 3, 300, 30
    vids = []
    for seed, F in ((9101, 120), (9102, 200), (9103, 90), (9104, 150)):
        vid, frame_to_det, annot = vid_shape_video(seed, F=F, Bmax=Bmax, C=C)
        b, s, counts = vio.arrays_from_frame_to_det(vid, frame_to_det)
        if b.shape[1] < Bmax:                                     # one B for the whole batch: more padding
            pad = Bmax - b.shape[1]
            b = np.concatenate([b, np.tile(vio.pad_boxes(Bmax)[None, b.shape[1]:], (F, 1, 1))], 1)
            s = np.concatenate([s, np.full((F, pad, C), -np.inf, np.float32)], 1)
        vids.append((annot, b, s))
    off = np.concatenate([[0], np.cumsum([b.shape[0] for _, b, _ in vids])])
    boxes = torch.from_numpy(np.concatenate([b for _, b, _ in vids])).cuda()
    scores = torch.from_numpy(np.concatenate([s for _, _, s in vids])).cuda()
    pooled_vol, _ = ops.volume_pass(scores, 3, frame_off=off)
    out = ops.video_batch(boxes, scores, off, nms_thres=0.3, thres=0.5, max_tracks=T, link_thres=0.4, overlap_thres=0.5, window=3)
    gpu_dets, cpu_dets, annots = [], [], []
    for v, (annot, b, s) in enumerate(vids):
        annots.append(annot)
        f0, f1 = int(off[v]), int(off[v + 1])
        assert np.array_equal(pooled_vol[f0:f1, :6].cpu().numpy(), oracle.temporal_maxpool(s[:, :6], 3))
        wtr, wnt, wsc, wbx = oracle.rescored_tubelets(b, s, 0.3, 0.5, T, 0.4, 0.5, 3)
        nt = out["ntracks"][v].cpu().numpy()
        assert np.array_equal(nt, wnt)
        tr, pooled, ob = out["tracks"][v].cpu().numpy(), out["pooled"][v].cpu().numpy(), out["tboxes"][v].cpu().numpy()
        for c in range(C):
            n = int(wnt[c])
            assert np.array_equal(tr[c, :n], wtr[c, :n], equal_nan=True)
            np.testing.assert_allclose(pooled[c, :n], wsc[c, :n], rtol=0, atol=1e-9, equal_nan=True)
            assert np.array_equal(ob[c, :n], wbx[c, :n], equal_nan=True)
        gpu_dets += vev.detections_from_tracks(annot['video'], tr, nt, pooled, ob)
        cpu_dets += vev.detections_from_tracks(annot['video'], wtr, wnt, wsc, wbx)
    gt = vev.ground_truth_from_annots(annots)
    aps_g, map_g = vev.evaluate(gpu_dets, gt)
    aps_c, map_c = vev.evaluate(cpu_dets, gt)
    assert gpu_dets == cpu_dets and aps_g == aps_c and map_g == map_c and map_g > 0.5
