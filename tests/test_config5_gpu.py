"""-m gpu: BASELINE config 5 in miniature -- VID-shaped annotation protos (synthetic: the dataset is
not available here), precomputed per-frame scores, greedy tubelets + re-scoring on the GPU, mAP with
the build's evaluator -- and the same numbers from the CPU oracle path (mAP parity)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synthetic_vid(seed, F, B, C, n_obj=3):
    """Ground-truth objects drifting over the frames; proposals = jittered copies of the objects
    (scored high for the object's class) + clutter."""
    rng = np.random.RandomState(seed)
    objs = []
    for k in range(n_obj):
        x, y = rng.uniform(50, 900), rng.uniform(50, 450)
        w, h = rng.uniform(60, 250), rng.uniform(60, 220)
        objs.append(dict(cls=int(rng.randint(1, C + 1)), box=np.array([x, y, x + w, y + h]), v=rng.uniform(-4, 4, 2)))
    boxes = np.zeros((F, B, 4), np.float32)
    scores = (0.05 * rng.rand(F, B, C)).astype(np.float32)
    annot = {'video': 'syn_%d' % seed, 'annotations': []}
    for k, o in enumerate(objs):
        annot['annotations'].append({'id': str(k), 'track': []})
    for f in range(F):
        clutter_x = rng.uniform(0, 1100, B); clutter_y = rng.uniform(0, 600, B)
        boxes[f] = np.stack([clutter_x, clutter_y, clutter_x + rng.uniform(20, 200, B), clutter_y + rng.uniform(20, 150, B)], 1)
        for k, o in enumerate(objs):
            gtb = np.round(o['box'] + np.tile(o['v'], 2) * f)
            annot['annotations'][k]['track'].append({'frame': f + 1, 'bbox': [int(v) for v in gtb], 'class_index': o['cls'],
                                                     'class': 'c%d' % o['cls']})
            for j in range(6):                                   # 6 jittered proposals per object
                b = k * 6 + j
                boxes[f, b] = gtb + rng.randint(-6, 7, 4)
                scores[f, b, o['cls'] - 1] = 0.6 + 0.39 * rng.rand()
    return np.round(boxes).astype(np.float32), scores, annot


def test_config5_map_parity(oracle):
    import torch
    from vdetlib_amd import ops
    from vdetlib_amd import eval as vev
    F, B, C, T = 20, 120, 5, 4
    gpu_dets, cpu_dets, annots = [], [], []
    for seed in (51, 52, 53):
        boxes, scores, annot = _synthetic_vid(seed, F, B, C)
        annots.append(annot)
        tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        tr, an, nt = ops.track_volume(tb, ts, nms_thres=0.3, thres=0.5, max_tracks=T, link_thres=0.4)
        det, pooled, ob = ops.rescore_tracks(tr, nt, tb, ts, overlap_thres=0.5, window=3)
        gpu_dets += vev.detections_from_tracks(annot['video'], tr.cpu().numpy(), nt.cpu().numpy(),
                                               pooled.cpu().numpy(), ob.cpu().numpy())
        # the same through the oracle
        wtr = np.full((C, T, F, 5), np.nan, np.float32); wnt = np.zeros(C, np.int32)
        wsc = np.full((C, T, F), np.nan); wbx = np.full((C, T, F, 4), np.nan, np.float32)
        for c in range(C):
            t_, a_, n_ = oracle.greedy_track_volume(boxes, scores[:, :, c], 0.3, 0.5, T, 0.4, 0)
            wtr[c], wnt[c] = t_, n_
            for t in range(n_):
                fr = [f for f in range(F) if not np.isnan(t_[t, f, 0])]
                s, bx = [], []
                for f in fr:
                    ss, bb, _ = oracle.spatial_maxpool([t_[t, f, :4]], boxes[f], scores[f, :, c], 0.5)
                    s.append(ss[0]); bx.append(bb[0])
                comp = oracle.score_completion(s)
                pool = [max(comp[g] if 0 <= g < len(comp) else -1e5 for g in (i - 1, i, i + 1)) for i in range(len(comp))]
                wsc[c, t, fr] = pool
                wbx[c, t, fr] = np.asarray(bx, np.float32)
        cpu_dets += vev.detections_from_tracks(annot['video'], wtr, wnt, wsc, wbx)
    gt = vev.ground_truth_from_annots(annots)
    aps_g, map_g = vev.evaluate(gpu_dets, gt)
    aps_c, map_c = vev.evaluate(cpu_dets, gt)
    assert gpu_dets == cpu_dets                       # identical tubelets ...
    assert aps_g == aps_c and map_g == map_c          # ... hence identical AP per class and mAP
    assert map_g > 0.5                                # and the pipeline does find the planted objects
