"""-m gpu: the reference's pipeline order on the device -- per-class threshold + top-k
(fast_rcnn_det_vid, vdet/video_det.py:89-99) THEN per-(frame, class) NMS (apply_image_nms,
vdet/image_det.py:117-123) -- through vdet_nms_volume_topk, against oracle.threshold_topk + oracle.nms."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _want(oracle, boxes, scores, thresh, score_thresh, topk):
    F, B, C = scores.shape
    idx = np.full((F, C, B), -1, np.int32)
    cnt = np.zeros((F, C), np.int32)
    for f in range(F):
        # oracle.threshold_topk speaks the reference's layout: class 0 = background, per-class box columns
        sc = np.hstack([np.zeros((B, 1), np.float32), scores[f]])
        bx = np.tile(boxes[f], (1, C + 1))
        per_cls = oracle.threshold_topk(sc, bx, score_thresh, topk)
        for c in range(C):
            d = np.asarray(per_cls[c + 1], np.float32).reshape(-1, 5)
            keep = oracle.nms(d, thresh)
            # map the kept rows back to box indices: rows are (box, score); the candidates in the reference's order
            inds = np.where(scores[f, :, c] > np.float32(score_thresh))[0]
            if len(inds) > topk:
                inds = inds[np.argsort(-scores[f, inds, c], kind='stable')[:topk]]
            cnt[f, c] = len(keep)
            idx[f, c, :len(keep)] = inds[keep]
    return idx, cnt


@pytest.mark.parametrize("cfg", [dict(seed=1, F=4, B=300, C=6, topk=100, st=0.05, kind="perm"),
                                 dict(seed=2, F=3, B=1000, C=4, topk=100, st=0.5, kind="perm"),
                                 dict(seed=3, F=2, B=500, C=8, topk=7, st=-1.0, kind="randn"),
                                 dict(seed=4, F=3, B=200, C=4, topk=300, st=0.05, kind="perm"),      # topk > B: no cut
                                 dict(seed=5, F=2, B=2500, C=3, topk=64, st=0.0, kind="randn")])
def test_threshold_topk_nms_volume(oracle, cfg):
    import torch
    from vdetlib_amd import ops
    boxes, scores = synth.video(5000 + cfg['seed'], cfg['F'], cfg['B'], cfg['C'], kind=cfg['kind'])
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    idx, cnt = ops.nms_volume(tb, ts, 0.3, score_thresh=cfg['st'], topk=cfg['topk'])
    widx, wcnt = _want(oracle, boxes, scores, 0.3, cfg['st'], cfg['topk'])
    assert np.array_equal(cnt.cpu().numpy(), wcnt)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert int(cnt.max()) <= cfg['topk']


def test_topk_ties_at_the_cut(oracle):
    """Scores tied across the k-th position: argsort(-scores)[:k] (stable) keeps the LOWEST indices of the tied
    run, while the NMS order inside the selection is the build's descending-index tie rule."""
    import torch
    from vdetlib_amd import ops
    rng = np.random.RandomState(77)
    F, B, C = 3, 400, 3
    boxes = np.stack([synth.boxes_1(rng, B) for _ in range(F)], 0)
    scores = (rng.randint(0, 12, (F, B, C)) / 12.0).astype(np.float32)      # ~33 boxes per level
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    for topk in (1, 10, 50, 101, 399):
        idx, cnt = ops.nms_volume(tb, ts, 0.3, score_thresh=0.05, topk=topk)
        idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
        for f in range(F):
            for c in range(C):
                inds = np.where(scores[f, :, c] > np.float32(0.05))[0]
                if len(inds) > topk:
                    inds = inds[np.argsort(-scores[f, inds, c], kind='stable')[:topk]]
                inds = np.sort(inds)               # rows in box-index order: the oracle's tie rule (descending ROW index
                d = np.hstack([boxes[f, inds], scores[f, inds, c][:, None]]).astype(np.float32)   # among equal scores) is then
                want = inds[oracle.nms(d, 0.3)]    # the device's (descending BOX index)
                got = idx[f, c, :cnt[f, c]]
                assert np.array_equal(got, want), (topk, f, c)
