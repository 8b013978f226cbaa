"""-m gpu: BASELINE config 5 in miniature -- VID-shaped annotation protos (synthetic: the dataset is
not available here), precomputed per-frame scores, greedy tubelets + re-scoring on the GPU, mAP with
the build's evaluator -- and the same numbers from the CPU oracle path (mAP parity)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def test_config5_map_parity(oracle):
    import torch
    from vdetlib_amd import ops
    from vdetlib_amd import eval as vev
    F, B, C, T = 20, 120, 5, 4
    gpu_dets, cpu_dets, annots = [], [], []
    for seed in (51, 52, 53):
        boxes, scores, annot = synth.vid_with_objects(seed, F, B, C)
        annots.append(annot)
        tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        tr, an, nt = ops.track_volume(tb, ts, nms_thres=0.3, thres=0.5, max_tracks=T, link_thres=0.4)
        det, pooled, ob = ops.rescore_tracks(tr, nt, tb, ts, overlap_thres=0.5, window=3)
        gpu_dets += vev.detections_from_tracks(annot['video'], tr.cpu().numpy(), nt.cpu().numpy(),
                                               pooled.cpu().numpy(), ob.cpu().numpy())
        # the same through the oracle
        wtr, wnt, wsc, wbx = oracle.rescored_tubelets(boxes, scores, 0.3, 0.5, T, 0.4, 0.5, 3)
        cpu_dets += vev.detections_from_tracks(annot['video'], wtr, wnt, wsc, wbx)
    gt = vev.ground_truth_from_annots(annots)
    aps_g, map_g = vev.evaluate(gpu_dets, gt)
    aps_c, map_c = vev.evaluate(cpu_dets, gt)
    assert gpu_dets == cpu_dets                       # identical tubelets ...
    assert aps_g == aps_c and map_g == map_c          # ... hence identical AP per class and mAP
    assert map_g > 0.5                                # and the pipeline does find the planted objects
