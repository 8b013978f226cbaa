"""Test infrastructure: the CPU oracle's per-class tubelet pipeline on several host processes (the oracle is a
single-threaded python / C restatement; classes are independent, the GPU box has hundreds of host threads).
Spawned workers only import numpy + the oracle (never HIP); the video travels as .npy files."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _class_job(job):
    boxes_path, scores_path, opts = job
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle
    boxes = np.load(boxes_path, mmap_mode="r")
    scores = np.load(scores_path)                        # [F, B] of one class
    o = opts
    tr, nt, pooled, bx, det = oracle.rescored_tubelets(np.asarray(boxes), scores[:, :, None], o["nms_thres"], o["thres"],
                                                       o["max_tracks"], o["link_thres"], o["pool_thres"], o["window"],
                                                       return_det=True)
    _, anchors, _ = (None, None, None)
    return tr[0], int(nt[0]), pooled[0], bx[0], det[0]


def rescored_tubelets_per_class(boxes, class_scores, opts, tmpdir, processes=None):
    """boxes [F,B,4] f32; class_scores {class id: [F,B] f32}.  Returns {class id: (tracks [T,F,5], ntracks, pooled [T,F],
    boxes [T,F,4], det [T,F])} -- oracle.rescored_tubelets of every class, one process each."""
    import multiprocessing as mp
    bp = os.path.join(str(tmpdir), "boxes.npy")
    np.save(bp, boxes)
    jobs, order = [], []
    for c, sc in class_scores.items():
        sp = os.path.join(str(tmpdir), "scores_%d.npy" % c)
        np.save(sp, sc)
        jobs.append((bp, sp, opts))
        order.append(c)
    n = processes or min(len(jobs), max(1, len(os.sched_getaffinity(0))))
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(_class_job, jobs, chunksize=1)
    return dict(zip(order, res))
