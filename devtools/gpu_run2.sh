# round 2, GPU call 2: full test suite + link memo / thread variants + volume-pass tilings + gating
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/r2_t2.log
timeout 120 devtools/valu_bench > $O/r2_valu2.csv 2>&1
B="timeout 300 python bench.py --no-cpu"
$B > $O/r2_b2.json 2> $O/r2_b2.err
$B --gate none > $O/r2_b2_nogate.json 2> $O/r2_b2_nogate.err
$B --streams 2 > $O/r2_b2_st2.json 2> $O/r2_b2_st2.err
$B --streams 3 > $O/r2_b2_st3.json 2> $O/r2_b2_st3.err
$B --streams 1 --steps 6 > $O/r2_b2_s1.json 2> $O/r2_b2_s1.err
VDET_LINK_MEMO=0 $B --streams 1 --steps 6 > $O/r2_b2_s1_nomemo.json 2> $O/r2_b2_s1_nomemo.err
VDET_LINK_THREADS=64 $B --streams 1 --steps 6 > $O/r2_b2_s1_lt64.json 2> $O/r2_b2_s1_lt64.err
VDET_LINK_THREADS=128 $B --streams 1 --steps 6 > $O/r2_b2_s1_lt128.json 2> $O/r2_b2_s1_lt128.err
VDET_LINK_MEMO=0 VDET_LINK_THREADS=64 $B --streams 1 --steps 6 > $O/r2_b2_s1_nomemo_lt64.json 2> $O/r2_b2_s1_nomemo_lt64.err
VDET_VPASS=256,16 $B --streams 1 --steps 6 > $O/r2_b2_s1_vp256.json 2> $O/r2_b2_s1_vp256.err
python - <<'PY' > $O/r2_memo_stats.txt 2>&1
import sys, torch
sys.path.insert(0, '.')
import bench
from vdetlib_amd import ops, _lib
dev = torch.device('cuda', 0)
b, s = bench.synth_video_cuda(torch, 2000, 300, 10000, 200, dev)
cx = _lib.Context(0)
for mt in (1, 2, 5, 10):
    cx.invalidate()
    ops.track_volume(b, s, nms_thres=0.3, thres=0.9, max_tracks=mt, link_thres=0.5, ctx=cx)
    print('max_tracks', mt, 'hits', cx.query(4), 'misses', cx.query(5))
PY
