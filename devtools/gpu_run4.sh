set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 900 python -m pytest tests/test_track_volume_gpu.py tests/test_link_golden_gpu.py tests/test_volume_pass_gpu.py tests/test_config2_full_gpu.py tests/test_config5_vidshape_gpu.py tests/test_async_gpu.py -m gpu -q 2>&1 | tail -30 > $O/r2_t4.log
B="timeout 300 python bench.py --no-cpu"
$B > $O/r2_b4.json 2> $O/r2_b4.err
$B --streams 1 --steps 6 > $O/r2_b4_s1.json 2> $O/r2_b4_s1.err
VDET_LINK_WARM=0 $B --streams 1 --steps 6 > $O/r2_b4_s1_w0.json 2> $O/r2_b4_s1_w0.err
VDET_LINK_WARM=16 $B --streams 1 --steps 6 > $O/r2_b4_s1_w16.json 2> $O/r2_b4_s1_w16.err
VDET_LINK_WARM=6 $B --streams 1 --steps 6 > $O/r2_b4_s1_w6.json 2> $O/r2_b4_s1_w6.err
VDET_LINK_WARM=0 $B > $O/r2_b4_w0.json 2> $O/r2_b4_w0.err
VDET_VPASS=256,16 $B --streams 1 --steps 6 > $O/r2_b4_s1_vp256.json 2> $O/r2_b4_s1_vp256.err
python - <<'PY' > $O/r2_memo_stats4.txt 2>&1
import sys, torch
sys.path.insert(0, '.')
import bench
from vdetlib_amd import ops, _lib
dev = torch.device('cuda', 0)
b, s = bench.synth_video_cuda(torch, 2000, 300, 10000, 200, dev)
cx = _lib.Context(0)
cx.invalidate()
ops.track_volume(b, s, nms_thres=0.3, thres=0.9, max_tracks=10, link_thres=0.5, ctx=cx)
print('loop hits', cx.query(4), 'loop misses', cx.query(5), 'warm hits', cx.query(6), 'warm misses', cx.query(7))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/r2_prof4 -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --streams 1 > $R/$O/r2_prof4.log 2>&1
cd $R
python profiles/dispatch_times.py $O/r2_prof4/k_results.db track_link 44 > $O/r2_link_dispatch4.txt 2>&1
python profiles/summarize.py $O/r2_prof4/k_results.db $O/r2_kernel_stats4.csv "python bench.py --steps 2 --warmup 1 --no-cpu --streams 1" > /dev/null 2>&1
rm -rf $O/r2_prof4
