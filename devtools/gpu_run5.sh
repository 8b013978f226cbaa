set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 900 python -m pytest tests/test_nms_gpu.py tests/test_volume_gpu.py tests/test_track_volume_gpu.py tests/test_link_golden_gpu.py tests/test_config2_full_gpu.py -m gpu -q 2>&1 | tail -30 > $O/r2_t5.log
B="timeout 300 python bench.py --no-cpu"
$B > $O/r2_b5.json 2> $O/r2_b5.err
$B --streams 1 --steps 6 > $O/r2_b5_s1.json 2> $O/r2_b5_s1.err
VDET_LINK_MAXB=16 $B --streams 1 --steps 6 > $O/r2_b5_s1_mb16.json 2> $O/r2_b5_s1_mb16.err
VDET_LINK_MAXB=16 $B > $O/r2_b5_mb16.json 2> $O/r2_b5_mb16.err
$B --no-link > $O/r2_b5_nolink.json 2> $O/r2_b5_nolink.err
$B --streams 2 > $O/r2_b5_st2.json 2> $O/r2_b5_st2.err
$B --streams 4 > $O/r2_b5_st4.json 2> $O/r2_b5_st4.err
