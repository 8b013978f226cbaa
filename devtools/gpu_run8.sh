set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 900 python -m pytest tests/test_track_volume_gpu.py tests/test_link_golden_gpu.py tests/test_config2_full_gpu.py tests/test_async_gpu.py tests/test_config5_vidshape_gpu.py -m gpu -q 2>&1 | tail -8 > $O/r2_t8.log
B="timeout 300 python bench.py --no-cpu"
$B > $O/r2_b8.json 2> $O/r2_b8.err
$B --streams 1 --steps 6 > $O/r2_b8_s1.json 2> $O/r2_b8_s1.err
$B --streams 2 > $O/r2_b8_st2.json 2> $O/r2_b8_st2.err
VDET_AUX_STREAM=0 $B > $O/r2_b8_noaux.json 2> $O/r2_b8_noaux.err
VDET_AUX_STREAM=0 $B --streams 1 --steps 6 > $O/r2_b8_s1_noaux.json 2> $O/r2_b8_s1_noaux.err
