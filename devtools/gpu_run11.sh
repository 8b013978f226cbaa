set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 1200 python -m pytest tests/test_nms_gpu.py tests/test_volume_gpu.py tests/test_track_volume_gpu.py tests/test_topk_gpu.py tests/test_config2_full_gpu.py -m gpu -q 2>&1 | tail -12 > $O/r2_t11.log
B="timeout 300 python bench.py --no-cpu"
$B --streams 1 --steps 6 > $O/r2_b11_s1.json 2> $O/r2_b11_s1.err
$B > $O/r2_b11.json 2> $O/r2_b11.err
