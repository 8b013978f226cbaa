set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_config5_vidshape_gpu.py tests/test_track_volume_gpu.py -m gpu -q 2>&1 | tail -30 > $O/r2_t9.log
