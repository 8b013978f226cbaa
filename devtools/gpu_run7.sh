set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 900 python -m pytest tests/test_nms_gpu.py tests/test_volume_gpu.py tests/test_track_volume_gpu.py tests/test_config2_full_gpu.py -m gpu -q 2>&1 | tail -8 > $O/r2_t7.log
B="timeout 300 python bench.py --no-cpu --streams 1 --steps 6"
$B > $O/r2_b7_s1.json 2> $O/r2_b7_s1.err
for w in 10 11 14; do VDET_LINK_WARM=$w $B > $O/r2_b7_s1_w$w.json 2> $O/r2_b7_s1_w$w.err; done
for ch in 4 8 10 13 16 25; do VDET_VPASS_CHUNKS=$ch $B --no-link > $O/r2_b7_vp_ch$ch.json 2> $O/r2_b7_vp_ch$ch.err; done
timeout 300 python bench.py --no-cpu > $O/r2_b7.json 2> $O/r2_b7.err
