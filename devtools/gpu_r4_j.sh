#!/bin/bash
# round 4, GPU call J: quad-lane link table, batched videos without predicted chains; r04 profiles at HEAD; whole suite
mkdir -p gpurun_out/r4j
export TMPDIR=/tmp
timeout 300 python devtools/bench_vid.py 64 > gpurun_out/r4j/vid.log 2>&1
VDET_BATCH_CHAINS=1 timeout 300 python devtools/bench_vid.py 64 > gpurun_out/r4j/vid_chains.log 2>&1
echo "vid rc=$?" | tee -a gpurun_out/r4j/rc.txt
timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_track_volume_gpu.py tests/test_config5_vidshape_gpu.py tests/test_config5_gpu.py tests/test_link_golden_gpu.py -q -x > gpurun_out/r4j/quick.log 2>&1; echo "quick rc=$?" | tee -a gpurun_out/r4j/rc.txt
bash devtools/gpu_profile_r4.sh > gpurun_out/r4j/profile.log 2>&1; echo "profile rc=$?" | tee -a gpurun_out/r4j/rc.txt
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4j/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4j/rc.txt
grep -n "batch of\|one video" gpurun_out/r4j/vid.log gpurun_out/r4j/vid_chains.log
