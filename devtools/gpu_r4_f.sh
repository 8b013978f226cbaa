#!/bin/bash
# round 4, GPU call F: bucket v5 (sorted head, aggregated work list) + new full-size variant tests
mkdir -p gpurun_out/r4f
export TMPDIR=/tmp
L=gpurun_out/r4f/stages.log
echo "== nms" >> $L; timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L
echo "== BK_DBG=1 nms" >> $L; VDET_BK_DBG=1 timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L
echo "== track (heads)" >> $L; timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
echo "stages rc=$?" | tee -a gpurun_out/r4f/rc.txt
timeout 600 python -m pytest tests/test_bucket_gpu.py tests/test_track_volume_gpu.py tests/test_pipeline_gpu.py -q -x > gpurun_out/r4f/quick.log 2>&1; echo "quick rc=$?" | tee -a gpurun_out/r4f/rc.txt
timeout 900 python bench.py --no-cpu --no-upload > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4f/rc.txt
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4f/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4f/rc.txt
VDET_BUCKETS=2 timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_config2_full_gpu.py > gpurun_out/r4f/suite_forced.log 2>&1; echo "suite_forced rc=$?" | tee -a gpurun_out/r4f/rc.txt
cat $L
