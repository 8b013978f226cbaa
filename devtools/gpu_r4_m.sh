#!/bin/bash
# round 4, GPU call M: the sorts with their key loads in flight together (LSD; counting sort with LDS-only barriers)
mkdir -p gpurun_out/r4m
export TMPDIR=/tmp
L=gpurun_out/r4m/stages.log
echo "== LSD" >> $L; timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
echo "== BINSORT=1" >> $L; VDET_BINSORT=1 timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
echo "== BUCKETS=1" >> $L; VDET_BUCKETS=1 timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
echo "stages rc=$?" | tee -a gpurun_out/r4m/rc.txt
timeout 600 python -m pytest tests/test_argsort_gpu.py tests/test_nms_gpu.py tests/test_volume_gpu.py tests/test_topk_gpu.py tests/test_detnms_gpu.py tests/test_bucket_gpu.py -q -x > gpurun_out/r4m/quick.log 2>&1; echo "quick rc=$?" | tee -a gpurun_out/r4m/rc.txt
VDET_BINSORT=1 timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_config3_multirank_gpu.py > gpurun_out/r4m/suite_binsort.log 2>&1; echo "suite_binsort rc=$?" | tee -a gpurun_out/r4m/rc.txt
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4m/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4m/rc.txt
timeout 1200 python bench.py --no-cpu --no-upload > gpurun_out/r4m/bench.json 2> gpurun_out/r4m/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4m/rc.txt
VDET_BINSORT=1 timeout 1200 python bench.py --no-cpu --no-upload --no-coherent > gpurun_out/r4m/bench_binsort.json 2> gpurun_out/r4m/bench_binsort.err; echo "bench_binsort rc=$?" | tee -a gpurun_out/r4m/rc.txt
cat $L
