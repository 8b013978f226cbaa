#!/bin/bash
# round 4, GPU call B: bucketed lists v2 (chunk-aligned flagged buckets) + per-class box NMS (compile only so far)
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bucket_gpu.py tests/test_alias_gpu.py -x -q > gpurun_out/r4b/bucket.log 2>&1; echo "bucket rc=$?" | tee -a gpurun_out/r4b/rc.txt
VDET_BUCKETS=2 timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4b/suite_forced.log 2>&1; echo "suite_forced rc=$?" | tee -a gpurun_out/r4b/rc.txt
timeout 600 python bench.py --no-cpu --no-upload > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4b/rc.txt
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4b/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4b/rc.txt
tail -n 3 gpurun_out/r4b/bucket.log gpurun_out/r4b/suite_default.log gpurun_out/r4b/suite_forced.log
