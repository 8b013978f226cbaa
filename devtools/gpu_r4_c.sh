#!/bin/bash
# round 4, GPU call C: bucket kernel v3 (persistent, prefetch, owner-thread ranking) + per-class box NMS tests
mkdir -p gpurun_out/r4c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bucket_gpu.py tests/test_detnms_gpu.py -x -q > gpurun_out/r4c/new.log 2>&1; echo "new rc=$?" | tee -a gpurun_out/r4c/rc.txt
timeout 600 python bench.py --no-cpu --no-upload > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4c/rc.txt
VDET_BUCKET_BLOCK=1024 timeout 600 python bench.py --no-cpu --no-upload > gpurun_out/r4c/bench_1024.json 2> gpurun_out/r4c/bench_1024.err; echo "bench1024 rc=$?" | tee -a gpurun_out/r4c/rc.txt
VDET_BUCKETS=2 timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4c/suite_forced.log 2>&1; echo "suite_forced rc=$?" | tee -a gpurun_out/r4c/rc.txt
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4c/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4c/rc.txt
tail -n 5 gpurun_out/r4c/new.log
