#!/bin/bash
# round 4, GPU call A: the bucketed lists -- new tests first, then the whole suite both ways, then the bench
mkdir -p gpurun_out/r4a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bucket_gpu.py -x -q > gpurun_out/r4a/bucket.log 2>&1; echo "bucket rc=$?" | tee -a gpurun_out/r4a/rc.txt
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4a/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4a/rc.txt
VDET_BUCKETS=2 timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r4a/suite_forced.log 2>&1; echo "suite_forced rc=$?" | tee -a gpurun_out/r4a/rc.txt
timeout 600 python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4a/rc.txt
VDET_BUCKETS=0 timeout 600 python bench.py --no-cpu --no-upload > gpurun_out/r4a/bench_lsd.json 2> gpurun_out/r4a/bench_lsd.err; echo "bench_lsd rc=$?" | tee -a gpurun_out/r4a/rc.txt
tail -3 gpurun_out/r4a/bucket.log gpurun_out/r4a/suite_default.log gpurun_out/r4a/suite_forced.log
