#!/bin/bash
# round 4, GPU call L: the packed walk as a three-set software pipeline (no register copies, no path-dependent load counts)
mkdir -p gpurun_out/r4l
export TMPDIR=/tmp
timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 > gpurun_out/r4l/stages.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4l/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4l/rc.txt
timeout 1200 python bench.py --no-cpu --no-upload > gpurun_out/r4l/bench.json 2> gpurun_out/r4l/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4l/rc.txt
cat gpurun_out/r4l/stages.log; tail -n 2 gpurun_out/r4l/suite_default.log
