#!/bin/bash
# round 4, GPU call N: counting sort as the default, K2 with two word sets + 16-byte copy-out, re-scoring with its loads up front
mkdir -p gpurun_out/r4n
export TMPDIR=/tmp
L=gpurun_out/r4n/stages.log
echo "== default" >> $L; timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
echo "== VID batch" >> $L; timeout 300 python devtools/bench_vid.py 2>&1 | tail -n 4 >> $L
echo "stages rc=$?" | tee -a gpurun_out/r4n/rc.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4n/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4n/rc.txt
timeout 1200 python bench.py --no-cpu --no-upload > gpurun_out/r4n/bench.json 2> gpurun_out/r4n/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4n/rc.txt
cat $L
