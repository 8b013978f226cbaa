#!/bin/bash
# round 4, GPU call H: defaults settled (LSD lists; bucket path, coherent anchor slots and table-up-front as knobs) -- whole suite
# both ways, the full bench
mkdir -p gpurun_out/r4h
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4h/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4h/rc.txt
VDET_BUCKETS=2 timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4h/suite_forced.log 2>&1; echo "suite_forced rc=$?" | tee -a gpurun_out/r4h/rc.txt
timeout 1200 python bench.py > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4h/rc.txt
tail -n 3 gpurun_out/r4h/suite_default.log gpurun_out/r4h/suite_forced.log
