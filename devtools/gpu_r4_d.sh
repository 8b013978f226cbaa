#!/bin/bash
# round 4, GPU call D: where the bucket kernel's and the bucketed walk's time goes (timing knobs), the link table up front,
# the whole suite both ways
mkdir -p gpurun_out/r4d
export TMPDIR=/tmp
L=gpurun_out/r4d/stages.log
for dbg in 0 1 3 7 15; do echo "== BK_DBG=$dbg nms" >> $L; VDET_BK_DBG=$dbg timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L; done
echo "== track (heads)" >> $L; timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
for dbg in 1 2; do echo "== WALK_DBG=$dbg" >> $L; VDET_WALK_DBG=$dbg timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L; done
echo "== LSD" >> $L; VDET_BUCKETS=0 timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L
echo "stages rc=$?" | tee -a gpurun_out/r4d/rc.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r4d/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4d/rc.txt
VDET_BUCKETS=2 timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r4d/suite_forced.log 2>&1; echo "suite_forced rc=$?" | tee -a gpurun_out/r4d/rc.txt
timeout 600 python bench.py --no-cpu --no-upload > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4d/rc.txt
cat $L
