#!/bin/bash
# round 4, GPU call E: entry layout v4 + lean rank phase; link table up front; sharded videos; coherent leg
mkdir -p gpurun_out/r4e
export TMPDIR=/tmp
L=gpurun_out/r4e/stages.log
echo "== nms" >> $L; timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L
echo "== BK_DBG=1 nms" >> $L; VDET_BK_DBG=1 timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L
echo "== track (heads)" >> $L; timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
echo "== WALK_DBG=1" >> $L; VDET_WALK_DBG=1 timeout 200 python devtools/bench_nms_stages.py 2>&1 | tail -n 2 >> $L
echo "stages rc=$?" | tee -a gpurun_out/r4e/rc.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r4e/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4e/rc.txt
VDET_BUCKETS=2 timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r4e/suite_forced.log 2>&1; echo "suite_forced rc=$?" | tee -a gpurun_out/r4e/rc.txt
timeout 900 python bench.py > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4e/rc.txt
cat $L
