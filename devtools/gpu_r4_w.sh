#!/bin/bash
# round 4, GPU call W (rows built with LDS atomics; one wave per list LSD sort): the small-list kernels (one wave sorts a list, one block walks a frame): tests, VID-shape stage times
mkdir -p gpurun_out/r4w
export TMPDIR=/tmp
O=gpurun_out/r4w
timeout 900 python -m pytest tests/test_small_gpu.py -q -x > $O/small.log 2>&1; echo "small rc=$?" | tee -a $O/rc.txt
tail -n 15 $O/small.log
echo "== small (default)" >> $O/vid.log; timeout 300 python devtools/bench_vid.py 64 2>&1 | tail -n 8 >> $O/vid.log
echo "== VDET_SMALL_LISTS=0" >> $O/vid.log; VDET_SMALL_LISTS=0 timeout 300 python devtools/bench_vid.py 64 2>&1 | tail -n 8 >> $O/vid.log
cat $O/vid.log | cut -c1-700
timeout 1500 python -m pytest tests -m gpu -q -x > $O/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a $O/rc.txt
tail -n 5 $O/suite_default.log
