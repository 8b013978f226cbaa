#!/bin/bash
# round 4, GPU call X: the link table from LDS-staged frames (link_fill_frame_kernel) against the quad version
mkdir -p gpurun_out/r4x
export TMPDIR=/tmp
O=gpurun_out/r4x
timeout 900 python -m pytest tests/test_track_volume_gpu.py tests/test_batch_gpu.py tests/test_config5_vidshape_gpu.py -q -x > $O/quick.log 2>&1; echo "quick rc=$?" | tee -a $O/rc.txt
tail -n 4 $O/quick.log
echo "== default" >> $O/vid.log; timeout 300 python devtools/bench_vid.py 64 2>&1 | tail -n 8 >> $O/vid.log
echo "== VDET_LINK_FILL_LDS=0" >> $O/vid.log; VDET_LINK_FILL_LDS=0 timeout 300 python devtools/bench_vid.py 64 2>&1 | tail -n 8 >> $O/vid.log
grep -v "^one video\|track length" $O/vid.log | cut -c1-420
timeout 1500 python -m pytest tests -m gpu -q -x > $O/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a $O/rc.txt
grep -h "passed\|failed" $O/suite_default.log
