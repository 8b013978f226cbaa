#!/bin/bash
# round 4, GPU call AB: small sort with the top-byte digits counted by ballots
mkdir -p gpurun_out/r4ab
export TMPDIR=/tmp
O=gpurun_out/r4ab
timeout 900 python -m pytest tests/test_small_gpu.py tests/test_argsort_gpu.py tests/test_batch_gpu.py tests/test_config5_vidshape_gpu.py tests/test_volume_gpu.py tests/test_nms_gpu.py -q -x > $O/quick.log 2>&1; echo "quick rc=$?" | tee -a $O/rc.txt
tail -n 3 $O/quick.log
timeout 300 python devtools/bench_vid.py 64 2>&1 | tail -n 8 | grep -v "^one video\|track length" | cut -c1-420
