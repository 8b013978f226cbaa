#!/bin/bash
# round 4, GPU call AG: K2 with one block per 256-row tile on small frames
mkdir -p gpurun_out/r4ag
export TMPDIR=/tmp
O=gpurun_out/r4ag
timeout 900 python -m pytest tests/test_small_gpu.py tests/test_volume_gpu.py tests/test_nms_gpu.py tests/test_batch_gpu.py tests/test_config5_vidshape_gpu.py tests/test_track_volume_gpu.py tests/test_pipeline_gpu.py tests/test_async_gpu.py -q -x > $O/quick.log 2>&1; echo "quick rc=$?" | tee -a $O/rc.txt
tail -n 2 $O/quick.log
timeout 300 python devtools/bench_vid.py 64 2>&1 | tail -n 8 | grep -v "^one video\|track length" | cut -c1-420
