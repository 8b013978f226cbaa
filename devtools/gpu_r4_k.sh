#!/bin/bash
# round 4, GPU call K: link table with loads up front, batch re-scoring from the graph, leaner walk; whole suite + bench
mkdir -p gpurun_out/r4k
export TMPDIR=/tmp
timeout 300 python devtools/bench_vid.py 64 > gpurun_out/r4k/vid.log 2>&1; echo "vid rc=$?" | tee -a gpurun_out/r4k/rc.txt
timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 > gpurun_out/r4k/stages.log
timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_track_volume_gpu.py tests/test_config5_vidshape_gpu.py tests/test_config5_gpu.py tests/test_link_golden_gpu.py tests/test_nms_gpu.py tests/test_volume_gpu.py -q -x > gpurun_out/r4k/quick.log 2>&1; echo "quick rc=$?" | tee -a gpurun_out/r4k/rc.txt
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4k/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a gpurun_out/r4k/rc.txt
timeout 1200 python bench.py > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4k/rc.txt
grep -n "batch of\|one video" gpurun_out/r4k/vid.log; cat gpurun_out/r4k/stages.log
