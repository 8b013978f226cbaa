#!/bin/bash
# round 4, GPU call Z: the alternative paths at HEAD, whole suite each
mkdir -p gpurun_out/r4z
export TMPDIR=/tmp
O=gpurun_out/r4z
VDET_SMALL_LISTS=0 VDET_LINK_FILL_LDS=0 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_config3_multirank_gpu.py > $O/suite_nosmall.log 2>&1; echo "suite_nosmall rc=$?" | tee -a $O/rc.txt
VDET_BUCKETS=2 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_config3_multirank_gpu.py > $O/suite_buckets.log 2>&1; echo "suite_buckets rc=$?" | tee -a $O/rc.txt
VDET_BINSORT=0 VDET_LINK_COHERENT=0 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_config3_multirank_gpu.py > $O/suite_lsd.log 2>&1; echo "suite_lsd rc=$?" | tee -a $O/rc.txt
timeout 900 python bench.py --videos 8 --no-cpu --no-upload > $O/bench_videos.json 2> $O/bench_videos.err; echo "videos rc=$?" | tee -a $O/rc.txt
grep -h "passed\|failed" $O/*.log; cut -c1-600 $O/bench_videos.json
