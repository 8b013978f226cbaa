#!/bin/bash
# round 4, GPU call G: rank loop unrolled; head size A/B; coherent anchor prediction
mkdir -p gpurun_out/r4g
export TMPDIR=/tmp
L=gpurun_out/r4g/stages.log
for hd in 192 32 96 320; do echo "== track HEAD=$hd" >> $L; VDET_BUCKET_HEAD=$hd timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L; done
echo "== LSD track" >> $L; VDET_BUCKETS=0 timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
echo "stages rc=$?" | tee -a gpurun_out/r4g/rc.txt
timeout 600 python -m pytest tests/test_bucket_gpu.py tests/test_track_volume_gpu.py -q -x > gpurun_out/r4g/quick.log 2>&1; echo "quick rc=$?" | tee -a gpurun_out/r4g/rc.txt
timeout 900 python bench.py --no-cpu --no-upload > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r4g/rc.txt
VDET_BUCKETS=0 timeout 900 python bench.py --no-cpu --no-upload --no-coherent > gpurun_out/r4g/bench_lsd.json 2> gpurun_out/r4g/bench_lsd.err; echo "bench_lsd rc=$?" | tee -a gpurun_out/r4g/rc.txt
VDET_LINK_COHERENT=0 timeout 900 python bench.py --no-cpu --no-upload > gpurun_out/r4g/bench_nocoh.json 2> gpurun_out/r4g/bench_nocoh.err; echo "bench_nocoh rc=$?" | tee -a gpurun_out/r4g/rc.txt
cat $L
