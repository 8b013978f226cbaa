#!/bin/bash
# round 4, GPU call P: suite at HEAD (K2 kept variant), videos in flight 3 / 5 / 6, the two latency options with 4 in flight
mkdir -p gpurun_out/r4p
export TMPDIR=/tmp
O=gpurun_out/r4p
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a $O/rc.txt
B="python bench.py --no-cpu --no-upload --no-coherent --no-latency-leg"
for s in 4 3 5 6; do timeout 600 $B --streams $s > $O/bench_s$s.json 2> $O/bench_s$s.err; echo "streams $s rc=$?" | tee -a $O/rc.txt; done
VDET_GRAPH_PIPE=1 VDET_AUX_STREAM=1 timeout 600 $B > $O/bench_lat.json 2> $O/bench_lat.err; echo "latopts rc=$?" | tee -a $O/rc.txt
VDET_BUCKETS=2 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_config3_multirank_gpu.py > $O/suite_buckets.log 2>&1; echo "suite_buckets rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4p/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), d.get('single_video_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
