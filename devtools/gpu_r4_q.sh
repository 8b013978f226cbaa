#!/bin/bash
# round 4, GPU call Q: the coherent-video predictor stopped at max_tracks distinct objects (VDET_LINK_COHERENT=1) against the default
mkdir -p gpurun_out/r4q
export TMPDIR=/tmp
O=gpurun_out/r4q
timeout 900 python -m pytest tests/test_track_volume_gpu.py tests/test_volume_gpu.py -q -x > $O/quick.log 2>&1; echo "quick rc=$?" | tee -a $O/rc.txt
B="python bench.py --no-cpu --no-upload --no-latency-leg"
timeout 900 $B > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?" | tee -a $O/rc.txt
VDET_LINK_COHERENT=1 timeout 900 $B > $O/bench_coh.json 2> $O/bench_coh.err; echo "coherent rc=$?" | tee -a $O/rc.txt
B2="python bench.py --no-cpu --no-upload --no-coherent --no-latency-leg"
for r in 1 2; do
  timeout 600 $B2 > $O/bench_plain$r.json 2> $O/bench_plain$r.err
  VDET_GRAPH_PIPE=1 VDET_AUX_STREAM=1 timeout 600 $B2 > $O/bench_lat$r.json 2> $O/bench_lat$r.err
  VDET_AUX_STREAM=1 timeout 600 $B2 > $O/bench_aux$r.json 2> $O/bench_aux$r.err
done
VDET_BUCKETS=2 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_config3_multirank_gpu.py > $O/suite_buckets.log 2>&1; echo "suite_buckets rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4q/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); vc=d.get('value_coherent') or {}
        print(f, round(d['ms_per_step'],3), d.get('single_video_ms'), 'coherent:', vc.get('ms_per_step'), vc.get('single_video_ms'), vc.get('link_steps_memo_scanned'), {k:vc.get('stage_ms_one_video',{}).get(k) for k in ('track_link','track_loop')})
    except Exception as e: print(f, 'ERR', e)
PY
