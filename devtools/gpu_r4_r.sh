#!/bin/bash
# round 4, GPU call R: threads per warm-up chain (64 / 128 / 256) with the coherent-video predictor on
mkdir -p gpurun_out/r4r
export TMPDIR=/tmp
O=gpurun_out/r4r
export VDET_LINK_COHERENT=1
VDET_WARM_THREADS=64 timeout 900 python -m pytest tests/test_track_volume_gpu.py tests/test_link_golden_gpu.py tests/test_config2_full_gpu.py -q -x > $O/quick64.log 2>&1; echo "quick64 rc=$?" | tee -a $O/rc.txt
B="python bench.py --no-cpu --no-upload --no-latency-leg"
for t in 256 128 64; do VDET_WARM_THREADS=$t timeout 900 $B > $O/bench_w$t.json 2> $O/bench_w$t.err; echo "w$t rc=$?" | tee -a $O/rc.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4r/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); vc=d.get('value_coherent') or {}
        print(f, round(d['ms_per_step'],3), d.get('single_video_ms'), d['roofline']['stages']['track_link']['ms_per_step'], 'coherent:', vc.get('ms_per_step'), vc.get('single_video_ms'), vc.get('link_steps_memo_scanned'), {k:vc.get('stage_ms_one_video',{}).get(k) for k in ('track_link','track_loop')})
    except Exception as e: print(f, 'ERR', e)
PY
