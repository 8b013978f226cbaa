#!/bin/bash
# round 4, GPU call Y: profile set at HEAD + the full bench line
mkdir -p gpurun_out/r4y
export TMPDIR=/tmp
O=gpurun_out/r4y
bash devtools/gpu_profile_r4.sh > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/rc.txt
timeout 1500 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4y/bench_full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('single_video_ms'))
print({k:round(v['ms_per_step'],3) for k,v in d['roofline']['stages'].items()})
print('vid', d.get('vid_shape')); print('coh', {k:(d.get('value_coherent') or {}).get(k) for k in ('ms_per_step','single_video_ms')})
print('c1', d.get('c1_reference_flow')); print('cpu', d.get('cpu_baseline')); print('timed', d.get('timed_check'))
PY
