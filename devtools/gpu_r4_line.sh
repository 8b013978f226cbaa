#!/bin/bash
# the complete bench line at the round's last code state
mkdir -p gpurun_out/r4line
export TMPDIR=/tmp
timeout 1500 python bench.py > gpurun_out/r4line/bench_full.json 2> gpurun_out/r4line/bench_full.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4line/bench_full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('single_video_ms'))
print('vid', (d.get('vid_shape') or {}).get('ms_per_video'), 'coh', (d.get('value_coherent') or {}).get('ms_per_step'), (d.get('value_coherent') or {}).get('single_video_ms'))
print('cpu', (d.get('cpu_baseline') or {}).get('value'), 'timed', (d.get('timed_check') or {}).get('timed_outputs_identical'))
PY
