#!/bin/bash
# round 4: a 300-step soak of the timed configuration (same line format; the steady-state number)
mkdir -p gpurun_out/r4soak
export TMPDIR=/tmp
timeout 900 python bench.py --steps 300 --warmup 6 --no-cpu --no-upload --no-coherent --no-latency-leg > gpurun_out/r4soak/bench_soak.json 2> gpurun_out/r4soak/bench_soak.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4soak/bench_soak.json').read().strip().splitlines()[-1])
print(d['steps'], d['ms_per_step'], d['value'], d['timed_check'])
PY
