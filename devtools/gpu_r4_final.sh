#!/bin/bash
# round 4, last GPU call: the suite at HEAD, then the profile set (kernel stats, HBM traffic, VID batch, bucket path)
mkdir -p gpurun_out/r4final
export TMPDIR=/tmp
O=gpurun_out/r4final
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a $O/rc.txt
grep -h "passed\|failed" $O/suite_default.log
bash devtools/gpu_profile_r4.sh > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/rc.txt
timeout 900 python bench.py --no-cpu --no-upload > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4final/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('single_video_ms'), (d.get('value_coherent') or {}).get('ms_per_step'))
PY
