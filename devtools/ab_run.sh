#!/bin/bash
# experiments only: A/B of library builds.  devtools/gpu.sh 'bash devtools/ab_run.sh <tag> <so> [<so> ...]'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out; mkdir -p $O
P=$1; shift
for so in "$@"; do
  n=$(basename $so .so)
  timeout 300 python devtools/ab_lib.py $so --profile --no-sharded-leg --streams 1 --steps 8 --warmup 3 --inputs device > $O/${P}_${n}_s1.json 2> $O/${P}_${n}_s1.err
  timeout 300 python devtools/ab_lib.py $so --profile --no-sharded-leg --steps 32 --warmup 8 --inputs device > $O/${P}_${n}_s4.json 2> $O/${P}_${n}_s4.err
  python - <<PY
import json
for tag in ("s1","s4"):
    try:
        d=json.loads(open("$O/${P}_${n}_%s.json"%tag).read().split("\n")[0])
        print("$n", tag, "ms_per_step %.3f"%d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["roofline"]["stages"].items() if v["ms_per_step"]>0.3}, d["timed_check"]["timed_outputs_identical"])
    except Exception as e:
        print("$n", tag, "failed", e, open("$O/${P}_${n}_%s.err"%tag).read()[-300:])
PY
done
