#!/bin/bash
# experiments only: A/B of library builds on ONE box, alternating (a b a b), one video at a time: per-stage ms
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out; mkdir -p $O; P=$1; shift
for rep in 1 2; do for so in "$@"; do
  n=$(basename $so .so)
  timeout 300 python devtools/ab_lib.py $so --profile --no-sharded-leg --streams 1 --steps 10 --warmup 3 > $O/${P}_${n}_$rep.json 2> $O/${P}_${n}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/${P}_${n}_$rep.json").read().split("\n")[0])
print("$n", $rep, "ms_per_step %.3f"%d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["roofline"]["stages"].items() if v["ms_per_step"]>0.3})
PY
done; done
