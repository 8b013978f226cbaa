#!/bin/bash
# experiments only: A/B of ENVIRONMENT settings on one box, alternating, 4 videos in flight and one at a time
#   devtools/ab_env.sh <tag> "VAR=a" "VAR=b" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out; mkdir -p $O; P=$1; shift
for rep in 1 2; do i=0; for ev in "$@"; do i=$((i+1))
  env $ev timeout 300 python bench.py --profile --no-sharded-leg --steps 32 --warmup 8 > $O/${P}_${i}_$rep.json 2> $O/${P}_${i}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/${P}_${i}_$rep.json").read().split("\n")[0])
print("$ev", $rep, "ms_per_step %.3f"%d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["roofline"]["stages"].items() if v["ms_per_step"]>0.3})
PY
done; done
