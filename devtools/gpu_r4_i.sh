#!/bin/bash
# round 4, GPU call I: (1) why do warm_order_kernel / track_warm_anchors_kernel read 60x / 2x longer under rocprofv3 than in round 3?
# (2) the drain's division-free edge rule
mkdir -p gpurun_out/r4i
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4i/p1 -o k -- python $R/devtools/bench_nms_stages.py track > $R/gpurun_out/r4i/p1.log 2>&1
VDET_LINK_LPT=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4i/p2 -o k -- python $R/devtools/bench_nms_stages.py track > $R/gpurun_out/r4i/p2.log 2>&1
cd $R
python profiles/summarize.py gpurun_out/r4i/p1/k_results.db gpurun_out/r4i/stats_track.csv "bench_nms_stages.py track" > /dev/null 2>&1
python profiles/summarize.py gpurun_out/r4i/p2/k_results.db gpurun_out/r4i/stats_track_nolpt.csv "VDET_LINK_LPT=0 bench_nms_stages.py track" > /dev/null 2>&1
python - <<'P' > gpurun_out/r4i/dispatches.txt 2>&1
import sqlite3, glob
db = glob.glob('gpurun_out/r4i/p1/*results.db')[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
print(tabs)
kt = [t for t in tabs if 'kernel_dispatch' in t]
print(kt)
for t in kt[:1]:
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
    print(cols)
P
rm -rf gpurun_out/r4i/p1 gpurun_out/r4i/p2
head -20 gpurun_out/r4i/stats_track.csv
timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 > gpurun_out/r4i/stages.log
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_config2_full_gpu.py > gpurun_out/r4i/suite.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/r4i/rc.txt
cat gpurun_out/r4i/stages.log
