set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/r2_prof6 -o k -- python $R/bench.py --steps 12 --warmup 3 --no-cpu > $R/$O/r2_prof6.log 2>&1
cd $R
python profiles/timeline.py $O/r2_prof6/k_results.db 0 5 14 > $O/r2_timeline.txt 2>&1
rm -rf $O/r2_prof6
