set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
B="timeout 300 python bench.py --no-cpu"
VDET_VPASS_NOKEYS=1 $B --streams 1 --steps 4 --no-link > $O/r2_b3_nokeys.json 2> $O/r2_b3_nokeys.err
$B --streams 1 --steps 4 --no-link > $O/r2_b3_nolink.json 2> $O/r2_b3_nolink.err
GPU_MAX_HW_QUEUES=8 $B > $O/r2_b3_q8.json 2> $O/r2_b3_q8.err
GPU_MAX_HW_QUEUES=8 $B --gate none > $O/r2_b3_q8_nogate.json 2> $O/r2_b3_q8_nogate.err
$B --streams 3 --steps 18 > $O/r2_b3_st3.json 2> $O/r2_b3_st3.err
$B --streams 3 --steps 18 --gate none > $O/r2_b3_st3_nogate.json 2> $O/r2_b3_st3_nogate.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/r2_prof3 -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --streams 1 > $R/$O/r2_prof3.log 2>&1
cd $R
python profiles/dispatch_times.py $O/r2_prof3/k_results.db track_link 40 > $O/r2_link_dispatch.txt 2>&1
python profiles/dispatch_times.py $O/r2_prof3/k_results.db track_pick 40 >> $O/r2_link_dispatch.txt 2>&1
python profiles/summarize.py $O/r2_prof3/k_results.db $O/r2_kernel_stats3.csv "python bench.py --steps 2 --warmup 1 --no-cpu --streams 1" > /dev/null 2>&1
rm -rf $O/r2_prof3
