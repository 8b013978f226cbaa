# round-4 profile set at HEAD: kernel stats of the bench command (one video at a time), HBM traffic (two PMC passes, kernel trace
# only), the VID-shape batch, and -- for the record of the measured-and-lost bucket path -- the same kernel stats with VDET_BUCKETS=1
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
P=r04
CMD1="python $R/bench.py --steps 3 --warmup 2 --no-cpu --no-upload --no-coherent --no-latency-leg --streams 1"
CMD2="python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-upload --no-coherent --no-latency-leg --streams 1"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/p_k -o k -- $CMD1 > $R/$O/p_k.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/p_f -o f -- $CMD2 > $R/$O/p_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/p_w -o w -- $CMD2 > $R/$O/p_w.log 2>&1
VDET_BUCKETS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/p_kb -o k -- $CMD1 > $R/$O/p_kb.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/p_v -o k -- python $R/devtools/bench_vid.py 64 > $R/$O/p_v.log 2>&1
cd $R
python profiles/summarize.py $O/p_k/k_results.db $O/${P}_kernel_stats.csv "python bench.py --steps 3 --warmup 2 --no-cpu --no-upload --no-coherent --no-latency-leg --streams 1 (one video at a time)" > /dev/null 2>> $O/p_sum.err
python profiles/pmc_summarize.py $O/p_f/f_results.db $O/p_w/w_results.db $O/${P}_pmc_hbm_traffic.csv $O/${P}_pmc_traffic.json > /dev/null 2>> $O/p_sum.err
python profiles/summarize.py $O/p_kb/k_results.db $O/${P}_bucket_path_kernel_stats.csv "VDET_BUCKETS=1 python bench.py --steps 3 --warmup 2 --no-cpu --no-upload --no-coherent --no-latency-leg --streams 1 (the bucketed lists: an option, not the default)" > /dev/null 2>> $O/p_sum.err
python profiles/summarize.py $O/p_v/k_results.db $O/${P}_vid_batch_kernel_stats.csv "python devtools/bench_vid.py 64 (2 batched runs of 64 VID-shaped videos + 2 single-video runs)" > /dev/null 2>> $O/p_sum.err
rm -rf $O/p_k $O/p_f $O/p_w $O/p_kb $O/p_v
tail -5 $O/p_sum.err
