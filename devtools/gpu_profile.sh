# round-3 profile set: kernel stats, HBM traffic (two PMC passes), SQ counters (two passes), TCP counters, HIP API trace,
# SQ counters of the sort alone (LSD vs the counting sort), the VID-shape batch
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
P=r03
CMD1="python $R/bench.py --steps 3 --warmup 2 --no-cpu --streams 1"
CMD2="python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/p_k -o k -- $CMD1 > $R/$O/p_k.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/p_f -o f -- $CMD2 > $R/$O/p_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/p_w -o w -- $CMD2 > $R/$O/p_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/$O/p_s -o s -- $CMD2 > $R/$O/p_s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES -d $R/$O/p_s2 -o s -- $CMD2 > $R/$O/p_s2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum -d $R/$O/p_t -o s -- $CMD2 > $R/$O/p_t.log 2>&1
timeout 600 rocprofv3 --hip-runtime-trace --stats -d $R/$O/p_h -o h -- python $R/bench.py --steps 8 --warmup 4 --no-cpu > $R/$O/p_h.log 2>&1
# the sort alone: LSD (default) and the counting sort, SQ counters
SORT="python $R/devtools/bench_sort.py 100 10000 200 rand"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/$O/p_q0 -o s -- $SORT > $R/$O/p_q0.log 2>&1
VDET_BINSORT=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/$O/p_q1 -o s -- $SORT > $R/$O/p_q1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/p_v -o k -- python $R/devtools/bench_vid.py 64 > $R/$O/p_v.log 2>&1
cd $R
python profiles/summarize.py $O/p_k/k_results.db $O/${P}_kernel_stats.csv "python bench.py --steps 3 --warmup 2 --no-cpu --streams 1 (one video at a time; 5 videos + 3 timing repetitions + 8 single-video steps)" > /dev/null 2>> $O/p_sum.err
python profiles/pmc_summarize.py $O/p_f/f_results.db $O/p_w/w_results.db $O/${P}_pmc_hbm_traffic.csv $O/${P}_pmc_traffic.json > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_s/s_results.db $O/${P}_pmc_sq.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_s2/s_results.db $O/${P}_pmc_sq2.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_t/s_results.db $O/${P}_pmc_tcp.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_q0/s_results.db $O/${P}_pmc_sort_lsd.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_q1/s_results.db $O/${P}_pmc_sort_counting.csv > /dev/null 2>> $O/p_sum.err
python profiles/summarize.py $O/p_v/k_results.db $O/${P}_vid_batch_kernel_stats.csv "python devtools/bench_vid.py 64 (2 batched runs of 64 VID-shaped videos + 2 single-video runs)" > /dev/null 2>> $O/p_sum.err
python profiles/hip_api_summarize.py $O/p_h/h_results.db $O/${P}_hip_api_stats.csv >> $O/p_sum.err 2>&1
rm -rf $O/p_k $O/p_f $O/p_w $O/p_s $O/p_s2 $O/p_h $O/p_t $O/p_q0 $O/p_q1 $O/p_v
tail -5 $O/p_sum.err
