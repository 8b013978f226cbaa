# latency counters of the NMS kernels: average VMEM / LDS / SMEM latency = SQ_INST_LEVEL_x / SQ_INSTS_x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
CMD2="python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1 --no-link"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $R/$O/p_s -o s -- $CMD2 > $R/$O/p_s3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum -d $R/$O/p_s2 -o s -- $CMD2 > $R/$O/p_s4.log 2>&1
cd $R
python profiles/sq_summarize.py $O/p_s/s_results.db $O/sq3.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_s2/s_results.db $O/sq4.csv > /dev/null 2>> $O/p_sum.err
tail -5 $O/p_s3.log $O/p_s4.log
rm -rf $O/p_s $O/p_s2
