# SQ counters of the sort alone (devtools/bench_sort.py): gpurun_out/r3_sq1.csv, r3_sq2.csv
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
CMD2="python $R/devtools/bench_sort.py 100 10000 200 rand"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/$O/p_s -o s -- $CMD2 > $R/$O/p_s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES -d $R/$O/p_s2 -o s -- $CMD2 > $R/$O/p_s2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG -d $R/$O/p_s3 -o s -- $CMD2 > $R/$O/p_s3.log 2>&1
cd $R
python profiles/sq_summarize.py $O/p_s/s_results.db $O/r3_sq1.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_s2/s_results.db $O/r3_sq2.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_s3/s_results.db $O/r3_sq3.csv > /dev/null 2>> $O/p_sum.err
rm -rf $O/p_s $O/p_s2 $O/p_s3
cat $O/r3_sq1.csv $O/r3_sq2.csv $O/r3_sq3.csv; tail -3 $O/p_s3.log
