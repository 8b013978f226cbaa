# round-4 SQ counter passes at HEAD (kernel trace only; 8 counters per pass): the c2 step and the VID-shape batch
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
P=r04
CMD2="python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-upload --no-coherent --no-latency-leg --streams 1"
VID="python $R/devtools/bench_vid.py 64"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/$O/p_s -o s -- $CMD2 > $R/$O/p_s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES -d $R/$O/p_s2 -o s -- $CMD2 > $R/$O/p_s2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/$O/p_vs -o s -- $VID > $R/$O/p_vs.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES -d $R/$O/p_vs2 -o s -- $VID > $R/$O/p_vs2.log 2>&1
cd $R
python profiles/sq_summarize.py $O/p_s/s_results.db $O/${P}_pmc_sq.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_s2/s_results.db $O/${P}_pmc_sq2.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_vs/s_results.db $O/${P}_pmc_vid_sq.csv > /dev/null 2>> $O/p_sum.err
python profiles/sq_summarize.py $O/p_vs2/s_results.db $O/${P}_pmc_vid_sq2.csv > /dev/null 2>> $O/p_sum.err
rm -rf $O/p_s $O/p_s2 $O/p_vs $O/p_vs2
tail -3 $O/p_sum.err
head -8 $O/${P}_pmc_sq.csv | cut -c1-250
head -10 $O/${P}_pmc_vid_sq.csv | cut -c1-250
