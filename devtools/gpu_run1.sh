# round 2, GPU call 1: parity of everything new + first measurements (run from the repo root on the GPU box)
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_config2_full_gpu.py 2>&1 | tail -25 > $O/r2_t1.log
timeout 900 python -m pytest tests/test_config2_full_gpu.py -m gpu -q 2>&1 | tail -25 > $O/r2_t1_full.log
timeout 120 devtools/valu_bench > $O/r2_valu.csv 2>&1
timeout 600 python bench.py > $O/r2_b1.json 2> $O/r2_b1.err
timeout 300 python bench.py --no-cpu --streams 1 --steps 6 > $O/r2_b1_s1.json 2> $O/r2_b1_s1.err
VDET_LINK_THREADS=64 timeout 300 python bench.py --no-cpu --streams 1 --steps 6 > $O/r2_b1_s1_lt64.json 2> $O/r2_b1_s1_lt64.err
VDET_LINK_THREADS=128 timeout 300 python bench.py --no-cpu --streams 1 --steps 6 > $O/r2_b1_s1_lt128.json 2> $O/r2_b1_s1_lt128.err
timeout 300 python bench.py --no-cpu --separate-pass > $O/r2_b1_sep.json 2> $O/r2_b1_sep.err
timeout 300 python bench.py --no-cpu --sync-build > $O/r2_b1_sync.json 2> $O/r2_b1_sync.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/r2_prof1 -o k -- python $R/bench.py --steps 3 --warmup 2 --no-cpu --streams 1 > $R/$O/r2_prof1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/r2_pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1 > $R/$O/r2_pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/r2_pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1 > $R/$O/r2_pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/$O/r2_pmc_sq -o s -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1 > $R/$O/r2_pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES -d $R/$O/r2_pmc_sq2 -o s -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1 > $R/$O/r2_pmc_sq2.log 2>&1
cd $R
python profiles/summarize.py $O/r2_prof1/k_results.db $O/r2_kernel_stats.csv "python bench.py --steps 3 --warmup 2 --no-cpu --streams 1" > /dev/null 2>> $O/r2_sum.err
python profiles/pmc_summarize.py $O/r2_pmc_f/f_results.db $O/r2_pmc_w/w_results.db $O/r2_pmc_hbm_traffic.csv $O/r2_pmc_traffic.json > /dev/null 2>> $O/r2_sum.err
python profiles/sq_summarize.py $O/r2_pmc_sq/s_results.db $O/r2_pmc_sq.csv > /dev/null 2>> $O/r2_sum.err
python profiles/sq_summarize.py $O/r2_pmc_sq2/s_results.db $O/r2_pmc_sq2.csv > /dev/null 2>> $O/r2_sum.err
find $O/r2_prof1 $O/r2_pmc_f $O/r2_pmc_w $O/r2_pmc_sq $O/r2_pmc_sq2 -type f | head -30 > $O/r2_files.txt
rm -rf $O/r2_prof1 $O/r2_pmc_f $O/r2_pmc_w $O/r2_pmc_sq $O/r2_pmc_sq2
