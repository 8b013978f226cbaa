#!/bin/bash
# The ONE script that runs on the GPU box (devtools/gpu.sh ships the tree and calls it):
#     devtools/gpu.sh [--timeout S] 'bash devtools/gpu_run.sh <tag> <what> [<what> ...]'
# <tag> names the outputs under gpurun_out/ (e.g. r05a); <what> is any of
#   tests            the whole -m gpu suite                     -> <tag>_tests.log
#   tests:<expr>     pytest -k <expr>                           -> <tag>_tests.log
#   smoke            __graft_entry__.smoke()                    -> <tag>_smoke.log
#   bench            python bench.py (the driver's command)     -> <tag>_bench.json / .err
#   bench:<args>     python bench.py <args> (use _ for spaces)  -> <tag>_bench_<n>.json
#   kstats           rocprofv3 --kernel-trace --stats of the one-video-at-a-time step  -> <tag>_kernel_stats.csv
#   hbm              two PMC passes (FETCH_SIZE, WRITE_SIZE)                              -> <tag>_pmc_hbm_traffic.csv + <tag>_pmc_traffic.json
#   sq               two SQ counter passes of the same step                               -> <tag>_pmc_sq.csv, <tag>_pmc_sq2.csv
#   tcp              L1 / address-unit counters of the same step                                 -> <tag>_pmc_tcp.csv
#   tcc              L2 hit / miss / request counters of the same step (per-kernel L2 hit rate)    -> <tag>_pmc_tcc.csv
#   vidstats         kernel stats of the VID-shape batch (devtools/bench_vid.py 64)       -> <tag>_vid_batch_kernel_stats.csv
# Counters are collected in their own runs with --kernel-trace only (gpurun refuses other trace domains next to --pmc).
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out
mkdir -p $O
P=$1; shift
STEP="python $R/bench.py --profile --no-sharded-leg --streams 1"
n=0
for what in "$@"; do
  case "$what" in
    tests)   timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/${P}_tests.log; tail -5 $O/${P}_tests.log ;;
    tests:*) timeout 2400 python -m pytest tests -m gpu -q -k "${what#tests:}" 2>&1 | tail -60 > $O/${P}_tests.log; tail -8 $O/${P}_tests.log ;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; tail -2 $O/${P}_smoke.log ;;
    bench)   t0=$SECONDS; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${P}_bench.json 2> $O/${P}_bench.err
             echo "bench wall $((SECONDS - t0)) s" | tee $O/${P}_bench.time; tail -c 600 $O/${P}_bench.err; head -c 400 $O/${P}_bench.json ;;
    bench:*) n=$((n+1)); a="${what#bench:}"; timeout 1200 python bench.py ${a//_/ } > $O/${P}_bench_$n.json 2> $O/${P}_bench_$n.err; head -c 300 $O/${P}_bench_$n.json ;;
    kstats)  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/p_k -o k -- $STEP --steps 3 --warmup 2 > $R/$O/${P}_p_k.log 2>&1)
             python profiles/summarize.py $O/p_k/k_results.db $O/${P}_kernel_stats.csv "python bench.py --profile --no-sharded-leg --streams 1 --steps 3 --warmup 2 (one video at a time; rocprim:: / at:: rows = the input generator and torch glue outside the timed region)" > /dev/null 2>> $O/${P}_sum.err
             rm -rf $O/p_k; head -14 $O/${P}_kernel_stats.csv | cut -c1-160 ;;
    hbm)     (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/p_f -o f -- $STEP --steps 1 --warmup 1 > $R/$O/${P}_p_f.log 2>&1
              timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/p_w -o w -- $STEP --steps 1 --warmup 1 > $R/$O/${P}_p_w.log 2>&1)
             python profiles/pmc_summarize.py $O/p_f/f_results.db $O/p_w/w_results.db $O/${P}_pmc_hbm_traffic.csv $O/${P}_pmc_traffic.json > /dev/null 2>> $O/${P}_sum.err
             rm -rf $O/p_f $O/p_w; head -12 $O/${P}_pmc_hbm_traffic.csv | cut -c1-160 ;;
    sq)      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/$O/p_s -o s -- $STEP --steps 1 --warmup 1 > $R/$O/${P}_p_s.log 2>&1
              timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES -d $R/$O/p_s2 -o s -- $STEP --steps 1 --warmup 1 > $R/$O/${P}_p_s2.log 2>&1)
             python profiles/sq_summarize.py $O/p_s/s_results.db $O/${P}_pmc_sq.csv > /dev/null 2>> $O/${P}_sum.err
             python profiles/sq_summarize.py $O/p_s2/s_results.db $O/${P}_pmc_sq2.csv > /dev/null 2>> $O/${P}_sum.err
             rm -rf $O/p_s $O/p_s2; head -8 $O/${P}_pmc_sq2.csv | cut -c1-200 ;;
    tcc)     (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/$O/p_t -o t -- $STEP --steps 1 --warmup 1 > $R/$O/${P}_p_t.log 2>&1)
             python profiles/sq_summarize.py $O/p_t/t_results.db $O/${P}_pmc_tcc.csv > /dev/null 2>> $O/${P}_sum.err
             rm -rf $O/p_t; head -8 $O/${P}_pmc_tcc.csv | cut -c1-200 ;;
    tcp)     (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $R/$O/p_c -o c -- $STEP --steps 1 --warmup 1 > $R/$O/${P}_p_c.log 2>&1)
             python profiles/sq_summarize.py $O/p_c/c_results.db $O/${P}_pmc_tcp.csv > /dev/null 2>> $O/${P}_sum.err
             rm -rf $O/p_c; head -8 $O/${P}_pmc_tcp.csv | cut -c1-200 ;;
    vidstats) (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/p_v -o k -- python $R/devtools/bench_vid.py 64 > $R/$O/${P}_p_v.log 2>&1)
             python profiles/summarize.py $O/p_v/k_results.db $O/${P}_vid_batch_kernel_stats.csv "python devtools/bench_vid.py 64 (2 batched runs of 64 VID-shaped videos + 2 single-video runs)" > /dev/null 2>> $O/${P}_sum.err
             rm -rf $O/p_v ;;
    *) echo "unknown step $what" ;;
  esac
done
