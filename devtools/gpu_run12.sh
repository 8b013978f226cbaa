set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r2_t12.log
B="timeout 300 python bench.py --no-cpu"
$B --streams 1 --steps 6 > $O/r2_b12_s1.json 2> $O/r2_b12_s1.err
$B > $O/r2_b12.json 2> $O/r2_b12.err
VDET_RESCORE_ADJ=0 $B > $O/r2_b12_noadj.json 2> $O/r2_b12_noadj.err
