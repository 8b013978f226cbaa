set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
B="timeout 300 python bench.py --no-cpu"
$B > $O/r2_b13_a.json 2> $O/r2_b13_a.err
$B > $O/r2_b13_b.json 2> $O/r2_b13_b.err
VDET_RESCORE_ADJ=0 $B > $O/r2_b13_noadj.json 2> $O/r2_b13_noadj.err
$B --streams 1 --steps 6 > $O/r2_b13_s1.json 2> $O/r2_b13_s1.err
$B --steps 24 > $O/r2_b13_c.json 2> $O/r2_b13_c.err
