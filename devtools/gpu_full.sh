set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r2_tfull.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_smoke.log 2>&1
timeout 900 python bench.py > $O/r2_bfull.json 2> $O/r2_bfull.err
timeout 600 python bench.py --scores randn --no-cpu > $O/r2_bfull_randn.json 2> $O/r2_bfull_randn.err
