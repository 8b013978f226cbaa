set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_argsort_gpu.py -x -q 2>&1 | tail -25 > $O/r3_t1.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/r3_t1_full.log
timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b1.json 2> $O/r3_b1.err
timeout 600 python bench.py --no-cpu --steps 6 --warmup 3 --streams 1 > $O/r3_b1_s1.json 2> $O/r3_b1_s1.err
VDET_BINSORT=0 timeout 600 python bench.py --no-cpu --steps 6 --warmup 3 --streams 1 > $O/r3_b1_s1_lsd.json 2> $O/r3_b1_s1_lsd.err
tail -5 $O/r3_t1.log $O/r3_t1_full.log
