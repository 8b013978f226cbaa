cd $GRAFT_REPO_ROOT
O=gpurun_out
VDET_BINSORT=1 timeout 300 python devtools/bench_sort.py 300 10000 200 rand > $O/r3_s2.log 2>&1
VDET_BINSORT=1 timeout 300 python devtools/bench_sort.py 300 10000 200 randn >> $O/r3_s2.log 2>&1
timeout 300 python devtools/bench_sort.py 300 10000 200 rand >> $O/r3_s2.log 2>&1
timeout 600 python -m pytest tests/test_argsort_gpu.py -q 2>&1 | tail -25 > $O/r3_t2.log
grep -v amdgpu.ids $O/r3_s2.log; tail -n 6 $O/r3_t2.log
