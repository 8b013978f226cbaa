cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_lazy_order_gpu.py tests/test_argsort_gpu.py -x -q 2>&1 | tail -30 > $O/r3_t9.log
tail -n 30 $O/r3_t9.log
