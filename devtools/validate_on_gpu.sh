set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/val_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/val_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/val_bench.json 2> gpurun_out/val_bench.err
timeout 600 python bench.py --no-link --no-cpu > gpurun_out/val_bench_nolink.json 2> gpurun_out/val_bench_nolink.err
# N > 1 control flow dry run on this single GPU (both ranks on device 0, gloo): barriers, exchange step, MAX over ranks
VDET_BENCH_ONE_GPU=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu --frames 20 > gpurun_out/val_bench_w2.json 2> gpurun_out/val_bench_w2.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o r1 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu --streams 1 > $R/gpurun_out/prof_r1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1 > $R/gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --streams 1 > $R/gpurun_out/pmc_w.log 2>&1
ls -la $R/gpurun_out/prof_r1 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
