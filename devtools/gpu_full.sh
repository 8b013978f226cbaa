set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/r3_tfull.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r3_smoke.log 2>&1
timeout 900 python bench.py > $O/r3_bfull.json 2> $O/r3_bfull.err
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > $O/r3_bfull2.json 2> $O/r3_bfull2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 12 --warmup 4 --no-cpu --force-exchange > $O/r3_bfull_x.json 2> $O/r3_bfull_x.err
tail -4 $O/r3_tfull.log; cat $O/r3_smoke.log | tail -2
