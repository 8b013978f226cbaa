cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_argsort_gpu.py -q 2>&1 | tail -8 > $O/r3_t3.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r3_t3_full.log
timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b3.json 2> $O/r3_b3.err
VDET_BINSORT=0 timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b3_lsd.json 2> $O/r3_b3_lsd.err
timeout 600 python bench.py --no-cpu --steps 6 --warmup 3 --streams 1 > $O/r3_b3_s1.json 2> $O/r3_b3_s1.err
VDET_BINSORT=0 timeout 600 python bench.py --no-cpu --steps 6 --warmup 3 --streams 1 > $O/r3_b3_s1_lsd.json 2> $O/r3_b3_s1_lsd.err
tail -n 4 $O/r3_t3.log $O/r3_t3_full.log
for f in r3_b3 r3_b3_lsd r3_b3_s1 r3_b3_s1_lsd; do python -c "
import json,sys
d=json.load(open('$O/$f.json'))
print('$f', round(d['ms_per_step'],3), {k:round(v['ms_per_step'],2) for k,v in d['roofline']['stages'].items()})
"; done
