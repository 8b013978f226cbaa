cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/r3_t5_full.log
timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b5.json 2> $O/r3_b5.err
VDET_WALK_PACKED=1 timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b5_p1.json 2> $O/r3_b5_p1.err
tail -n 8 $O/r3_t5_full.log
for f in r3_b5 r3_b5_p1; do python -c "
import json,sys
d=json.load(open('$O/$f.json'))
print('$f', round(d['ms_per_step'],3), 'single', round(d['single_video_ms'],3), {k:round(v['ms_per_step'],2) for k,v in d['roofline']['stages'].items()})
"; done
