cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -30 > $O/r3_t4_full.log
timeout 900 python bench.py > $O/r3_b4.json 2> $O/r3_b4.err
VDET_TRACK_LOOP=0 timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b4_noloop.json 2> $O/r3_b4_noloop.err
tail -n 30 $O/r3_t4_full.log
for f in r3_b4 r3_b4_noloop; do python -c "
import json,sys
d=json.load(open('$O/$f.json'))
print('$f', round(d['ms_per_step'],3), 'single', round(d['single_video_ms'],3), d.get('value_other_scores'), d.get('exchange'), {k:round(v['ms_per_step'],2) for k,v in d['roofline']['stages'].items()})
"; done
tail -3 $O/r3_b4.err
