cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r3_t11_full.log
tail -n 5 $O/r3_t11_full.log | head -3
for m in 1 0; do
VDET_LINK_YBANDS=$m timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b11_$m.json 2> $O/r3_b11_$m.err
python -c "
import json
d=json.load(open('$O/r3_b11_$m.json'))
print('ybands=$m', round(d['ms_per_step'],3), 'single', round(d['single_video_ms'],3), {k:round(v['ms_per_step'],2) for k,v in d['roofline']['stages'].items()})
"
done
