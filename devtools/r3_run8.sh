cd $GRAFT_REPO_ROOT
O=gpurun_out
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b8_$i.json 2> $O/r3_b8_$i.err
python -c "
import json
d=json.load(open('$O/r3_b8_$i.json'))
print(round(d['ms_per_step'],3), 'single', round(d['single_video_ms'],3))
"
done
