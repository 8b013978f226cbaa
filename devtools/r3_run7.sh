cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_nms_gpu.py tests/test_volume_gpu.py -x -q 2>&1 | tail -4 > $O/r3_t7.log
timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 > $O/r3_b7.json 2> $O/r3_b7.err
tail -n 4 $O/r3_t7.log
python -c "
import json
d=json.load(open('$O/r3_b7.json'))
print(round(d['ms_per_step'],3), 'single', round(d['single_video_ms'],3), {k:round(v['ms_per_step'],2) for k,v in d['roofline']['stages'].items()})
"
