cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_volume_pass_gpu.py tests/test_track_volume_gpu.py -x -q 2>&1 | tail -25 > $O/r3_t6.log
timeout 900 python bench.py --steps 6 --warmup 3 --no-upload > $O/r3_b6.json 2> $O/r3_b6.err
tail -n 25 $O/r3_t6.log
python -c "
import json
d=json.load(open('$O/r3_b6.json'))
print(round(d['ms_per_step'],3), json.dumps(d['vid_shape'], indent=1))
"
tail -5 $O/r3_b6.err
