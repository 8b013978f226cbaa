#!/bin/bash
# round 4, GPU call S: the suite at HEAD (coherent predictor on), then the profile set
mkdir -p gpurun_out/r4s
export TMPDIR=/tmp
O=gpurun_out/r4s
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_default.log 2>&1; echo "suite_default rc=$?" | tee -a $O/rc.txt
bash devtools/gpu_profile_r4.sh > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/rc.txt
ls -la gpurun_out/r04_* | tee -a $O/rc.txt
