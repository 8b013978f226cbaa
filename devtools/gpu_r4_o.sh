#!/bin/bash
# round 4, GPU call O: K2 variants (old/new word loop x old/new copy-out, guarded prefetch, batches of 8)
mkdir -p gpurun_out/r4o
export TMPDIR=/tmp
L=gpurun_out/r4o/stages.log
cp vdetlib_amd/libvdet_hip.so /tmp/libD.so
for v in D B C E G H; do
  if [ $v != D ]; then cp devtools/k2ab/lib$v.so vdetlib_amd/libvdet_hip.so; fi
  echo "== $v" >> $L; timeout 200 python devtools/bench_nms_stages.py track 2>&1 | tail -n 2 >> $L
done
cp /tmp/libD.so vdetlib_amd/libvdet_hip.so
cat $L
