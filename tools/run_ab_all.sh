#!/bin/bash
# run tools/ab_step.py against every experimental build under build/exp (GPU box)
mkdir -p gpurun_out
for so in build/exp/libt1d_*.so; do
  tag=$(basename $so .so)
  T1D_LIB_PATH=$PWD/$so timeout -k 10 200 python tools/ab_step.py "$@" > gpurun_out/ab_$tag.log 2>&1
  echo "== $tag rc=$?"; grep -v "^{" gpurun_out/ab_$tag.log | grep -v amdgpu.ids
done
