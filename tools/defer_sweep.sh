#!/bin/bash
# deferred against in-place refinement of the single-minute kernel over batch sizes (GPU box); episodes start at
# random times of day, so every launch sees the day's mix of meal phases
mkdir -p gpurun_out
for n in 1024 16384 65536 131072 262144 524288 2097152; do
  echo "== envs $n"
  timeout -k 10 120 python tools/ab_step.py --envs $n --variants split_adapt_defer_reg,split_adapt_inplace_lds,split_reg --rounds 3 --steps 200 2>&1 | grep "^mod30"
done
