#!/bin/bash
mkdir -p gpurun_out/r2o
for l in 2 4 8 16; do
  echo "== lanes per block $l"
  MW_LANES_PER_BLOCK=$l python tools/per_task_timing.py 82 100 fp64 250 box-close-v3 peg-unplug-side-v3 door-open-v3 assembly-v3 plate-slide-back-side-v3 hammer-v3 sweep-into-v3 reach-v3 2>&1 | grep -v amdgpu.ids
done
