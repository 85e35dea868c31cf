#!/bin/bash
mkdir -p gpurun_out/r2m
for v in timing v1 v2; do
  echo "== variant $v (fp64)"
  MW_LIB=libmwgpu_$v.so MW_LANES_PER_BLOCK=4 MW_PREC=fp64 MW_NWIN=7 timeout 400 python tools/solver_timing.py 82 peg-unplug-side-v3 door-open-v3 plate-slide-v3 2>&1 | grep "steps 250-300\|steps 300-350" | cut -c1-330
done
echo "== baseline fp32"
MW_LIB=libmwgpu_timing.so MW_LANES_PER_BLOCK=4 MW_PREC=fp32 MW_NWIN=7 timeout 400 python tools/solver_timing.py 82 peg-unplug-side-v3 door-open-v3 plate-slide-v3 2>&1 | grep "steps 250-300\|steps 300-350" | cut -c1-330
