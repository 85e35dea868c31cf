#!/bin/bash
# debug / measurement variants of the library (never loaded by the product path):
#   libmwgpu_timing.so   -DMW_SOLVER_TIMING   s_memtime phase timers (tools/solver_timing.py)
#   libmwgpu_bounds.so   -DMW_BOUNDS -DMW_SOLVER_TIMING   range-checked column store / scratchpad accesses (fault hunting)
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -ffp-contract=off"
$HIPCC $F -DMW_SOLVER_TIMING -o metaworld_amd/libmwgpu_timing.so metaworld_amd/csrc/mwgpu.hip &
$HIPCC $F -DMW_BOUNDS -DMW_SOLVER_TIMING -o metaworld_amd/libmwgpu_bounds.so metaworld_amd/csrc/mwgpu.hip &
wait
ls -la metaworld_amd/*.so
