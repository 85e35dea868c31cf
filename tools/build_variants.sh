#!/bin/bash
# debug / measurement variants of the library (never loaded by the product path); pass the names you want, default: timing
#   timing    libmwgpu_timing.so        -DMW_SOLVER_TIMING                  s_memtime clocks: 8 solver phases + 6 pipeline stages (tools/mix_timing.py, tools/experiments/outlier_probe.py)
#   fine      libmwgpu_timing_fine.so   -DMW_SOLVER_TIMING -DMW_SOLVE_FINE   the solver slots hold the pieces of solve_wave (mw_solve_wave.hpp), slot 0 the whole step
#   step      libmwgpu_timing_step.so   -DMW_SOLVER_TIMING -DMW_STEP_FINE    the solver slots hold integrate / observation / reward / outputs / the halves of make_constraints
#   bounds    libmwgpu_bounds.so        -DMW_BOUNDS -DMW_SOLVER_TIMING       range-checked column store / scratchpad accesses (fault hunting)
#   split     libmwgpu_split.so         -DMW_SPLIT_COLLISION                 the split-collision experiment (also built by __graft_entry__.build for its test)
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -ffp-contract=off"
[ $# -eq 0 ] && set -- timing
for v in "$@"; do
  case $v in
    timing) $HIPCC $F -DMW_SOLVER_TIMING -o metaworld_amd/libmwgpu_timing.so metaworld_amd/csrc/mwgpu.hip & ;;
    fine)   $HIPCC $F -DMW_SOLVER_TIMING -DMW_SOLVE_FINE -o metaworld_amd/libmwgpu_timing_fine.so metaworld_amd/csrc/mwgpu.hip & ;;
    step)   $HIPCC $F -DMW_SOLVER_TIMING -DMW_STEP_FINE -o metaworld_amd/libmwgpu_timing_step.so metaworld_amd/csrc/mwgpu.hip & ;;
    bounds) $HIPCC $F -DMW_BOUNDS -DMW_SOLVER_TIMING -o metaworld_amd/libmwgpu_bounds.so metaworld_amd/csrc/mwgpu.hip & ;;
    split)  $HIPCC $F -DMW_SPLIT_COLLISION -o metaworld_amd/libmwgpu_split.so metaworld_amd/csrc/mwgpu.hip & ;;
    *) echo "unknown variant $v"; exit 2;;
  esac
done
wait
ls -la metaworld_amd/*.so
