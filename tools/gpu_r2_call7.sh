#!/bin/bash
mkdir -p gpurun_out/r2d
python tools/experiments/probe_solver.py reach-v3 box-close-v3 2>&1 | grep -v amdgpu.ids | cut -c1-330
python -m pytest tests/test_gpu_parity.py -x -q -k "not policy_episode" > gpurun_out/r2d/pytest_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest_parity.log
tail -6 gpurun_out/r2d/pytest_parity.log
MW_PREC=fp64 MW_NWIN=6 timeout 600 python tools/solver_timing.py 82 box-close-v3 peg-unplug-side-v3 > gpurun_out/r2d/solver_fp64.txt 2>&1
cat gpurun_out/r2d/solver_fp64.txt | grep -v amdgpu.ids
MW_PREC=fp32 MW_NWIN=6 timeout 600 python tools/solver_timing.py 82 box-close-v3 > gpurun_out/r2d/solver_fp32.txt 2>&1
cat gpurun_out/r2d/solver_fp32.txt | grep -v amdgpu.ids
