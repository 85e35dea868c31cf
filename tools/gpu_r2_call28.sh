#!/bin/bash
mkdir -p gpurun_out/r2l
MW_LANES_PER_BLOCK=4 MW_PREC=fp64 MW_NWIN=7 timeout 600 python tools/solver_timing.py 82 plate-slide-back-side-v3 plate-slide-v3 door-open-v3 peg-unplug-side-v3 assembly-v3 > gpurun_out/r2l/solver_fp64.txt 2>&1
grep -v amdgpu.ids gpurun_out/r2l/solver_fp64.txt | cut -c1-330
python tools/policy_gate_gpu.py fp64 > gpurun_out/r2l/policy_gate_gpu_fp64.txt 2>&1
grep -v "50/50\|amdgpu.ids" gpurun_out/r2l/policy_gate_gpu_fp64.txt | cut -c1-200
python tools/policy_gate_gpu.py fp32 > gpurun_out/r2l/policy_gate_gpu_fp32.txt 2>&1
grep -v "50/50\|amdgpu.ids" gpurun_out/r2l/policy_gate_gpu_fp32.txt | cut -c1-200
