#!/bin/bash
# GPU call 3: new solver (matrix-core Hessian pass) -- parity tests, stage timing, capacity planning, fault hunt
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_parity.py -x -q -k "not policy_episode" > gpurun_out/r2c/pytest_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/pytest_parity.log
tail -12 gpurun_out/r2c/pytest_parity.log
MW_PREC=fp64 MW_NWIN=6 timeout 600 python tools/solver_timing.py 82 box-close-v3 peg-unplug-side-v3 > gpurun_out/r2c/solver_fp64.txt 2>&1
cat gpurun_out/r2c/solver_fp64.txt | grep -v amdgpu.ids
MW_PREC=fp32 MW_NWIN=6 timeout 600 python tools/solver_timing.py 82 box-close-v3 > gpurun_out/r2c/solver_fp32.txt 2>&1
cat gpurun_out/r2c/solver_fp32.txt | grep -v amdgpu.ids
timeout 900 python tools/measure_caps_gpu.py 4096 1500 > gpurun_out/r2c/caps.log 2>&1; tail -40 gpurun_out/r2c/caps.log
timeout 600 python bench.py --allow-status --no-cpu-baseline > gpurun_out/r2c/bench.log 2>&1; tail -1 gpurun_out/r2c/bench.log | cut -c1-1500
MW_LIB=libmwgpu_timing.so timeout 300 python tools/experiments/fault_hunt.py stick-pull-v3 fp64 8 > gpurun_out/r2c/fault_hunt.log 2>&1; tail -12 gpurun_out/r2c/fault_hunt.log
MW_LIB=libmwgpu_timing.so timeout 300 python tools/experiments/fault_hunt.py hammer-v3 fp64 8 >> gpurun_out/r2c/fault_hunt.log 2>&1; tail -10 gpurun_out/r2c/fault_hunt.log
