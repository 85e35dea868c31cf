#!/bin/bash
mkdir -p gpurun_out/r2j
python -m pytest tests -m gpu -q > gpurun_out/r2j/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest_gpu.log
tail -6 gpurun_out/r2j/pytest_gpu.log | cut -c1-250
MW_PREC=fp64 MW_NWIN=3 timeout 600 python tools/solver_timing.py 82 peg-unplug-side-v3 box-close-v3 hammer-v3 > gpurun_out/r2j/solver_fp64.txt 2>&1
grep -v amdgpu.ids gpurun_out/r2j/solver_fp64.txt | cut -c1-330
python bench.py > gpurun_out/r2j/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2j/bench_default.log
tail -2 gpurun_out/r2j/bench_default.log | cut -c1-400
python bench.py --precision fp32 --no-cpu-baseline > gpurun_out/r2j/bench_fp32.log 2>&1
tail -1 gpurun_out/r2j/bench_fp32.log | cut -c1-400
