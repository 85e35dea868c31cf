#!/bin/bash
mkdir -p gpurun_out/r2n
python -m pytest tests -m gpu -q > gpurun_out/r2n/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2n/pytest_gpu.log
tail -8 gpurun_out/r2n/pytest_gpu.log | cut -c1-250
python bench.py > gpurun_out/r2n/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2n/bench_default.log
tail -2 gpurun_out/r2n/bench_default.log | cut -c1-400
python bench.py --precision fp32 --no-cpu-baseline > gpurun_out/r2n/bench_fp32.log 2>&1
tail -1 gpurun_out/r2n/bench_fp32.log | cut -c1-400
MW_LANES_PER_BLOCK=4 MW_PREC=fp64 MW_NWIN=7 timeout 400 python tools/solver_timing.py 82 peg-unplug-side-v3 box-close-v3 door-open-v3 2>&1 | grep "steps 250-300\|steps 300-350" | cut -c1-330
