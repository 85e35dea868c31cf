#!/bin/bash
# GPU call 2 of round 2: full GPU test-suite, the default bench line, then the fault hunt (bounds-checked build)
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest_gpu.log
tail -15 gpurun_out/r2b/pytest_gpu.log
python bench.py > gpurun_out/r2b/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2b/bench_default.log
tail -3 gpurun_out/r2b/bench_default.log | cut -c1-3000
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/bench_short.log 2>&1
tail -1 gpurun_out/r2b/bench_short.log | cut -c1-600
MW_LIB=libmwgpu_bounds.so MW_PREC=fp64 MW_NWIN=3 timeout 300 python tools/solver_timing.py 82 stick-pull-v3 stick-push-v3 > gpurun_out/r2b/bounds_stick.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/bounds_stick.log
cat gpurun_out/r2b/bounds_stick.log | tail -12
MW_PREC=fp64 MW_NWIN=3 timeout 300 python tools/solver_timing.py 82 stick-pull-v3 assembly-v3 hammer-v3 > gpurun_out/r2b/timing_stick.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/timing_stick.log
cat gpurun_out/r2b/timing_stick.log | tail -14
