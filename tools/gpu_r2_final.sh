#!/bin/bash
# end-of-round evidence run (GPU box): GPU tests, default bench, the reference's gate on the device, rocprof passes of both precisions
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest_gpu.log
tail -4 gpurun_out/final/pytest_gpu.log | cut -c1-250
python bench.py > gpurun_out/final/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/final/bench_default.log
tail -2 gpurun_out/final/bench_default.log | cut -c1-300
python tools/policy_gate_gpu.py fp64 > gpurun_out/final/policy_gate_gpu_fp64.txt 2>&1; tail -1 gpurun_out/final/policy_gate_gpu_fp64.txt
python tools/policy_gate_gpu.py fp32 > gpurun_out/final/policy_gate_gpu_fp32.txt 2>&1; tail -1 gpurun_out/final/policy_gate_gpu_fp32.txt
bash tools/profile_bench.sh r02_mt50_fp64 > gpurun_out/final/prof_fp64.log 2>&1; tail -1 gpurun_out/final/prof_fp64.log
bash tools/profile_bench.sh r02_mt50_fp32 --precision fp32 > gpurun_out/final/prof_fp32.log 2>&1; tail -1 gpurun_out/final/prof_fp32.log
