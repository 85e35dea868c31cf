#!/bin/bash
mkdir -p gpurun_out/r2k
python -m pytest tests -m gpu -q > gpurun_out/r2k/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest_gpu.log
tail -6 gpurun_out/r2k/pytest_gpu.log | cut -c1-250
python bench.py > gpurun_out/r2k/bench_default.log 2>&1; echo "bench rc=$?" >> gpurun_out/r2k/bench_default.log
tail -2 gpurun_out/r2k/bench_default.log | cut -c1-400
python bench.py --precision fp32 --no-cpu-baseline > gpurun_out/r2k/bench_fp32.log 2>&1
tail -1 gpurun_out/r2k/bench_fp32.log | cut -c1-400
python tools/policy_eval_device.py 2500 1 fp64 MT50 > gpurun_out/r2k/policy_gate_device_fp64.txt 2>&1
tail -3 gpurun_out/r2k/policy_gate_device_fp64.txt
python tools/policy_eval_device.py 2500 1 fp32 MT50 > gpurun_out/r2k/policy_gate_device_fp32.txt 2>&1
tail -3 gpurun_out/r2k/policy_gate_device_fp32.txt
MW_LANES_PER_BLOCK=4 python tools/per_task_timing.py 82 100 fp64 250 > gpurun_out/r2k/per_task.txt 2>&1
tail -25 gpurun_out/r2k/per_task.txt
