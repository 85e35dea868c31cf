#!/bin/bash
# round-4 evidence call: GPU suite, default bench line, rocprofv3 kernel trace + PMC passes (fp64, fp32), the bench-state recordings for
# tests/golden/, the reference's 50-goal policy gate on the device, the stage / solver-phase split.  Summaries -> gpurun_out/r04/.
set -u
O=gpurun_out/r04; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -q -rA > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc $?"; grep -E "^[0-9]+ passed|passed|failed" $O/pytest_gpu_full.txt | tail -n 2
timeout 400 python bench.py > $O/bench_default.txt 2> $O/bench_default.err; tail -c 1500 $O/bench_default.txt | head -c 600; echo
timeout 420 bash tools/profile_bench.sh r04_fp64 --no-boundary --no-saturation > $O/profile_fp64.log 2>&1; tail -n 2 $O/profile_fp64.log
timeout 300 python tools/dump_bench_states.py MT50 4096 MT10 10240 > $O/dump_bench_states.txt 2>&1; tail -n 2 $O/dump_bench_states.txt
timeout 400 python tools/policy_gate_gpu.py fp64 > $O/policy_gate_gpu_fp64.txt 2>&1; tail -n 1 $O/policy_gate_gpu_fp64.txt
timeout 420 bash tools/profile_bench.sh r04_fp32 --precision fp32 --no-boundary --no-saturation > $O/profile_fp32.log 2>&1; tail -n 2 $O/profile_fp32.log
MW_LIB=libmwgpu_v_timing.so timeout 240 python tools/mix_timing.py 100 > $O/mix_timing_fp64.txt 2>&1; head -n 3 $O/mix_timing_fp64.txt | cut -c1-160
timeout 400 python tools/policy_gate_gpu.py fp32 > $O/policy_gate_gpu_fp32.txt 2>&1; tail -n 1 $O/policy_gate_gpu_fp32.txt
