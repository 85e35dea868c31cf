#!/bin/bash
# round 3, final evidence call: the GPU suite, the default bench line, rocprofv3 kernel trace + PMC passes in both precisions,
# the stage split inside the bench workload, the other BASELINE configurations + batch-size curve, the reference's policy gate
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
timeout 600 python bench.py > $O/bench_default.txt 2>&1
timeout 900 bash tools/profile_bench.sh r03 > $O/profile_fp64.log 2>&1
timeout 900 bash tools/profile_bench.sh r03_fp32 --precision fp32 > $O/profile_fp32.log 2>&1
MW_VERBOSE=1 MW_MIX_NPZ=$O/mix_timing_fp64.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
MW_MAX_ENVS=32768 timeout 900 python tools/experiments/config_table.py fp64 > $O/config_table_fp64.txt 2>&1
MW_MAX_ENVS=16384 timeout 600 python tools/experiments/config_table.py fp32 > $O/config_table_fp32.txt 2>&1
timeout 900 python tools/policy_gate_gpu.py fp64 > $O/policy_gate_gpu_fp64.txt 2>&1
timeout 900 python tools/policy_gate_gpu.py fp32 > $O/policy_gate_gpu_fp32.txt 2>&1
tail -n 3 $O/pytest_gpu.txt
grep -h -o '"value": [0-9.]*' $O/bench_default.txt | head -3
tail -n 12 $O/config_table_fp64.txt
tail -n 2 $O/policy_gate_gpu_fp64.txt
