#!/bin/bash
# round 3, last evidence call (the tree after the box-box change): GPU suite, default bench line, rocprofv3 kernel trace + PMC passes
# in both precisions (source hash = this tree), stage split.  (BASELINE configs table + policy gate: tools/gpu_r3_final.sh, one commit
# earlier; the change in between is result-identical.)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final2
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
timeout 600 python bench.py > $O/bench_default.txt 2>&1
timeout 600 bash tools/profile_bench.sh r03 > $O/profile_fp64.log 2>&1
timeout 600 bash tools/profile_bench.sh r03_fp32 --precision fp32 > $O/profile_fp32.log 2>&1
MW_VERBOSE=1 MW_MIX_NPZ=$O/mix_timing_fp64.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
tail -n 3 $O/pytest_gpu.txt
grep -h -o '"value": [0-9.]*' $O/bench_default.txt | head -3
