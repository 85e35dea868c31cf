#!/bin/bash
# round 3, GPU call 4 (first call of the re-entered session): the GPU suite on the current tree, the default bench line, the
# rocprofv3 evidence (kernel trace + PMC passes incl. the fp64 instruction counters), the per-task stage split inside the bench
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c4
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
timeout 600 python bench.py > $O/bench_default.txt 2>&1
timeout 900 bash tools/profile_bench.sh r03 > $O/profile.log 2>&1
MW_VERBOSE=1 MW_MIX_NPZ=$O/mix_timing_fp64.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
(cd /tmp && rocprofv3 -L 2>&1 | grep -o -E "SQ_[A-Z0-9_]+" | sort -u | tr '\n' ' ') > $O/sq_counters.txt 2>&1
tail -n 3 $O/pytest_gpu.txt
grep -h -o '"value": [0-9.]*' $O/bench_default.txt | head -3
cat gpurun_out/prof_r03/pmc_summary.txt | tail -n 40
