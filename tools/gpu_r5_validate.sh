#!/bin/bash
# round-5 validation call: the GPU suite, then the default bench line (with boundary / saturation / configs / cpu_baseline)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_validate; rm -rf $O; mkdir -p $O
rm -f gpurun_out/policy200_relaxed_gpu.txt gpurun_out/bench_states_relaxed.txt
timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -20
cp gpurun_out/policy200_relaxed_gpu.txt gpurun_out/bench_states_relaxed.txt $O/ 2>/dev/null
timeout 900 python bench.py --steps ${1:-200} > $O/bench_default.txt 2> $O/bench_default.err
tail -c 3000 $O/bench_default.txt; tail -5 $O/bench_default.err
