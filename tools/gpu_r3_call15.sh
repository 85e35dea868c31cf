#!/bin/bash
# round 3, GPU call 15: mixed-precision iterative refinement of the Newton direction: A/B inside one call against the previous tree
# and the session-start library; parity subset; widened GPU tests (v1 on all 50 tasks, device invariants on all 50 tasks)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c15
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-precision --steps 300"
for v in libmwgpu_v_ref.so libmwgpu_v_pre.so libmwgpu.so libmwgpu_v_pre.so libmwgpu.so; do
  MW_LIB=$v timeout 300 python bench.py $B >> $O/bench_$v.txt 2>&1
done
timeout 300 python bench.py $B --precision fp32 > $O/bench_fp32.txt 2>&1
MW_VERBOSE=1 MW_MIX_NPZ=$O/mix_timing_fp64.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -n 3 $O/pytest_gpu.txt
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
head -4 $O/mix_timing_fp64.txt
