#!/bin/bash
# round 3, GPU call 16: refinement in its own non-inlined function: A/B inside one call + stage split
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c16
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-precision --steps 300"
for v in libmwgpu_v_pre.so libmwgpu.so libmwgpu_v_pre.so libmwgpu.so; do
  MW_LIB=$v timeout 300 python bench.py $B >> $O/bench_$v.txt 2>&1
done
MW_VERBOSE=1 MW_MIX_NPZ=$O/mix_timing_fp64.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
grep -H -o '"solver_stalls": [0-9]*' $O/bench_*.txt
head -4 $O/mix_timing_fp64.txt
