#!/bin/bash
# round 3, GPU call 17: the polishing Newton iteration on the kept factor (newton_resolve): A/B inside one call
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c17
mkdir -p $O
B="--no-cpu-baseline --no-extra-precision --steps 300"
for v in libmwgpu_v_pre.so libmwgpu_v_reuse.so libmwgpu_v_pre.so libmwgpu_v_reuse.so; do
  MW_LIB=$v timeout 300 python bench.py $B >> $O/bench_$v.txt 2>&1
done
MW_LIB=libmwgpu_v_reuse.so timeout 300 python bench.py $B --precision fp32 > $O/bench_fp32_reuse.txt 2>&1
MW_LIB=libmwgpu_v_pre.so timeout 300 python bench.py $B --precision fp32 > $O/bench_fp32_pre.txt 2>&1
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
grep -H -o '"solver_stalls": [0-9]*' $O/bench_*.txt
