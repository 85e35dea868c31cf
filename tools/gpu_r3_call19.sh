#!/bin/bash
# round 3, GPU call 19: register-resident box-box (no dynamically indexed arrays) on top of call 18: A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c19
mkdir -p $O
B="--no-cpu-baseline --no-extra-precision --steps 300"
for v in libmwgpu_v_pre.so libmwgpu_v_new.so libmwgpu_v_new2.so libmwgpu_v_new.so libmwgpu_v_new2.so; do
  MW_LIB=$v timeout 300 python bench.py $B >> $O/bench_$v.txt 2>&1
done
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
