#!/bin/bash
# round 3, GPU call 18: box-box clipping polygons in a thread-private LDS slice + early exit of the box face-axis test: A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c18
mkdir -p $O
B="--no-cpu-baseline --no-extra-precision --steps 300"
for v in libmwgpu_v_pre.so libmwgpu_v_new.so libmwgpu_v_pre.so libmwgpu_v_new.so; do
  MW_LIB=$v timeout 300 python bench.py $B >> $O/bench_$v.txt 2>&1
done
MW_LIB=libmwgpu_v_new.so timeout 300 python bench.py $B --precision fp32 > $O/bench_fp32_new.txt 2>&1
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
grep -H -o '"flags": [0-9]*' $O/bench_*.txt | sort | uniq -c
