#!/bin/bash
# round 3, GPU call 11: A/B inside one call (boxes differ by up to 25 %): session-start library, previous commit, current tree
# (adjacency coordinates + the start vertex's neighbour block stored with its cube-map cell), and the session-start library again
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c11
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-precision --steps 300"
for v in libmwgpu_v_ref.so libmwgpu_v_c9.so libmwgpu.so libmwgpu_v_ref.so libmwgpu.so; do
  MW_LIB=$v timeout 300 python bench.py $B >> $O/bench_$v.txt 2>&1
done
rocm-smi --showclocks --showpower --showtemp > $O/smi.txt 2>&1
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
