#!/bin/bash
# A/B of the split collision (mw_split.inl) against the fused step kernel INSIDE ONE gpurun call, interleaved, at the metric's batch
# (MT50 @ 4096) and at the saturation point (16 384):   gpurun -- bash tools/gpu_ab_split.sh [ROUNDS=2]
set -u
cd "$(dirname "$0")/.."
rounds=${1:-2}
O=gpurun_out/absplit_$(date +%H%M%S); mkdir -p $O
A="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --no-configs"
timeout 600 python -m pytest tests/test_split_collision.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for r in $(seq $rounds); do
  for sc in 0 1; do
    timeout 300 python bench.py $A --steps 300 --split-collision $sc >> $O/n4096_sc$sc.txt 2>&1
    timeout 300 python bench.py $A --steps 100 --warmup 10 --envs 16384 --split-collision $sc >> $O/n16384_sc$sc.txt 2>&1
  done
done
for f in n4096_sc0 n4096_sc1 n16384_sc0 n16384_sc1; do
  echo "$f: value $(grep -h -o '"value": [0-9.]*' $O/$f.txt | cut -d' ' -f2 | tr '\n' ' ') | kernel_ms/step $(grep -h -o '"kernel_ms_per_launch": [0-9.]*' $O/$f.txt | cut -d' ' -f2 | tr '\n' ' ') | flags $(grep -h -o '"flags": [0-9]*' $O/$f.txt | cut -d' ' -f2 | sort -u | tr '\n' ' ')"
  grep -h -v '^{' $O/$f.txt | tail -3
done | tee $O/summary.txt
