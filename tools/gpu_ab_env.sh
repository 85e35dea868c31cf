#!/bin/bash
# A/B of environment-variable variants of the runtime INSIDE ONE gpurun call, interleaved:
#   gpurun -- bash tools/gpu_ab_env.sh "VAR=a VAR=b ..." [ROUNDS=2] [bench args]
set -u
cd "$(dirname "$0")/.."
variants=$1; rounds=${2:-2}; shift 2 || true
O=gpurun_out/abenv_$(date +%H%M%S); mkdir -p $O
A="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --no-configs ${*:---steps 300}"
for r in $(seq $rounds); do
  for v in $variants; do
    env $v MW_VERBOSE=1 timeout 300 python bench.py $A >> $O/$v.txt 2>&1
  done
done
for v in $variants; do
  echo "$v: value $(grep -h -o '"value": [0-9.]*' $O/$v.txt | cut -d' ' -f2 | cut -c1-9 | tr '\n' ' ') | kernel_ms $(grep -h -o '"kernel_ms_per_launch": [0-9.]*' $O/$v.txt | cut -d' ' -f2 | cut -c1-6 | tr '\n' ' ') | flags $(grep -h -o '"flags": [0-9]*' $O/$v.txt | cut -d' ' -f2 | sort -u | tr '\n' ' ') | $(grep -h 'lanes per workgroup' $O/$v.txt | head -1)"
done | tee $O/summary.txt
