#!/bin/bash
# A/B of the working tree against a frozen copy of an earlier tree (ab_ref/: `git archive <rev> | tar -x -C ab_ref` + its own
# built libmwgpu.so; git-ignored, travels with gpurun) INSIDE ONE gpurun call, interleaved -- needed when the model tables change
# together with the library, so that tools/ab_bench.sh's MW_LIB switch is not enough.
#   gpurun -- bash tools/gpu_ab_ref.sh [ROUNDS=2]        (bench args via AB_ARGS)
set -u
cd "$(dirname "$0")/.."
rounds=${1:-2}
O=gpurun_out/abref_$(date +%H%M%S); mkdir -p $O
args=${AB_ARGS:---no-cpu-baseline --no-extra-precision --steps 300}
for r in $(seq $rounds); do
  (cd ab_ref && timeout 300 python bench.py $args) >> $O/ref.txt 2>&1
  timeout 300 python bench.py $args >> $O/new.txt 2>&1
done
for v in ref new; do
  echo "$v: $(grep -h -o '"value": [0-9.]*' $O/$v.txt | cut -d' ' -f2 | tr '\n' ' ')  flags: $(grep -h -o '"flags": [0-9]*' $O/$v.txt | cut -d' ' -f2 | sort -u | tr '\n' ' ')  kernel_ms: $(grep -h -o '"kernel_ms": [0-9.]*' $O/$v.txt | cut -d' ' -f2 | tr '\n' ' ')"
done
