#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/abenv2_$(date +%H%M%S); mkdir -p $O
A="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --no-configs --steps 300"
for r in 1 2; do
  timeout 300 python bench.py $A >> $O/base.txt 2>&1
  MW_LPB_EXCHANGE=1 timeout 300 python bench.py $A >> $O/exch.txt 2>&1
  MW_OVERSUBSCRIBE=1.15 timeout 300 python bench.py $A >> $O/f115.txt 2>&1
done
for v in base exch f115; do echo "$v: $(grep -h -o '"value": [0-9.]*' $O/$v.txt | cut -d' ' -f2 | cut -c1-9 | tr '\n' ' ') | $(grep -h -o '"median": [0-9.]*' $O/$v.txt | cut -d' ' -f2 | cut -c1-6 | tr '\n' ' ')"; done
