#!/bin/bash
# BASELINE.json configs 2 / 3 and the batch-size curve of the headline workload on one MI355X (config 5: tests/test_cfg5_policies.py)
set -u
O=gpurun_out/r04_config_table.txt; : > $O
A="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation"
run() { echo "== python bench.py $*" >> $O; timeout 120 python bench.py $A "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['config']['workload'], '|', round(j['value']), 'env-steps/s | kernel ms/launch', round(j['roofline']['kernel_ms_per_launch'], 3), '| flags', j['config']['status_flags']['flags'])
" >> $O; }
run --benchmark MT1 --precision fp32 --envs 4096 --steps 300
run --benchmark MT10 --envs 10240 --steps 200
run --envs 8192 --steps 150
run --envs 32768 --steps 60 --warmup 5
cat $O
