#!/bin/bash
# round 3, GPU call 13: lane assignment A/B inside one call: current table, table re-measured in call 10 (uniform 4 / 8 lanes),
# the same with the exchange pass, 4 lanes everywhere
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c13
mkdir -p $O
B="--no-cpu-baseline --no-extra-precision --steps 300"
MW_VERBOSE=1 timeout 300 python bench.py $B > $O/bench_a_current.txt 2>&1
MW_VERBOSE=1 MW_MODEL_CAPS=tools/experiments/model_caps_c10.json timeout 300 python bench.py $B > $O/bench_b_newcaps.txt 2>&1
MW_VERBOSE=1 MW_MODEL_CAPS=tools/experiments/model_caps_c10.json MW_LPB_EXCHANGE=1 timeout 300 python bench.py $B > $O/bench_c_newcaps_exchange.txt 2>&1
MW_VERBOSE=1 MW_LANES_PER_BLOCK=4 timeout 300 python bench.py $B > $O/bench_d_uniform4.txt 2>&1
MW_VERBOSE=1 timeout 300 python bench.py $B > $O/bench_e_current_again.txt 2>&1
grep -H "lanes per workgroup" $O/bench_*.txt
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
