#!/bin/bash
# round-4 evidence call (second, after the fused-rollout entry points were added): GPU suite, default bench line, rocprofv3 kernel trace +
# PMC passes (fp64), the bench-state recordings for tests/golden/.  Summaries -> gpurun_out/r04b/.  (The first evidence call of the round
# also ran the policy gates, the fp32 profile and the stage split: same lane programs, profiles/r04_*.)
set -u
O=gpurun_out/r04b; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -q -rA > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_gpu_full.txt | tail -n 2
timeout 400 python bench.py > $O/bench_default.txt 2> $O/bench_default.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04b/bench_default.txt') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); print(j['value'], j.get('fused_rollout',{}).get('value'), j.get('boundary',{}).get('host_numpy',{}).get('value'), j.get('saturation',{}).get('value'), j.get('throughput_mode',{}).get('value'))
PY
timeout 420 bash tools/profile_bench.sh r04b_fp64 --no-boundary --no-saturation > $O/profile_fp64.log 2>&1; tail -n 1 $O/profile_fp64.log
timeout 300 python tools/dump_bench_states.py MT50 4096 MT10 10240 > $O/dump_bench_states.txt 2>&1; tail -n 2 $O/dump_bench_states.txt | cut -c1-120
timeout 300 bash tools/profile_bench.sh r04b_fp32 --precision fp32 --no-boundary --no-saturation > $O/profile_fp32.log 2>&1; tail -n 1 $O/profile_fp32.log
