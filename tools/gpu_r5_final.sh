#!/bin/bash
# round-5 evidence call: bench-state recordings for tests/golden/, the GPU suite, the default bench line, rocprofv3 kernel trace + PMC
# passes (fp64: incl. the matrix-core counters; fp32), summaries -> gpurun_out/r05/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05; rm -rf $O; mkdir -p $O
rm -f gpurun_out/policy200_relaxed_gpu.txt gpurun_out/bench_states_relaxed.txt
timeout 300 python tools/dump_bench_states.py MT50 4096 MT10 10240 > $O/dump_bench_states.txt 2>&1; tail -n 2 $O/dump_bench_states.txt | cut -c1-160
timeout 900 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_gpu_full.txt | tail -n 2
cp gpurun_out/policy200_relaxed_gpu.txt gpurun_out/bench_states_relaxed.txt $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench_default.txt 2> $O/bench_default.err; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05/bench_default.txt') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); print("value", j['value'], j['roofline']['kernel_ms'], "fused", j.get('fused_rollout',{}).get('value'), "boundary", {k: (round(v['value']), round(v['median_ms_per_step'], 3)) for k, v in j.get('boundary',{}).items() if isinstance(v, dict)}, "sat", j.get('saturation',{}).get('value'), "fp32", j.get('throughput_mode',{}).get('value'), "cpu", j.get('cpu_baseline',{}).get('value'), j.get('cpu_baseline',{}).get('single_core_value'))
    for c in j.get("configs", []): print("config", c["config"], round(c["value"]), c["flags"], c.get("mean_success"))
PY
timeout 700 bash tools/profile_bench.sh r05_fp64 --no-boundary --no-saturation > $O/profile_fp64.log 2>&1; tail -n 1 $O/profile_fp64.log
cat gpurun_out/prof_r05_fp64/mfma_counters_available.txt
timeout 500 bash tools/profile_bench.sh r05_fp32 --precision fp32 --no-boundary --no-saturation > $O/profile_fp32.log 2>&1; tail -n 1 $O/profile_fp32.log
