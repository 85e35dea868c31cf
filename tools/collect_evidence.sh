#!/bin/bash
# after `gpurun -- bash tools/gpu_run.sh final <tag>`: copy the summaries the round is judged on from gpurun_out/ (scratch) into
# profiles/ (tracked) and the bench-state recordings into tests/golden/.   usage: tools/collect_evidence.sh [tag=r06]
set -u
cd "$(dirname "$0")/.."
tag=${1:-r06}
F=$(ls -d gpurun_out/final_* | tail -1)
cp gpurun_out/benchstate_MT50_4096.npz gpurun_out/benchstate_MT10_10240.npz tests/golden/
for p in fp64 fp32; do
  for f in kernel_stats.csv pmc.json pmc_summary.txt bench_line_under_kernel_trace.json; do
    cp gpurun_out/prof_${tag}_$p/$f profiles/${tag}_mt50_${p}_$f
  done
done
cp gpurun_out/prof_${tag}_fp64/mfma_counters_available.txt profiles/${tag}_mfma_counters_available.txt
{ echo "# python -m pytest tests -m gpu -q -rA   (1x MI355X, final tree of the round: tools/gpu_run.sh final)"; grep -E "passed|failed" $F/pytest_gpu_full.txt | tail -n 1; grep -c "^PASSED" $F/pytest_gpu_full.txt | sed 's/^/    /;s/$/ PASSED/'; echo; cat $F/bench_states_relaxed.txt 2>/dev/null; } > profiles/${tag}_pytest_gpu_summary.txt
cp $F/policy200_branches_gpu.txt profiles/${tag}_policy200_branches_gpu.txt
cp $F/bench_states_relaxed.txt profiles/${tag}_bench_states_branches.txt
{ echo "# python bench.py (defaults) on the final tree of the round, 1x MI355X"; grep '^{' $F/bench_default.txt | tail -1; } > profiles/${tag}_bench_default.txt
{ echo "# tools/policy_gate_gpu.py fp64 on the final tree of the round: the reference's 50-goal scripted-policy gate, device policies"; grep -v amdgpu.ids $F/policy_gate_gpu_fp64.txt; } > profiles/${tag}_policy_gate_gpu_fp64.txt
ls -la profiles/${tag}_* | awk '{print $5, $9}'
