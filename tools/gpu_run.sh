#!/bin/bash
# ONE parametrised runner for everything that goes to the GPU box (replaces the one-shot gpu_*.sh call scripts of rounds 3-5).
# Boxes of the pool differ by up to 25 % on this latency-bound kernel, so every comparison is made INSIDE one gpurun call, interleaved.
#   gpurun -- bash tools/gpu_run.sh ab   [ROUNDS] LIB_A LIB_B ...      library variants (metaworld_amd/<name>), bench args via AB_ARGS
#   gpurun -- bash tools/gpu_run.sh env  [ROUNDS] "VAR=a" "VAR=b" ...  environment-variable variants of the runtime (MW_LIB=... is one of them)
#   gpurun -- bash tools/gpu_run.sh ref  [ROUNDS]                      working tree against the frozen tree in ab_ref/ (its own models + library)
#   gpurun -- bash tools/gpu_run.sh test [pytest args]                 pytest -m gpu (default: the whole GPU suite), summary under gpurun_out/
#   gpurun -- bash tools/gpu_run.sh mix  [LIB]                         stage / solver-phase clocks inside the bench workload (timing build)
#   gpurun -- bash tools/gpu_run.sh final [TAG]                        the round's evidence on the final tree (recordings, GPU suite, bench line, rocprofv3 stats + PMC, policy gate)
# Several jobs in one call: separate them with "--", e.g.  tools/gpu_run.sh test tests/test_gpu_parity.py -- ab 2 libmwgpu_v_r5.so libmwgpu.so
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AB_DEFAULT="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --no-configs --steps 300"
summ() {   # file -> value / kernel ms / flags
  echo "value $(grep -h -o '"value": [0-9.]*' $1 | head -${2:-99} | cut -d' ' -f2 | cut -c1-9 | tr '\n' ' ')| kernel_ms(median) $(grep -h -o '"median": [0-9.]*' $1 | cut -d' ' -f2 | cut -c1-6 | tr '\n' ' ')| flags $(grep -h -o '"flags": [0-9]*' $1 | cut -d' ' -f2 | sort -u | tr '\n' ' ')| stalls $(grep -h -o '"solver_stalls": [0-9]*' $1 | cut -d' ' -f2 | tr '\n' ' ')"
}
job() {
  local cmd=$1; shift
  local rounds=2
  case "$cmd" in ab|env|ref) case "${1:-}" in ''|*[!0-9]*) ;; *) rounds=$1; shift;; esac;; esac
  local args=${AB_ARGS:-$AB_DEFAULT}
  local O=gpurun_out/${cmd}_$(date +%H%M%S); mkdir -p $O
  case "$cmd" in
  ab)
    for r in $(seq $rounds); do for v in "$@"; do MW_LIB=$v timeout 400 python bench.py $args >> $O/$v.txt 2>&1; done; done
    for v in "$@"; do echo "$v: $(summ $O/$v.txt)"; done | tee $O/summary.txt;;
  env)
    for r in $(seq $rounds); do for v in "$@"; do f=$(echo "$v" | tr '/ ' '__'); env $v MW_VERBOSE=1 timeout 400 python bench.py $args >> "$O/$f.txt" 2>&1; done; done
    for v in "$@"; do f=$(echo "$v" | tr '/ ' '__'); echo "$v: $(summ "$O/$f.txt") | $(grep -h 'lanes per workgroup' "$O/$f.txt" | head -1)"; done | tee $O/summary.txt;;
  ref)
    for r in $(seq $rounds); do
      (cd ab_ref && timeout 400 python bench.py $args) >> $O/ref.txt 2>&1
      timeout 400 python bench.py $args >> $O/new.txt 2>&1
    done
    for v in ref new; do echo "$v: $(summ $O/$v.txt)"; done | tee $O/summary.txt;;
  test)
    if [ $# -eq 0 ]; then set -- tests; fi
    timeout ${TEST_TIMEOUT:-3000} python -m pytest "$@" -m gpu ${TEST_X--x} -q -p no:cacheprovider 2>&1 | tail -40 | tee $O/pytest_tail.txt;;
  mix)
    MW_LIB=${1:-libmwgpu_timing.so} timeout 900 python tools/mix_timing.py 2>&1 | tee $O/mix_timing.txt | tail -5;;
  final)          # the evidence of a round on the final tree: bench-state recordings, GPU suite, default bench line, rocprofv3 kernel stats + PMC (fp64 / fp32), policy gate
    tag=${1:-r06}
    rm -f gpurun_out/policy200_branches_gpu.txt gpurun_out/bench_states_relaxed.txt
    timeout 300 python tools/dump_bench_states.py MT50 4096 MT10 10240 > $O/dump_bench_states.txt 2>&1; tail -n 2 $O/dump_bench_states.txt | cut -c1-160
    timeout 1200 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_gpu_full.txt | tail -n 2
    cp gpurun_out/policy200_branches_gpu.txt gpurun_out/bench_states_relaxed.txt $O/ 2>/dev/null
    timeout 900 python bench.py > $O/bench_default.txt 2> $O/bench_default.err; grep -o '"value": [0-9.]*' $O/bench_default.txt | head -1
    timeout 900 bash tools/profile_bench.sh ${tag}_fp64 --no-boundary --no-saturation > $O/profile_fp64.log 2>&1; tail -n 1 $O/profile_fp64.log
    timeout 700 bash tools/profile_bench.sh ${tag}_fp32 --precision fp32 --no-boundary --no-saturation > $O/profile_fp32.log 2>&1; tail -n 1 $O/profile_fp32.log
    timeout 900 python tools/policy_gate_gpu.py fp64 > $O/policy_gate_gpu_fp64.txt 2>&1; tail -n 1 $O/policy_gate_gpu_fp64.txt;;
  *) echo "unknown job $cmd"; return 2;;
  esac
}
cur=()
for a in "$@"; do
  if [ "$a" = "--" ]; then job "${cur[@]}"; cur=(); else cur+=("$a"); fi
done
[ ${#cur[@]} -gt 0 ] && job "${cur[@]}"
