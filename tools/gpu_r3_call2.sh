#!/bin/bash
# round 3, GPU call 2: the IPRA probe one variant per process (ipra_off first), bench lines of the current tree, the
# scratchpad-row-store variant and a no-IPRA build, stage / solver-phase split of both, GPU tests of the variant
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c2
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/experiments/want_probe.py run ipra_off > $O/want_probe_ipra_off.txt 2>&1
timeout 300 python tools/experiments/want_probe.py run ipra_on > $O/want_probe_ipra_on.txt 2>&1
for v in libmwgpu.so libmwgpu_v_lds.so libmwgpu_v_noipra.so; do
  MW_LIB=$v timeout 300 python bench.py --no-cpu-baseline --no-extra-precision --steps 300 > $O/bench_$v.txt 2>&1
done
MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_main.txt 2>&1
MW_LIB=libmwgpu_timing_lds.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_lds.txt 2>&1
MW_LIB_OVERRIDE=libmwgpu_v_lds.so timeout 900 python -m pytest -m gpu -q tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_self_consistency.py > $O/pytest_lds.txt 2>&1
timeout 600 python -m pytest -m gpu -q tests/test_self_consistency.py "tests/test_gpu_fullsize.py::test_bench_states_match_the_oracle" "tests/test_gpu_parity.py::test_gpu_task_matches_reference_trace" > $O/pytest_main_subset.txt 2>&1
tail -n 3 $O/want_probe_*.txt $O/pytest_*.txt
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
