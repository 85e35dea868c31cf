#!/bin/bash
# round 3, GPU call 1: (1) the IPRA question of DESIGN.md 5 on the round-2 form of the narrow phase (tools/experiments/want_probe.py,
# libraries built from 66130b6), (2) the GPU suite on the current tree (convergent narrow-phase calls + canary + the three merged
# narrow-phase branches), (3) bench lines of the round-2 library, the convergent-call library and the current one, (4) the
# per-task stage split inside the bench workload, (5) the bench-state recording for tests/test_bench_state_parity.py
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c1
mkdir -p $O
export TMPDIR=/tmp
python tools/experiments/want_probe.py run > $O/want_probe.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
for v in libmwgpu_v_r2.so libmwgpu_v_conv.so libmwgpu.so; do
  MW_LIB=$v timeout 300 python bench.py --no-cpu-baseline --no-extra-precision --steps 300 > $O/bench_$v.txt 2>&1
done
MW_LIB=libmwgpu.so timeout 300 python bench.py --no-cpu-baseline --no-extra-precision --steps 300 --precision fp32 > $O/bench_fp32.txt 2>&1
MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
timeout 600 python tools/dump_bench_states.py MT50 4096 MT10 10240 > $O/dump.txt 2>&1
cp gpurun_out/benchstate_*.npz $O/ 2>/dev/null
tail -3 $O/want_probe.txt $O/pytest_gpu.txt $O/dump.txt
grep -h -o '"value": [0-9.]*' $O/bench_*.txt
