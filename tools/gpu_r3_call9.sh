#!/bin/bash
# round 3, GPU call 9: solver / staging / contact-walk load batching, qpos in the scratchpad: bench + stage split
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c9
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-precision --steps 300"
timeout 300 python bench.py $B > $O/bench_new.txt 2>&1
timeout 300 python bench.py $B --precision fp32 > $O/bench_fp32.txt 2>&1
MW_VERBOSE=1 MW_MIX_NPZ=$O/mix_timing_fp64.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_parity.py tests/test_gpu_fullsize.py > $O/pytest_subset.txt 2>&1
tail -n 3 $O/pytest_subset.txt
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
head -8 $O/mix_timing_fp64.txt
