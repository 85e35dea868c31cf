#!/bin/bash
# round 3, GPU call 7: load batching (Euler step, smooth-force tail, snapshot copy), wavefront-scope MW_SYNC, support() without the
# final vertex fetch, body-level chains in the scratchpad: parity subset, bench A/B (chains on / off, strong sync), stage split
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c7
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_parity.py tests/test_self_consistency.py tests/test_gpu_fullsize.py tests/test_lazy_forward.py > $O/pytest_subset.txt 2>&1
B="--no-cpu-baseline --no-extra-precision --steps 300"
timeout 300 python bench.py $B > $O/bench_new.txt 2>&1
MW_CHAIN_LDS=0 timeout 300 python bench.py $B > $O/bench_chain0.txt 2>&1
MW_LIB=libmwgpu_v_syncstrong.so timeout 300 python bench.py $B > $O/bench_syncstrong.txt 2>&1
timeout 300 python bench.py $B --precision fp32 > $O/bench_fp32.txt 2>&1
MW_VERBOSE=1 MW_MIX_NPZ=$O/mix_timing_fp64.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_timing_fp64.txt 2>&1
tail -n 3 $O/pytest_subset.txt
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
head -8 $O/mix_timing_fp64.txt
