#!/bin/bash
# the library built with -ffp-contract=off (libmwgpu_nofma.so) against the default build: the GPU suite on it, then an interleaved bench A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05_nofma; rm -rf $O; mkdir -p $O
rm -f gpurun_out/policy200_relaxed_gpu.txt
MW_LIB_OVERRIDE=libmwgpu_nofma.so timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_bench_state_parity.py > $O/pytest_gpu_nofma.txt 2>&1
tail -4 $O/pytest_gpu_nofma.txt; grep -E "^(FAILED|ERROR)" $O/pytest_gpu_nofma.txt | head
cp gpurun_out/policy200_relaxed_gpu.txt $O/policy200_relaxed_gpu_nofma.txt 2>/dev/null
grep "steps relaxed" $O/policy200_relaxed_gpu_nofma.txt | grep -v "  0 of"
A="--no-cpu-baseline --no-boundary --no-saturation --no-configs --steps 300"
for r in 1 2; do
  timeout 300 python bench.py $A >> $O/fma.txt 2>/dev/null
  MW_LIB=libmwgpu_nofma.so timeout 300 python bench.py $A >> $O/nofma.txt 2>/dev/null
done
for v in fma nofma; do echo "$v: fp64 $(grep -h -o '"value": [0-9.]*' $O/$v.txt | sed -n '1p;3p' | cut -d' ' -f2 | cut -c1-9 | tr '\n' ' ') fp32 $(grep -h -o '"value": [0-9.]*' $O/$v.txt | sed -n '2p;4p' | cut -d' ' -f2 | cut -c1-9 | tr '\n' ' ')"; done | tee $O/summary.txt
