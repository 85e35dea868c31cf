# round-4 development call: wave-cooperative Newton direction (MFMA), fourth version (branch-free Cholesky, DPP broadcasts, VGPR accumulators)
set -u
O=gpurun_out/c5; mkdir -p $O
tools/experiments/_build/mfma_layout_probe 2>&1 | tail -n 3
for lpb in 4 8 1; do
  MW_LANES_PER_BLOCK=$lpb timeout 120 python tools/experiments/ab_physics.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
done > $O/ab_physics.txt 2>&1
cat $O/ab_physics.txt
export AB_ARGS="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --steps 300"
for k in 1 2; do
MW_LIB=libmwgpu_v_nowave.so timeout 300 python bench.py $AB_ARGS >> $O/bench_nowave.txt 2>&1
timeout 300 python bench.py $AB_ARGS >> $O/bench_new.txt 2>&1
done
for v in nowave new; do echo "$v: $(grep -h -o '"value": [0-9.]*' $O/bench_$v.txt | head -n 4 | cut -d' ' -f2 | tr '\n' ' ') flags $(grep -h -o '"flags": [0-9]*' $O/bench_$v.txt | sort -u | tr '\n' ' ')"; grep -h -i "error\|Traceback" $O/bench_$v.txt | head -n 3; done
MW_LIB=libmwgpu_v_timing.so timeout 300 python tools/mix_timing.py 100 > $O/mix_timing.txt 2>&1; head -n 4 $O/mix_timing.txt | cut -c1-200; grep "stick-push\|sweep-v3 \|reach-v3 \|box-close" $O/mix_timing.txt | cut -c1-200
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_cfg5_policies.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; echo "pytest subset rc $?"; grep -E "passed|failed|Error|assert" $O/pytest_subset.txt | tail -n 6
