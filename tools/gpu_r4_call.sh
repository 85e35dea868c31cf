# round-4 development call: wave-cooperative Newton direction (MFMA), third version (per-row coefficients, two reads per row)
set -u
O=gpurun_out/c4; mkdir -p $O
for v in libmwgpu.so; do for lpb in 4 8; do
  MW_LIB=$v MW_LANES_PER_BLOCK=$lpb timeout 120 python tools/experiments/ab_physics.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
done; done > $O/ab_physics.txt 2>&1
cat $O/ab_physics.txt
export AB_ARGS="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --steps 300"
for k in 1 2; do
MW_LIB=libmwgpu_v_nowave.so timeout 300 python bench.py $AB_ARGS >> $O/bench_nowave.txt 2>&1
timeout 300 python bench.py $AB_ARGS >> $O/bench_new.txt 2>&1
done
for v in nowave new; do echo "$v: $(grep -h -o '"value": [0-9.]*' $O/bench_$v.txt | head -n 4 | cut -d' ' -f2 | tr '\n' ' ') flags $(grep -h -o '"flags": [0-9]*' $O/bench_$v.txt | sort -u | tr '\n' ' ')"; grep -h -i "error\|Traceback" $O/bench_$v.txt | head -n 3; done
MW_LIB=libmwgpu_v_timing.so timeout 300 python tools/mix_timing.py 100 > $O/mix_timing.txt 2>&1; head -n 6 $O/mix_timing.txt | cut -c1-200; grep -A3 "solver phases" $O/mix_timing.txt | cut -c1-200; grep "stick-push\|sweep-v3 \|reach-v3 " $O/mix_timing.txt | cut -c1-200
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_self_consistency.py tests/test_cfg5_policies.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; echo "pytest subset rc $?"; grep -E "passed|failed|Error|assert" $O/pytest_subset.txt | tail -n 6
