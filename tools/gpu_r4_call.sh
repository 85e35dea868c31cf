# round-4 development call: wave-cooperative Newton direction, fifth version (batched M loads, scratchpad fast path of the coefficient pass)
set -u
O=gpurun_out/c6; mkdir -p $O
MW_LANES_PER_BLOCK=4 timeout 120 python tools/experiments/ab_physics.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $O/ab_physics.txt
export AB_ARGS="--no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --steps 300"
for k in 1 2; do
MW_LIB=libmwgpu_v_nowave.so timeout 300 python bench.py $AB_ARGS >> $O/bench_nowave.txt 2>&1
timeout 300 python bench.py $AB_ARGS >> $O/bench_new.txt 2>&1
done
for v in nowave new; do echo "$v: $(grep -h -o '"value": [0-9.]*' $O/bench_$v.txt | head -n 4 | cut -d' ' -f2 | tr '\n' ' ') flags $(grep -h -o '"flags": [0-9]*' $O/bench_$v.txt | sort -u | tr '\n' ' ')"; grep -h -i "error\|Traceback" $O/bench_$v.txt | head -n 3; done
