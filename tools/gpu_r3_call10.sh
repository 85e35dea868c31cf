#!/bin/bash
# round 3, GPU call 10: (1) per-step tail: which env sets each launch's time; (2) the in-batch critical path of every scene with 4
# lanes per workgroup everywhere (re-calibration of model_caps.json step_ms_lpb4); (3) bench with the minimax exchange on / off
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c10
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-extra-precision --steps 300"
MW_VERBOSE=1 timeout 300 python bench.py $B > $O/bench_exchange.txt 2>&1
MW_VERBOSE=1 MW_LPB_EXCHANGE=0 timeout 300 python bench.py $B > $O/bench_noexchange.txt 2>&1
MW_LANES_PER_BLOCK=4 MW_MIX_JSON=$O/mix_lpb4.json MW_MIX_NPZ=$O/mix_lpb4.npz MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 100 fp64 > $O/mix_lpb4.txt 2>&1
MW_LANES_PER_BLOCK=8 MW_MIX_JSON=$O/mix_lpb8.json MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 60 fp64 > $O/mix_lpb8.txt 2>&1
MW_LANES_PER_BLOCK=2 MW_MIX_JSON=$O/mix_lpb2.json MW_LIB=libmwgpu_timing.so timeout 300 python tools/mix_timing.py 60 fp64 > $O/mix_lpb2.txt 2>&1
MW_LIB=libmwgpu_timing.so timeout 400 python tools/experiments/tail_probe.py 24 $O/tail.npz > $O/tail_probe.txt 2>&1
grep -H "lanes per workgroup" $O/bench_*.txt
grep -H -o '"value": [0-9.]*' $O/bench_*.txt
head -3 $O/mix_lpb4.txt $O/mix_lpb8.txt $O/mix_lpb2.txt | cut -c1-200
cat $O/tail_probe.txt | cut -c1-330
