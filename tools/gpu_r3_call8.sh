#!/bin/bash
# round 3, GPU call 8: the bench-state parity sample under the new build, with chains off, and with the strong sync
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c8
mkdir -p $O
timeout 400 python tools/experiments/bench_state_probe.py $O/worst_default.npz > $O/probe_default.txt 2>&1
MW_CHAIN_LDS=0 timeout 400 python tools/experiments/bench_state_probe.py $O/worst_chain0.npz > $O/probe_chain0.txt 2>&1
MW_LIB=libmwgpu_v_syncstrong.so timeout 400 python tools/experiments/bench_state_probe.py $O/worst_syncstrong.npz > $O/probe_syncstrong.txt 2>&1
tail -n 14 $O/probe_*.txt
