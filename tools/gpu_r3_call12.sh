#!/bin/bash
# round 3, GPU call 12: collision-stage split of the current tree with the raw per-env numbers (wave-level analysis)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c12
mkdir -p $O
MW_COLL_NPZ=$O/coll.npz MW_LIB=libmwgpu_colltiming.so timeout 300 python tools/experiments/coll_timing.py 100 fp64 > $O/coll_timing.txt 2>&1
head -3 $O/coll_timing.txt | cut -c1-200
