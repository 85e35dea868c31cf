#!/bin/bash
# round 3, GPU call 3: where does the narrow phase spend its time (per-branch shader clocks inside collide_pair), the no-IPRA
# fault with the device buffers' address ranges logged (and a range-checked build of the same), instruction-cache counter names
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c3
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L 2>&1 | grep -i -E "icache|ifetch|inst_fetch|SQC_" | head -60) > $O/counters.txt 2>&1
MW_LIB=libmwgpu_colltiming.so timeout 300 python tools/experiments/coll_timing.py 100 fp64 > $O/coll_timing.txt 2>&1
for k in 1 2; do
  MW_VERBOSE=2 MW_LIB=libmwgpu_v_noipra.so timeout 300 python bench.py --no-cpu-baseline --no-extra-precision --steps 100 > $O/noipra_$k.txt 2>&1
done
MW_VERBOSE=2 MW_LIB=libmwgpu_v_noipra_bounds.so timeout 400 python bench.py --no-cpu-baseline --no-extra-precision --steps 100 > $O/noipra_bounds.txt 2>&1
tail -n 4 $O/noipra_1.txt $O/noipra_2.txt $O/noipra_bounds.txt | cut -c1-300
