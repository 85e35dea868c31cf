#!/bin/bash
# round 3, GPU call 5: what is the wave waiting for?  (1) instruction-cache probe (straight-line code vs loops, one wave per SIMD),
# (2) PMC pass with the in-flight LEVEL counters (average latency of instruction fetches, vector / scalar memory and LDS
# instructions), (3) the collision-stage split inside the bench workload (per-branch clocks inside collide_pair)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c5
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/experiments/_build/icache_probe > $O/icache_probe.txt 2>&1
(cd /tmp && rocprofv3 -L 2>&1 | grep -o -E "SQC_[A-Z0-9_]+|TCP_[A-Z0-9_]+" | sort -u | tr '\n' ' ') > $O/sqc_tcp_counters.txt 2>&1
root=$(pwd)
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  name=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d "$root/$O/pmc_$name" -- python $root/bench.py --no-cpu-baseline --no-extra-precision --steps 200 > "$root/$O/pmc_$name.log" 2>&1)
done
find $O -name "*.db" -delete
python tools/summarize_pmc.py "$O/pmc_*/**/*counter_collection.csv" --kernel step_device_only > $O/level_counters.txt 2>&1
rm -rf $O/pmc_*/
MW_LIB=libmwgpu_colltiming.so timeout 300 python tools/experiments/coll_timing.py 100 fp64 > $O/coll_timing.txt 2>&1
cat $O/icache_probe.txt $O/level_counters.txt
