#!/bin/bash
# round 3, GPU call 6: the PMC passes of call 5 again (the library had not been rebuilt after the ABI addition)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/c6
mkdir -p $O
export TMPDIR=/tmp
root=$(pwd)
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ"; do
  name=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d "$root/$O/pmc_$name" -- python $root/bench.py --no-cpu-baseline --no-extra-precision --steps 200 > "$root/$O/pmc_$name.log" 2>&1)
done
find $O -name "*.db" -delete
python tools/summarize_pmc.py "$O/pmc_*/**/*counter_collection.csv" --kernel step_device_only > $O/level_counters.txt 2>&1
rm -rf $O/pmc_*/
cat $O/level_counters.txt; tail -n 3 $O/pmc_SQC_ICACHE_REQ.log | cut -c1-300
