#!/bin/bash
# rocprofv3 evidence for the bench workload (run on the GPU box through gpurun):
#   kernel-trace stats (its own pass) + separate PMC passes, all CSV under gpurun_out/prof_<tag>/
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
args="--no-cpu-baseline --no-parity-mode $*"     # bench.py defaults (1000 timed steps after 100) unless overridden
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python $root/bench.py $args > "$out/kt.log" 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d "$out/pmc_$name" -- python $root/bench.py $args > "$out/pmc_$name.log" 2>&1
done
cd $root
# keep only the small CSVs
find "$out" -name "*.db" -delete
du -sh "$out"
