#!/bin/bash
# rocprofv3 evidence for the bench workload (run on the GPU box through gpurun):
#   kernel-trace stats (its own pass) + separate PMC passes, all CSV under gpurun_out/prof_<tag>/
# usage: tools/profile_bench.sh <tag> [bench args...]      (the bench runs with --no-cpu-baseline --no-extra-precision)
set -u
tag=${1:-r03}; shift || true
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
args="--no-cpu-baseline --no-extra-precision --no-configs $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python $root/bench.py $args > "$out/kt.log" 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d "$out/pmc_$name" -- python $root/bench.py $args > "$out/pmc_$name.log" 2>&1
done
# matrix-core counters: whichever of the candidate names this rocprofv3 lists for gfx950, in one extra pass
avail=$(rocprofv3 -L 2>/dev/null | grep -o -E 'SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*' | sort -u | tr '\n' ' ')
echo "MFMA counters listed by rocprofv3 -L: $avail" > "$out/mfma_counters_available.txt"
mf=""
for c in SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F64; do
  echo " $avail " | grep -q " $c " && mf="$mf $c"
done
if [ -n "$mf" ]; then
  rocprofv3 --pmc $mf SQ_BUSY_CU_CYCLES --output-format csv -d "$out/pmc_MFMA" -- python $root/bench.py $args > "$out/pmc_MFMA.log" 2>&1 || \
  rocprofv3 --pmc $mf --output-format csv -d "$out/pmc_MFMA" -- python $root/bench.py $args > "$out/pmc_MFMA.log" 2>&1
fi
cd $root
find "$out" -name "*.db" -delete
stats=$(find "$out/kt" -name "*kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp "$stats" "$out/kernel_stats.csv"
kms=$(python - "$out/kernel_stats.csv" <<'PY'
import csv, sys
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "launch_step" in r["Name"]:
            print(float(r["AverageNs"]) / 1e6); break
except Exception:
    print(0)
PY
)
prec=$(echo "$args" | grep -q "fp32" && echo fp32 || echo fp64)
python tools/summarize_pmc.py "$out/pmc_*/**/*counter_collection.csv" --kernel launch_step --json "$out/pmc.json" \
  --workload "MT50 sync-vector, 4096 envs/GPU, $prec, random actions" --kernel-ms "$kms" > "$out/pmc_summary.txt" 2>&1
grep -h '^{' "$out/kt.log" | tail -1 > "$out/bench_line_under_kernel_trace.json"
# keep the summaries only (gpurun copies at most 64 MiB back; the raw kernel trace + counter CSVs are ~40 MB per run)
rm -rf "$out/kt" "$out"/pmc_*/
du -sh "$out"
