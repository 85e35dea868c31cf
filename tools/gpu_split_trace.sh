#!/bin/bash
# per-kernel durations of the split-collision step (rocprofv3 kernel trace): gpurun -- bash tools/gpu_split_trace.sh [envs]
set -u
cd "$(dirname "$0")/.."
root=$(pwd); envs=${1:-4096}
O=$root/gpurun_out/split_trace_$envs; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $root/bench.py --no-cpu-baseline --no-extra-precision --no-boundary --no-saturation --no-configs --steps 100 --envs $envs --split-collision 1 > $O/kt.log 2>&1
cd $root
stats=$(find $O/kt -name "*kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp $stats $O/kernel_stats.csv
trace=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python - "$trace" > $O/step_timeline.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 % of the trace = the timed region; print one step's timeline (kernel, duration, gap to the previous kernel)
n = len(rows); tail = rows[int(n * 0.7):]
def short(name):
    for k in ("mid_phase", "narrow", "lane_phase", "launch_collision", "phase_kernel", "lane_step"):
        if k in name: return k
    return name[:40]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
lines = []
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3; agg[k][2] += gap
    lines.append(f"{k:18s} dur {(e - s) / 1e3:9.1f} us   gap {gap:8.1f} us")
    prev_end = e
for k, (c, d, g) in agg.items():
    print(f"{k:18s} n {c:6d}  mean dur {d / c:9.1f} us   mean gap before {g / c:8.1f} us")
print("\n".join(lines[:60]))
PY
cat $O/step_timeline.txt | head -80
grep -h '^{' $O/kt.log | tail -1 | cut -c1-300
rm -rf $O/kt
