#!/bin/bash
set -u
cd "$(dirname "$0")/.."
root=$(pwd); O=$root/gpurun_out/torch_trace; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
python $root/tools/experiments/torch_boundary_trace.py 2>&1 | tail -1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt -- python $root/tools/experiments/torch_boundary_trace.py > $O/log.txt 2>&1
tail -1 $O/log.txt
python - $O <<'PY'
import csv, sys, glob
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K STEP" if "launch_step" in n else "K " + n.replace("void at::native::", "")[:90]))
for f in glob.glob(O + "/kt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", "") + " " + r.get("Bytes", "")))
ev.sort()
# last two step kernels and everything between them
idx = [i for i, e in enumerate(ev) if e[2] == "K STEP"]
a, b = idx[-3], idx[-1]
t0 = ev[a][0]
for s, e, n in ev[a:b + 1]:
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  {n}")
PY
rm -rf $O/kt
