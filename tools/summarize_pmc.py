#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_*/p_counter_collection.csv) for the step kernel:
mean counter value per dispatch.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide
coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- both raw and corrected values are printed."""
import collections
import csv
import glob
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_*/p_counter_collection.csv"
kernel = sys.argv[2] if len(sys.argv) > 2 else "step_device_only"
print(f"# kernel filter: {kernel}")
for f in sorted(glob.glob(pat)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = None
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
    for c, d in sorted(acc.items()):
        v = list(d.values())
        m = sum(v) / len(v)
        extra = ""
        if c == "FETCH_SIZE":
            extra = f"  = {m / 1024:.1f} MiB raw, {2 * m / 1024:.1f} MiB with the gfx950 x2 correction"
        if c == "WRITE_SIZE":
            extra = f"  = {m / 1024:.1f} MiB"
        print(f"{c:24s} dispatches={len(v):3d} mean/dispatch={m:16.1f}{extra}")
    if meta:
        print("  launch:", meta)
