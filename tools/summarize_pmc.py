#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (one directory per counter group, tools/profile_bench.sh) for the step kernel: mean counter value
per dispatch, printed as text; with --json OUT also the small JSON bench.py reads for `roofline.traffic` and `roofline.alu_issue`.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports WIDE coalesced reads by 2x (MI355X_MICROARCH.md, HBM section):
the step kernel's requests are 4-32 bytes per lane group, so the raw value is used and the corrected one is printed beside it."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metaworld_amd import native  # noqa: E402  (source_hash: which sources the profile belongs to)

ap = argparse.ArgumentParser()
ap.add_argument("pattern", nargs="?", default="gpurun_out/pmc_*/**/*counter_collection.csv")
ap.add_argument("--kernel", default="launch_step")
ap.add_argument("--json")
ap.add_argument("--workload", default="")
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--kernel-ms", type=float, default=0.0, help="average launch duration of the kernel trace of the same command")
args = ap.parse_args()
print(f"# kernel filter: {args.kernel}")
mean = {}
meta = None
for f in sorted(glob.glob(args.pattern, recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if args.kernel not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size") if k in r}
    for c, d in sorted(acc.items()):
        v = list(d.values())
        m = sum(v) / len(v)
        mean[c] = m
        extra = ""
        if c == "FETCH_SIZE":
            extra = f"  = {m / 1024:.1f} MiB raw, {2 * m / 1024:.1f} MiB with the gfx950 x2 wide-read correction"
        if c == "WRITE_SIZE":
            extra = f"  = {m / 1024:.1f} MiB"
        print(f"{c:24s} dispatches={len(v):4d} mean/dispatch={m:16.1f}{extra}")
if meta:
    print("  launch:", meta)
if args.json and mean:
    waves = mean.get("SQ_WAVES", 0.0)
    f64 = sum(mean.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    out = {"workload": args.workload, "kernel": args.kernel, "launch": meta, "source_hash": native.source_hash(),
           "valu_f64_wave_instr_per_launch": f64 if any(k.endswith("_F64") for k in mean) else None,
           "valu_active_frac": mean.get("SQ_ACTIVE_INST_VALU", 0.0) / mean["SQ_WAVE_CYCLES"] if mean.get("SQ_ACTIVE_INST_VALU") and mean.get("SQ_WAVE_CYCLES") else None,
           "busy_cycles_frac": mean.get("SQ_BUSY_CYCLES", 0.0) / mean["GRBM_GUI_ACTIVE"] if mean.get("SQ_BUSY_CYCLES") and mean.get("GRBM_GUI_ACTIVE") else None,
           "fetch_bytes_per_launch": mean.get("FETCH_SIZE", 0.0) * 1024, "write_bytes_per_launch": mean.get("WRITE_SIZE", 0.0) * 1024,
           "valu_wave_instr_per_launch": mean.get("SQ_INSTS_VALU", 0.0), "salu_wave_instr_per_launch": mean.get("SQ_INSTS_SALU", 0.0),
           "vmem_rd_wave_instr_per_launch": mean.get("SQ_INSTS_VMEM_RD", 0.0), "vmem_wr_wave_instr_per_launch": mean.get("SQ_INSTS_VMEM_WR", 0.0),
           "lds_wave_instr_per_launch": mean.get("SQ_INSTS_LDS", 0.0), "waves_per_launch": waves,
           # matrix cores (the wave-cooperative Newton direction, v_mfma_f32_16x16x1_4b_f32): instructions / MOPS / busy cycles per launch
           # from whichever counters this rocprofv3 offers on gfx950; utilisation = busy cycles / (1024 SIMDs x kernel cycles)
           "mfma": {k: mean[k] for k in sorted(mean) if "MFMA" in k} or None,
           "wait_frac": mean.get("SQ_WAIT_ANY", 0.0) / mean["SQ_WAVE_CYCLES"] if mean.get("SQ_WAVE_CYCLES") else None,
           "tcc_hit_rate": mean.get("TCC_HIT_sum", 0.0) / (mean.get("TCC_HIT_sum", 0.0) + mean.get("TCC_MISS_sum", 1.0)) if mean.get("TCC_HIT_sum") else None,
           # envs x sub-lanes all do useful (distinct or replicated-by-design) work only where the lanes hold an env: lanes / (waves x 64)
           "lane_utilisation": None, "wave_slot_occupancy": None, "kernel_ms_per_launch": args.kernel_ms or None,
           "note": "rocprofv3 --pmc, separate passes per counter group, mean over the step-kernel launches of `python bench.py --no-cpu-baseline "
                   "--no-extra-precision`; FETCH_SIZE / WRITE_SIZE in KiB x 1024 (raw: narrow requests, no x2 correction)"}
    if waves:
        # distinct environments per lane slot: 1 = one env per lane; below that the other lanes are sub-lanes of the same env
        # (they split the row / pair sweeps and replicate the serial parts)
        out["lane_utilisation"] = args.envs / (64.0 * waves)
    if out["mfma"] and args.kernel_ms and mean.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        out["mfma"]["utilisation"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * args.kernel_ms * 1e-3 * 2.4e9)
    if mean.get("SQ_WAVE_CYCLES") and args.kernel_ms:
        cycles = args.kernel_ms * 1e-3 * 2.4e9
        out["wave_slot_occupancy"] = mean["SQ_WAVE_CYCLES"] * 4 / (1024 * cycles)      # SQ_WAVE_CYCLES counts quad-cycles
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.json)
