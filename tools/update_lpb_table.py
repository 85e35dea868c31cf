#!/usr/bin/env python
"""Refresh the two numbers per scene that the runtime's lanes-per-workgroup assignment ranks the groups by (mw_runtime.hpp finalize,
metaworld_amd/data/model_caps.json): step_ms_lpb4 = the largest per-environment cycle count of one step of the scene INSIDE the
MT50 @ 4096 bench workload with 4 lanes per workgroup everywhere (MW_LANES_PER_BLOCK=4 MW_MIX_JSON=<file> tools/mix_timing.py, timing
build) / 2.4e6, maximum over the tasks that share the scene; step_ms_lpb8 keeps the scene's measured 8-lane / 4-lane ratio.
usage: tools/update_lpb_table.py gpurun_out/mix4.json [out.json]      (default: rewrite metaworld_amd/data/model_caps.json)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaworld_amd import tasks as T  # noqa: E402

mix = json.load(open(sys.argv[1]))
path = os.path.join(ROOT, "metaworld_amd", "data", "model_caps.json")
caps = json.load(open(path))
t4 = {}
for task, v in mix.items():
    model = T.TASK_CONST[task]["model"]
    t4[model] = max(t4.get(model, 0.0), v["max_kcyc"] / 2.4e3)
for model, ms in sorted(t4.items(), key=lambda kv: -kv[1]):
    c = caps[model]
    ratio = c["step_ms_lpb8"] / c["step_ms_lpb4"] if c.get("step_ms_lpb4") else 1.3
    print(f"{model:34s} step_ms_lpb4 {c.get('step_ms_lpb4', 0):6.3f} -> {ms:6.3f}   (8 / 4 ratio {ratio:.2f})")
    c["step_ms_lpb4"] = round(ms, 3)
    c["step_ms_lpb8"] = round(ms * ratio, 3)
out = sys.argv[2] if len(sys.argv) > 2 else path
json.dump(caps, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out)
