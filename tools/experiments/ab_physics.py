#!/usr/bin/env python
"""A/B of the raw physics against the oracle engine: scenario of tests/test_gpu_parity.py::test_gpu_physics_matches_oracle for a
given library variant (MW_LIB) and lanes-per-workgroup (MW_LANES_PER_BLOCK)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metaworld_amd import native  # noqa: E402
from tests.helpers import make_env, oracle_for  # noqa: E402

lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ.get("MW_LIB", "libmwgpu.so")))
for prec in ("fp64", "fp32"):
    env = make_env(lib, n=70, precision=prec)
    om, d = oracle_for("sawyer_reach_v3")
    d.mocap_pos[:] = [0, 0.6, 0.2]; d.mocap_quat[:] = [1, 0, 1, 0]; d.ctrl[:] = [-1, 1]
    env.ctx.debug("reset_data")
    for e in (0, 69):
        env.ctx.write(e, "mocap", [0, 0.6, 0.2]); env.ctx.write(e, "ctrl", [-1, 1])
    out = []
    for n in [int(x) for x in os.environ.get('MW_SEQ', '1,1,1,7,40').split(',')]:
        d.step(n); env.ctx.debug("substeps", n)
        ic = env.ctx.read_int(0, "icount", 24)
        out.append(f"{max(np.abs(env.ctx.read(e, 'qpos') - d.qpos).max() for e in (0, 69)):.1e}(it{ic[2]},nefc{ic[1]}/{d.nefc},ncon{ic[0]}/{d.ncon})")
    print(os.environ.get("MW_LIB", "libmwgpu.so"), "lpb", os.environ.get("MW_LANES_PER_BLOCK", "auto"), prec, "qpos err after 1,2,3,10,50 substeps:", " ".join(out), flush=True)
    env.close()
