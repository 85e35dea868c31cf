"""The three memory placements of the body-level chains (Env::chain_lds, mw_common.hpp: 0 = everything through the column store,
1 = cdof / qvel / qpos in the scratchpad slots in front of the constraint rows + the composite inertias, 2 = also the body frames of
the kinematics walk and the velocities / accelerations / forces of the recursive Newton-Euler passes) are three ways of moving the
SAME numbers: random-action rollouts of contact-rich tasks on the host build must be bit-identical in all of them, in both
precisions and with emulated sub-lanes.  (The level is fixed per process: one subprocess per level.)"""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
from metaworld_amd import native
from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
lib = native.load("mwh_", g.build_host_harness())
for prec in ("fp64", "fp32"):
    for t in ("hammer-v3", "door-unlock-v3", "stick-pull-v3"):
        env = MetaWorldGpuVectorEnv("MT1", t, num_envs=3, seed=3, precision=prec, lib=lib, max_episode_steps=60)
        env.reset()
        rng = np.random.default_rng(1)
        h = hashlib.sha256()
        for s in range(70):
            o, r, te, tr, info = env.step(rng.uniform(-1, 1, (3, 4)).astype(np.float32))
            h.update(o.tobytes()); h.update(r.tobytes())
        for e in range(3):
            for c in ("qpos", "qvel", "warm", "cdof", "qM", "qacc", "qfrc_constraint"):
                h.update(env.ctx.read(e, c).tobytes())
        assert env.status()["flags"] == 0
        print(prec, t, h.hexdigest())
        env.close()
""" % ROOT


def _run(level, nsub):
    env = dict(os.environ, MW_CHAIN_LDS=str(level), MW_NSUB=str(nsub))
    p = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("nsub", [1, 8])
def test_chain_levels_are_bit_identical(nsub):
    import __graft_entry__ as g
    g.build_host_harness()
    ref = _run(0, nsub)
    assert len(ref.strip().splitlines()) == 6
    for level in (1, 2):
        assert _run(level, nsub) == ref, f"chain level {level} changed a result (nsub {nsub})"
