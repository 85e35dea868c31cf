"""Engine pin against the REAL MuJoCo.  tools/dump_reference_traces.py records traces with the genuine
mujoco + gymnasium + metaworld wherever they can be installed; these tests replay them one step from a synchronised state.
They SKIP while tests/golden_mujoco/ is empty -- which it is in this repo: no mujoco wheel exists in the build container or on
the GPU box, so parity with MuJoCo 3.3.0 is unpinned and these tests are the hook that pins it."""
import glob
import os

import numpy as np
import pytest

from metaworld_amd import tasks as T
from tests.helpers import ROOT, make_env, replay_trace

DIR = os.path.join(ROOT, "tests", "golden_mujoco")
HAVE = sorted(glob.glob(os.path.join(DIR, "trace_*_seed42.npz")))
needs_traces = pytest.mark.skipif(not HAVE, reason="no MuJoCo traces recorded: run `pip install -r tools/pin/requirements.txt && bash tools/pin/run_pin.sh` where the mujoco wheel installs")


def test_dump_tool_reports_the_missing_stack_instead_of_faking_it():
    from tools import dump_reference_traces as D
    ok, why = D.real_stack_available()
    if not ok:
        assert "Error" in why or "stand-in" in why


def _check(lib, task, precision, tol_obs, tol_rew):
    G = dict(np.load(os.path.join(DIR, f"trace_{task}_seed42.npz")))
    env = make_env(lib, task, n=len(G["goal_idx"]), precision=precision)
    r = replay_trace(env, G, sync=True, steps=30)
    env.close()
    assert r["obs"] < tol_obs and r["reward"] < tol_rew and r["success_mismatch"] == 0, r


@needs_traces
@pytest.mark.parametrize("task", T.ALL_V3)
def test_host_build_reproduces_mujoco(hostsim, task):
    _check(hostsim, task, "fp64", 1e-5, 1e-5)


@needs_traces
@pytest.mark.gpu
@pytest.mark.parametrize("task", T.ALL_V3)
def test_gpu_reproduces_mujoco(gpulib, task):
    _check(gpulib, task, "fp64", 1e-5, 1e-5)
