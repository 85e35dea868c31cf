"""The two facts about gfx950 the lane programs rely on beyond the HIP language, checked on the device the suite runs on:
(a) the lane / register layout of v_mfma_f32_16x16x1_4b_f32 and the DPP controls row_newbcast / row_ror that newton_direction_wave
    (csrc/mw_phys.hpp) uses for the Hessian, the Cholesky and the triangular solves;
(b) a non-inlined 256-VGPR callee with SGPR spills into VGPR lanes, entered under a partial EXEC mask, leaves the caller's values alone
    (DESIGN.md 5 "register hazard": the generic pattern is handled by the calling convention).
The probes are small stand-alone HIP programs (tools/experiments/*.hip), compiled here with hipcc and run.

Lab notes, not product tests: they run only with MW_RUN_ISA_PROBES=1 (`MW_RUN_ISA_PROBES=1 pytest tests/test_gpu_isa_probes.py -m gpu`);
the default GPU suite skips them."""
import os
import shutil
import subprocess

import pytest

from tests.helpers import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _run(src, tmp_path, extra=()):
    if not os.environ.get("MW_RUN_ISA_PROBES"):
        pytest.skip("ISA probes run on request only (MW_RUN_ISA_PROBES=1)")
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "probe")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", *extra, "-o", exe, os.path.join(ROOT, "tools", "experiments", src)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(p.stdout)
    return p


@pytest.mark.gpu
def test_mfma_4block_layout_and_dpp_controls(tmp_path):
    p = _run("mfma_layout_probe.hip", tmp_path)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 of 1024 entries contradict" in p.stdout and "0 of 64 lanes wrong" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("-mllvm", "-enable-ipra=0")], ids=["default", "no-ipra"])
def test_partially_masked_call_of_a_full_register_callee_preserves_the_callers_values(tmp_path, extra):
    p = _run("partial_exec_call_probe.hip", tmp_path, extra)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "that sat out: 0" in p.stdout
