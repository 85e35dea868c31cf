"""libmwgpu.so loads and exports every symbol include/mwgpu.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    import __graft_entry__ as g
    from metaworld_amd import native
    path = g.build_gpu()
    hdr = open(os.path.join(ROOT, "include", "mwgpu.h")).read()
    declared = set(re.findall(r"\b(mw_[a-z_]+)\s*\(", hdr))
    import ctypes
    dll = ctypes.CDLL(path)
    for sym in declared:
        assert hasattr(dll, sym), sym
    assert {"mw_" + s for s in native.EXPORTED_SYMBOLS} <= declared


def test_product_path_fails_loudly_without_gpu_or_library(tmp_path):
    from metaworld_amd import native
    with pytest.raises(RuntimeError):
        native.Lib(str(tmp_path / "missing.so"), "mw_")
    import torch
    if not torch.cuda.is_available():
        lib = native.load()
        with pytest.raises(RuntimeError):
            native.Context(lib)     # hipGetDeviceCount == 0 -> error, never a CPU fallback


def test_bench_cpu_baseline_leg_runs_on_the_oracle():
    """bench.py's `cpu_baseline` (the oracle engine timed on a bounded sample of the task mix) works without a GPU"""
    import bench
    r = bench.cpu_baseline(["reach-v3", "box-close-v3"], seconds=0.5)
    assert r["kind"] == "port" and r["cores"] == (os.cpu_count() or 1) and r["unit"] == "env-steps/s" and 1e2 < r["value"] < 1e7
    assert 1e2 < r["single_core_value"] < 1e6 and "2 tasks in equal shares" in r["sample"]
    assert r["full_step_port"]["value"] > 0          # physics + obs + reward on the host cores (host build of the lane programs)


def test_every_entry_point_is_mapped_to_the_reference_in_integration_md():
    hdr = open(os.path.join(ROOT, "include", "mwgpu.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"\b(mw_[a-z_]+)\s*\(", hdr))
    assert not [s for s in sorted(declared) if s not in doc]
