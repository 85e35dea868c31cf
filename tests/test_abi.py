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
    assert r["kind"] == "port" and r["cores"] == __import__("bench").usable_cores() and r["parallel_speedup"] > 0 and r["unit"] == "env-steps/s" and 1e2 < r["value"] < 1e7
    assert 1e2 < r["single_core_value"] < 1e6 and "2 tasks in equal shares" in r["sample"]
    assert r["full_step_port"]["value"] > 0          # physics + obs + reward on the host cores (host build of the lane programs)


def test_every_entry_point_is_mapped_to_the_reference_in_integration_md():
    hdr = open(os.path.join(ROOT, "include", "mwgpu.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"\b(mw_[a-z_]+)\s*\(", hdr))
    assert not [s for s in sorted(declared) if s not in doc]


def test_bench_quotes_counters_only_from_a_profile_of_the_same_sources(tmp_path):
    """bench.py's roofline.traffic / alu_issue come from a committed rocprofv3 summary ONLY if it was taken on the same workload with
    the same device sources (content hash); anything else gives traffic = null (round 2 replayed a stale profile keyed by the
    workload string)"""
    import json
    import bench
    from metaworld_amd import native
    h = native.source_hash()
    assert len(h) == 16 and h == native.source_hash()
    wl = "MT50 sync-vector, 4096 envs/GPU, fp64, random actions"
    (tmp_path / "r02_mt50_fp64_pmc.json").write_text(json.dumps({"workload": wl, "fetch_bytes_per_launch": 1.0}))                      # no hash at all
    (tmp_path / "r03_mt50_fp64_pmc.json").write_text(json.dumps({"workload": wl, "source_hash": "0" * 16}))                            # another tree
    assert bench.matching_profile("fp64", wl, h, str(tmp_path)) == (None, None)
    (tmp_path / "r03b_mt50_fp64_pmc.json").write_text(json.dumps({"workload": "other", "source_hash": h}))                            # another workload
    assert bench.matching_profile("fp64", wl, h, str(tmp_path))[0] is None
    (tmp_path / "r03a_mt50_fp64_pmc.json").write_text(json.dumps({"workload": wl, "source_hash": h, "fetch_bytes_per_launch": 2.0}))
    pj, path = bench.matching_profile("fp64", wl, h, str(tmp_path))
    assert pj["fetch_bytes_per_launch"] == 2.0 and path.endswith("r03a_mt50_fp64_pmc.json")
    assert bench.matching_profile("fp32", wl, h, str(tmp_path)) == (None, None)
    # the committed profiles of this tree: whichever matches must belong to these sources
    real, _ = bench.matching_profile("fp64", wl, h)
    assert real is None or real["source_hash"] == h
