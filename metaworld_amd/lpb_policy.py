"""Lanes per workgroup from measured wave times.

The step kernel runs one wave per SIMD and a launch lasts as long as its slowest wave, so the batch is spread over the
chip's 1024 wave slots (256 CUs x 4 SIMDs) by giving every model group its own "lanes per workgroup" l: a wave then
carries l environments of that model and 64/l cooperating sub-lanes per environment.  Heavy scenes want
few environments per wave, light scenes can share one, and the slots are a budget.  `data/lpb_costs.json` holds tau(model, l):
the slowest wave of each model's group, in ms per step, measured on an MI355X by tools/calibrate_lpb.py (MT50 @ 4096
envs, random actions, every group forced to the same l).  `choose` picks, for the batch at hand, the assignment that
minimises the slowest wave subject to the slot budget; models without a measurement, or batches that do not fit one round
of waves at any measured l, fall back to the runtime's own proxy (mw_runtime.hpp finalize).

Status (round 1): NO table is shipped, so `choose` returns {} and the runtime's proxy decides.  The first calibration
(profiles/r01_wave_times_by_lpb.json: mean wave 2.3 ms, slowest wave 5.8 ms at 4 lanes -- the kernel is bound by the
serial chain of its heaviest scene, sawyer_box, whatever its lanes) predicted 5.8 ms per launch with only 262 waves but
measured 9.5 ms (the proxy's 1017 waves: 6.2 ms): the table holds per-wave time AVERAGES over an early window of the
episode, while a launch lasts as long as the slowest wave OF THAT LAUNCH, and packing every group up to the critical path
puts hundreds of 16-32-lane waves within reach of it.  The knob (`lanes_per_block` model option), the per-workgroup clock
(mw_wave_profile) and the calibration tool stay for the next attempt (per-launch maxima, whole-episode window)."""
from __future__ import annotations

import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
WAVE_SLOTS = 1024          # MI355X: 256 CUs x 4 SIMDs, one wave of the 512-VGPR lane programs per SIMD
_table = None


def table():
    global _table
    if _table is None:
        path = os.path.join(_HERE, "data", "lpb_costs.json")
        _table = json.load(open(path)) if os.path.exists(path) else {}
    return _table


def choose(envs_per_model: dict, precision: str = "fp32", wave_slots: int = WAVE_SLOTS, costs: dict | None = None) -> dict:
    """{model name: lanes per workgroup}, or {} = leave the choice to the runtime"""
    costs = table().get(precision, {}) if costs is None else costs
    if not envs_per_model or any(m not in costs for m in envs_per_model):
        return {}
    tau = {m: {int(l): float(t) for l, t in costs[m].items()} for m in envs_per_model}
    for theta in sorted({t for m in tau for t in tau[m].values()}):
        pick, waves = {}, 0
        for m, n in envs_per_model.items():
            ok = [l for l, t in tau[m].items() if t <= theta]
            if not ok:
                break
            pick[m] = max(ok)          # the fewest waves that still meet theta
            waves += -(-n // pick[m])
        else:
            if waves <= wave_slots:
                return pick
    return {}
