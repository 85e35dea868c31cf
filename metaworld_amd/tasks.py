"""Host-side task registry: which compiled model, which frames ("probes") and which constants each
Meta-World v3 task needs.  Mirrors the per-class data of metaworld/envs/sawyer_*_v3.py
(`_get_pos_objects`, `_get_quat_objects`, hand/goal boxes); the numeric
constants come from metaworld_amd/data/task_constants.json, dumped from the reference classes by
tools/gen_goal_tables.py.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import native
from .mjcf import load_model
from .pack import pack_model

_HERE = os.path.dirname(os.path.abspath(__file__))
QUAT_SCIPY, QUAT_MUJOCO, QUAT_ZERO, QUAT_IDENT, QUAT_NONE = range(5)

COMMON_PROBES = [("body", "hand"), ("body", "rightclaw"), ("body", "leftclaw"), ("body", "rightpad"),
                 ("body", "leftpad"), ("site", "rightEndEffector"), ("site", "leftEndEffector")]

# name -> dict(objs=[(pos_probe, quat_probe, quat_mode, offset)], extra probes, ...).  Only tasks listed here
# have device-side reward/reset code (metaworld_amd/csrc/mw_tasks.hpp `task_supported`).
def _obj(pos, quat, mode, off=(0, 0, 0)):
    return (pos, quat, mode, off)


B, G, S = "body", "geom", "site"
_PUCK = [_obj((B, "obj"), (G, "objGeom"), QUAT_SCIPY)]
_GEOMPUCK = [_obj((G, "objGeom"), (G, "objGeom"), QUAT_SCIPY)]
TASK_DEFS = {
    "reach-v3": dict(objs=_PUCK),
    "reach-wall-v3": dict(objs=_PUCK),
    "push-v3": dict(objs=_PUCK),
    "pick-place-v3": dict(objs=_PUCK),
    "push-back-v3": dict(objs=_GEOMPUCK),
    "push-wall-v3": dict(objs=_GEOMPUCK),
    "pick-place-wall-v3": dict(objs=_GEOMPUCK),
    "sweep-v3": dict(objs=[_obj((B, "obj"), (B, "obj"), QUAT_MUJOCO)]),
    "sweep-into-v3": dict(objs=_PUCK),
    "soccer-v3": dict(objs=[_obj((B, "soccer_ball"), (B, "soccer_ball"), QUAT_SCIPY)], reloc=["goal_whole"]),
    "hand-insert-v3": dict(objs=[_obj((B, "obj"), (B, "obj"), QUAT_MUJOCO)]),
    "bin-picking-v3": dict(objs=[_obj((B, "obj"), (B, "obj"), QUAT_MUJOCO)], extra=[(B, "bin_goal")]),
    "button-press-topdown-v3": dict(objs=[_obj((B, "button"), (B, "button"), QUAT_MUJOCO, (0, 0, 0.193))],
                                    extra=[(S, "hole"), (S, "buttonStart")], reloc=["box"], geom="btnGeom"),
    "button-press-topdown-wall-v3": dict(objs=[_obj((B, "button"), (B, "button"), QUAT_MUJOCO, (0, 0, 0.193))],
                                         extra=[(S, "hole"), (S, "buttonStart")], reloc=["box"], geom="btnGeom"),
    "button-press-v3": dict(objs=[_obj((B, "button"), (B, "button"), QUAT_MUJOCO, (0, -0.193, 0))],
                            extra=[(S, "hole"), (S, "buttonStart")], reloc=["box"], geom="btnGeom"),
    "button-press-wall-v3": dict(objs=[_obj((B, "button"), (B, "button"), QUAT_MUJOCO, (0, -0.193, 0))],
                                 extra=[(S, "hole"), (S, "buttonStart")], reloc=["box"], geom="btnGeom"),
    "coffee-button-v3": dict(objs=[_obj((S, "buttonStart"), None, QUAT_IDENT)], reloc=["coffee_machine"], geom="mug"),
    "coffee-pull-v3": dict(objs=[_obj((B, "obj"), (G, "mug"), QUAT_SCIPY)], reloc=["coffee_machine"], geom="mug"),
    "coffee-push-v3": dict(objs=[_obj((B, "obj"), (G, "mug"), QUAT_SCIPY)], reloc=["coffee_machine"], geom="mug"),
    "dial-turn-v3": dict(objs=[_obj((B, "dial"), (B, "dial"), QUAT_MUJOCO)], extra=[(B, "dial")], reloc=["dial"],
                         joints=["knob_Joint_1"], dial=True),
    "door-close-v3": dict(objs=[_obj((G, "handle"), (G, "handle"), QUAT_SCIPY)], reloc=["door"], joints=["doorjoint"]),
    "door-open-v3": dict(objs=[_obj((G, "handle"), (G, "handle"), QUAT_SCIPY)], reloc=["door"], joints=["doorjoint"]),
    "door-lock-v3": dict(objs=[_obj((S, "lockStartLock"), (B, "door_link"), QUAT_MUJOCO)], extra=[(B, "lock_link")], reloc=["door"]),
    "door-unlock-v3": dict(objs=[_obj((S, "lockStartUnlock"), (B, "door_link"), QUAT_MUJOCO)], extra=[(B, "lock_link")], reloc=["door"]),
    "drawer-close-v3": dict(objs=[_obj((B, "drawer_link"), None, QUAT_ZERO, (0, -0.16, 0.05))], reloc=["drawer"]),
    "drawer-open-v3": dict(objs=[_obj((B, "drawer_link"), (B, "drawer_link"), QUAT_MUJOCO, (0, -0.16, 0))], reloc=["drawer"]),
    "faucet-open-v3": dict(objs=[_obj((S, "handleStartOpen"), (B, "faucetBase"), QUAT_MUJOCO, (0, 0, -0.01))], reloc=["faucetBase"]),
    "faucet-close-v3": dict(objs=[_obj((S, "handleStartClose"), (B, "faucetBase"), QUAT_MUJOCO, (0, 0, -0.01))], reloc=["faucetBase"]),
    "handle-press-side-v3": dict(objs=[_obj((S, "handleStart"), None, QUAT_ZERO)], extra=[(S, "goalPress")], reloc=["box"]),
    "handle-press-v3": dict(objs=[_obj((S, "handleStart"), None, QUAT_ZERO)], extra=[(S, "goalPress")], reloc=["box"]),
    "handle-pull-side-v3": dict(objs=[_obj((S, "handleCenter"), None, QUAT_ZERO)], extra=[(S, "goalPull")], reloc=["box"],
                                extra_v1=[(S, "handleStart")]),
    "handle-pull-v3": dict(objs=[_obj((S, "handleRight"), None, QUAT_ZERO)], extra=[(S, "goalPull")], reloc=["box"],
                           c_model_v1=[("site_pos", "handleStart", 2)]),
    "lever-pull-v3": dict(objs=[_obj((S, "leverStart"), (G, "objGeom"), QUAT_SCIPY)], reloc=["lever"], joints=["LeverAxis"]),
    "window-open-v3": dict(objs=[_obj((S, "handleOpenStart"), None, QUAT_ZERO)], reloc=["window"], joints=["window_slide"]),
    "window-close-v3": dict(objs=[_obj((S, "handleCloseStart"), None, QUAT_ZERO)], reloc=["window"], joints=["window_slide"]),
    "plate-slide-v3": dict(objs=[_obj((G, "puck"), (G, "puck"), QUAT_SCIPY)], reloc=["puck_goal"], geom="puck"),
    "plate-slide-side-v3": dict(objs=[_obj((G, "puck"), (G, "puck"), QUAT_SCIPY)], geom="puck"),
    # (-side and -back only write data.body("puck_goal").xpos, which the next mj_forward overwrites: the goal body stays where the
    #  XML puts it; they must NOT share a relocatable-body slot with the variant that moves it, see model_key)
    "plate-slide-back-v3": dict(objs=[_obj((G, "puck"), (G, "puck"), QUAT_SCIPY)], geom="puck"),
    "plate-slide-back-side-v3": dict(objs=[_obj((G, "puck"), (G, "puck"), QUAT_SCIPY)], reloc=["puck_goal"], geom="puck"),
    "assembly-v3": dict(objs=[_obj((S, "RoundNut-8"), (B, "RoundNut"), QUAT_MUJOCO)], extra=[(S, "RoundNut")], reloc=["peg"], geom="WrenchHandle"),
    "disassemble-v3": dict(objs=[_obj((S, "RoundNut-8"), (B, "RoundNut"), QUAT_MUJOCO)], extra=[(S, "RoundNut")], reloc=["peg"], geom="WrenchHandle"),
    "hammer-v3": dict(objs=[_obj((B, "hammer"), (B, "hammer"), QUAT_MUJOCO), _obj((B, "nail_link"), (B, "nail_link"), QUAT_MUJOCO)],
                      extra=[(S, "goal")], reloc=["box"], joints=["NailSlideJoint"], geom="HammerHandle",
                      extra_v1=[(G, "HammerHead"), (S, "nailHead")]),
    "basketball-v3": dict(objs=[_obj((B, "bsktball"), (B, "bsktball"), QUAT_MUJOCO)], extra=[(S, "goal")], reloc=["basket_goal"],
                          extra_v1=[(G, "objGeom")]),
    "box-close-v3": dict(objs=[_obj((B, "top_link"), (B, "top_link"), QUAT_MUJOCO)], reloc=["boxbody"], geom="BoxHandleGeom",
                         c_model=[("body_pos", "boxbody", 2)], extra_v1=[(G, "BoxHandleGeom")]),
    "pick-out-of-hole-v3": dict(objs=[_obj((B, "obj"), (B, "obj"), QUAT_MUJOCO)], extra_v1=[(G, "objGeom")]),
    "shelf-place-v3": dict(objs=[_obj((B, "obj"), (G, "objGeom"), QUAT_SCIPY)], reloc=["shelf"], c_model=[("site_pos", "goal", None)]),
    "peg-insert-side-v3": dict(objs=[_obj((S, "pegGrasp"), (S, "pegGrasp"), QUAT_SCIPY)], reloc=["box"],
                               extra=[(S, "pegHead"), (S, "bottom_right_corner_collision_box_1"), (S, "top_left_corner_collision_box_1"),
                                      (S, "bottom_right_corner_collision_box_2"), (S, "top_left_corner_collision_box_2")]),
    "peg-unplug-side-v3": dict(objs=[_obj((S, "pegEnd"), (B, "plug1"), QUAT_MUJOCO)], reloc=["box"]),
    "stick-push-v3": dict(objs=[_obj((B, "stick"), (B, "stick"), QUAT_SCIPY), _obj((S, "insertion"), None, QUAT_ZERO, (0, 0.09, 0))],
                          extra=[(B, "object"), (S, "stick_end")], c=[0.02]),
    "stick-pull-v3": dict(objs=[_obj((B, "stick"), (B, "stick"), QUAT_SCIPY), _obj((S, "insertion"), None, QUAT_ZERO)],
                          extra=[(B, "object"), (S, "stick_end")], c=[0.02]),
}

with open(os.path.join(_HERE, "data", "task_constants.json")) as _f:
    _CONST = json.load(_f)
ALL_V3 = _CONST["all_v3"]          # index = MT50 one-hot id (metaworld/env_dict.py:217-270)
MT10 = _CONST["mt10"]              # MT10 order (metaworld/env_dict.py:278-291)
TASK_CONST = _CONST["tasks"]


# tasks whose `reward_function_version="v1"` branch is restated on the device (csrc/mw_tasks_v1.hpp), by MT50 id there
V1_TASKS = ["reach-v3", "reach-wall-v3",
            "button-press-topdown-v3", "button-press-topdown-wall-v3", "button-press-v3", "button-press-wall-v3", "coffee-button-v3",
            "dial-turn-v3", "door-close-v3", "door-lock-v3", "door-open-v3", "door-unlock-v3", "drawer-close-v3", "drawer-open-v3",
            "faucet-open-v3", "faucet-close-v3", "handle-press-side-v3", "handle-press-v3", "handle-pull-side-v3", "handle-pull-v3",
            "lever-pull-v3", "plate-slide-v3", "plate-slide-side-v3", "plate-slide-back-v3", "plate-slide-back-side-v3",
            "window-open-v3", "window-close-v3",
            "push-v3", "push-back-v3", "push-wall-v3", "coffee-push-v3", "coffee-pull-v3", "soccer-v3", "sweep-v3", "sweep-into-v3",
            "hand-insert-v3",
            "pick-place-v3", "pick-place-wall-v3", "shelf-place-v3", "pick-out-of-hole-v3", "basketball-v3", "box-close-v3",
            "peg-insert-side-v3", "bin-picking-v3", "peg-unplug-side-v3", "assembly-v3", "disassemble-v3", "hammer-v3",
            "stick-push-v3", "stick-pull-v3"]


def supported_tasks():
    return [t for t in ALL_V3 if t in TASK_DEFS]


_goal_cache = {}


def benchmark_task_names(benchmark: str, env_name=None):
    """Task names of a reference benchmark split in construction order: MT1 / ML1-train / ML1-test (one task), MT10, MT25,
    MT50, ML10-train, ML10-test, ML25-*, ML45-* (metaworld/env_dict.py; dumped to data/benchmarks.json)."""
    if benchmark in ("MT1", "ML1-train", "ML1-test"):
        assert env_name is not None
        return [env_name]
    with open(os.path.join(_HERE, "data", "benchmarks.json")) as f:
        B = json.load(f)
    if benchmark not in B:
        raise ValueError(f"unsupported benchmark {benchmark}")
    return list(B[benchmark])


def goal_table(benchmark: str, task: str, seed: int = 42) -> np.ndarray:
    """[50][6] rand_vecs the reference's `benchmark`(seed) assigns to `task`, for any seed: the physics-free restatement
    of `_make_tasks` in metaworld_amd/goals.py (pinned against tables dumped from the reference itself,
    metaworld_amd/data/goals_seed*.npz, by tests/test_goal_tables.py).  ML1's test split uses seed + 1, every other
    split of every benchmark its own stream started at `seed` (metaworld/__init__.py:185-400)."""
    from . import goals
    single = benchmark in ("MT1", "ML1-train", "ML1-test")
    s = seed + 1 if benchmark == "ML1-test" else seed
    key = (benchmark, s, task) if single else (benchmark, s)
    if key not in _goal_cache:
        with open(os.path.join(_HERE, "data", "task_constants.json")) as f:
            C = json.load(f)
        _goal_cache[key] = goals.make_tables(benchmark_task_names(benchmark, task), s, C["tasks"])
    tab = _goal_cache[key]
    if task not in tab:
        raise KeyError(f"{benchmark}/{task}")
    return tab[task]


def custom_goal_tables(task_list, seed: int) -> dict:
    """{task: [50][6]} of `_make_tasks` over an arbitrary class list (CustomML `metaworld/__init__.py:370-395`; one-task lists
    give the MT1(seed) tables the custom MT entry point builds per env, `:741-767`)."""
    from . import goals
    key = ("custom", tuple(task_list), seed)
    if key not in _goal_cache:
        with open(os.path.join(_HERE, "data", "task_constants.json")) as f:
            C = json.load(f)
        _goal_cache[key] = goals.make_tables(list(task_list), seed, C["tasks"])
    return _goal_cache[key]


_model_cache = {}


def compiled_model(name):
    if name not in _model_cache:
        _model_cache[name] = load_model(os.path.join(_HERE, "models", name + ".npz"))
    return _model_cache[name]


def model_probes(model_name, v1=False):
    """union of the probes of every task that uses this model -> (list, {task: [probe ids by role]}); v1: + the frames only the
    v1 reward functions read (`extra_v1`)"""
    probes = list(COMMON_PROBES)
    roles = {}
    for task, d in TASK_DEFS.items():
        if TASK_CONST[task]["model"] != model_name:
            continue
        ids = list(range(len(COMMON_PROBES)))
        for (pp, qp, qm, off) in d["objs"]:
            for p in (pp, qp):
                if p is None:
                    ids.append(-1)
                    continue
                if p not in probes:
                    probes.append(p)
                ids.append(probes.index(p))
        while len(ids) < 7 + 4:
            ids.append(-1)
        for p in d.get("extra", []) + (d.get("extra_v1", []) if v1 else []):
            if p not in probes:
                probes.append(p)
            ids.append(probes.index(p))
        while len(ids) < native.NPROBE:
            ids.append(-1)
        roles[task] = ids
    return probes, roles


with open(os.environ.get("MW_MODEL_CAPS") or os.path.join(_HERE, "data", "model_caps.json")) as _f:          # (MW_MODEL_CAPS: experiments with another table)
    MODEL_CAPS = json.load(_f)     # per model: contact / constraint-row capacities = 2 x the demand measured on the GPU over whole episodes of random actions at MT50 @ 4096 (tools/measure_caps_gpu.py); step_ms_lpb4 = critical-path weight of the scene inside the MT50 @ 4096 bench workload at 4 lanes per workgroup (largest per-env cycle count of a step / 2.4e6, tools/mix_timing.py, max over the tasks sharing the scene), step_ms_lpb8 = the same scaled by the scene's isolated 8-lane / 4-lane step-time ratio (tools/per_task_timing.py)


def model_key(task):
    """Tasks share one packed model (= one group of the runtime) when they use the same scene AND relocate the same bodies: a body
    that is relocatable in the tables reads its position from a per-env slot, which only the tasks that write
    `model.body(X).pos` at reset fill in (plate-slide-side / -back never do: their goal body must keep its XML position)."""
    return TASK_CONST[task]["model"], tuple(TASK_DEFS[task].get("reloc", []))


def packed_model(model_name, maxcon=None, maxefc=None, v1=False, reloc_bodies=None, **kw):
    caps = MODEL_CAPS.get(model_name, dict(maxcon=64, maxefc=256))
    maxcon = maxcon or caps["maxcon"]
    maxefc = maxefc or caps["maxefc"]
    probes, roles = model_probes(model_name, v1=v1)
    reloc = list(reloc_bodies) if reloc_bodies is not None else []
    if reloc_bodies is None:          # (union over the scene's tasks: only right when they all relocate the same bodies, see model_key)
        for task, d in TASK_DEFS.items():
            if TASK_CONST[task]["model"] == model_name:
                for b in d.get("reloc", []):
                    if b not in reloc:
                        reloc.append(b)
    pk = pack_model(compiled_model(model_name), probes, reloc_bodies=reloc, maxcon=maxcon, maxefc=maxefc, **kw)
    for k in ("step_ms_lpb4", "step_ms_lpb8", "lanes_per_block"):          # (lanes_per_block: an explicit assignment, experiments) measured step time of the scene at 4 / 8 lanes per workgroup: the runtime's
        if k in caps:                                     # lanes-per-workgroup assignment ranks the groups by it (mw_runtime.hpp finalize)
            pk["options"][k] = caps[k]
    return pk, roles, reloc


def task_struct(task, model_index, roles, reloc, onehot_id, partially_observable=False, v1=False) -> native.MwTask:
    c, d = TASK_CONST[task], TASK_DEFS[task]
    t = native.MwTask()
    t.kind, t.model, t.onehot_id = c["id"], model_index, onehot_id
    for i, p in enumerate(roles[task]):
        t.probe[i] = p
    t.nobj = len(d["objs"])
    for i in range(2):
        t.quat_mode[i] = d["objs"][i][2] if i < t.nobj else QUAT_NONE
        for k in range(3):
            t.obj_off[i][k] = d["objs"][i][3][k] if i < t.nobj else 0.0
    for i in range(4):
        t.qadr[i] = t.dadr[i] = t.geom[i] = -1
    m = compiled_model(c["model"])
    gn = m.names["geom"]
    main = d.get("geom", "objGeom")
    t.geom[0] = gn.get(main, -1)
    t.geom[1], t.geom[2] = gn["leftpad_geom"], gn["rightpad_geom"]
    for i, jn in enumerate(d.get("joints", [])):
        j = m.names["joint"][jn]
        t.qadr[i], t.dadr[i] = int(m.arrays["jnt_qposadr"][j]), int(m.arrays["jnt_dofadr"][j])
    for i in range(2):
        t.reloc[i] = -1
    for i, b in enumerate(d.get("reloc", [])):
        t.reloc[i] = reloc.index(b)
    t.partially_observable = int(partially_observable)
    t.max_path_length = c["max_path_length"]
    t.hand_init[:] = c["hand_init_pos"]
    t.mocap_low[:] = c["mocap_low"]
    t.mocap_high[:] = c["mocap_high"]
    t.goal_low[:] = c["goal_low"]
    t.goal_high[:] = c["goal_high"]
    oi = c.get("init_config", {}).get("obj_init_pos", [0, 0, 0])
    for i in range(3):
        t.c[i] = oi[i] if i < len(oi) else 0.0
        t.c[3 + i] = c["goal"][i] if i < len(c["goal"]) else 0.0
    k = 6
    for v in d.get("c", []):
        t.c[k] = v
        k += 1
    for arr, name, comp in d.get("c_model", []) + (d.get("c_model_v1", []) if v1 else []):      # constants read from the compiled model (XML values)
        kind = "body" if arr == "body_pos" else "site"
        vals = m.arrays[arr][m.names[kind][name]]
        for v in ([vals[comp]] if comp is not None else vals):
            t.c[k] = float(v)
            k += 1
    return t
