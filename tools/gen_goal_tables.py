#!/usr/bin/env python
"""Dump the goal (`rand_vec`) tables the reference's benchmark builders draw
(metaworld/__init__.py:114-179 `_make_tasks`) and the per-task constants of its env classes
(hand_init_pos, mocap box, goal_space, ...) by running the reference's own code on the oracle engine.

Outputs (committed, generated data):
  metaworld_amd/data/goals_seed<seed>.npz   MT1/<task>, MT10, MT25, MT50, ML10/ML45 train+test, ML1 (pick-place) tables: [50][6] per task
  metaworld_amd/data/task_constants.json     per task: model, hand_init_pos, hand_low/high, goal_low/high, ...
"""
import json
import os
import pickle
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


def table(bench):
    out = {}
    for t in bench.train_tasks:
        d = pickle.loads(t.data)
        rv = np.asarray(d["rand_vec"], dtype=np.float64)
        if rv.size == 3:
            rv = np.concatenate([rv, np.zeros(3)])
        out.setdefault(t.env_name, []).append(rv)
    return {k: np.array(v) for k, v in out.items()}


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 42
    consts_only = "--consts-only" in sys.argv
    from oracle import refshim
    refshim.install()
    import metaworld
    from metaworld.env_dict import ALL_V3_ENVIRONMENTS, MT10_V3
    names = list(ALL_V3_ENVIRONMENTS.keys())
    data = {}
    consts = {}
    for i, name in enumerate(names):
        mt1 = metaworld.MT1(name, seed=seed)
        data[f"MT1/{name}"] = table(mt1)[name]
        env = mt1.train_classes[name]()
        env.set_task(mt1.train_tasks[0])
        env.reset()
        consts[name] = dict(
            id=i, cls=type(env).__name__, model=os.path.splitext(os.path.basename(env.model_name))[0],
            hand_init_pos=list(map(float, env.hand_init_pos)), mocap_low=list(map(float, env.mocap_low)),
            mocap_high=list(map(float, env.mocap_high)), goal_low=list(map(float, env.goal_space.low)),
            goal_high=list(map(float, env.goal_space.high)),
            reset_low=list(map(float, env._random_reset_space.low)), reset_high=list(map(float, env._random_reset_space.high)),
            max_path_length=int(env.max_path_length),
            init_config={k: (list(map(float, np.ravel(v))) if np.ndim(v) else float(v)) for k, v in getattr(env, "init_config", {}).items()
                         if isinstance(v, (int, float, np.ndarray, list, tuple))},
            goal=list(map(float, np.ravel(getattr(env, "goal", [0, 0, 0])))),
            class_constants={k: float(getattr(env, k)) for k in ("TARGET_RADIUS", "OBJ_RADIUS", "liftThresh", "max_dist", "PAD_SUCCESS_MARGIN", "LIFT_THRESH", "LEVER_RADIUS")
                             if isinstance(getattr(env, k, None), (int, float))})
        print(name, "ok", flush=True)
    for bname, cls in (() if consts_only else (("MT10", metaworld.MT10), ("MT25", metaworld.MT25), ("MT50", metaworld.MT50))):
        tb = table(cls(seed=seed))
        for k, v in tb.items():
            data[f"{bname}/{k}"] = v
        print(bname, "ok", flush=True)

    def table_of(tasks):
        out = {}
        for t in tasks:
            d = pickle.loads(t.data)
            rv = np.asarray(d["rand_vec"], dtype=np.float64)
            out.setdefault(t.env_name, []).append(np.concatenate([rv, np.zeros(3)]) if rv.size == 3 else rv)
        return {k: np.array(v) for k, v in out.items()}
    for bname, mk in (() if consts_only else (("ML10", lambda: metaworld.ML10(seed=seed)), ("ML45", lambda: metaworld.ML45(seed=seed)),
                                              ("ML1", lambda: metaworld.ML1("pick-place-v3", seed=seed)))):
        b = mk()
        for split, tasks in (("train", b.train_tasks), ("test", b.test_tasks)):
            for k, v in table_of(tasks).items():
                data[f"{bname}-{split}/{k}"] = v
        print(bname, "ok", flush=True)
    os.makedirs(os.path.join(ROOT, "metaworld_amd", "data"), exist_ok=True)
    if not consts_only:
        np.savez_compressed(os.path.join(ROOT, "metaworld_amd", "data", f"goals_seed{seed}.npz"), **data)
    with open(os.path.join(ROOT, "metaworld_amd", "data", "task_constants.json"), "w") as f:
        json.dump(dict(all_v3=names, mt10=list(MT10_V3.keys()), tasks=consts), f, indent=1)


if __name__ == "__main__":
    main()
