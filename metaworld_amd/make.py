"""The reference's construction surface for the hot path, returning the GPU VectorEnv.

`make_mt_envs` / `make_ml_envs` / `make_ml_envs_train` / `make_ml_envs_test` mirror metaworld/__init__.py:460-618 (same names,
argument meaning and errors); `register_mw_envs` registers the same "Meta-World/..." vector ids (`:621-823`) under the
namespace "Meta-World-GPU/" when gymnasium is importable, so `gym.make_vec("Meta-World-GPU/MT50", num_envs=4096, seed=42)` is
the one-line switch.  As in the reference one `seed` feeds both the benchmark's goal tables and every sub-env's task-selection
stream.  `vector_strategy` is accepted and ignored (there is one kernel launch per step, not a worker pool); only
SAME_STEP auto-reset exists, the mode the reference's entry points use.
"""
from __future__ import annotations

from functools import partial

from . import tasks as T
from .vector_env import MetaWorldGpuVectorEnv

_MT = ("MT10", "MT25", "MT50")
_ML = ("ML10", "ML25", "ML45")


def _check_autoreset(mode):
    name = getattr(mode, "value", mode)
    if str(name).replace("_", "").lower() not in ("samestep",):
        raise NotImplementedError(f"autoreset_mode {mode!r}: the step kernel fuses the SAME_STEP auto-reset")


def make_mt_envs(name, seed=None, num_tasks=None, vector_strategy="sync", autoreset_mode="SameStep", num_envs=None, **kwargs):
    """metaworld/__init__.py:460-513.  `name` is a task name (the MT1 case; the reference returns one wrapped env, here a
    VectorEnv of `num_envs` copies) or "MT10" / "MT25" / "MT50"."""
    _check_autoreset(autoreset_mode)
    gs = 42 if seed is None else seed          # the reference draws unseeded goals for seed=None; a fixed table is reproducible
    if name in T.ALL_V3:          # `num_tasks=num_tasks or 1` (metaworld/__init__.py:477)
        return MetaWorldGpuVectorEnv("MT1", name, num_envs=num_envs or 1, seed=seed, goal_seed=gs, num_tasks=num_tasks or 1, **kwargs)
    if name in _MT:               # `num_tasks=num_tasks or default_num_tasks` (:501): the one-hot may be wider than the benchmark
        return MetaWorldGpuVectorEnv(name, num_envs=num_envs, seed=seed, goal_seed=gs, num_tasks=num_tasks, **kwargs)
    raise ValueError("Invalid MT env name. Must either be a valid Metaworld task name (e.g. 'reach-v3'), 'MT10' or 'MT50'.")


def make_ml_envs(name, seed=None, meta_batch_size=20, total_tasks_per_cls=None, split="train", vector_strategy="sync",
                 autoreset_mode="SameStep", num_envs=None, **kwargs):
    """metaworld/__init__.py:565-593 (+ `_make_ml_envs_inner` :516-562): `meta_batch_size` sub-envs, each class's goals dealt
    round-robin over its sub-envs."""
    _check_autoreset(autoreset_mode)
    if split not in ("train", "test"):
        raise ValueError(split)
    gs = 42 if seed is None else seed
    if name in T.ALL_V3:
        return MetaWorldGpuVectorEnv(f"ML1-{split}", name, seed=seed, goal_seed=gs, meta_batch_size=meta_batch_size,
                                     total_tasks_per_cls=total_tasks_per_cls, **kwargs)
    if name in _ML:
        return MetaWorldGpuVectorEnv(f"{name}-{split}", seed=seed, goal_seed=gs, meta_batch_size=meta_batch_size,
                                     total_tasks_per_cls=total_tasks_per_cls, **kwargs)
    raise ValueError("Invalid ML env name. Must either be a valid Metaworld task name (e.g. 'reach-v3'), 'ML10', 'ML25', or 'ML45'.")


def make_goal_observable(env_name, seed=None, num_envs=1, **kwargs):
    """The single-goal classes `ALL_V3_ENVIRONMENTS_GOAL_OBSERVABLE[name](seed)` ("Meta-World/goal_observable",
    metaworld/env_dict.py:171-212, metaworld/__init__.py:683-693): seed the legacy RNG, build the env, reset once, freeze -- the
    very draws that make goal 0 of MT1(name, seed), so this is the MT1 env restricted to that goal, goal visible."""
    name = env_name[:-len("-goal-observable")] if env_name.endswith("-goal-observable") else env_name
    if name not in T.ALL_V3:
        raise KeyError(env_name)
    return MetaWorldGpuVectorEnv("MT1", name, num_envs=num_envs, seed=seed, goal_seed=42 if seed is None else seed,
                                 total_tasks_per_cls=1, partially_observable=False, **kwargs)


def make_goal_hidden(env_name, seed=None, num_envs=1, **kwargs):
    """`ALL_V3_ENVIRONMENTS_GOAL_HIDDEN[name](seed)` ("Meta-World/goal_hidden", metaworld/__init__.py:671-681): the same single
    goal with the goal slots of the observation zeroed."""
    name = env_name[:-len("-goal-hidden")] if env_name.endswith("-goal-hidden") else env_name
    if name not in T.ALL_V3:
        raise KeyError(env_name)
    return MetaWorldGpuVectorEnv("MT1", name, num_envs=num_envs, seed=seed, goal_seed=42 if seed is None else seed,
                                 total_tasks_per_cls=1, partially_observable=True, **kwargs)


def make_custom_mt_envs(envs_list, seed=None, use_one_hot=False, vector_strategy="sync", autoreset_mode="SameStep", num_envs=None, **kwargs):
    """The "Meta-World/custom-mt-envs" entry point (metaworld/__init__.py:741-781): env idx is `make_mt_envs(envs_list[idx],
    seed=seed + idx, env_id=idx, num_tasks=len(envs_list))`, i.e. MT1 goal tables and a task-selection stream per class."""
    _check_autoreset(autoreset_mode)
    return MetaWorldGpuVectorEnv("custom-mt", envs_list=list(envs_list), num_envs=num_envs, seed=seed or None,
                                 goal_seed=seed or 42, use_one_hot=use_one_hot, **kwargs)


def make_custom_ml_envs(train_envs, test_envs, seed=None, meta_batch_size=20, total_tasks_per_cls=None, split="train",
                        vector_strategy="sync", autoreset_mode="SameStep", num_envs=None, **kwargs):
    """The "Meta-World/custom-ml-envs" entry point (metaworld/__init__.py:783-821) over CustomML (`:370-395`)."""
    _check_autoreset(autoreset_mode)
    if set(train_envs) & set(test_envs):
        raise ValueError("The test tasks cannot contain any of the train tasks.")
    return MetaWorldGpuVectorEnv("custom-ml", envs_list=list(train_envs if split == "train" else test_envs), seed=seed,
                                 goal_seed=42 if seed is None else seed, meta_batch_size=meta_batch_size,
                                 total_tasks_per_cls=total_tasks_per_cls, partially_observable=kwargs.pop("partially_observable", True), **kwargs)


# metaworld/__init__.py:596-604
make_ml_envs_train = partial(make_ml_envs, terminate_on_success=False, task_select="pseudorandom", split="train")
make_ml_envs_test = partial(make_ml_envs, terminate_on_success=True, task_select="pseudorandom", split="test")


def register_mw_envs(namespace="Meta-World-GPU"):
    """Register the vector ids of metaworld/__init__.py:655-823 that sit on the hot path.  No-op (returns False) when
    gymnasium is not installed.  Unlike the reference's lambdas (`:711-718`), `num_envs` is honoured.

    `register_mw_envs("Meta-World")` is the opt-in drop-in: it registers the REFERENCE's own ids ("Meta-World/MT50", ...), so an
    unmodified `gym.make_vec("Meta-World/MT50", seed=..., use_one_hot=True)` resolves to the GPU VectorEnv (call it after, or
    instead of, `import metaworld`, whose import-time registration `:823` it overrides).  The default namespace keeps both
    registrations side by side."""
    try:
        from gymnasium.envs.registration import register
    except Exception:
        return False

    def mt(bench, env_name=None, vector_strategy="sync", autoreset_mode="SameStep", seed=None, use_one_hot=False, num_envs=None, **kw):
        return make_mt_envs(env_name or bench, seed=seed, use_one_hot=use_one_hot, vector_strategy=vector_strategy,
                            autoreset_mode=autoreset_mode, num_envs=num_envs, **kw)

    def ml(bench, split, env_name=None, vector_strategy="sync", autoreset_mode="SameStep", total_tasks_per_cls=None, seed=None,
           meta_batch_size=20, num_envs=None, **kw):
        gen = make_ml_envs_train if split == "train" else make_ml_envs_test
        return gen(env_name or bench, seed=seed, meta_batch_size=meta_batch_size, total_tasks_per_cls=total_tasks_per_cls,
                   vector_strategy=vector_strategy, autoreset_mode=autoreset_mode, **kw)

    # "Meta-World/MT1" is a plain (non-vector) entry point in the reference (`:655-667`); here MT1 is a VectorEnv of num_envs
    # copies of the one task (BASELINE config 2 has no reference spelling), so it is registered as a vector entry point too
    register(id=f"{namespace}/MT1", vector_entry_point=lambda env_name, **kw: mt(env_name, **kw), kwargs={})
    for b in _MT:
        register(id=f"{namespace}/{b}", vector_entry_point=partial(mt, b), kwargs={})
    for split in ("train", "test"):
        register(id=f"{namespace}/ML1-{split}", vector_entry_point=partial(ml, "ML1", split), kwargs={})
        for b in _ML:
            register(id=f"{namespace}/{b}-{split}", vector_entry_point=partial(ml, b, split), kwargs={})
    register(id=f"{namespace}/goal_observable", vector_entry_point=lambda env_name, seed=None, **kw: make_goal_observable(env_name, seed, **kw), kwargs={})
    register(id=f"{namespace}/goal_hidden", vector_entry_point=lambda env_name, seed=None, **kw: make_goal_hidden(env_name, seed, **kw), kwargs={})
    register(id=f"{namespace}/custom-mt-envs", kwargs={},
             vector_entry_point=lambda envs_list, vector_strategy="sync", **kw: make_custom_mt_envs(envs_list, **kw))
    register(id=f"{namespace}/custom-ml-envs", kwargs={},
             vector_entry_point=lambda train_envs, test_envs, vector_strategy="sync", **kw: make_custom_ml_envs(train_envs, test_envs, **kw))
    return True
