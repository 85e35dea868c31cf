"""MetaWorldGpuVectorEnv -- the drop-in boundary.

Presents the Gymnasium `VectorEnv` surface the reference returns from
`gym.make_vec("Meta-World/MT50", ...)` -> `make_mt_envs` (metaworld/__init__.py:460-513):
`reset`, `step` with SAME_STEP auto-reset and dict-of-arrays infos, `get_attr`, `call`, `close`,
`single_observation_space`, ... -- but every sub-environment lives on the GPU and one `step()` is a
single kernel launch through the C ABI of libmwgpu.so (include/mwgpu.h).

Differences that are extensions, not breaks: `num_envs` is honoured (the reference
accepts and ignores it); the env batch can be sharded over ranks (one process per GPU).
"""
from __future__ import annotations

import time

import os

import numpy as np

from . import native, tasks as T

INFO_KEYS = ["near_object", "grasp_success", "grasp_reward", "in_place_reward", "obj_to_target", "unscaled_reward"]


class Box:
    """Minimal stand-in for gymnasium.spaces.Box (used only if gymnasium is not installed)."""

    def __init__(self, low, high, dtype):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)


def _box(low, high, dtype):
    try:
        from gymnasium.spaces import Box as GBox  # type: ignore
        return GBox(np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype), dtype=dtype)
    except Exception:
        return Box(low, high, dtype)


def benchmark_tasks(benchmark, env_name=None):
    """(task names, goal-table key) of a benchmark split: MT1, MT10, MT25, MT50, ML1-train/-test, ML10-train/-test,
    ML25-*, ML45-* (the reference's `metaworld.MT50(seed).train_tasks` etc.)."""
    return T.benchmark_task_names(benchmark, env_name), benchmark


class _RunningMeanStd:
    """gymnasium.wrappers.utils.RunningMeanStd (epsilon 1e-4; Chan's parallel update), one independent instance per sub-env
    held as arrays, each update feeding one sample per selected env (batch_count 1, batch_var 0).  gymnasium is an
    external dependency of the reference that cannot be installed here, so this restatement is unpinned."""

    def __init__(self, n, shape, dtype):
        self.mean, self.var, self.count = np.zeros((n,) + shape, dtype=dtype), np.ones((n,) + shape, dtype=dtype), np.full(n, 1e-4)

    def update(self, x, mask):
        i = np.flatnonzero(mask)
        cnt = self.count[i].reshape((-1,) + (1,) * (self.mean.ndim - 1))
        delta = x[i] - self.mean[i]
        tot = cnt + 1
        m2 = self.var[i] * cnt + np.square(delta) * cnt * 1 / tot
        self.mean[i] = self.mean[i] + delta * 1 / tot
        self.var[i] = m2 / tot
        self.count[i] += 1


def _vector_env_base():
    """gymnasium.vector.VectorEnv when gymnasium is importable -- `gym.make_vec("Meta-World/MT50")` (metaworld/__init__.py:707-722),
    vector wrappers and `isinstance` checks then see a real VectorEnv --, a plain object otherwise (this container, the GPU box);
    the class below defines every attribute of the base itself, so both spellings behave alike."""
    try:
        from gymnasium.vector import VectorEnv  # type: ignore
        return VectorEnv
    except Exception:
        return object


def _autoreset_same_step():
    try:
        from gymnasium.vector import AutoresetMode  # type: ignore
        return AutoresetMode.SAME_STEP
    except Exception:
        return "SameStep"


class MetaWorldGpuVectorEnv(_vector_env_base()):
    # gymnasium.vector.VectorEnv's class attributes (gymnasium 1.1, vector/vector_env.py): make_vec writes `env.unwrapped.spec`,
    # vector wrappers read `metadata["autoreset_mode"]`, `render_mode`, `closed`, `np_random`
    metadata = {"autoreset_mode": _autoreset_same_step(), "render_modes": []}
    spec = None
    render_mode = None
    closed = False
    _np_random = None
    _np_random_seed = None

    def __init__(self, benchmark="MT1", env_name=None, num_envs=None, seed=None, use_one_hot=False,
                 max_episode_steps=None, terminate_on_success=False, precision="fp64", device_id=0,
                 rank=0, world_size=1, goal_seed=42, task_names=None, lib=None, maxcon=None, maxefc=None,
                 partially_observable=None, task_select="random", meta_batch_size=None, total_tasks_per_cls=None,
                 recurrent_info_in_obs=False, normalize_reward_in_recurrent_info=True, reward_function_version="v2",
                 reward_normalization_method=None, reward_alpha=0.001, normalize_observations=False, envs_list=None,
                 lanes_per_block=None, raise_on_status=False, full_forward=False, num_tasks=None):
        """The keyword set of the reference's `_init_each_env` / `make_ml_envs` (metaworld/__init__.py:398-460, :516-618):
        `task_select` "random" = RandomTaskSelectWrapper, "pseudorandom" = PseudoRandomTaskSelectWrapper;
        `meta_batch_size` / `total_tasks_per_cls` = the ML split of each class's goals over sub-envs (`tasks[i::k]`);
        `recurrent_info_in_obs` = RNNBasedMetaRLWrapper; `reward_normalization_method` / `normalize_observations` = the
        normalisation wrappers.  Only the v2 reward functions have device code.
        `precision`: "fp64" (default) = the reference's own arithmetic (float64 state, solver, observations): the mode whose GPU
        tests hold obs / reward <= 1e-5 against the reference traces; "fp32" is the opt-in throughput mode (success flags exact,
        obs / reward within the single-precision contact-geometry floor).
        `full_forward`: True = every step ends with the complete mj_forward (callers that read ncon / nefc / contact forces
        of the final state through `ctx.read*`); the default stops after the kinematics where no reward reads them."""
        if reward_function_version not in ("v1", "v2"):
            raise ValueError(f"reward_function_version must be 'v1' or 'v2', got {reward_function_version!r}")
        v1 = reward_function_version == "v1"
        if task_select not in ("random", "pseudorandom"):
            raise ValueError(f"task_select must be 'random' or 'pseudorandom', got {task_select!r}")
        if reward_normalization_method not in (None, "gymnasium", "exponential"):
            raise ValueError(f"unknown reward_normalization_method {reward_normalization_method!r}")
        custom = benchmark in ("custom-mt", "custom-ml")          # arbitrary class lists (metaworld/__init__.py:741-821)
        if custom:
            names, goal_key = list(envs_list), benchmark
            if len(set(names)) != len(names) or any(n not in T.ALL_V3 for n in names):
                raise ValueError(f"envs_list must hold distinct v3 task names, got {names}")
        else:
            names, goal_key = benchmark_tasks(benchmark, env_name)
        if task_names is not None:          # restrict a benchmark to the tasks that have device code (tests)
            names = [n for n in names if n in task_names]
        missing = [n for n in names if n not in T.TASK_DEFS]
        if missing:
            raise NotImplementedError(f"no device-side task code yet for: {missing}")
        self.task_list = names
        ntask = len(names)
        # OneHotWrapper(env, env_id, num_tasks) (metaworld/wrappers.py:14-32; make_mt_envs passes `num_tasks or <benchmark size>`,
        # metaworld/__init__.py:434-436, :501): the one-hot may be WIDER than the benchmark; a narrower one fails like the
        # reference's `self.one_hot[task_idx] = 1.0`
        width = ntask if num_tasks is None else int(num_tasks)
        if use_one_hot and width < ntask:
            raise IndexError(f"index {width} is out of bounds for axis 0 with size {width}")
        if meta_batch_size is not None:
            # _make_ml_envs_inner (metaworld/__init__.py:527-531): meta_batch_size sub-envs, meta_batch_size/ntask per class
            assert meta_batch_size % ntask == 0, "meta_batch_size must be divisible by envs_per_task"
            if num_envs not in (None, meta_batch_size):
                raise ValueError("num_envs and meta_batch_size disagree")
            num_envs = meta_batch_size
        self.num_envs = int(num_envs) if num_envs else ntask
        if self.num_envs < ntask:
            raise ValueError("num_envs must be >= number of tasks")
        self.rank, self.world_size = rank, world_size
        self.use_one_hot = bool(use_one_hot)
        # ML benchmarks hide the goal (metaworld/__init__.py `_ML_OVERRIDE`), MT ones show it; the reference's own policy test
        # makes it visible for ML too (tests/metaworld/test_evaluation.py:70-82) -> overridable
        self.partially_observable = benchmark.startswith("ML") if partially_observable is None else bool(partially_observable)
        self.seed_value = seed
        if v1:
            # the v1 reward functions (csrc/mw_tasks_v1.hpp) are selected per context (mw_config.reward_version); tasks whose
            # v1 branch has no device restatement yet are refused here rather than silently evaluated with v2
            missing_v1 = [n for n in names if n not in T.V1_TASKS]
            if missing_v1:
                raise NotImplementedError(f"reward_function_version='v1' has no device code yet for: {missing_v1}")
        self.reward_function_version = reward_function_version
        self._lib = lib or native.load()
        self.ctx = native.Context(self._lib, precision=1 if precision in ("fp64", 1) else 0, device_id=device_id,
                                  rank=rank, world_size=world_size, max_episode_steps=max_episode_steps or 500,
                                  terminate_on_success=terminate_on_success, one_hot=use_one_hot, num_tasks=width,
                                  full_forward=full_forward, reward_version=1 if v1 else 2)
        # env -> task (task-major contiguous blocks, like the reference's enumerate order)
        per, rem = divmod(self.num_envs, ntask)
        env_task_names = []
        for i, name in enumerate(names):
            env_task_names += [name] * (per + (1 if i < rem else 0))
        self.env_task_names = env_task_names
        # lanes per workgroup of every model group: None = the runtime's own choice, or an explicit {model name: lanes}
        self.lanes_per_block = dict(lanes_per_block or {})
        # models and tasks
        model_index, roles_of, reloc_of = {}, {}, {}
        self._task_index = {}
        self.goal_tables = {}
        for oh, name in enumerate(names):
            mkey = T.model_key(name)          # (scene, relocated bodies): tasks share a group only if both agree
            mname = mkey[0]
            if mkey not in model_index:
                pk, roles, reloc = T.packed_model(mname, maxcon=maxcon, maxefc=maxefc, v1=v1, reloc_bodies=mkey[1],
                                                  tolerance=None if precision in ("fp64", 1) else 1e-6)
                if mname in self.lanes_per_block:
                    pk["options"]["lanes_per_block"] = self.lanes_per_block[mname]
                model_index[mkey] = self.ctx.add_model(pk)
                roles_of[mkey], reloc_of[mkey] = roles, reloc
            if benchmark == "custom-mt":          # env idx is built as MT1(name, seed + idx)
                goals = T.custom_goal_tables((name,), goal_seed + oh)[name]
            elif benchmark == "custom-ml":        # one `_make_tasks` stream over the class list
                goals = T.custom_goal_tables(tuple(names), goal_seed)[name]
            else:
                goals = T.goal_table(goal_key, name, goal_seed)
            if total_tasks_per_cls is not None:          # only the first total_tasks_per_cls goals of a class are ever selected
                goals = goals[:max(1, int(total_tasks_per_cls))]
            self.goal_tables[name] = goals
            ts = T.task_struct(name, model_index[mkey], roles_of[mkey], reloc_of[mkey], onehot_id=oh, v1=v1,
                               partially_observable=self.partially_observable)
            self._task_index[name] = self.ctx.add_task(ts, goals)
        self.model_index = model_index
        self.ctx.set_envs([self._task_index[n] for n in env_task_names])
        self.ctx.finalize()
        # the goals each sub-env may be set to (indices into its task's table).  ML: the sub-envs of one class share its goals
        # as `tasks[i::k]` (metaworld/__init__.py:533-543); otherwise every env sees the whole table.
        self._goal_lists = []
        for i, name in enumerate(names):
            k = per + (1 if i < rem else 0)
            ng = len(self.goal_tables[name]) if total_tasks_per_cls is None else min(total_tasks_per_cls, len(self.goal_tables[name]))
            for j in range(k):
                if meta_batch_size is None:
                    self._goal_lists.append(np.arange(ng))
                else:
                    sub = np.arange(ng)[j::k]
                    assert len(sub) == ng // k, f"Invalid division of subtasks, expected {ng // k} got {len(sub)}"
                    self._goal_lists.append(sub)
        # RandomTaskSelectWrapper / PseudoRandomTaskSelectWrapper: every sub-env is seeded with the SAME seed
        # (metaworld/__init__.py:430-431), so the n-th draw of any env is the n-th value of one PCG64 stream.
        # (the custom MT entry point seeds env idx with seed + idx, `:761`)
        self._env_seed = [None if seed is None else (seed + names.index(n) if benchmark == "custom-mt" and seed else seed) for n in env_task_names]
        self.task_select = task_select
        self._stream, self._shuffles = {}, {}
        # envs grouped by (goal-list length, seed): one cached selection stream serves a whole group (_random_goals)
        self._groups, self._group_of = [], []
        for e in range(self.num_envs):
            k = (len(self._goal_lists[e]), self._env_seed[e])
            if k not in self._groups:
                self._groups.append(k)
            self._group_of.append(self._groups.index(k))
        self._lists_are_ranges = all(np.array_equal(l, np.arange(len(l))) for l in self._goal_lists)
        self._reset_count = np.zeros(self.num_envs, dtype=np.int64)    # random: draws made; pseudorandom: shuffles made
        self._task_idx = np.full(self.num_envs, -1, dtype=np.int64)     # pseudorandom: current_task_idx (wrappers.py:180)
        self._cur_goal = np.full(self.num_envs, -1, dtype=np.int32)     # goal of the running episode (-1 = no task set yet)
        self._next_goal = np.zeros(self.num_envs, dtype=np.int32)       # goal the next (auto-)reset will use
        self.sample_tasks_on_reset = task_select == "random"            # wrappers.py:96 / :154
        self.terminate_on_success = bool(terminate_on_success)
        self.max_episode_steps = int(max_episode_steps or 500)
        D = self.ctx.D
        lo = np.concatenate([[-0.525, 0.348, -0.0525, -1.0], np.full(14, -np.inf)] * 2 + [np.zeros(3)])
        hi = np.concatenate([[0.525, 1.025, 0.7, 1.0], np.full(14, np.inf)] * 2 + [np.zeros(3)])
        if use_one_hot:
            lo, hi = np.concatenate([lo, np.zeros(width)]), np.concatenate([hi, np.ones(width)])
        self.obs_dtype = np.float32 if use_one_hot else np.float64
        self._raw_dtype = self.obs_dtype
        self.recurrent_info_in_obs = bool(recurrent_info_in_obs)
        self._normalize_reward_in_recurrent_info = bool(normalize_reward_in_recurrent_info)
        if self.recurrent_info_in_obs:      # RNNBasedMetaRLWrapper (wrappers.py:50-88): obs ++ action ++ reward ++ done, Box default dtype f32
            lo, hi = np.full(len(lo) + 6, -np.inf), np.full(len(hi) + 6, np.inf)
            self.obs_dtype = np.float32
        self.reward_normalization_method, self.reward_alpha = reward_normalization_method, float(reward_alpha)
        self._rew_mean, self._rew_var = np.zeros(self.num_envs), np.ones(self.num_envs)      # NormalizeRewardsExponential (wrappers.py:233-237)
        self._disc_ret = np.zeros(self.num_envs)                                               # gymnasium NormalizeReward.discounted_reward
        self._ret_rms = _RunningMeanStd(self.num_envs, (), np.float64)
        self.normalize_observations = bool(normalize_observations)
        if self.normalize_observations:     # gymnasium NormalizeObservation: statistics in the wrapped space's dtype, f32 output
            self._obs_rms = _RunningMeanStd(self.num_envs, (len(lo),), self.obs_dtype)
            lo, hi = np.full(len(lo), -np.inf), np.full(len(hi), np.inf)
            self.obs_dtype = np.float32
        self._ep_ret = np.zeros(self.num_envs)          # RecordEpisodeStatistics sits outside the normalisers: it sums what they return
        self.single_observation_space = _box(lo, hi, self.obs_dtype)
        self.single_action_space = _box(-np.ones(4), np.ones(4), np.float32)
        self.observation_space = _box(np.tile(lo, (self.num_envs, 1)), np.tile(hi, (self.num_envs, 1)), self.obs_dtype)
        self.action_space = _box(-np.ones((self.num_envs, 4)), np.ones((self.num_envs, 4)), np.float32)
        self._episode_start = np.full(self.num_envs, time.perf_counter())
        self.closed = False
        self.raise_on_status = raise_on_status
        assert D == len(lo) - (6 if self.recurrent_info_in_obs else 0)

    # ---- RandomTaskSelectWrapper stream (wrappers.py:98-100: self.np_random.choice(len(tasks))) ----
    def _draw(self, n_goals, k, seed):
        """k-th draw of Generator(PCG64(seed)).choice(n_goals): every sub-env owns an identically seeded
        generator and only ever calls choice(len(tasks)), so one cached stream per distinct (n, seed) serves all."""
        if (n_goals, seed) not in self._stream:
            self._stream[n_goals, seed] = (np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))), [])
        gen, s = self._stream[n_goals, seed]
        while len(s) <= k:
            s.append(int(gen.choice(n_goals)))
        return s[k]

    def _shuffle_perm(self, n, k, seed):
        """Index permutation of the k-th `np_random.shuffle(tasks)` of an n-task list (wrappers.py:158-161): Generator.shuffle
        draws do not depend on the list's contents, so one cached stream per distinct n serves all identically seeded sub-envs."""
        if (n, seed) not in self._shuffles:
            self._shuffles[n, seed] = (np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))), [])
        gen, s = self._shuffles[n, seed]
        while len(s) <= k:
            order = list(range(n))
            gen.shuffle(order)          # a Python list, as in the reference (list[Task])
            s.append(np.array(order, dtype=np.int64))
        return s[k]

    def _select(self, e, commit):
        """Goal the next task selection of env e yields (`_set_random_task` wrappers.py:98-100 / `_set_pseudo_random_task`
        :157-162); with commit=False nothing advances (used to tell the kernel which goal a SAME_STEP auto-reset takes)."""
        lst = self._goal_lists[e]
        if self.task_select == "random":
            g = lst[self._draw(len(lst), int(self._reset_count[e]), self._env_seed[e])]
            if commit:
                self._reset_count[e] += 1
            return int(g)
        idx = (int(self._task_idx[e]) + 1) % len(lst)
        if idx == 0:
            lst = lst[self._shuffle_perm(len(lst), int(self._reset_count[e]), self._env_seed[e])]
        if commit:
            if idx == 0:
                self._goal_lists[e] = lst
                self._reset_count[e] += 1
            self._task_idx[e] = idx
        return int(lst[idx])

    def _random_goals(self, idx, ahead=0):
        """`_select` of the "random" selection for MANY envs at once, without committing: the goal of draw number
        `_reset_count[e] + ahead` of every env in idx.  One numpy gather per (list length, seed) group -- a synchronised reset of
        4096 sub-envs is not 4096 Python iterations."""
        idx = np.asarray(idx, dtype=np.int64)
        out = np.empty(len(idx), dtype=np.int64)
        if not len(idx):
            return out
        key = np.array([self._group_of[e] for e in idx]) if len(self._groups) > 1 else np.zeros(len(idx), dtype=np.int64)
        for g, (n_goals, seed) in enumerate(self._groups):
            sel = np.flatnonzero(key == g)
            if not len(sel):
                continue
            k = self._reset_count[idx[sel]] + ahead
            self._draw(n_goals, int(k.max()), seed)                      # extends the cached stream
            draws = np.asarray(self._stream[n_goals, seed][1], dtype=np.int64)[k]
            if self._lists_are_ranges:
                out[sel] = draws
            else:
                out[sel] = [self._goal_lists[e][d] for e, d in zip(idx[sel], draws)]
        return out

    def _begin_episodes(self, mask, force=False):
        """what `reset()` of the task-select wrapper does to the envs in mask (a new task iff sample_tasks_on_reset, or always
        for `sample_tasks`), then the look-ahead for their next auto-reset"""
        sample = self.sample_tasks_on_reset or force
        idx = np.flatnonzero(mask)
        if self.task_select == "random":
            if sample:
                self._cur_goal[idx] = self._random_goals(idx)
                self._reset_count[idx] += 1
        else:
            for e in idx:
                if sample:
                    self._cur_goal[e] = self._select(e, commit=True)
        assert (self._cur_goal[idx] >= 0).all(), "no task set: call('sample_tasks') first (sawyer_xyz_env.py:699-701)"
        self._look_ahead(mask)

    def _look_ahead(self, mask):
        idx = np.flatnonzero(mask)
        if not self.sample_tasks_on_reset:
            self._next_goal[idx] = self._cur_goal[idx]
        elif self.task_select == "random":
            self._next_goal[idx] = self._random_goals(idx)
        else:
            for e in idx:
                self._next_goal[e] = self._select(e, commit=False)

    # ---- the resident loop (mw_step_resident): K steps on pre-uploaded actions, outputs left in HBM ----
    _MAX_SCHEDULE_ROWS = 4096          # default cap on the goal-schedule rows of one call (16 KB per row at 4096 envs)
    def step_resident(self, nsteps, gather=False, schedule_rows=None, steps_per_launch=None):
        """`nsteps` steps of the whole batch on the actions uploaded with `ctx.upload_actions`, no host round trip in between;
        returns the HIP-event kernel time in ms.  The auto-resets that happen inside draw a NEW task per reset like
        `RandomTaskSelectWrapper.reset` (metaworld/wrappers.py:116-119): the next `schedule_rows` selections of every sub-env are
        taken from its task-selection stream and handed to the kernel (mw_set_goal_schedule), the consumed counts are read back
        and the streams advanced by them, so that a following `step()` / `reset()` continues exactly where the reference's
        wrappers would be.  (Row K-1 repeats if an env resets more than K times: only with episodes of a few steps.)
        `steps_per_launch` = k runs k consecutive steps of every environment per kernel launch (mw_step_resident_fused): the same
        final state and outputs bit for bit, one batch-wide synchronisation per launch instead of per step -- for rollouts nobody
        observes between steps."""
        if steps_per_launch and gather:
            raise ValueError("the per-step cross-rank gather needs one launch per step")
        if (self._cur_goal < 0).any():
            raise RuntimeError("step_resident() called before reset(): no task has been set for some sub-envs")
        sched = None
        if self.sample_tasks_on_reset and self.task_select == "random":
            # rows needed = the most auto-resets one env can make in nsteps: one per max_episode_steps without early termination,
            # up to one per step with terminate_on_success (an env can succeed in its first step)
            worst = nsteps + 1 if self.terminate_on_success else nsteps // max(1, min(self.max_episode_steps, 500)) + 3   # (+ slack for instability truncations)
            # K follows the bound: a K x N int32 table is cheap, and a default smaller than `worst` turned legitimate
            # configurations (short episodes, terminate_on_success) into a failure after the kernel had already run
            if schedule_rows is not None and int(schedule_rows) < 1:
                raise ValueError("schedule_rows must be >= 1")
            K = int(schedule_rows) if schedule_rows is not None else max(2, min(self._MAX_SCHEDULE_ROWS, worst))
            every = np.arange(self.num_envs)
            sched = np.stack([self._random_goals(every, ahead=k) for k in range(K)]).astype(np.int32)
            self.ctx.set_goal_schedule(sched)
        if steps_per_launch:
            ms = self.ctx.step_resident_fused(nsteps, steps_per_launch)
        else:
            ms = self.ctx.step_resident_gather(nsteps) if gather else self.ctx.step_resident(nsteps)
        if sched is not None:
            used = self.ctx.goal_schedule_pos().astype(np.int64)
            self.ctx.set_goal_schedule(None)
            # the host bookkeeping follows the device for the rows that were really consumed -- also when the table overflowed, so
            # that the env object stays usable (its streams are then ahead of the reference's by the surplus resets: reported below)
            took = np.minimum(used, len(sched))
            hit = np.flatnonzero(took > 0)
            self._cur_goal[hit] = sched[took[hit] - 1, hit]
            self._reset_count += took
            self._look_ahead(np.ones(self.num_envs, dtype=bool))
            if (used > len(sched)).any():
                # the kernel repeated the last row for the surplus resets: those draws never came from the selection streams, so the
                # trajectory after them is not the reference wrapper's
                raise RuntimeError(f"step_resident: an env auto-reset {int(used.max())} times but only {len(sched)} schedule rows were drawn; "
                                   "pass schedule_rows >= the number of resets per env (short episodes with terminate_on_success)")
        return ms

    # ---- the observation / reward wrappers between the env and the vectoriser (metaworld/__init__.py:438-449) ----
    def _wrap_reset_obs(self, obs, mask):
        """obs of freshly reset envs through RNNBasedMetaRLWrapper.reset (wrappers.py:82-88) and NormalizeObservation"""
        if self.recurrent_info_in_obs:
            obs = np.concatenate([obs, np.zeros((len(obs), 6))], axis=1)
        if self.normalize_observations:
            obs = self._normalize_obs(obs, mask)
        return obs

    def _normalize_obs(self, obs, mask):
        obs = obs.astype(self._obs_rms.mean.dtype)
        self._obs_rms.update(obs, mask)
        out = obs.astype(np.float32)
        idx = np.flatnonzero(mask)
        out[idx] = np.float32((obs[idx] - self._obs_rms.mean[idx]) / np.sqrt(self._obs_rms.var[idx] + 1e-8))
        return out

    def _normalize_reward(self, rew, term):
        if self.reward_normalization_method == "exponential":
            # NormalizeRewardsExponential.step (wrappers.py:251-258) updates the estimate twice per step: once itself, once
            # inside _apply_normalize_reward
            a = self.reward_alpha
            for _ in range(2):
                self._rew_mean = (1 - a) * self._rew_mean + a * rew
                self._rew_var = (1 - a) * self._rew_var + a * np.square(rew - self._rew_mean)
            return rew / (np.sqrt(self._rew_var) + 1e-8)
        if self.reward_normalization_method == "gymnasium":
            self._disc_ret = self._disc_ret * 0.99 * (1 - term) + rew
            self._ret_rms.update(self._disc_ret, np.ones(self.num_envs, dtype=bool))
            return rew / np.sqrt(self._ret_rms.var + 1e-8)
        return rew

    # ---- VectorEnv API ----
    def reset(self, *, seed=None, options=None):
        """`seed` is accepted and has no effect, as in the reference: `SawyerXYZEnv.reset` drops it (`super().reset()` is called
        without it, metaworld/sawyer_xyz_env.py:664-678), so neither the task-selection stream (seeded once by `env.seed(seed)`
        at construction, metaworld/__init__.py:428-429) nor the goal tables change."""
        mask = np.ones(self.num_envs, dtype=bool)
        self._begin_episodes(mask)
        obs = self._wrap_reset_obs(self.ctx.reset(self._cur_goal).astype(self._raw_dtype), mask)
        self._episode_start[:] = time.perf_counter()
        self._ep_ret[:] = 0
        return obs.astype(self.obs_dtype, copy=True), {}

    def step(self, actions):
        a = np.asarray(actions)
        # SawyerXYZEnv.step: `assert len(action) == 4` (metaworld/sawyer_xyz_env.py:591); the vectoriser iterates over the batch
        assert a.shape == (self.num_envs, 4), f"Actions should be size 4, got {a.shape[1:] if a.ndim > 1 else a.shape}"
        if (self._cur_goal < 0).any():          # SawyerXYZEnv._get_state_rand_vec without a task (`:699-701`) / step before reset
            raise RuntimeError("step() called before reset(): no task has been set for some sub-envs")
        obs, rew, term, trunc, succ, info = self.ctx.step(a, self._next_goal)
        if self.raise_on_status:
            self.check_status()
        term_b, trunc_b = term.astype(bool), trunc.astype(bool)
        done = term_b | trunc_b
        obs = obs.astype(self._raw_dtype)
        wrapped = self.recurrent_info_in_obs or self.normalize_observations
        final = self.ctx.final_obs
        if wrapped:          # the terminal obs goes through the wrappers' step(), the reset obs through their reset()
            step_obs = np.where(done[:, None], final, obs).astype(self._raw_dtype)
            if self.recurrent_info_in_obs:
                a = np.asarray(actions, dtype=np.float64).reshape(self.num_envs, 4)
                ro = rew / 10.0 if self._normalize_reward_in_recurrent_info else rew
                step_obs = np.concatenate([step_obs, a, ro[:, None], done[:, None].astype(np.float64)], axis=1)
            if self.normalize_observations:
                step_obs = self._normalize_obs(step_obs, np.ones(self.num_envs, dtype=bool))
            final = step_obs
            obs = step_obs.copy()
            if done.any():
                obs[done] = self._wrap_reset_obs(self.ctx.obs.astype(self._raw_dtype), done)[done]
        rew = self._normalize_reward(rew.copy(), term_b)
        self._ep_ret += rew
        infos = {"success": succ.astype(np.float64), "_success": np.ones(self.num_envs, dtype=bool)}
        for k, key in enumerate(INFO_KEYS):
            infos[key] = info[:, k].astype(np.float64)
            infos["_" + key] = infos["_success"]
        if done.any():
            now = time.perf_counter()
            fi = {k: np.where(done, v, 0) for k, v in infos.items() if not k.startswith("_")}
            for k in list(fi):
                fi["_" + k] = done.copy()
            ep_ret = self.ctx.ep_ret if self.reward_normalization_method is None else self._ep_ret
            fi["episode"] = {"r": np.where(done, ep_ret, 0.0), "l": np.where(done, self.ctx.ep_len, 0),
                             "t": np.where(done, np.round(now - self._episode_start, 6), 0.0),
                             "_r": done.copy(), "_l": done.copy(), "_t": done.copy()}
            fi["_episode"] = done.copy()
            infos["final_info"], infos["_final_info"] = fi, done.copy()
            fo = np.empty(self.num_envs, dtype=object)
            for e in np.flatnonzero(done):
                fo[e] = final[e].astype(self.obs_dtype)
            infos["final_obs"], infos["_final_obs"] = fo, done.copy()
            self._episode_start[done] = now
            self._ep_ret[done] = 0
            self._begin_episodes(done)          # the auto-reset the kernel just did used _next_goal == this selection
        # (obs is already a fresh array of this step -- the astype / concatenate above made it -- unless dtypes matched and it is still a
        #  view of the context's pinned output buffer, which the next step overwrites)
        if obs.dtype != self.obs_dtype or obs.base is not None or obs is self.ctx.obs:
            obs = obs.astype(self.obs_dtype, copy=True)
        return obs, rew, term_b, trunc_b, infos

    def get_attr(self, name):
        if name == "task_name":
            return tuple(T.TASK_CONST[n]["cls"] for n in self.env_task_names)
        if name == "terminate_on_success":
            return (self.terminate_on_success,) * self.num_envs
        if name == "_partially_observable":
            return (self.partially_observable,) * self.num_envs
        if name == "_last_rand_vec":
            return tuple(self.goal_tables[n][g] if g >= 0 else None for n, g in zip(self.env_task_names, self._cur_goal))
        if name == "tasks":
            return tuple(self.goal_tables[n][l] for n, l in zip(self.env_task_names, self._goal_lists))
        if name == "sample_tasks_on_reset":
            return (self.sample_tasks_on_reset,) * self.num_envs
        if name == "current_task_idx" and self.task_select == "pseudorandom":
            return tuple(int(i) for i in self._task_idx)
        raise AttributeError(name)

    def set_attr(self, name, values):
        """gymnasium VectorEnv.set_attr for the attributes of the reference's wrapper stack that are state here"""
        vals = list(values) if isinstance(values, (list, tuple, np.ndarray)) else [values] * self.num_envs
        if name == "terminate_on_success":
            assert len(set(bool(v) for v in vals)) == 1, "terminate_on_success is one flag per vector env"
            self.call("toggle_terminate_on_success", bool(vals[0]))
        elif name == "sample_tasks_on_reset":
            self.call("toggle_sample_tasks_on_reset", bool(vals[0]))
        else:
            raise NotImplementedError(name)

    def call(self, name, *args, **kwargs):
        """gymnasium VectorEnv.call for the methods the reference's wrappers expose (metaworld/wrappers.py:107-142, :222-223;
        used by metaworld/evaluation.py:53-125)."""
        if name == "toggle_sample_tasks_on_reset":
            self.sample_tasks_on_reset = bool(args[0])
            self._look_ahead(np.ones(self.num_envs, dtype=bool))
            return (None,) * self.num_envs
        if name == "toggle_terminate_on_success":
            self.terminate_on_success = bool(args[0])
            self.ctx.set_terminate_on_success(self.terminate_on_success)
            return (None,) * self.num_envs
        if name == "sample_tasks":          # (Pseudo)RandomTaskSelectWrapper.sample_tasks: a new task for every env, then reset it
            mask = np.ones(self.num_envs, dtype=bool)
            self._begin_episodes(mask, force=True)
            obs = self._wrap_reset_obs(self.ctx.reset(self._cur_goal).astype(self._raw_dtype), mask).astype(self.obs_dtype, copy=True)
            self._episode_start[:] = time.perf_counter()
            self._ep_ret[:] = 0
            return tuple((obs[e], {}) for e in range(self.num_envs))
        if name == "get_checkpoint":
            return self._get_checkpoint()
        if name == "load_checkpoint":
            self._load_checkpoint(args[0])
            return (None,) * self.num_envs
        return self.get_attr(name)

    # ---- checkpoints in the reference's shape (CheckpointWrapper, metaworld/wrappers.py:275-301) ----
    def _env_id(self, e):
        """`f"{env_cls}_{env_id}"` (metaworld/__init__.py:455): the class as the reference prints it, then the sub-env's index"""
        name = self.env_task_names[e]
        cls = T.TASK_CONST[name]["cls"]
        # the reference's module names do not all follow the task name (metaworld/envs/__init__.py)
        module = {"assembly-v3": "sawyer_assembly_peg_v3", "disassemble-v3": "sawyer_disassemble_peg_v3", "door-open-v3": "sawyer_door_v3",
                  "peg-insert-side-v3": "sawyer_peg_insertion_side_v3", "sweep-into-v3": "sawyer_sweep_into_goal_v3"}.get(
                      name, f"sawyer_{name[:-3].replace('-', '_')}_v3")
        # (the suffix is the sub-env's index as in the MT / ML paths of the reference; its MT1 path prints "_None": a deliberate
        #  deviation of this id, which only has to round-trip through this package's own checkpoints)
        return f"<class 'metaworld.envs.{module}.{cls}'>_{e}"

    def _selection_rng_states(self):
        """bit-generator state of every sub-env's task-selection stream after the draws it has made (RandomTaskSelectWrapper keeps
        `self.np_random.bit_generator.state`, wrappers.py:128), replayed from the seeds.  One generator per distinct (list length, seed)
        is advanced ONCE through the sorted draw counts of the envs that share it and snapshotted as it passes each of them (ADVICE r5:
        the per-env replay with an unbounded cache cost ~4e7 Python-level draws for one checkpoint of a long run); nothing is cached."""
        groups = {}
        for e in range(self.num_envs):
            groups.setdefault((len(self._goal_lists[e]), self._env_seed[e]), []).append(e)
        out = [None] * self.num_envs
        for (n, seed), envs in groups.items():
            gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
            done = 0
            for e in sorted(envs, key=lambda i: int(self._reset_count[i])):
                for _ in range(int(self._reset_count[e]) - done):
                    gen.choice(n)
                done = int(self._reset_count[e])
                out[e] = gen.bit_generator.state
        return out

    def _get_checkpoint(self):
        """`envs.call("get_checkpoint")`: one `(env_id, ckpt)` pair per sub-env like the reference's CheckpointWrapper; `ckpt` has the
        keys of (Pseudo)RandomTaskSelectWrapper.get_checkpoint (wrappers.py:125-131, :187-193) -- tasks (env_name + base64 of the
        pickled task data), rng_state / current_task_idx, sample_tasks_on_reset, env_rng_state (empty: the batched env has no
        per-env gymnasium space generators) -- plus, under "mwgpu", what the reference does not checkpoint: the simulation state."""
        import base64
        import pickle
        rng_states = self._selection_rng_states() if self.task_select == "random" else None
        states = self.ctx.get_state()
        out = []
        for e in range(self.num_envs):
            name = self.env_task_names[e]
            tasks = [{"env_name": name, "data": base64.b64encode(pickle.dumps(
                {"rand_vec": np.asarray(self.goal_tables[name][g]), "partially_observable": self.partially_observable})).decode("ascii")}
                for g in self._goal_lists[e]]
            ck = {"tasks": tasks, "sample_tasks_on_reset": self.sample_tasks_on_reset, "env_rng_state": {}}
            if self.task_select == "random":
                ck["rng_state"] = rng_states[e]
            else:
                ck["current_task_idx"] = int(self._task_idx[e])
            ck["mwgpu"] = dict(reset_count=int(self._reset_count[e]), cur_goal=int(self._cur_goal[e]), goal_list=np.asarray(self._goal_lists[e]).copy(),
                               seed=self._env_seed[e], task_select=self.task_select, state=states[e],
                               normalizers=dict(rew_mean=float(self._rew_mean[e]), rew_var=float(self._rew_var[e]), disc_ret=float(self._disc_ret[e]),
                                                ep_ret=float(self._ep_ret[e]),
                                                ret_rms=(self._ret_rms.mean[e].copy(), self._ret_rms.var[e].copy(), self._ret_rms.count[e].copy()),
                                                obs_rms=(self._obs_rms.mean[e].copy(), self._obs_rms.var[e].copy(), self._obs_rms.count[e].copy())
                                                if self.normalize_observations else None))
            out.append((self._env_id(e), ck))
        return tuple(out)

    def _load_checkpoint(self, ckpts):
        """`envs.call("load_checkpoint", ckpts)`: every sub-env looks its own id up in the list and raises the reference's ValueError
        when it is missing (wrappers.py:290-301)"""
        if isinstance(ckpts, tuple) and len(ckpts) == 2 and isinstance(ckpts[0], str):
            ckpts = [ckpts]
        by_id = {}
        for env_id, ck in ckpts:
            by_id.setdefault(env_id, ck)
        rows = []
        for e in range(self.num_envs):
            eid = self._env_id(e)
            if eid not in by_id:
                raise ValueError(f"Could not load checkpoint, no checkpoint found with id {eid}. Checkpoint IDs: ", [i for i, _ in ckpts])
            ck = by_id[eid]
            for key in ("tasks", "sample_tasks_on_reset", "env_rng_state"):
                assert key in ck
            assert ("rng_state" if self.task_select == "random" else "current_task_idx") in ck
            x = ck.get("mwgpu")
            if x is None:
                raise ValueError("checkpoint without the 'mwgpu' section: written by the reference's wrappers, which do not store the simulation state")
            assert x["task_select"] == self.task_select and x["seed"] == self._env_seed[e], "checkpoint of another seed / selection rule"
            assert [t["env_name"] for t in ck["tasks"]] == [self.env_task_names[e]] * len(x["goal_list"]), "checkpoint of another benchmark"
            self._reset_count[e], self._cur_goal[e] = x["reset_count"], x["cur_goal"]
            self._goal_lists[e] = np.asarray(x["goal_list"]).copy()
            if self.task_select != "random":
                self._task_idx[e] = ck["current_task_idx"]
            nz = x["normalizers"]
            self._rew_mean[e], self._rew_var[e], self._disc_ret[e], self._ep_ret[e] = nz["rew_mean"], nz["rew_var"], nz["disc_ret"], nz["ep_ret"]
            self._ret_rms.mean[e], self._ret_rms.var[e], self._ret_rms.count[e] = nz["ret_rms"]
            if self.normalize_observations and nz["obs_rms"] is not None:
                self._obs_rms.mean[e], self._obs_rms.var[e], self._obs_rms.count[e] = nz["obs_rms"]
            rows.append(np.asarray(x["state"], dtype=np.float64))
        self.sample_tasks_on_reset = bool(by_id[self._env_id(0)]["sample_tasks_on_reset"])
        self._lists_are_ranges = all(np.array_equal(l, np.arange(len(l))) for l in self._goal_lists)
        self.ctx.set_state(rows)
        self._look_ahead(np.ones(self.num_envs, dtype=bool))

    # ---- run-time status (no reference counterpart; SURVEY.md 5 "failure detection") ----
    def status(self, clear=False):
        """dict(flags, row_overflow_steps, contact_overflow_steps, unstable_steps, diverged_steps, solver_stalls)
        accumulated since the last clear: flag 1 / 2 = the constraint-row / contact capacity of a scene
        (metaworld_amd/data/model_caps.json) was exceeded and rows / contacts were DROPPED; 4 = a non-finite state was caught
        and the env reset (the intent of sawyer_xyz_env.py:603-619); 8 = the step kernel's redundancy canary fired (the
        threads sharing one env disagreed on a value they all compute: a code-generation / hardware fault, never expected)."""
        return self.ctx.status(clear)

    def check_status(self):
        st = self.ctx.status(clear=True)
        if st["flags"] & 3:
            raise RuntimeError(f"contact / constraint-row capacity exceeded, results differ from the reference: {st} "
                               "(raise maxcon / maxefc)")
        if st["flags"] & 8:
            raise RuntimeError(f"the step kernel's redundancy canary fired (threads sharing one env disagreed): {st}")
        if st["flags"] & 16:
            raise RuntimeError(f"a column-store / scratchpad index was out of range (MW_BOUNDS build of the library): {st}")
        if st["flags"] & 4:
            import warnings
            warnings.warn(f"non-finite simulation state caught; the affected envs were truncated and reset: {st}")
        return st

    def bookkeeping(self):
        """[N, 5] float64 record (done, success, task_id, episode_return, episode_length) of the last step:
        what the cross-rank gather exchanges."""
        c = self.ctx
        tid = np.array([T.TASK_CONST[n]["id"] for n in self.env_task_names], dtype=np.float64)
        done = (c.terminated | c.truncated).astype(np.float64)
        return np.stack([done, c.success.astype(np.float64), tid, c.ep_ret * done, c.ep_len * done], axis=1)

    # ---- the rest of gymnasium.vector.VectorEnv's surface ----
    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        """gymnasium's lazily created generator (seeded by reset(seed=...) there; the reference's reset drops the seed, so this
        one is seeded from the construction seed and never drives the task selection: `_draw` / `_shuffle_perm` do)"""
        if self._np_random is None:
            self._np_random_seed = self.seed_value
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(self.seed_value)))
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random, self._np_random_seed = value, -1

    @property
    def np_random_seed(self):
        if self._np_random is None:
            self.np_random  # noqa: B018  (creates it)
        return self._np_random_seed

    def render(self):
        return None          # render_modes = []: the batched env has no renderer (out of scope)

    def close_extras(self, **kwargs):
        self.ctx.close()

    def close(self, **kwargs):
        if not self.closed:
            self.close_extras(**kwargs)
            self.closed = True

    def __repr__(self):
        return f"MetaWorldGpuVectorEnv({self.task_list[0] if len(self.task_list) == 1 else str(len(self.task_list)) + ' tasks'}, num_envs={self.num_envs})"


def gather_bookkeeping(local: np.ndarray, device=None):
    """All-gather the per-step bookkeeping record over the ranks of the default process group
    (RCCL when tensors are on the GPU, gloo on CPU).  Returns [world, N_local, 5]."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    world = dist.get_world_size()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # concatenated along dim 0
    dist.all_gather_into_tensor(out, t)
    return out.reshape((world,) + tuple(t.shape)).cpu().numpy()
