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


class MetaWorldGpuVectorEnv:
    metadata = {"autoreset_mode": "SameStep", "render_modes": []}

    def __init__(self, benchmark="MT1", env_name=None, num_envs=None, seed=None, use_one_hot=False,
                 max_episode_steps=None, terminate_on_success=False, precision="fp32", device_id=0,
                 rank=0, world_size=1, goal_seed=42, task_names=None, lib=None, maxcon=None, maxefc=None,
                 partially_observable=None):
        names, goal_key = benchmark_tasks(benchmark, env_name)
        if task_names is not None:          # restrict a benchmark to the tasks that have device code (tests)
            names = [n for n in names if n in task_names]
        missing = [n for n in names if n not in T.TASK_DEFS]
        if missing:
            raise NotImplementedError(f"no device-side task code yet for: {missing}")
        self.task_list = names
        ntask = len(names)
        self.num_envs = int(num_envs) if num_envs else ntask
        if self.num_envs < ntask:
            raise ValueError("num_envs must be >= number of tasks")
        self.rank, self.world_size = rank, world_size
        self.use_one_hot = bool(use_one_hot)
        # ML benchmarks hide the goal (metaworld/__init__.py `_ML_OVERRIDE`), MT ones show it; the reference's own policy test
        # makes it visible for ML too (tests/metaworld/test_evaluation.py:70-82) -> overridable
        self.partially_observable = benchmark.startswith("ML") if partially_observable is None else bool(partially_observable)
        self.seed_value = seed
        self._lib = lib or native.load()
        self.ctx = native.Context(self._lib, precision=1 if precision in ("fp64", 1) else 0, device_id=device_id,
                                  rank=rank, world_size=world_size, max_episode_steps=max_episode_steps or 500,
                                  terminate_on_success=terminate_on_success, one_hot=use_one_hot, num_tasks=ntask)
        # models and tasks
        model_index, roles_of, reloc_of = {}, {}, {}
        self._task_index = {}
        self.goal_tables = {}
        for oh, name in enumerate(names):
            mname = T.TASK_CONST[name]["model"]
            if mname not in model_index:
                pk, roles, reloc = T.packed_model(mname, maxcon=maxcon, maxefc=maxefc,
                                                  tolerance=None if precision in ("fp64", 1) else 1e-6)
                model_index[mname] = self.ctx.add_model(pk)
                roles_of[mname], reloc_of[mname] = roles, reloc
            goals = T.goal_table(goal_key, name, goal_seed)
            self.goal_tables[name] = goals
            ts = T.task_struct(name, model_index[mname], roles_of[mname], reloc_of[mname], onehot_id=oh,
                               partially_observable=self.partially_observable)
            self._task_index[name] = self.ctx.add_task(ts, goals)
        # env -> task (task-major contiguous blocks, like the reference's enumerate order)
        per, rem = divmod(self.num_envs, ntask)
        env_task_names = []
        for i, name in enumerate(names):
            env_task_names += [name] * (per + (1 if i < rem else 0))
        self.env_task_names = env_task_names
        self.ctx.set_envs([self._task_index[n] for n in env_task_names])
        self.ctx.finalize()
        self._ngoals = np.array([len(self.goal_tables[n]) for n in env_task_names])
        # RandomTaskSelectWrapper: every sub-env is seeded with the SAME seed (metaworld/__init__.py:499),
        # so the n-th reset of any env draws the n-th value of one PCG64 stream.
        self._stream = {}
        self._reset_count = np.zeros(self.num_envs, dtype=np.int64)
        self._next_goal = np.zeros(self.num_envs, dtype=np.int32)
        self.sample_tasks_on_reset = True
        self.terminate_on_success = bool(terminate_on_success)
        D = self.ctx.D
        lo = np.concatenate([[-0.525, 0.348, -0.0525, -1.0], np.full(14, -np.inf)] * 2 + [np.zeros(3)])
        hi = np.concatenate([[0.525, 1.025, 0.7, 1.0], np.full(14, np.inf)] * 2 + [np.zeros(3)])
        if use_one_hot:
            lo, hi = np.concatenate([lo, np.zeros(ntask)]), np.concatenate([hi, np.ones(ntask)])
        self.obs_dtype = np.float32 if use_one_hot else np.float64
        self.single_observation_space = _box(lo, hi, self.obs_dtype)
        self.single_action_space = _box(-np.ones(4), np.ones(4), np.float32)
        self.observation_space = _box(np.tile(lo, (self.num_envs, 1)), np.tile(hi, (self.num_envs, 1)), self.obs_dtype)
        self.action_space = _box(-np.ones((self.num_envs, 4)), np.ones((self.num_envs, 4)), np.float32)
        self._episode_start = np.full(self.num_envs, time.perf_counter())
        self.closed = False
        assert D == len(lo)

    # ---- RandomTaskSelectWrapper stream (wrappers.py:98-100: self.np_random.choice(len(tasks))) ----
    def _draw(self, n_goals, k):
        """k-th draw of Generator(PCG64(seed)).choice(n_goals): every sub-env owns an identically seeded
        generator and only ever calls choice(len(tasks)), so one cached stream per distinct n serves all."""
        if n_goals not in self._stream:
            self._stream[n_goals] = (np.random.Generator(np.random.PCG64(np.random.SeedSequence(self.seed_value))), [])
        gen, s = self._stream[n_goals]
        while len(s) <= k:
            s.append(int(gen.choice(n_goals)))
        return s[k]

    def _advance_goals(self, mask):
        for e in np.flatnonzero(mask):
            if self.sample_tasks_on_reset:
                self._next_goal[e] = self._draw(int(self._ngoals[e]), int(self._reset_count[e]))
                self._reset_count[e] += 1

    # ---- VectorEnv API ----
    def reset(self, *, seed=None, options=None):
        mask = np.ones(self.num_envs, dtype=bool)
        self._advance_goals(mask)
        obs = self.ctx.reset(self._next_goal)
        self._advance_goals(mask)          # pre-draw the goal of the next auto-reset
        self._episode_start[:] = time.perf_counter()
        return obs.astype(self.obs_dtype, copy=True), {}

    def step(self, actions):
        obs, rew, term, trunc, succ, info = self.ctx.step(actions, self._next_goal)
        term_b, trunc_b = term.astype(bool), trunc.astype(bool)
        infos = {"success": succ.astype(np.float64), "_success": np.ones(self.num_envs, dtype=bool)}
        for k, key in enumerate(INFO_KEYS):
            infos[key] = info[:, k].astype(np.float64)
            infos["_" + key] = infos["_success"]
        done = term_b | trunc_b
        if done.any():
            now = time.perf_counter()
            fi = {k: np.where(done, v, 0) for k, v in infos.items() if not k.startswith("_")}
            for k in list(fi):
                fi["_" + k] = done.copy()
            fi["episode"] = {"r": np.where(done, self.ctx.ep_ret, 0.0), "l": np.where(done, self.ctx.ep_len, 0),
                             "t": np.where(done, np.round(now - self._episode_start, 6), 0.0),
                             "_r": done.copy(), "_l": done.copy(), "_t": done.copy()}
            fi["_episode"] = done.copy()
            infos["final_info"], infos["_final_info"] = fi, done.copy()
            fo = np.empty(self.num_envs, dtype=object)
            for e in np.flatnonzero(done):
                fo[e] = self.ctx.final_obs[e].astype(self.obs_dtype)
            infos["final_obs"], infos["_final_obs"] = fo, done.copy()
            self._episode_start[done] = now
            self._advance_goals(done)
        return obs.astype(self.obs_dtype, copy=True), rew.copy(), term_b, trunc_b, infos

    def get_attr(self, name):
        if name == "task_name":
            return tuple(T.TASK_CONST[n]["cls"] for n in self.env_task_names)
        if name == "terminate_on_success":
            return (self.terminate_on_success,) * self.num_envs
        if name == "_partially_observable":
            return (self.partially_observable,) * self.num_envs
        if name == "_last_rand_vec":
            return tuple(self.goal_tables[n][g] for n, g in zip(self.env_task_names, self._current_goals()))
        if name == "tasks":
            return tuple(self.goal_tables[n] for n in self.env_task_names)
        raise AttributeError(name)

    def _current_goals(self):
        return [int(self.ctx.read(e, "task", 2)[1]) for e in range(self.num_envs)]

    def set_attr(self, name, values):
        """gymnasium VectorEnv.set_attr for the attributes of the reference's wrapper stack that are state here"""
        vals = list(values) if isinstance(values, (list, tuple, np.ndarray)) else [values] * self.num_envs
        if name == "terminate_on_success":
            assert len(set(bool(v) for v in vals)) == 1, "terminate_on_success is one flag per vector env"
            self.call("toggle_terminate_on_success", bool(vals[0]))
        elif name == "sample_tasks_on_reset":
            self.sample_tasks_on_reset = bool(vals[0])
        else:
            raise NotImplementedError(name)

    def call(self, name, *args, **kwargs):
        """gymnasium VectorEnv.call for the methods the reference's wrappers expose (metaworld/wrappers.py:107-142, :222-223;
        used by metaworld/evaluation.py:53-125)."""
        if name == "toggle_sample_tasks_on_reset":
            self.sample_tasks_on_reset = bool(args[0])
            return (None,) * self.num_envs
        if name == "toggle_terminate_on_success":
            self.terminate_on_success = bool(args[0])
            self.ctx.set_terminate_on_success(self.terminate_on_success)
            return (None,) * self.num_envs
        if name == "sample_tasks":          # RandomTaskSelectWrapper.sample_tasks: draw a task for every env, then reset it
            saved, self.sample_tasks_on_reset = self.sample_tasks_on_reset, True
            mask = np.ones(self.num_envs, dtype=bool)
            self._advance_goals(mask)
            self.sample_tasks_on_reset = saved
            obs = self.ctx.reset(self._next_goal).astype(self.obs_dtype, copy=True)
            self._advance_goals(mask)
            self._episode_start[:] = time.perf_counter()
            return tuple((obs[e], {}) for e in range(self.num_envs))
        if name == "get_checkpoint":
            ck = dict(tasks={n: t.copy() for n, t in self.goal_tables.items()}, reset_count=self._reset_count.copy(),
                      next_goal=self._next_goal.copy(), seed=self.seed_value, sample_tasks_on_reset=self.sample_tasks_on_reset,
                      state=[self.ctx.read(e, "state") for e in range(self.num_envs)])
            return (ck,) + (None,) * (self.num_envs - 1)
        if name == "load_checkpoint":
            ck = args[0][0] if isinstance(args[0], (list, tuple)) else args[0]
            assert all(np.array_equal(ck["tasks"][n], t) for n, t in self.goal_tables.items()), "checkpoint of another benchmark/seed"
            self._reset_count[:] = ck["reset_count"]; self._next_goal[:] = ck["next_goal"]
            self.sample_tasks_on_reset = ck["sample_tasks_on_reset"]
            for e in range(self.num_envs):
                self.ctx.write(e, "state", ck["state"][e])
            return (None,) * self.num_envs
        return self.get_attr(name)

    def bookkeeping(self):
        """[N, 5] float64 record (done, success, task_id, episode_return, episode_length) of the last step:
        what the cross-rank gather exchanges."""
        c = self.ctx
        tid = np.array([T.TASK_CONST[n]["id"] for n in self.env_task_names], dtype=np.float64)
        done = (c.terminated | c.truncated).astype(np.float64)
        return np.stack([done, c.success.astype(np.float64), tid, c.ep_ret * done, c.ep_len * done], axis=1)

    def close(self):
        if not self.closed:
            self.ctx.close()
            self.closed = True


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
