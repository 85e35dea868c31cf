"""Run the *unmodified* reference Python (metaworld/*.py from /root/reference) on top of
the oracle physics engine.

TEST INFRASTRUCTURE ONLY.  `install()` injects two stand-in packages into
sys.modules:

  * `mujoco`     -- the subset of the MuJoCo Python bindings the reference touches
                    (call sites: SURVEY.md section 8c), backed by oracle/mjlite.py;
  * `gymnasium`  -- the subset of Gymnasium >= 1.1 the reference touches
                    (MujocoEnv, spaces.Box, Wrapper family, TimeLimit,
                    RecordEpisodeStatistics, SyncVectorEnv with SAME_STEP autoreset),
                    restated from Gymnasium's documented behaviour [EXT].

With these in place `import metaworld` executes the reference's own sources, so
its task classes, wrappers, scripted policies and benchmark builders can be used
to (a) validate the restated task layer and (b) generate golden fixtures
(tools/gen_golden.py).  Only usable where /root/reference exists (this
container); nothing here travels to the GPU box except the fixtures it produced.
"""
from __future__ import annotations

import enum
import sys
import time
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


# =============================================================================
# mujoco stand-in
# =============================================================================
def _make_mujoco():
    from metaworld_amd.mjcf import compile_mjcf
    from oracle.mjlite import OracleData, OracleModel

    mj = types.ModuleType("mujoco")

    class mjtObj(enum.IntEnum):
        mjOBJ_BODY = 1
        mjOBJ_JOINT = 3
        mjOBJ_GEOM = 5
        mjOBJ_SITE = 6

    class mjtEq(enum.IntEnum):
        mjEQ_CONNECT = 0
        mjEQ_WELD = 1

    _KIND = {mjtObj.mjOBJ_BODY: "body", mjtObj.mjOBJ_JOINT: "joint", mjtObj.mjOBJ_GEOM: "geom",
             mjtObj.mjOBJ_SITE: "site"}

    class _Opt:
        def __init__(self, m):
            self.timestep = m.opt_timestep

    class _ModelBody:
        def __init__(self, model, i):
            self._m, self.id = model, i

        @property
        def pos(self):
            return self._m.body_pos[self.id]

        @pos.setter
        def pos(self, v):
            self._m.body_pos[self.id] = v

    class _ModelSite(_ModelBody):
        @property
        def pos(self):
            return self._m.site_pos[self.id]

        @pos.setter
        def pos(self, v):
            self._m.site_pos[self.id] = v

    class _ModelJoint:
        def __init__(self, model, i):
            A = model._src.arrays
            self.id = i
            self.qposadr = np.array([A["jnt_qposadr"][i]])
            self.dofadr = np.array([A["jnt_dofadr"][i]])
            self.type = np.array([A["jnt_type"][i]])

    class _ModelGeom:
        def __init__(self, model, i):
            self.id = i

    class MjModel:
        def __init__(self, src):
            self._src = src
            self._om = OracleModel(src)
            A = src.arrays
            self.nq, self.nv, self.nu = len(A["qpos0"]), len(A["dof_bodyid"]), len(A["act_dofid"])
            self.na = 0
            self.nbody, self.ngeom, self.nsite = len(A["body_parentid"]), len(A["geom_type"]), len(A["site_bodyid"])
            self.nmocap = int(A["body_mocap"].sum())
            self.opt = _Opt(src)
            self.body_pos = self._om.view("body_pos", (self.nbody, 3))
            self.site_pos = self._om.view("site_pos", (self.nsite, 3))
            self.eq_data = self._om.view("eq_data", (len(A["eq_body1"]), 11))
            self.eq_type = np.full(len(A["eq_body1"]), int(mjtEq.mjEQ_WELD))
            self.body_mocapid = np.where(A["body_mocap"] > 0, np.cumsum(A["body_mocap"]) - 1, -1)
            self.jnt_qposadr = A["jnt_qposadr"]
            self.jnt_dofadr = A["jnt_dofadr"]
            self.actuator_ctrlrange = A["act_ctrlrange"].copy()
            self.qpos0 = A["qpos0"].copy()

        @classmethod
        def from_xml_path(cls, path):
            return cls(compile_mjcf(path))

        def body(self, name):
            return _ModelBody(self, self._src.names["body"][name])

        def site(self, name):
            return _ModelSite(self, self._src.names["site"][name])

        def joint(self, name):
            return _ModelJoint(self, self._src.names["joint"][name])

        def geom(self, name):
            return _ModelGeom(self, self._src.names["geom"][name])

    class _DataBody:
        def __init__(self, d, i):
            object.__setattr__(self, "_d", d)
            object.__setattr__(self, "id", i)

        xpos = property(lambda s: s._d.xpos[s.id])
        xquat = property(lambda s: s._d.xquat[s.id])
        xmat = property(lambda s: s._d.xmat[s.id])

        def __setattr__(self, k, v):
            getattr(self, k)[...] = v   # data-level writes (erased by the next FK, like the real thing)

    class _DataGeom(_DataBody):
        xpos = property(lambda s: s._d.geom_xpos[s.id])
        xmat = property(lambda s: s._d.geom_xmat[s.id])

    class _DataSite(_DataBody):
        xpos = property(lambda s: s._d.site_xpos[s.id])
        xmat = property(lambda s: s._d.site_xmat[s.id])

    class _DataJoint:
        def __init__(self, d, j):
            A = d.model._src.arrays
            object.__setattr__(self, "_d", d)
            object.__setattr__(self, "id", j)
            n = 7 if A["jnt_type"][j] == 0 else 1
            nd = 6 if A["jnt_type"][j] == 0 else 1
            object.__setattr__(self, "_q", slice(A["jnt_qposadr"][j], A["jnt_qposadr"][j] + n))
            object.__setattr__(self, "_v", slice(A["jnt_dofadr"][j], A["jnt_dofadr"][j] + nd))

        qpos = property(lambda s: s._d.qpos[s._q])
        qvel = property(lambda s: s._d.qvel[s._v])

        def __setattr__(self, k, v):
            getattr(self, k)[...] = v

    class _Contact:
        __slots__ = ("geom1", "geom2", "geom", "efc_address", "dist", "pos", "frame", "dim")

    _ARRAYS = ("qpos", "qvel", "ctrl", "mocap_pos", "mocap_quat", "qacc", "qacc_warmstart")

    class MjData:
        def __init__(self, model):
            object.__setattr__(self, "model", model)
            od = OracleData(model._om)
            object.__setattr__(self, "_od", od)
            for k in ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "xpos", "xquat", "xmat", "xipos",
                      "geom_xpos", "geom_xmat", "site_xpos", "site_xmat", "qfrc_constraint"):
                object.__setattr__(self, k, getattr(od, k))
            object.__setattr__(self, "mocap_pos", od.mocap_pos.reshape(1, 3))
            object.__setattr__(self, "mocap_quat", od.mocap_quat.reshape(1, 4))
            object.__setattr__(self, "act", np.zeros(0))

        def __setattr__(self, k, v):
            if k in _ARRAYS or k in ("xpos", "site_xpos", "geom_xpos"):
                getattr(self, k)[...] = v
            else:
                object.__setattr__(self, k, v)

        time = property(lambda s: s._od.time)
        efc_force = property(lambda s: s._od.efc_force)
        ncon = property(lambda s: s._od.ncon)
        nefc = property(lambda s: s._od.nefc)

        @property
        def contact(self):
            out = []
            for c in self._od.contacts():
                o = _Contact()
                o.geom1, o.geom2, o.efc_address, o.dist = c["geom1"], c["geom2"], c["efc_address"], c["dist"]
                o.geom = (o.geom1, o.geom2)
                o.pos, o.frame, o.dim = c["pos"], c["frame"], c["dim"]
                out.append(o)
            return out

        def body(self, name):
            return _DataBody(self, self.model._src.names["body"][name])

        def geom(self, name):
            return _DataGeom(self, self.model._src.names["geom"][name])

        def site(self, name):
            return _DataSite(self, self.model._src.names["site"][name])

        def joint(self, name):
            return _DataJoint(self, self.model._src.names["joint"][name])

    def mj_step(model, data, nstep=1):
        data._od.step(nstep)

    def mj_forward(model, data):
        data._od.forward()

    def mj_resetData(model, data):
        data._od.reset()

    def mj_rnePostConstraint(model, data):
        pass

    def mj_name2id(model, objtype, name):
        return model._src.names[_KIND[mjtObj(objtype)]].get(name, -1)

    mj.MjModel, mj.MjData = MjModel, MjData
    mj.mj_step, mj.mj_forward, mj.mj_resetData = mj_step, mj_forward, mj_resetData
    mj.mj_rnePostConstraint, mj.mj_name2id = mj_rnePostConstraint, mj_name2id
    mj.mjtObj, mj.mjtEq = mjtObj, mjtEq
    mj.__version__ = "3.3.0-oracle-standin"
    return mj


# =============================================================================
# gymnasium stand-in
# =============================================================================
def _make_gymnasium(mujoco_mod):
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")
    utils = types.ModuleType("gymnasium.utils")
    seeding = types.ModuleType("gymnasium.utils.seeding")
    ezpickle = types.ModuleType("gymnasium.utils.ezpickle")
    envs = types.ModuleType("gymnasium.envs")
    envs_mujoco = types.ModuleType("gymnasium.envs.mujoco")
    registration = types.ModuleType("gymnasium.envs.registration")
    wrappers = types.ModuleType("gymnasium.wrappers")
    vector = types.ModuleType("gymnasium.vector")

    def np_random(seed=None):
        ss = np.random.SeedSequence(seed)
        return np.random.Generator(np.random.PCG64(ss)), ss.entropy

    seeding.np_random = np_random
    seeding.RandomNumberGenerator = np.random.Generator

    class Space:
        def __init__(self, shape=None, dtype=None, seed=None):
            self._shape, self.dtype, self._np_random = shape, None if dtype is None else np.dtype(dtype), None
            if seed is not None:
                self.seed(seed)

        @property
        def shape(self):
            return self._shape

        @property
        def np_random(self):
            if self._np_random is None:
                self.seed()
            return self._np_random

        def seed(self, seed=None):
            self._np_random, s = np_random(seed)
            return s

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
            super().__init__(tuple(shape), dtype, seed)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e3)
            hi = np.where(np.isfinite(self.high), self.high, 1e3)
            return self.np_random.uniform(lo, hi, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class Discrete(Space):
        def __init__(self, n, seed=None):
            self.n = int(n)
            super().__init__((), np.int64, seed)

        def sample(self):
            return int(self.np_random.integers(self.n))

    spaces.Space, spaces.Box, spaces.Discrete = Space, Box, Discrete

    class EzPickle:
        def __init__(self, *a, **k):
            self._ezpickle_args, self._ezpickle_kwargs = a, k

    class RecordConstructorArgs:
        def __init__(self, **k):
            pass

    ezpickle.EzPickle = EzPickle
    utils.EzPickle, utils.RecordConstructorArgs = EzPickle, RecordConstructorArgs
    utils.seeding, utils.ezpickle = seeding, ezpickle

    class Env:
        metadata = {"render_modes": []}
        render_mode = None
        spec = None
        _np_random = None
        _np_random_seed = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random, self._np_random_seed = np_random(seed)

        @property
        def unwrapped(self):
            return self

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random, self._np_random_seed = np_random()
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

        def close(self):
            pass

        def get_wrapper_attr(self, name):
            return getattr(self, name)

        def has_wrapper_attr(self, name):
            return hasattr(self, name)

        def set_wrapper_attr(self, name, value, *, force=True):
            if force or hasattr(self, name):
                setattr(self, name, value)
                return True
            return False

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env
            self._action_space = self._observation_space = None

        def __getattr__(self, name):
            if name.startswith("_") and name not in ("_np_random",):
                raise AttributeError(name)
            return getattr(self.env, name)

        @property
        def action_space(self):
            return self.env.action_space if self._action_space is None else self._action_space

        @action_space.setter
        def action_space(self, v):
            self._action_space = v

        @property
        def observation_space(self):
            return self.env.observation_space if self._observation_space is None else self._observation_space

        @observation_space.setter
        def observation_space(self, v):
            self._observation_space = v

        @property
        def unwrapped(self):
            return self.env.unwrapped

        @property
        def np_random(self):
            return self.env.np_random

        @np_random.setter
        def np_random(self, v):
            self.env.np_random = v

        def reset(self, *, seed=None, options=None):
            return self.env.reset(seed=seed, options=options)

        def step(self, action):
            return self.env.step(action)

        def close(self):
            return self.env.close()

        def get_wrapper_attr(self, name):
            if name in self.__dict__ or hasattr(type(self), name):
                return getattr(self, name)
            return self.env.get_wrapper_attr(name)

        def has_wrapper_attr(self, name):
            return name in self.__dict__ or hasattr(type(self), name) or self.env.has_wrapper_attr(name)

        def set_wrapper_attr(self, name, value, *, force=True):
            if name in self.__dict__ or hasattr(type(self), name):
                setattr(self, name, value)
                return True
            done = self.env.set_wrapper_attr(name, value, force=False)
            if done:
                return True
            if force:
                setattr(self, name, value)
                return True
            return False

    class ObservationWrapper(Wrapper):
        def reset(self, *, seed=None, options=None):
            obs, info = self.env.reset(seed=seed, options=options)
            return self.observation(obs), info

        def step(self, action):
            obs, r, te, tr, info = self.env.step(action)
            return self.observation(obs), r, te, tr, info

    class TimeLimit(Wrapper):
        def __init__(self, env, max_episode_steps):
            super().__init__(env)
            self._max_episode_steps = max_episode_steps
            self._elapsed_steps = None

        def step(self, action):
            obs, r, te, tr, info = self.env.step(action)
            self._elapsed_steps += 1
            if self._elapsed_steps >= self._max_episode_steps:
                tr = True
            return obs, r, te, tr, info

        def reset(self, *, seed=None, options=None):
            self._elapsed_steps = 0
            return self.env.reset(seed=seed, options=options)

    class RecordEpisodeStatistics(Wrapper):
        def __init__(self, env, buffer_length=100, stats_key="episode"):
            super().__init__(env)
            self._stats_key = stats_key
            self.episode_count = 0
            self.episode_start_time = -1.0
            self.episode_returns, self.episode_lengths = 0.0, 0

        def step(self, action):
            obs, r, te, tr, info = self.env.step(action)
            self.episode_returns += r
            self.episode_lengths += 1
            if te or tr:
                info = dict(info)
                info[self._stats_key] = {"r": self.episode_returns, "l": self.episode_lengths,
                                         "t": round(time.perf_counter() - self.episode_start_time, 6)}
                self.episode_count += 1
            return obs, r, te, tr, info

        def reset(self, *, seed=None, options=None):
            out = self.env.reset(seed=seed, options=options)
            self.episode_start_time = time.perf_counter()
            self.episode_returns, self.episode_lengths = 0.0, 0
            return out

    def _unsupported(*a, **k):
        raise NotImplementedError("not provided by the gymnasium stand-in")

    wrappers.TimeLimit, wrappers.RecordEpisodeStatistics = TimeLimit, RecordEpisodeStatistics
    wrappers.NormalizeReward = wrappers.NormalizeObservation = _unsupported

    class MujocoEnv(Env):
        def __init__(self, model_path, frame_skip, observation_space, render_mode=None, width=480, height=480,
                     camera_id=None, camera_name=None, default_camera_config=None, max_geom=1000,
                     visual_options=None):
            self.fullpath = model_path
            self.width, self.height = width, height
            self.model = mujoco_mod.MjModel.from_xml_path(self.fullpath)
            self.data = mujoco_mod.MjData(self.model)
            self.init_qpos = self.data.qpos.ravel().copy()
            self.init_qvel = self.data.qvel.ravel().copy()
            self.frame_skip = frame_skip
            if observation_space is not None:
                self.observation_space = observation_space
            cr = self.model.actuator_ctrlrange.astype(np.float32)
            self.action_space = Box(cr[:, 0], cr[:, 1], dtype=np.float32)
            self.render_mode, self.camera_name, self.camera_id = render_mode, camera_name, camera_id

        @property
        def dt(self):
            return self.model.opt.timestep * self.frame_skip

        def set_state(self, qpos, qvel):
            assert qpos.shape == (self.model.nq,) and qvel.shape == (self.model.nv,)
            self.data.qpos[:] = np.copy(qpos)
            self.data.qvel[:] = np.copy(qvel)
            mujoco_mod.mj_forward(self.model, self.data)

        def do_simulation(self, ctrl, n_frames):
            if np.array(ctrl).shape != (self.model.nu,):
                raise ValueError(f"Action dimension mismatch. Expected {(self.model.nu,)}, found {np.array(ctrl).shape}")
            self.data.ctrl[:] = ctrl
            mujoco_mod.mj_step(self.model, self.data, nstep=n_frames)
            mujoco_mod.mj_rnePostConstraint(self.model, self.data)

        def reset(self, *, seed=None, options=None):
            Env.reset(self, seed=seed)
            mujoco_mod.mj_resetData(self.model, self.data)
            ob = self.reset_model()
            return ob, {}

        def get_body_com(self, body_name):
            return self.data.body(body_name).xpos

        def state_vector(self):
            return np.concatenate([self.data.qpos.flat, self.data.qvel.flat])

        def render(self):
            return None

    envs_mujoco.MujocoEnv = MujocoEnv
    envs.mujoco = envs_mujoco
    # minimal registry: what `register(id, entry_point=..., vector_entry_point=..., kwargs=...)` / `gym.make_vec(id, num_envs=..., **kw)`
    # do for a spec that has a vector entry point (gymnasium 1.1 `make_vec`: `env_creator(num_envs=num_envs, **spec_kwargs)`)
    registration.registry = {}

    def _register(id, entry_point=None, vector_entry_point=None, kwargs=None, **_ignored):
        registration.registry[id] = dict(entry_point=entry_point, vector_entry_point=vector_entry_point, kwargs=dict(kwargs or {}))

    def _make_vec(id, num_envs=None, vectorization_mode=None, vector_kwargs=None, wrappers=None, **kwargs):
        spec = registration.registry.get(id)
        if spec is None:
            raise KeyError(f"Environment `{id}` doesn't exist.")
        if spec["vector_entry_point"] is None:
            raise NotImplementedError("the gymnasium stand-in only builds specs that have a vector_entry_point")
        kw = dict(spec["kwargs"]); kw.update(kwargs)
        env = spec["vector_entry_point"](num_envs=num_envs, **kw)
        # what gymnasium 1.1's make_vec does with the object it gets back (envs/registration.py): the spec is written through
        # `unwrapped`, and a vector env without an autoreset mode in its metadata draws a warning
        env.unwrapped.spec = types.SimpleNamespace(id=id, kwargs=kw, vector_entry_point=spec["vector_entry_point"])
        assert "autoreset_mode" in env.metadata, "vector env without metadata['autoreset_mode']"
        return env

    registration.register = _register
    envs.registration = registration

    # ---- vector: SyncVectorEnv with SAME_STEP / NEXT_STEP autoreset (dict-of-arrays infos) ----
    class AutoresetMode(enum.Enum):
        NEXT_STEP = "NextStep"
        SAME_STEP = "SameStep"
        DISABLED = "Disabled"

    class VectorEnv:
        pass

    def _add_info(vinfo, info, i, n):
        for k, v in info.items():
            if isinstance(v, dict):
                vinfo[k] = _add_info(vinfo.get(k, {}), v, i, n)
                continue
            if k not in vinfo:
                if isinstance(v, (int, float, bool, np.number, np.bool_)):
                    vinfo[k] = np.zeros(n, dtype=np.asarray(v).dtype)
                elif isinstance(v, np.ndarray):
                    vinfo[k] = np.zeros((n,) + v.shape, dtype=v.dtype)
                else:
                    vinfo[k] = np.full(n, None, dtype=object)
                vinfo["_" + k] = np.zeros(n, dtype=bool)
            vinfo[k][i] = v
            vinfo["_" + k][i] = True
        return vinfo

    class SyncVectorEnv(VectorEnv):
        def __init__(self, env_fns, copy=True, observation_mode="same", autoreset_mode=AutoresetMode.NEXT_STEP):
            self.envs = [fn() for fn in env_fns]
            self.num_envs = len(self.envs)
            self.autoreset_mode = AutoresetMode(autoreset_mode) if not isinstance(autoreset_mode, AutoresetMode) else autoreset_mode
            self.metadata = {"autoreset_mode": self.autoreset_mode}
            self.single_observation_space = self.envs[0].observation_space
            self.single_action_space = self.envs[0].action_space
            so = self.single_observation_space
            self.observation_space = Box(np.stack([so.low] * self.num_envs), np.stack([so.high] * self.num_envs), dtype=so.dtype)
            sa = self.single_action_space
            self.action_space = Box(np.stack([sa.low] * self.num_envs), np.stack([sa.high] * self.num_envs), dtype=sa.dtype)
            self._autoreset = np.zeros(self.num_envs, dtype=bool)

        def reset(self, *, seed=None, options=None):
            seeds = [None] * self.num_envs if seed is None else ([seed + i for i in range(self.num_envs)] if isinstance(seed, int) else list(seed))
            obs, infos = [], {}
            for i, (e, s) in enumerate(zip(self.envs, seeds)):
                o, info = e.reset(seed=s, options=options)
                obs.append(o)
                infos = _add_info(infos, info, i, self.num_envs)
            self._autoreset[:] = False
            return np.stack(obs).astype(self.single_observation_space.dtype), infos

        def step(self, actions):
            n = self.num_envs
            obs, infos = [None] * n, {}
            rew, term, trunc = np.zeros(n, dtype=np.float64), np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
            for i, e in enumerate(self.envs):
                if self.autoreset_mode == AutoresetMode.NEXT_STEP and self._autoreset[i]:
                    o, info = e.reset()
                    r, te, tr = 0.0, False, False
                else:
                    o, r, te, tr, info = e.step(actions[i])
                    if (te or tr) and self.autoreset_mode == AutoresetMode.SAME_STEP:
                        final_o, final_info = o, info
                        o, info = e.reset()
                        info = dict(info)
                        info["final_obs"], info["final_info"] = final_o, final_info
                obs[i], rew[i], term[i], trunc[i] = o, r, te, tr
                infos = _add_info(infos, info, i, n)
            self._autoreset = term | trunc
            return np.stack(obs).astype(self.single_observation_space.dtype), rew, term, trunc, infos

        def call(self, name, *a, **k):
            out = []
            for e in self.envs:
                f = e.get_wrapper_attr(name)
                out.append(f(*a, **k) if callable(f) else f)
            return tuple(out)

        def get_attr(self, name):
            return self.call(name)

        def set_attr(self, name, values):
            if not isinstance(values, (list, tuple)):
                values = [values] * self.num_envs
            for e, v in zip(self.envs, values):
                e.set_wrapper_attr(name, v)

        def close(self, **k):
            for e in self.envs:
                e.close()

    vector.AutoresetMode, vector.VectorEnv, vector.SyncVectorEnv = AutoresetMode, VectorEnv, SyncVectorEnv
    vector.AsyncVectorEnv = SyncVectorEnv  # single-process stand-in

    gym.Env, gym.Wrapper, gym.ObservationWrapper = Env, Wrapper, ObservationWrapper
    gym.spaces, gym.utils, gym.envs, gym.wrappers, gym.vector = spaces, utils, envs, wrappers, vector
    gym.Space = Space
    gym.make = _unsupported
    gym.make_vec, gym.register = _make_vec, _register
    gym.__version__ = "1.1-oracle-standin"
    mods = {"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.utils": utils,
            "gymnasium.utils.seeding": seeding, "gymnasium.utils.ezpickle": ezpickle, "gymnasium.envs": envs,
            "gymnasium.envs.mujoco": envs_mujoco, "gymnasium.envs.registration": registration,
            "gymnasium.wrappers": wrappers, "gymnasium.vector": vector}
    return mods


_installed = False


def install(reference_root=REFERENCE_ROOT):
    """Make `import metaworld` resolve to the reference sources running on the oracle engine."""
    global _installed
    if _installed:
        return
    import os
    if not os.path.isdir(os.path.join(reference_root, "metaworld")):
        raise RuntimeError(f"reference sources not found under {reference_root}")
    mj = _make_mujoco()
    sys.modules["mujoco"] = mj
    sys.modules.update(_make_gymnasium(mj))
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    _installed = True
