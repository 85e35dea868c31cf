"""ctypes binding of the C ABI in include/mwgpu.h.

`load()` returns the HIP library (metaworld_amd/libmwgpu.so) and raises if it is missing or
if no GPU is visible -- the product path has no CPU fallback.  `load(prefix="mwh_", path=...)`
is used by the CPU test-suite to drive tests/host_harness.cpp (same lane code, host loops).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmwgpu.so")
LIB_PATH_V1 = LIB_PATH          # rounds 2-4 built the v1 reward functions into a second library; they are a run-time flag now (mw_config.reward_version)
NPROBE = 16
STATUS_WORDS = 8          # MW_STATUS_WORDS (include/mwgpu.h)


def source_hash():
    """sha256 over the device sources (csrc/*, include/mwgpu.h): written into every profile summary (tools/summarize_pmc.py) and
    compared by bench.py, which only quotes counter-derived numbers (roofline.traffic / alu_issue) of a profile taken on the SAME
    sources (the .git directory does not travel to the GPU box, so this is a content hash, not a commit id)"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for f in sorted(os.listdir(src)) + [os.path.join("..", "..", "include", "mwgpu.h")]:
        with open(os.path.join(src, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


class MwConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("precision", "device_id", "rank", "world_size", "max_episode_steps",
                                         "terminate_on_success", "one_hot", "num_tasks", "full_forward", "reward_version")]


class MwTask(C.Structure):
    _fields_ = [("kind", C.c_int32), ("model", C.c_int32), ("onehot_id", C.c_int32), ("probe", C.c_int32 * NPROBE),
                ("nobj", C.c_int32), ("quat_mode", C.c_int32 * 2), ("qadr", C.c_int32 * 4), ("dadr", C.c_int32 * 4),
                ("geom", C.c_int32 * 4), ("reloc", C.c_int32 * 2), ("partially_observable", C.c_int32),
                ("max_path_length", C.c_int32), ("hand_init", C.c_double * 3), ("mocap_low", C.c_double * 3),
                ("mocap_high", C.c_double * 3), ("goal_low", C.c_double * 3), ("goal_high", C.c_double * 3),
                ("obj_off", (C.c_double * 3) * 2), ("c", C.c_double * 15)]


class MwDeviceOut(C.Structure):
    """mw_device_out (include/mwgpu.h): caller-owned DEVICE output buffers of mw_step_device"""
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("flags", C.c_void_p), ("info", C.c_void_p),
                ("final_obs", C.c_void_p), ("episode_return", C.c_void_p), ("episode_length", C.c_void_p)]


# mw_bookkeeping (include/mwgpu.h): 12 bytes per env and step
BOOKKEEPING_DTYPE = np.dtype([("done", np.uint8), ("success", np.uint8), ("task_id", np.int16), ("episode_return", np.float32),
                              ("episode_length", np.int32)])
assert BOOKKEEPING_DTYPE.itemsize == 12


class Lib:
    def __init__(self, path, prefix):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        if prefix == "mw_":
            # One HIP runtime per process: torch wheels bundle their own libamdhip64, and whichever copy is loaded first owns
            # the device.  Importing torch first (when present) makes libmwgpu.so bind to that copy, so the library and
            # torch tensors (MetaWorldTorchVectorEnv, RCCL in bench.py) can share the GPU whatever the caller's import order.
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        self.dll = C.CDLL(path)
        self.path = path
        self.prefix = prefix
        f = self._f
        f("model_new", C.c_void_p)
        f("model_free", None, C.c_void_p)
        f("model_set_int", C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int)
        f("model_set_real", C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int)
        f("model_set_option", C.c_int, C.c_void_p, C.c_char_p, C.c_double)
        f("alloc_host", C.c_void_p, C.c_size_t)
        f("free_host", None, C.c_void_p)
        f("create", C.c_int, C.POINTER(MwConfig), C.POINTER(C.c_void_p))
        f("add_model", C.c_int, C.c_void_p, C.c_void_p)
        f("add_task", C.c_int, C.c_void_p, C.POINTER(MwTask), C.c_void_p, C.c_int)
        f("set_envs", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        f("finalize", C.c_int, C.c_void_p)
        f("set_terminate_on_success", C.c_int, C.c_void_p, C.c_int)
        f("destroy", None, C.c_void_p)
        f("last_error", C.c_char_p, C.c_void_p)
        f("num_envs", C.c_int, C.c_void_p)
        f("obs_dim", C.c_int, C.c_void_p)
        f("reset", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
        f("step", C.c_int, *([C.c_void_p] * 12))
        f("upload_actions", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        f("step_resident", C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float))
        f("step_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MwDeviceOut))
        f("reset_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
        f("step_device_on", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MwDeviceOut), C.c_void_p)
        f("wait_done", C.c_int, C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)))
        f("policy_actions", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
        f("policy_rollout", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float))
        f("policy_rollout_fused", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float))
        f("step_resident_gather", C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float))
        f("comm_unique_id", C.c_int, C.c_void_p)
        f("comm_init", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
        f("comm_info", C.c_int, C.c_void_p, C.c_void_p)
        f("gather_bookkeeping", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        f("status", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
        f("launch_times", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        f("set_option", C.c_int, C.c_void_p, C.c_char_p, C.c_double)
        f("set_episode_phase", C.c_int, C.c_void_p, C.c_void_p)
        f("step_resident_fused", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float))
        f("set_goal_schedule", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        f("goal_schedule_pos", C.c_int, C.c_void_p, C.c_void_p)
        f("column_size", C.c_int, C.c_void_p, C.c_int, C.c_char_p)
        f("read", C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int)
        f("write", C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int)
        f("get_state", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        f("set_state", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        f("read_int", C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int)
        f("debug", C.c_int, C.c_void_p, C.c_int, C.c_int)

    def _f(self, name, restype, *argtypes):
        fn = getattr(self.dll, self.prefix + name)
        fn.restype = restype
        fn.argtypes = list(argtypes)
        setattr(self, name, fn)


_libs = {}


def load(prefix="mw_", path=None) -> Lib:
    if path is None and prefix == "mw_" and os.environ.get("MW_LIB_OVERRIDE"):          # experiments: run anything (tests, tools) on a variant build of the library
        path = os.path.join(_HERE, os.environ["MW_LIB_OVERRIDE"])
    path = path or LIB_PATH
    key = (prefix, path)
    if key not in _libs:
        _libs[key] = Lib(path, prefix)
    return _libs[key]


EXPORTED_SYMBOLS = ["alloc_host", "free_host", "model_new", "model_free", "model_set_int", "model_set_real", "model_set_option", "create",
                    "add_model", "add_task", "set_envs", "finalize", "set_terminate_on_success", "destroy", "last_error", "num_envs", "obs_dim",
                    "reset", "step", "step_device", "step_device_on", "wait_done", "reset_device", "policy_actions", "policy_rollout", "policy_rollout_fused", "upload_actions", "step_resident", "step_resident_fused", "step_resident_gather", "comm_unique_id", "comm_init", "comm_info", "gather_bookkeeping", "status", "launch_times", "set_option", "set_episode_phase", "set_goal_schedule", "goal_schedule_pos", "column_size", "read", "write", "get_state", "set_state", "read_int",
                    "debug"]


class _PinnedBlock:
    """one mw_alloc_host block; released when the last numpy view of it is garbage-collected"""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.free_host(self.ptr)
        except Exception:
            pass


class Context:
    """Thin object wrapper over mw_ctx."""

    def __init__(self, lib: Lib, precision=0, device_id=0, rank=0, world_size=1, max_episode_steps=500,
                 terminate_on_success=False, one_hot=False, num_tasks=1, full_forward=False, reward_version=2):
        self.lib = lib
        cfg = MwConfig(int(precision), device_id, rank, world_size, max_episode_steps, int(terminate_on_success),
                       int(one_hot), num_tasks, int(full_forward), int(reward_version))
        self.ptr = C.c_void_p()
        rc = lib.create(C.byref(cfg), C.byref(self.ptr))
        self._check(rc)
        self.N = 0
        self.D = 39 + (num_tasks if one_hot else 0)

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError("mwgpu: " + self.lib.last_error(self.ptr).decode())
        return rc

    def add_model(self, packed) -> int:
        """packed: dict(ints={name: int32 array}, reals={name: float64 array}, options={name: float})"""
        L = self.lib
        m = C.c_void_p(L.model_new())
        keep = []
        for k, v in packed["ints"].items():
            a = np.ascontiguousarray(v, dtype=np.int32)
            keep.append(a)
            L.model_set_int(m, k.encode(), a.ctypes.data, a.size)
        for k, v in packed["reals"].items():
            a = np.ascontiguousarray(v, dtype=np.float64)
            keep.append(a)
            L.model_set_real(m, k.encode(), a.ctypes.data, a.size)
        for k, v in packed["options"].items():
            if L.model_set_option(m, k.encode(), float(v)) != 0:
                raise RuntimeError(f"unknown model option {k}")
        idx = self._check(L.add_model(self.ptr, m))
        L.model_free(m)
        return idx

    def add_task(self, task: MwTask, goals) -> int:
        g = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 6)
        return self._check(self.lib.add_task(self.ptr, C.byref(task), g.ctypes.data, len(g)))

    def set_envs(self, env_task):
        a = np.ascontiguousarray(env_task, dtype=np.int32)
        self.N = len(a)
        self._check(self.lib.set_envs(self.ptr, a.ctypes.data, len(a)))

    def finalize(self):
        self._check(self.lib.finalize(self.ptr))
        N, D = self.N, self.D
        # the step / reset buffers live in page-locked memory (mw_alloc_host): the outputs of mw_step arrive as asynchronous DMA copies
        z = self._host_zeros
        self.obs = z((N, D), np.float64); self.final_obs = z((N, D), np.float64)
        self.reward = z((N,), np.float64); self.ep_ret = z((N,), np.float64); self.ep_len = z((N,), np.int32)
        self.terminated = z((N,), np.uint8); self.truncated = z((N,), np.uint8)
        self.success = z((N,), np.uint8); self.info = z((N, 6), np.float32)
        self._act_stage = z((N, 4), np.float32)

    def _host_zeros(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.alloc_host(n) if n else None
        if not p:
            return np.zeros(shape, dtype=dtype)
        buf = (C.c_char * n).from_address(p)
        buf._mw_owner = _PinnedBlock(self.lib, p)          # freed when the last numpy view of the block is gone (views may outlive close())
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        a[...] = 0
        return a

    def set_terminate_on_success(self, on):
        self._check(self.lib.set_terminate_on_success(self.ptr, int(bool(on))))

    def reset(self, goal_idx, mask=None):
        g = np.ascontiguousarray(goal_idx, dtype=np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.lib.reset(self.ptr, None if mk is None else mk.ctypes.data, g.ctypes.data, self.obs.ctypes.data))
        return self.obs

    def step(self, actions, next_goal=None):
        assert np.shape(actions) == (self.N, 4)
        a = self._act_stage
        a[...] = actions          # (pinned staging: the upload is an asynchronous copy too)
        ng = None if next_goal is None else np.ascontiguousarray(next_goal, dtype=np.int32)
        self._check(self.lib.step(self.ptr, a.ctypes.data, None if ng is None else ng.ctypes.data, self.obs.ctypes.data,
                                  self.reward.ctypes.data, self.terminated.ctypes.data, self.truncated.ctypes.data,
                                  self.success.ctypes.data, self.info.ctypes.data, self.final_obs.ctypes.data,
                                  self.ep_ret.ctypes.data, self.ep_len.ctypes.data))
        return self.obs, self.reward, self.terminated, self.truncated, self.success, self.info

    def step_device(self, actions_ptr, next_goal_ptr=None, out: MwDeviceOut | None = None):
        """mw_step_device: raw device pointers in, outputs into the caller's device buffers"""
        self._check(self.lib.step_device(self.ptr, actions_ptr, next_goal_ptr, None if out is None else C.byref(out)))

    def step_device_on(self, actions_ptr, next_goal_ptr, out, stream_handle):
        """mw_step_device_on: the step ordered against the caller's stream by events; returns at once"""
        self._check(self.lib.step_device_on(self.ptr, actions_ptr, next_goal_ptr, None if out is None else C.byref(out), stream_handle))

    def wait_done(self):
        """mw_wait_done -> uint8 [N] view of the pinned host copy of the last step's `done` row (valid until the next step)"""
        p = C.POINTER(C.c_uint8)()
        self._check(self.lib.wait_done(self.ptr, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(self.N,))

    def reset_device(self, goal_idx_ptr, mask_ptr=None, obs_ptr=None):
        self._check(self.lib.reset_device(self.ptr, mask_ptr, goal_idx_ptr, obs_ptr))

    def policy_actions(self, policy_id, obs):
        """mw_policy_actions: the device scripted policies on host observations -> float32 [N, 4]"""
        pid = np.ascontiguousarray(policy_id, dtype=np.int32)
        o = np.ascontiguousarray(obs, dtype=np.float64)
        assert pid.shape == (self.N,) and o.shape == (self.N, self.D)
        act = np.zeros((self.N, 4), dtype=np.float32)
        self._check(self.lib.policy_actions(self.ptr, pid.ctypes.data, o.ctypes.data, act.ctypes.data))
        return act

    def policy_rollout(self, policy_id, goal_schedule, nsteps, steps_per_launch=None):
        """mw_policy_rollout (or mw_policy_rollout_fused with steps_per_launch (policy, step) pairs per kernel launch)
        -> (episodes [N], successes [N], kernel ms)"""
        pid = np.ascontiguousarray(policy_id, dtype=np.int32)
        sch = np.ascontiguousarray(goal_schedule, dtype=np.int32)
        assert pid.shape == (self.N,) and sch.ndim == 2 and sch.shape[1] == self.N
        ep, su, ms = np.zeros(self.N, dtype=np.int32), np.zeros(self.N, dtype=np.int32), C.c_float(0)
        if steps_per_launch:
            self._check(self.lib.policy_rollout_fused(self.ptr, pid.ctypes.data, sch.ctypes.data, sch.shape[0], int(nsteps), int(steps_per_launch),
                                                      ep.ctypes.data, su.ctypes.data, C.byref(ms)))
        else:
            self._check(self.lib.policy_rollout(self.ptr, pid.ctypes.data, sch.ctypes.data, sch.shape[0], int(nsteps),
                                                ep.ctypes.data, su.ctypes.data, C.byref(ms)))
        return ep, su, ms.value

    # ---- cross-rank bookkeeping gather (RCCL inside the library) ----
    def comm_unique_id(self):
        """mw_comm_unique_id -> 128 bytes (an ncclUniqueId); create on rank 0, hand the same bytes to every rank"""
        buf = np.zeros(128, dtype=np.uint8)
        if self.lib.comm_unique_id(buf.ctypes.data) != 0:
            raise RuntimeError("mwgpu: cannot create a communicator id (is RCCL loadable?)")
        return buf

    def comm_info(self):
        """mw_comm_info: what the collective library itself reports for this context's communicator"""
        out = np.zeros(4, dtype=np.int32)
        self._check(self.lib.comm_info(self.ptr, out.ctypes.data))
        return {"comm_count": int(out[0]), "comm_rank": int(out[1]), "rccl": bool(out[2]), "device": int(out[3])}

    def comm_init(self, unique_id, rank, world_size):
        uid = None if unique_id is None else np.ascontiguousarray(unique_id, dtype=np.uint8)
        self._check(self.lib.comm_init(self.ptr, None if uid is None else uid.ctypes.data, int(rank), int(world_size)))
        self.world_size = int(world_size)

    def gather_bookkeeping(self):
        """mw_gather_bookkeeping -> structured array [world, N] of the LAST step's records on every rank"""
        out = np.zeros((getattr(self, "world_size", 1), self.N), dtype=BOOKKEEPING_DTYPE)
        self._check(self.lib.gather_bookkeeping(self.ptr, out.ctypes.data, 0))
        return out

    def step_resident_fused(self, nsteps, steps_per_launch):
        """mw_step_resident_fused: nsteps steps on the uploaded actions, steps_per_launch of them per kernel launch -> kernel ms"""
        ms = C.c_float(0)
        self._check(self.lib.step_resident_fused(self.ptr, int(nsteps), self._resident_steps, int(steps_per_launch), C.byref(ms)))
        return ms.value

    def step_resident_gather(self, nsteps):
        ms = C.c_float(0)
        self._check(self.lib.step_resident_gather(self.ptr, nsteps, self._resident_steps, C.byref(ms)))
        return ms.value

    def set_episode_phase(self, elapsed):
        e = np.ascontiguousarray(elapsed, dtype=np.int32)
        assert e.shape == (self.N,)
        self._check(self.lib.set_episode_phase(self.ptr, e.ctypes.data))

    def set_goal_schedule(self, schedule):
        """mw_set_goal_schedule: [K, N] goal indices for the auto-resets inside step_resident / step_resident_gather (None clears)"""
        if schedule is None:
            self._check(self.lib.set_goal_schedule(self.ptr, None, 0))
            return
        g = np.ascontiguousarray(schedule, dtype=np.int32)
        assert g.ndim == 2 and g.shape[1] == self.N and g.shape[0] >= 1
        self._check(self.lib.set_goal_schedule(self.ptr, g.ctypes.data, g.shape[0]))

    def goal_schedule_pos(self):
        """mw_goal_schedule_pos -> [N] auto-resets of every env since set_goal_schedule"""
        out = np.zeros(self.N, dtype=np.int32)
        self._check(self.lib.goal_schedule_pos(self.ptr, out.ctypes.data))
        return out

    def status(self, clear=False):
        """mw_status -> dict(flags, row_overflow_steps, contact_overflow_steps, unstable_steps, diverged_steps, solver_stalls); flags: 1 / 2 capacity exceeded, 4 non-finite state, 8 sub-lane divergence canary (include/mwgpu.h)"""
        out = np.zeros(STATUS_WORDS, dtype=np.int32)
        self._check(self.lib.status(self.ptr, out.ctypes.data, STATUS_WORDS, int(bool(clear))))
        return dict(flags=int(out[0]), row_overflow_steps=int(out[1]), contact_overflow_steps=int(out[2]), unstable_steps=int(out[3]),
                    diverged_steps=int(out[4]), solver_stalls=int(out[5]))

    def launch_times(self, cap=8192):
        """mw_launch_times -> float32 [n] ms of every launch (step) of the last step_resident / step_resident_gather call"""
        out = np.zeros(cap, dtype=np.float32)
        n = self._check(self.lib.launch_times(self.ptr, out.ctypes.data, cap))
        return out[:n]

    def set_option(self, name, value):
        """mw_set_option: run-time options of the context ("split_collision")"""
        self._check(self.lib.set_option(self.ptr, name.encode(), float(value)))

    def upload_actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        assert a.ndim == 3 and a.shape[1:] == (self.N, 4)
        self._check(self.lib.upload_actions(self.ptr, a.ctypes.data, a.shape[0]))
        self._resident_steps = a.shape[0]

    def step_resident(self, nsteps):
        ms = C.c_float(0)
        self._check(self.lib.step_resident(self.ptr, nsteps, self._resident_steps, C.byref(ms)))
        return ms.value

    def read(self, env, what, n=None):
        if n is None:
            n = self._check(self.lib.column_size(self.ptr, env, what.encode()))
        out = np.zeros(n)
        self._check(self.lib.read(self.ptr, env, what.encode(), out.ctypes.data, n))
        return out

    def read_int(self, env, what, n=None):
        if n is None:
            n = self._check(self.lib.column_size(self.ptr, env, what.encode()))
        out = np.zeros(n, dtype=np.int32)
        self._check(self.lib.read_int(self.ptr, env, what.encode(), out.ctypes.data, n))
        return out

    def write(self, env, what, values):
        a = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self.lib.write(self.ptr, env, what.encode(), a.ctypes.data, a.size))

    def get_state(self):
        """mw_get_state -> list of N float64 arrays (the rows of read(e, "state")), one device round trip per model group"""
        sizes = np.array([self._check(self.lib.column_size(self.ptr, e, b"state")) for e in range(self.N)])
        buf = np.zeros((self.N, int(sizes.max())))
        self._check(self.lib.get_state(self.ptr, buf.ctypes.data, buf.shape[1]))
        return [buf[e, :sizes[e]].copy() for e in range(self.N)]

    def set_state(self, rows):
        """mw_set_state: rows[e] = the state vector of env e as get_state returned it"""
        stride = max(len(r) for r in rows)
        buf = np.zeros((self.N, stride))
        for e, r in enumerate(rows):
            buf[e, :len(r)] = r
        self._check(self.lib.set_state(self.ptr, buf.ctypes.data, stride))

    def debug(self, what, n=0):
        code = what if isinstance(what, int) else {"forward": 0, "substeps": 1, "reset_data": 2, "kinematics": 3}[what]
        self._check(self.lib.debug(self.ptr, code, n))

    def close(self):
        if self.ptr:
            self.lib.destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
