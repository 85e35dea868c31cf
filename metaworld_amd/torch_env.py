"""MetaWorldTorchVectorEnv -- the same VectorEnv with the batch left in HBM.

For a learner whose policy lives on the same GPU: `step(actions)` takes a CUDA tensor and returns CUDA tensors; actions,
observations, rewards and flags never visit the host (C ABI: mw_step_device / mw_reset_device in include/mwgpu.h, SURVEY.md 8b
"outputs_on_device").  torch is plumbing here (device memory + the caller's stream), the step is the same single kernel launch.
Only the N-byte `done` row is read back each step, because task selection (RandomTaskSelectWrapper semantics, wrappers.py:91-142)
stays on the host.

Differences from the numpy class: arrays are tensors; `infos["final_obs"]` is one [N, D] tensor whose rows are valid where
`infos["_final_obs"]`; the recurrent-info / normalisation wrappers are not offered (apply them on the device in the learner).
"""
from __future__ import annotations

import time

import numpy as np

from .native import MwDeviceOut
from .vector_env import INFO_KEYS, MetaWorldGpuVectorEnv


class MetaWorldTorchVectorEnv(MetaWorldGpuVectorEnv):
    def __init__(self, *args, device=None, **kwargs):
        import torch
        super().__init__(*args, **kwargs)
        if self.recurrent_info_in_obs or self.normalize_observations or self.reward_normalization_method:
            raise NotImplementedError("wrappers that post-process obs / reward on the host are not part of the device-resident env")
        self._torch = torch
        # "cpu" only with the host test harness, whose "device" memory is host memory
        self.device = torch.device(device if device is not None else f"cuda:{kwargs.get('device_id', 0)}")
        N, D = self.num_envs, self.ctx.D
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)
        self._obs, self._final = z((N, D), torch.float64), z((N, D), torch.float64)
        self._reward, self._epret = z((N,), torch.float64), z((N,), torch.float64)
        self._flags, self._info, self._eplen = z((4, N), torch.uint8), z((N, 6), torch.float32), z((N,), torch.int32)
        self._goal = z((N,), torch.int32)
        self._out = MwDeviceOut(obs=self._obs.data_ptr(), reward=self._reward.data_ptr(), flags=self._flags.data_ptr(),
                                info=self._info.data_ptr(), final_obs=self._final.data_ptr(),
                                episode_return=self._epret.data_ptr(), episode_length=self._eplen.data_ptr())
        self._tdtype = torch.float32 if self.obs_dtype == np.float32 else torch.float64

    def _fence(self):
        """the context launches on its own stream: the caller's pending writes (actions, goal upload) must have landed"""
        if self.device.type == "cuda":
            self._torch.cuda.current_stream(self.device).synchronize()

    def _upload_goals(self, which):
        self._goal.copy_(self._torch.from_numpy(np.ascontiguousarray(which, dtype=np.int32)))

    def _fresh(self, t):
        return t.to(self._tdtype) if t.dtype != self._tdtype else t.clone()

    def reset(self, *, seed=None, options=None):
        self._begin_episodes(np.ones(self.num_envs, dtype=bool))
        self._upload_goals(self._cur_goal)
        self._fence()
        self.ctx.reset_device(self._goal.data_ptr(), None, self._obs.data_ptr())
        self._upload_goals(self._next_goal)
        self._episode_start[:] = time.perf_counter()
        return self._fresh(self._obs), {}

    def step(self, actions):
        torch = self._torch
        a = torch.as_tensor(actions, device=self.device).to(torch.float32).contiguous()
        assert tuple(a.shape) == (self.num_envs, 4)
        self._fence()
        self.ctx.step_device(a.data_ptr(), self._goal.data_ptr(), self._out)
        term, trunc, succ, done = self._flags[0].bool(), self._flags[1].bool(), self._flags[2].to(torch.float64), self._flags[3].bool()
        ones = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        infos = {"success": succ, "_success": ones}
        info = self._info.to(torch.float64)
        for k, key in enumerate(INFO_KEYS):
            infos[key], infos["_" + key] = info[:, k], ones
        done_h = done.cpu().numpy()
        if done_h.any():
            now = time.perf_counter()
            zero = torch.zeros((), dtype=torch.float64, device=self.device)
            fi = {k: torch.where(done, v, zero) for k, v in infos.items() if not k.startswith("_")}
            for k in list(fi):
                fi["_" + k] = done
            t = torch.as_tensor(np.where(done_h, np.round(now - self._episode_start, 6), 0.0), device=self.device)
            fi["episode"] = {"r": torch.where(done, self._epret, zero), "l": torch.where(done, self._eplen, torch.zeros_like(self._eplen)),
                             "t": t, "_r": done, "_l": done, "_t": done}
            fi["_episode"] = done
            infos["final_info"], infos["_final_info"] = fi, done
            infos["final_obs"], infos["_final_obs"] = self._fresh(self._final), done
            self._episode_start[done_h] = now
            self._begin_episodes(done_h)
            self._upload_goals(self._next_goal)
        return self._fresh(self._obs), self._reward.clone(), term, trunc, infos

    def call(self, name, *args, **kwargs):
        out = super().call(name, *args, **kwargs)          # host-side bookkeeping (and, for sample_tasks, a host-path reset)
        self._upload_goals(self._next_goal)
        return out
