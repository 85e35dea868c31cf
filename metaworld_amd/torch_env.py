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
        self._ones = torch.ones(N, dtype=torch.bool, device=self.device)
        self._zero, self._zero_i = z((), torch.float64), z((), torch.int32)
        # staging buffer of the look-ahead goals: pinned, so that the upload is an asynchronous copy on the caller's stream
        self._goal_host = torch.zeros(N, dtype=torch.int32, pin_memory=self.device.type == "cuda")
        self._out = MwDeviceOut(obs=self._obs.data_ptr(), reward=self._reward.data_ptr(), flags=self._flags.data_ptr(),
                                info=self._info.data_ptr(), final_obs=self._final.data_ptr(),
                                episode_return=self._epret.data_ptr(), episode_length=self._eplen.data_ptr())
        self._tdtype = torch.float32 if self.obs_dtype == np.float32 else torch.float64

    def _fence(self):
        """the context launches on its own stream: the caller's pending writes (actions, goal upload) must have landed"""
        if self.device.type == "cuda":
            self._torch.cuda.current_stream(self.device).synchronize()

    def _upload_goals(self, which):
        # (the kernel that reads self._goal is ordered behind this copy: mw_step_device_on waits for the caller's stream)
        self._goal_host.numpy()[:] = which
        self._goal.copy_(self._goal_host, non_blocking=True)

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
        """Stream-ordered (mw_step_device_on): the step kernel is ordered behind the caller's pending work and in front of
        everything queued afterwards by EVENTS; the host only waits for the pinned `done` row (mw_wait_done), and it queues the
        tensor ops below while the kernel is still running."""
        torch = self._torch
        a = torch.as_tensor(actions, device=self.device).to(torch.float32).contiguous()
        assert tuple(a.shape) == (self.num_envs, 4)
        if self.device.type == "cuda":
            self.ctx.step_device_on(a.data_ptr(), self._goal.data_ptr(), self._out, torch.cuda.current_stream(self.device).cuda_stream)
        else:          # host test harness: "device" memory is host memory, the call is synchronous
            self.ctx.step_device_on(a.data_ptr(), self._goal.data_ptr(), self._out, None)
        # ---- queued behind the kernel on the caller's stream (the host does not wait here) ----
        fl = self._flags.bool()                                   # [4, N]: terminated, truncated, success, done
        term, trunc, done = fl[0], fl[1], fl[3]
        vals = torch.cat([self._flags[2:3].T, self._info], dim=1).to(torch.float64)      # [N, 7]: success ++ the six info floats
        ones = self._ones
        infos = {"success": vals[:, 0], "_success": ones}
        for k, key in enumerate(INFO_KEYS):
            infos[key], infos["_" + key] = vals[:, 1 + k], ones
        obs, rew = self._fresh(self._obs), self._reward.clone()
        fvals = torch.where(done.unsqueeze(1), vals, self._zero)   # the same seven columns, zero where the env goes on (final_info)
        epr, epl, fobs = torch.where(done, self._epret, self._zero), torch.where(done, self._eplen, self._zero_i), self._fresh(self._final)
        # ---- the one host wait of the step: the pinned copy of the done row ----
        done_h = self.ctx.wait_done().astype(bool)
        if done_h.any():
            now = time.perf_counter()
            fi = {"success": fvals[:, 0], "_success": done}
            for k, key in enumerate(INFO_KEYS):
                fi[key], fi["_" + key] = fvals[:, 1 + k], done
            t = torch.as_tensor(np.where(done_h, np.round(now - self._episode_start, 6), 0.0), device=self.device)
            fi["episode"] = {"r": epr, "l": epl, "t": t, "_r": done, "_l": done, "_t": done}
            fi["_episode"] = done
            infos["final_info"], infos["_final_info"] = fi, done
            infos["final_obs"], infos["_final_obs"] = fobs, done
            self._episode_start[done_h] = now
            self._begin_episodes(done_h)
            self._upload_goals(self._next_goal)
        return obs, rew, term, trunc, infos

    def call(self, name, *args, **kwargs):
        out = super().call(name, *args, **kwargs)          # host-side bookkeeping (and, for sample_tasks, a host-path reset)
        self._upload_goals(self._next_goal)
        return out
