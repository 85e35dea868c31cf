"""MI355X-native batched Meta-World: the `SawyerXYZEnv.step()/reset()` hot path of Farama-Foundation/Metaworld as HIP kernels
behind the reference's VectorEnv surface.  Importing the package loads nothing native; `libmwgpu.so` is opened
when the first environment is built and there is no CPU fallback."""
from .make import (make_custom_ml_envs, make_custom_mt_envs, make_goal_hidden, make_goal_observable, make_ml_envs, make_ml_envs_test, make_ml_envs_train,  # noqa: F401
                   make_mt_envs, register_mw_envs)
from .vector_env import MetaWorldGpuVectorEnv  # noqa: F401

__all__ = ["MetaWorldGpuVectorEnv", "make_mt_envs", "make_ml_envs", "make_ml_envs_train", "make_ml_envs_test",
           "make_custom_mt_envs", "make_custom_ml_envs", "make_goal_observable", "make_goal_hidden", "register_mw_envs"]
