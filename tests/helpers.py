import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WELD = [0, 0, 0, 0, 0, 0, -1, 0, 0, 0, 5.0]


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def oracle_for(model_name):
    from metaworld_amd import tasks as T
    from oracle.mjlite import OracleData, OracleModel
    om = OracleModel(T.compiled_model(model_name))
    om.view("eq_data")[:] = WELD
    return om, OracleData(om)


def make_env(lib, task="reach-v3", n=4, precision="fp64", **kw):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    return MetaWorldGpuVectorEnv("MT1", task, num_envs=n, seed=0, precision=precision, lib=lib, **kw)


def replay_trace(env, G, sync, steps=None):
    """returns max abs errors (obs, reward, info) and #success mismatches replaying golden trace G on env.ctx"""
    ctx = env.ctx
    E, T = G["actions"].shape[:2]
    T = steps or T
    obs = ctx.reset(G["goal_idx"]).copy()
    e_reset = np.abs(obs - G["reset_obs"]).max()
    eo = er = ei = 0.0
    es = 0
    for t in range(T):
        if sync and t > 0:
            for e in range(E):
                ctx.write(e, "qpos", G["qpos"][e, t - 1]); ctx.write(e, "qvel", G["qvel"][e, t - 1])
                ctx.write(e, "mocap", G["mocap"][e, t - 1]); ctx.write(e, "warm", G["warm"][e, t - 1])
                tk = ctx.read(e, "task"); tk[15:33] = G["obs"][e, t - 1][:18]; ctx.write(e, "task", tk)
        o, r, te, tr, su, info = ctx.step(G["actions"][:, t])
        eo = max(eo, np.abs(o - G["obs"][:, t]).max()); er = max(er, np.abs(r - G["reward"][:, t]).max())
        ei = max(ei, np.abs(info - G["info"][:, t]).max()); es += int((su != G["success"][:, t]).sum())
    return dict(reset=e_reset, obs=eo, reward=er, info=ei, success_mismatch=es)
