"""Device-resident boundary (mw_step_device / mw_reset_device): same results as the host-buffer boundary."""
import numpy as np
import pytest

from metaworld_amd.vector_env import INFO_KEYS, MetaWorldGpuVectorEnv


def _twin_run(lib, device, steps=40):
    import torch
    from metaworld_amd.torch_env import MetaWorldTorchVectorEnv
    kw = dict(num_envs=24, seed=5, use_one_hot=True, precision="fp32", lib=lib, max_episode_steps=9)
    ref = MetaWorldGpuVectorEnv("MT10", **kw)
    env = MetaWorldTorchVectorEnv("MT10", device=device, **kw)
    o1, _ = ref.reset()
    o2, _ = env.reset()
    assert o2.device.type == torch.device(device).type and o2.dtype == torch.float32
    assert np.array_equal(o1, o2.cpu().numpy())
    rng = np.random.default_rng(0)
    ndone = 0
    for t in range(steps):
        a = rng.uniform(-1, 1, (24, 4)).astype(np.float32)
        o1, r1, te1, tr1, i1 = ref.step(a)
        o2, r2, te2, tr2, i2 = env.step(torch.from_numpy(a).to(device))
        assert np.array_equal(o1, o2.cpu().numpy()) and np.array_equal(r1, r2.cpu().numpy())
        assert np.array_equal(te1, te2.cpu().numpy()) and np.array_equal(tr1, tr2.cpu().numpy())
        for k in ["success"] + INFO_KEYS:
            assert np.array_equal(i1[k], i2[k].cpu().numpy())
        assert ("final_info" in i1) == ("final_info" in i2)
        if "final_info" in i1:
            d = i1["_final_info"]
            ndone += int(d.sum())
            assert np.array_equal(d, i2["_final_info"].cpu().numpy())
            fo = i2["final_obs"].cpu().numpy()
            for e in np.flatnonzero(d):
                assert np.array_equal(i1["final_obs"][e], fo[e])
            for k in ("r", "l"):
                assert np.array_equal(i1["final_info"]["episode"][k], i2["final_info"]["episode"][k].cpu().numpy())
            assert np.array_equal(i1["final_info"]["success"], i2["final_info"]["success"].cpu().numpy())
        assert np.array_equal(np.stack(ref.get_attr("_last_rand_vec")), np.stack(env.get_attr("_last_rand_vec")))
    assert ndone >= 24 * (steps // 9)
    ref.close(); env.close()


def test_device_boundary_on_host_harness(hostsim):
    _twin_run(hostsim, "cpu")


@pytest.mark.gpu
def test_device_boundary_on_gpu(gpulib):
    _twin_run(gpulib, "cuda:0")
