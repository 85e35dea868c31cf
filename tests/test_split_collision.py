"""Split collision (mw_set_option("split_collision", 1), csrc/mw_split.inl): the narrow phase of every dynamics evaluation runs as
batch-wide kernels over (environment, candidate pair) work items between the lane kernels instead of inside the fused step
kernel.  It moves the SAME numbers: contacts in the same order, identical state and outputs.

CPU half: host build of the lane programs, fused vs split bit for bit (observations, rewards, flags, state, contact records) over
random-action rollouts that include auto-resets and the lazy final dynamics of the touching_object tasks.
GPU half (-m gpu): the same comparison through libmwgpu.so (within 1e-9: the two code paths may place fused multiply-adds
differently) + the status word."""
import numpy as np
import pytest

from metaworld_amd.vector_env import MetaWorldGpuVectorEnv

# contact-rich scenes incl. tasks whose reward calls touching_object (push, pick-place, hammer) and shared scenes
TASKS = ["push-v3", "hammer-v3", "door-unlock-v3", "stick-pull-v3", "box-close-v3", "coffee-button-v3", "pick-place-v3", "plate-slide-back-v3"]


def _rollout(lib, split, precision, steps, nenv=16, full_forward=False):
    env = MetaWorldGpuVectorEnv("custom-mt", envs_list=TASKS, num_envs=nenv, seed=5, precision=precision, lib=lib, use_one_hot=True,
                                max_episode_steps=37, total_tasks_per_cls=3, full_forward=full_forward)
    env.ctx.set_option("split_collision", 1 if split else 0)
    env.reset()
    rng = np.random.default_rng(2)
    out = []
    for s in range(steps):
        a = rng.uniform(-1, 1, (nenv, 4)).astype(np.float32)
        if s > 25:
            a[:, 2] = -abs(a[:, 2])          # push the hand down onto the table / the objects: many contacts
        o, r, te, tr, info = env.step(a)
        out.append((o.copy(), r.copy(), te.copy(), tr.copy(), info["success"].copy(), info["grasp_success"].copy()))
    state = []
    for e in range(nenv):
        ncon = int(env.ctx.read_int(e, "icount")[0])
        state.append((env.ctx.read(e, "qpos"), env.ctx.read(e, "qvel"), env.ctx.read(e, "warm"), ncon,
                      env.ctx.read(e, "con")[:26 * ncon], env.ctx.read_int(e, "icon")[:4 * ncon]))
    st = env.status()
    env.close()
    return out, state, st


def _compare(a, b, tol):
    (oa, sa, fa), (ob, sb, fb) = a, b
    assert fa["flags"] == 0 and fb["flags"] == 0, (fa, fb)
    for t, (x, y) in enumerate(zip(oa, ob)):
        for k in range(len(x)):
            if tol == 0:
                assert np.array_equal(x[k], y[k]), (t, k, np.abs(np.asarray(x[k], float) - np.asarray(y[k], float)).max())
            elif x[k].dtype.kind == "f":
                assert np.abs(x[k] - y[k]).max() <= tol, (t, k, np.abs(x[k] - y[k]).max())
            else:
                assert np.array_equal(x[k], y[k]), (t, k)
    for e, (x, y) in enumerate(zip(sa, sb)):
        assert x[3] == y[3], (e, x[3], y[3])          # same number of contacts ...
        assert np.array_equal(x[5], y[5]), e          # ... between the same geoms in the same order
        for k in (0, 1, 2, 4):
            if tol == 0:
                assert np.array_equal(x[k], y[k]), (e, k)
            else:
                assert np.abs(x[k] - y[k]).max() <= 1e3 * tol, (e, k, np.abs(x[k] - y[k]).max())


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_split_collision_is_bit_identical_on_the_host_build(hostsim, precision):
    fused = _rollout(hostsim, False, precision, 90)
    split = _rollout(hostsim, True, precision, 90)
    assert any(s[3] > 0 for s in fused[1]), "the rollout must end with contacts"
    assert sum(int(o[3].sum() + o[2].sum()) for o in fused[0]) > 0, "the rollout must contain auto-resets"
    _compare(fused, split, 0)


def test_split_collision_with_the_eager_final_forward(hostsim):
    _compare(_rollout(hostsim, False, "fp64", 40, full_forward=True), _rollout(hostsim, True, "fp64", 40, full_forward=True), 0)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_split_collision_matches_the_fused_kernel_on_the_gpu(gpulib_split, precision):
    gpulib = gpulib_split          # (built with -DMW_SPLIT_COLLISION; the default library refuses the option)
    fused = _rollout(gpulib, False, precision, 60, nenv=64)
    split = _rollout(gpulib, True, precision, 60, nenv=64)
    # one step from identical states the two paths agree to rounding; over a chaotic rollout only while no contact has amplified it
    for t in range(5):
        for k in range(2):
            assert np.abs(fused[0][t][k] - split[0][t][k]).max() <= (1e-9 if precision == "fp64" else 1e-4), (t, k)
    assert fused[2]["flags"] == 0 and split[2]["flags"] == 0


@pytest.mark.gpu
def test_default_library_refuses_the_split_option(gpulib):
    """round 6: the split-collision kernels live in -DMW_SPLIT_COLLISION builds only; the product library accepts 0 and refuses 1 loudly"""
    env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=4, seed=0, precision="fp64", lib=gpulib)
    env.ctx.set_option("split_collision", 0)
    with pytest.raises(Exception, match="MW_SPLIT_COLLISION"):
        env.ctx.set_option("split_collision", 1)
    env.reset()
    o, r, *_ = env.step(np.zeros((4, 4), dtype=np.float32))          # the context is still usable
    assert np.isfinite(o).all() and np.isfinite(r).all()
    env.close()


@pytest.mark.gpu
def test_split_collision_one_step_from_contact_rich_states_on_the_gpu(gpulib_split):
    """the rollout comparison above only holds over the first steps (chaos), before the hand is pushed onto the objects.  Here
    the fused kernel rolls 45 steps into the contact regime, its state is copied into a second context (mw_get_state / mw_set_state),
    which takes ONE step with the split kernels while the first takes the same step fused: same contact lists (geoms, order), same
    row counts, observations / rewards / state to rounding -- the device's mid-phase compaction, class work lists, narrow_wave and
    collision_gather under real contact load."""
    nenv = 64
    def mk():
        return MetaWorldGpuVectorEnv("custom-mt", envs_list=TASKS, num_envs=nenv, seed=5, precision="fp64", lib=gpulib_split, use_one_hot=True,
                                     max_episode_steps=200, total_tasks_per_cls=3, full_forward=True)
    a, b = mk(), mk()
    a.reset(); b.reset()
    rng = np.random.default_rng(2)
    for s in range(45):
        act = rng.uniform(-1, 1, (nenv, 4)).astype(np.float32)
        if s > 25:
            act[:, 2] = -abs(act[:, 2])
        a.step(act)
    b.ctx.set_state(a.ctx.get_state())
    b.ctx.set_option("split_collision", 1)
    act = rng.uniform(-1, 1, (nenv, 4)).astype(np.float32); act[:, 2] = -abs(act[:, 2])
    oa, ra, *_ = a.step(act)
    ob, rb, *_ = b.step(act)
    assert np.abs(oa - ob).max() <= 1e-9 and np.abs(ra - rb).max() <= 1e-7, (np.abs(oa - ob).max(), np.abs(ra - rb).max())
    ncon_total = 0
    for e in range(nenv):
        ia, ib = a.ctx.read_int(e, "icount"), b.ctx.read_int(e, "icount")
        assert ia[0] == ib[0] and ia[1] == ib[1], (e, ia[:2], ib[:2])
        n = int(ia[0]); ncon_total += n
        assert np.array_equal(a.ctx.read_int(e, "icon")[:4 * n].reshape(-1, 4)[:, :3], b.ctx.read_int(e, "icon")[:4 * n].reshape(-1, 4)[:, :3]), e
        assert np.abs(a.ctx.read(e, "con")[:26 * n] - b.ctx.read(e, "con")[:26 * n]).max(initial=0) <= 1e-9, e
        assert np.abs(a.ctx.read(e, "qpos") - b.ctx.read(e, "qpos")).max() <= 1e-9, e
    assert ncon_total > nenv, ncon_total          # (the states really are in contact)
    assert a.status()["flags"] == 0 and b.status()["flags"] == 0
    a.close(); b.close()
