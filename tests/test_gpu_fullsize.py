"""Parity at the BASELINE.json sizes (MT50 @ 4096 envs and MT10 @ 10 240 envs, built exactly as bench.py builds them: the
runtime's own lanes-per-workgroup choice per scene, one-hot on, fp64 = the headline precision).  The golden traces cannot be
replayed at this size, so the full-size batch is tied to what IS pinned at small size:
  (a) one env per task, taken from the big batch, must reproduce -- to summation-order accuracy -- a small MT1 batch of the
      same (task, goal, action stream), the configuration the per-task golden-trace tests pin against the reference Python;
  (b) after ~50 steps of random actions, the oracle engine (oracle/mjl_core.c) synchronised to the device state of one env per
      task must produce the same next physics step (qpos / qvel to 1e-7 / 1e-5, identical contact and constraint-row counts);
  (c) no capacity-overflow / instability flag may be raised anywhere in the batch.
"""
import os

import numpy as np
import pytest

from metaworld_amd import tasks as T
from tests.helpers import ROOT, WELD

pytestmark = pytest.mark.gpu

CONFIGS = [("MT50", 4096), ("MT10", 10240)]


def _oracle_synced_to(ctx, e, task):
    from oracle.mjlite import OracleData, OracleModel
    mname = T.TASK_CONST[task]["model"]
    pk, roles, reloc = T.packed_model(mname, reloc_bodies=T.model_key(task)[1])
    cm = T.compiled_model(mname)
    om = OracleModel(cm)
    om.view("eq_data")[:] = WELD
    rel = ctx.read(e, "reloc")
    bp = om.view("body_pos").reshape(-1, 3)
    for slot, body_name in enumerate(reloc):          # the per-env `model.body(X).pos` overrides (the device model renumbers bodies)
        bp[cm.names["body"][body_name]] = rel[3 * slot:3 * slot + 3]
    d = OracleData(om)
    d.qpos[:] = ctx.read(e, "qpos"); d.qvel[:] = ctx.read(e, "qvel"); d.qacc_warmstart[:] = ctx.read(e, "warm")
    d.mocap_pos[:] = ctx.read(e, "mocap"); d.mocap_quat[:] = [1, 0, 1, 0]; d.ctrl[:] = ctx.read(e, "ctrl")
    return om, d


def _oracle_response(ctx, e, task, state, eps=1e-12, trials=3):
    """max |change| of the oracle's own qpos / qvel, 5 substeps after the synchronised state `state` (columns read BEFORE the device
    stepped), when qpos is perturbed by eps: the conditioning of the reference computation at that state"""
    class _Cols:          # _oracle_synced_to reads the state through ctx.read
        def read(self, _e, col):
            return state[col]
    out = []
    rng = np.random.default_rng(0)
    for k in range(trials + 1):
        om, d = _oracle_synced_to(_Cols(), e, task)
        if k:
            d.qpos[:] += eps * rng.standard_normal(len(d.qpos))
        d.step(5)
        out.append((d.qpos.copy(), d.qvel.copy()))
    return (max(np.abs(q - out[0][0]).max() for q, _ in out[1:]), max(np.abs(v - out[0][1]).max() for _, v in out[1:]))


@pytest.mark.parametrize("bench,n", CONFIGS)
def test_fullsize_batch_matches_small_batches_and_oracle(gpulib, bench, n):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    from tests.test_tasks_parity import TOL
    big = MetaWorldGpuVectorEnv(bench, num_envs=n, seed=42, use_one_hot=True, precision="fp64", lib=gpulib)
    obs, _ = big.reset()
    names = big.task_list
    first = {name: big.env_task_names.index(name) for name in names}          # one env per task
    acts = np.random.default_rng(5).uniform(-1, 1, (60, n, 4)).astype(np.float32)
    NS = 12
    rec = []
    for t in range(NS):
        o, r, te, tr, infos = big.step(acts[t])
        rec.append((o.copy(), r.copy(), infos["success"].copy(), np.stack([infos[k] for k in
                    ("near_object", "grasp_success", "grasp_reward", "in_place_reward", "obj_to_target", "unscaled_reward")], 1)))
    assert big.status()["flags"] == 0
    # (a) the same (task, goal, actions) in a small MT1 batch
    worst = {}
    for name, e in first.items():
        # the same task WITH THE BENCHMARK'S goal table (MT50's goals of a task are not MT1's: one RNG stream over all classes)
        small = MetaWorldGpuVectorEnv(bench, num_envs=2, seed=42, precision="fp64", lib=gpulib, task_names=[name])
        o0 = small.ctx.reset(np.array([big._cur_goal[e]] * 2, dtype=np.int32)).copy()
        assert np.abs(o0[0] - obs[e, :39]).max() < 1e-6, name            # (the big batch returns float32 one-hot observations)
        eo = er = ei = 0.0
        for t in range(NS):
            o, r, te, tr, su, info = small.ctx.step(np.stack([acts[t, e]] * 2))
            eo = max(eo, np.abs(o[0] - rec[t][0][e, :39]).max()); er = max(er, abs(r[0] - rec[t][1][e]))
            ei = max(ei, np.abs(info[0] - rec[t][3][e]).max())
            assert su[0] == rec[t][2][e], (name, t)
        small.close()
        worst[name] = (eo, er, ei)
        # float32 observation dtype of the one-hot space (metaworld/wrappers.py:27-29) bounds the comparison at ~1e-7.
        # The big batch and the 2-env batch run this task with different numbers of cooperating sub-lanes (different
        # summation order, ~1e-16); the three tasks whose states amplify 1e-12 to 1e-5 ... 1e-3 in ONE step
        # (tests/test_ill_conditioning.py) carry that through 12 open-loop steps (measured: peg-unplug 1.8e-4 / 1.7e-3)
        lim = (2e-3, 2e-2, 2e-2) if name in TOL else (2e-6, 1e-7, 1e-5)
        assert eo < lim[0] and er < lim[1] and ei < lim[2], (name, worst[name])
    # (b) oracle one substep-batch from the synchronised device state, deep into contact-rich motion
    for t in range(NS, 60):
        big.ctx.step(acts[t], big._next_goal)
    assert big.status()["flags"] == 0
    synced = {name: _oracle_synced_to(big.ctx, e, name) for name, e in first.items()}
    big.ctx.debug("substeps", 5)
    for name, e in first.items():
        om, d = synced[name]
        d.step(5)
        ic = big.ctx.read_int(e, "icount")
        assert ic[0] == d.ncon and ic[1] == d.nefc, (name, ic[:2], d.ncon, d.nefc)
        # two implementations (dense AoS C vs 16 cooperating sub-lanes with fused multiply-adds) of a contact-rich state: 5 substeps
        # apart they agree to ~1e-8 (the contact-free test_gpu_physics_matches_oracle holds 1e-9 over 150 substeps)
        assert np.abs(big.ctx.read(e, "qpos") - d.qpos).max() < 1e-7, name
        assert np.abs(big.ctx.read(e, "qvel") - d.qvel).max() < 1e-5, name
    big.close()


@pytest.mark.parametrize("bench_name,n", CONFIGS)
def test_bench_states_match_the_oracle(gpulib, bench_name, n):
    """parity ON THE STATES bench.py TIMES.  The batch is built and pre-rolled exactly like bench.py::prepare
    (staggered episode phases, one untimed 500-step horizon of random actions + warm-up: late-episode states with mesh
    contacts, where the narrow phase dominates); for 4 envs of every task, spread over the episode phases, the oracle engine is
    synchronised to the device state and both advance 5 substeps: same contact / constraint-row counts, qpos 1e-7, qvel 1e-5.
    (The obs / reward / success layer on these states is checked against the reference's own Python by
    tests/test_bench_state_parity.py from the recording tools/dump_bench_states.py makes on the GPU.)"""
    from types import SimpleNamespace
    import bench
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    from tools.dump_bench_states import pick_envs
    from tests.test_tasks_parity import TOL
    env = MetaWorldGpuVectorEnv(bench_name, num_envs=n, seed=42, use_one_hot=True, precision="fp64", lib=gpulib)
    bench.prepare(env, SimpleNamespace(no_stagger=False, warmup=20, allow_status=False, fixed_goals=False), 0)      # raises on any status flag
    elapsed = (np.arange(n, dtype=np.int64) * 7919 + bench.HORIZON + 20) % bench.HORIZON          # TimeLimit phase of every env now
    chosen = pick_envs(env, elapsed)
    assert len(chosen) == 4 * len(env.task_list)
    for e in chosen[::17]:
        assert int(env.ctx.read(e, "task")[3]) == elapsed[e], e
    synced = [(e, env.env_task_names[e], _oracle_synced_to(env.ctx, e, env.env_task_names[e])) for e in chosen]
    before = {e: {c: env.ctx.read(e, c) for c in ("qpos", "qvel", "warm", "mocap", "ctrl", "reloc")} for e in chosen}
    env.ctx.debug("substeps", 5)
    bad, errs, relaxed = [], [], []
    for e, name, (om, d) in synced:
        d.step(5)
        ic = env.ctx.read_int(e, "icount")
        eq, ev = np.abs(env.ctx.read(e, "qpos") - d.qpos).max(), np.abs(env.ctx.read(e, "qvel") - d.qvel).max()
        ncon_dev, ncon_orc = int(ic[0]), d.ncon
        if ncon_dev != ncon_orc:
            # two surfaces that touch EXACTLY (the faucet's coaxial cylinders, dist = 0 +- 1e-17, no constraint row either way) are
            # listed or not depending on the last rounding: compare the contacts that are not such ties
            dist_dev = env.ctx.read(e, "con").reshape(-1, 26)[:ncon_dev, 0]
            dist_orc = np.array([c["dist"] for c in d.contacts()])
            ncon_dev, ncon_orc = int((np.abs(dist_dev) > 1e-12).sum()), int((np.abs(dist_orc) > 1e-12).sum())
        # one env-step (5 substeps) from a synchronised state: the reference tolerance of the observation, 1e-5, is the hard limit
        # (velocities 1e-3); typical agreement is 1e-9 (asserted below for 90 % of the sample); the two tasks whose states amplify
        # 1e-12 to 1e-5 ... 1e-3 in ONE step (tests/test_ill_conditioning.py) get the limits their trace tests use
        lq, lv = (1e-3, 1e-1) if name in TOL else (1e-5, 1e-3)
        errs.append(eq)
        if ncon_dev == ncon_orc and ic[1] == d.nefc and not (eq < lq and ev < lv):
            # Over the limit: is THIS STATE ill-conditioned in the reference computation itself?  (Which states the sample holds
            # depends on the last bit of 520 chaotic steps, so any change of the device code draws new ones; a contact sitting at its
            # activation margin turns 1e-12 into 1e-6 ... 1e-5 within one env-step, tests/test_ill_conditioning.py.)  The oracle is
            # re-run from the synchronised state with qpos perturbed by 1e-12: where its OWN answer moves by more than a tenth of the
            # limit no implementation can hold the limit, and the device may deviate by 10x that response (at most 1e-3 / 1e-1).
            rq, rv = _oracle_response(env.ctx, e, name, before[e])
            relaxed.append((name, e, float(eq), float(rq)))
            lq, lv = min(max(lq, 10 * rq), 1e-3), min(max(lv, 10 * rv), 1e-1)
        if not (ncon_dev == ncon_orc and ic[1] == d.nefc and eq < lq and ev < lv):
            bad.append((name, e, int(elapsed[e]), (int(ic[0]), d.ncon), (int(ic[1]), d.nefc), float(eq), float(ev)))
    # the states whose limit followed the oracle's own conditioning are REPORTED even when the test passes (a self-adjusting
    # tolerance must not be silent: VERDICT r3): stdout (pytest -rA) and gpurun_out/bench_states_relaxed.txt, committed under profiles/
    report = [f"{len(relaxed)} of {len(synced)} sampled states got a relaxed limit (task, env, |dq| device vs oracle, oracle's own response to 1e-12)"]
    report += [f"  {n:28s} env {e:5d}  dq {q:.3e}  oracle response {r:.3e}" for n, e, q, r in relaxed]
    report.append("error quantiles 0.5 / 0.9 / 0.99 / 1.0: " + " ".join(f"{x:.2e}" for x in np.quantile(errs, [0.5, 0.9, 0.99, 1.0])))
    print("\n".join(report))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_states_relaxed.txt"), "w") as f:
            f.write("\n".join(report) + "\n")
    except OSError:
        pass
    assert not bad, bad
    assert len(relaxed) <= max(2, len(synced) // 50), relaxed          # ill-conditioned states are the exception (<= 2 % of the sample)
    assert np.quantile(errs, 0.9) < 1e-7, np.quantile(errs, [0.5, 0.9, 0.99, 1.0])
    assert env.status()["flags"] == 0
    env.close()


def test_fullsize_gather_and_status_through_the_abi(gpulib, monkeypatch):
    """MT50 @ 4096: the per-step bookkeeping record of the resident loop (world size 1: no communicator needed) and the status word;
    then the same loop over a REAL one-rank RCCL communicator (ncclCommInitRank + ncclAllGather on the side stream)"""
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    env = MetaWorldGpuVectorEnv("MT50", num_envs=4096, seed=1, use_one_hot=True, precision="fp32", lib=gpulib, max_episode_steps=25)
    env.reset()
    env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (8, 4096, 4)).astype(np.float32))
    env.ctx.step_resident_gather(25)
    book = env.ctx.gather_bookkeeping()
    assert book.shape == (1, 4096) and (book["done"] == 1).all() and (book["episode_length"] == 25).all()
    ids = np.array([T.TASK_CONST[n]["id"] for n in env.env_task_names])
    assert (book["task_id"][0] == ids).all() and np.isfinite(book["episode_return"]).all() and (book["episode_return"] >= 0).all()
    assert env.status()["flags"] == 0
    assert env.ctx.comm_info()["rccl"] is False
    # world size 1 through mw_comm_init: still no communicator (device copy), same records
    env.ctx.comm_init(env.ctx.comm_unique_id(), 0, 1)
    env.ctx.step_resident_gather(3)
    b2 = env.ctx.gather_bookkeeping()
    assert (b2["episode_length"] == 3).all() and (b2["done"] == 0).all()
    # a REAL RCCL communicator of one rank (MW_COMM_FORCE_RCCL): ncclCommInitRank, then ncclAllGather per step on the side stream
    # with the gather-done back-edge, checked against the records of the plain loop continued from the same state
    monkeypatch.setenv("MW_COMM_FORCE_RCCL", "1")
    env.ctx.comm_init(env.ctx.comm_unique_id(), 0, 1)
    info = env.ctx.comm_info()
    assert info["rccl"] is True and info["comm_count"] == 1 and info["comm_rank"] == 0, info
    env.ctx.step_resident_gather(5)
    b3 = env.ctx.gather_bookkeeping()
    assert b3.shape == (1, 4096) and (b3["episode_length"] == 8).all() and (b3["done"] == 0).all() and (b3["task_id"][0] == ids).all()
    env.ctx.step_resident_gather(17)                       # ... to the end of the 25-step episode: every env done in the gathered record
    b4 = env.ctx.gather_bookkeeping()
    assert (b4["done"] == 1).all() and (b4["episode_length"] == 25).all() and np.isfinite(b4["episode_return"]).all()
    assert env.status()["flags"] == 0
    env.close()


def test_staggered_episode_phases(gpulib):
    """mw_set_episode_phase: env i truncates after max_episode_steps - elapsed[i] steps (bench.py's whole-episode sampling)"""
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=64, seed=0, precision="fp32", lib=gpulib, max_episode_steps=16)
    env.reset()
    env.ctx.set_episode_phase(np.arange(64, dtype=np.int32) % 16)
    a = np.zeros((64, 4), dtype=np.float32)
    for t in range(1, 17):
        o, r, te, tr, su, info = env.ctx.step(a)
        assert (tr.astype(bool) == ((np.arange(64) % 16) == (16 - t) % 16)).all(), t
    env.close()
