"""N > 1 path on CPU: two ranks (gloo), each owning its own env shard (host build of the lane programs), no data-path
collective; the per-step bookkeeping record is all-gathered (the one real exchange of this path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, lib_path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metaworld_amd import native
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv, gather_bookkeeping
    lib = native.load("mwh_", lib_path)
    env = MetaWorldGpuVectorEnv("MT10", num_envs=6, seed=7, use_one_hot=True, precision="fp32", lib=lib, rank=rank,
                                world_size=world, max_episode_steps=3, task_names=["reach-v3", "push-v3", "window-open-v3"])
    env.reset()
    rng = np.random.default_rng(100 + rank)          # different actions per shard
    for _ in range(3):
        obs, rew, term, trunc, infos = env.step(rng.uniform(-1, 1, (6, 4)).astype(np.float32))
    book = gather_bookkeeping(env.bookkeeping())
    np.save(os.path.join(out_dir, f"book{rank}.npy"), book)
    np.save(os.path.join(out_dir, f"rew{rank}.npy"), rew)
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
    assert b0.shape == (2, 6, 5) and np.array_equal(b0, b1)          # every rank sees the whole node's record
    assert (b0[:, :, 0] == 1).all()                                   # all envs truncated at step 3 on both shards
    assert set(b0[0, :, 2]) == {43.0, 40.0, 48.0}                     # MT50 task ids of the three tasks
    r0, r1 = np.load(tmp_path / "rew0.npy"), np.load(tmp_path / "rew1.npy")
    assert not np.allclose(r0, r1)                                    # shards are independent (different actions)
    assert np.allclose(b0[0, :, 4], 3) and np.allclose(b0[1, :, 4], 3)
