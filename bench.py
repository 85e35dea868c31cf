#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched Meta-World step kernel on MI355X.

One "step" = one VectorEnv.step over the rank's env batch (mocap update + 5 physics substeps + forward +
obs/reward + wrappers + SAME_STEP auto-reset), actions resident in HBM, outputs left in HBM.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 needs one process per GPU: when the script is not already running under
torch.distributed.run (no WORLD_SIZE in the environment) it re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`; when it is, WORLD_SIZE must equal N.
Weak scaling: --envs per GPU; the ranks own independent env shards (no data-path collective) and exchange only the
12-byte per-env bookkeeping record, all-gathered over RCCL INSIDE the timed loop (mw_step_resident_gather: the gather of
step k overlaps the kernel of step k+1 on a side stream).  Rank 0 prints ONE JSON line.

The headline precision is fp64 ("parity mode": the reference computes in float64 and this is the precision whose GPU tests
assert obs / reward <= 1e-5 against the reference traces); the fp32 throughput mode is reported beside it at N = 1.
The episode phases of the batch are staggered uniformly over the 500-step horizon (mw_set_episode_phase + an untimed
500-step pre-roll), so every timed window samples whole episodes -- resets, free motion and late-episode contacts in their
true proportions -- whatever --steps is.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# action + persistent state in, state + observation out, per env-step.  fp32: 0.9 KB + the 50-wide one-hot
# (200 B) = 1.1 KB; fp64: 1.6 KB + the one-hot as float64 (400 B) = 2.0 KB.  Without one-hot (MT1): 0.9 / 1.6 KB.
ALGO_BYTES_PER_ENV_STEP = {("fp32", True): 1100.0, ("fp64", True): 2000.0, ("fp32", False): 900.0, ("fp64", False): 1600.0}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md
N_SIMD, F_CLK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs
# issue cost of one wave64 VALU instruction on its SIMD (MI355X_MICROARCH.md "Wave scheduling" + constants table): 2 cycles for
# single-precision / integer / moves (v_fma_f32: 2), 4 for double precision (78.6 TFLOP/s fp64 = half rate)
VALU_CYCLES_F64, VALU_CYCLES_OTHER = 4.0, 2.0
HORIZON = 500                   # SawyerXYZEnv.max_path_length (sawyer_xyz_env.py:153) = TimeLimit default


# ------------------------------------------------------------------------------------------------ CPU baseline
def _oracle_worker(task_names, seconds, seed):
    """The CPU oracle (independent fp64 C restatement of the engine) stepping the workload on ONE core: every task of the
    benchmark an equal share of the env-steps, random actions, 5 substeps + forward per env-step (physics only).  Timing build
    (oracle/mjlite.py: -O3 -march=native, the hulls' support cells instead of the checker's all-vertex scan -- same answers by
    construction) and the action loop in C (mjl_bench_env_steps): no Python or ctypes call per env-step."""
    import ctypes
    from metaworld_amd import tasks as T
    from oracle.mjlite import OracleData, OracleModel
    sims = []
    for task in task_names:
        c = T.TASK_CONST[task]
        om = OracleModel(T.compiled_model(c["model"]), timing=True)
        om.view("eq_data")[:] = [0, 0, 0, 0, 0, 0, -1, 0, 0, 0, 5.0]
        d = OracleData(om)
        d.mocap_pos[:] = np.array(c["hand_init_pos"]); d.mocap_quat[:] = [1, 0, 1, 0]; d.ctrl[:] = [-1, 1]
        d.step(100)
        sims.append((om, d, np.array(c["mocap_low"], dtype=np.float64), np.array(c["mocap_high"], dtype=np.float64)))
    state = ctypes.c_ulonglong(88172645463325252 + 7919 * seed)
    chunk = 10
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for om, d, lo_, hi_ in sims:          # one round = `chunk` env-steps of every task
            d.bench_env_steps(chunk, state, lo_, hi_)
            n += chunk
    return n, time.perf_counter() - t0


def host_cores():
    """(cores this process may run on, cgroup CPU quota in cores or None, os.cpu_count()): what 'all host cores' means here"""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        quota = q / float(f.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    return aff, quota, os.cpu_count() or 1


def usable_cores():
    """the cores this process can really use at once: the affinity mask capped by the cgroup CPU quota"""
    aff, quota, _ = host_cores()
    return max(1, min(aff, int(quota) if quota and quota >= 1 else aff))


def _full_step_port(task_names, seconds):
    """The WHOLE step (physics + observation + reward + wrappers, the product's own lane programs compiled for the host,
    tests/host_harness.cpp, OpenMP over envs) on all host cores: the stand-in for the reference's Python obs / reward layer,
    which cannot run on the GPU box (no mujoco / gymnasium / metaworld there)."""
    so = os.path.join(ROOT, "tests", "_build", "libmw_hostsim.so")
    if not os.path.exists(so):
        return None
    from metaworld_amd import native
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    cores = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))          # (read when the OpenMP runtime of the harness starts: one thread per usable core)
    lib = native.load("mwh_", so)
    n = 32 * cores          # (enough envs per OpenMP thread that the per-step fork / join does not dominate)
    n = max(n, len(task_names))
    old_nsub = os.environ.get("MW_NSUB")
    os.environ["MW_NSUB"] = "1"          # (host harness knob: no emulated sub-lanes -- one plain lane program per env and core)
    # the task mix as a custom MT benchmark with 2 goals per task (every goal costs one faithful 500-substep reset on the host)
    env = MetaWorldGpuVectorEnv("custom-mt", envs_list=list(task_names), num_envs=n, seed=1, precision="fp64", lib=lib,
                                use_one_hot=len(task_names) > 1, total_tasks_per_cls=2)
    env.reset()
    rng = np.random.default_rng(0)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        env.ctx.step(rng.uniform(-1, 1, (n, 4)).astype(np.float32))
        steps += 1
    dt = time.perf_counter() - t0
    env.close()
    if old_nsub is None:
        del os.environ["MW_NSUB"]
    else:
        os.environ["MW_NSUB"] = old_nsub
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores,
            "sample": f"{n} envs x {steps} full steps (physics + obs + reward + wrappers, host build of the lane programs, OpenMP) in {dt:.1f}s"}


def cpu_baseline(task_names, seconds=10.0):
    """cpu_baseline of the JSON line: the oracle on 1 core and on all cores (one process per core), physics only; plus the
    full step (with obs / reward) of the host build of the lane programs on all cores.  All three are stand-ins for the
    reference's own SyncVectorEnv / AsyncVectorEnv, which needs mujoco + gymnasium (absent here and on the GPU box)."""
    import multiprocessing as mp
    from oracle import mjlite
    mjlite.build(fast=True)          # (once, before the workers race for it)
    n1, dt1 = _oracle_worker(task_names, seconds, 0)
    aff, quota, ncpu = host_cores()
    cores = usable_cores()          # one worker per core this process may really use
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.starmap(_oracle_worker, [(task_names, seconds, 1 + r) for r in range(cores)])
    rate_all = sum(n / dt for n, dt in res)
    what = task_names[0] if len(task_names) == 1 else f"{len(task_names)} tasks in equal shares"
    out = {"value": rate_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "single_core_value": n1 / dt1, "parallel_speedup": rate_all / (n1 / dt1),
           "host": {"sched_affinity_cores": aff, "cgroup_cpu_quota_cores": quota, "os_cpu_count": ncpu},
           "sample": f"oracle engine (fp64 C restatement, timing build: -O3 -march=native, support cells, C action loop; stand-in, not "
                     f"Farama/MuJoCo): {sum(n for n, _ in res)} env-steps of {what} "
                     f"(random actions, 5 substeps + forward each, physics only) in {seconds:.0f}s on {cores} processes "
                     f"(= the cores in this process's affinity mask / cgroup quota; measured speed-up over 1 process {rate_all / (n1 / dt1):.1f}x); "
                     f"1 process: {n1} env-steps in {dt1:.1f}s"}
    full = _full_step_port(task_names, min(seconds, 8.0))
    if full:
        out["full_step_port"] = full
    return out


# ------------------------------------------------------------------------------------------------ profile lookup
def matching_profile(precision, workload, src_hash, profiles_dir=None):
    """(summary dict, path) of the newest committed rocprofv3 PMC summary (tools/profile_bench.sh -> profiles/r*_mt50_<precision>_pmc.json)
    that was taken on THIS workload with THESE device sources (content hash of csrc/ + include/mwgpu.h), or (None, None): counters of
    another kernel are never quoted."""
    import glob
    d = profiles_dir or os.path.join(ROOT, "profiles")
    for prof in sorted(glob.glob(os.path.join(d, f"r*_mt50_{precision}_pmc.json")), reverse=True):
        try:
            with open(prof) as f:
                cand = json.load(f)
        except (OSError, ValueError):
            continue
        if cand.get("workload") == workload and cand.get("source_hash") == src_hash:
            return cand, prof
    return None, None


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(argv, n):
    """the torch.distributed.run command line `python bench.py --gpus n ...` re-executes itself under"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500, help="timed steps (the staggered batch makes any window a whole-episode sample)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--benchmark", default="MT50")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"], help="headline precision (fp64 = the parity-qualified mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-precision", action="store_true", help="skip the measurement of the other precision at N=1")
    ap.add_argument("--no-stagger", action="store_true", help="all envs start their episodes together (early-episode window)")
    ap.add_argument("--allow-status", action="store_true", help="do not abort on capacity overflow / instability flags")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the launcher path (gloo: CPU test of the launcher)")
    ap.add_argument("--host-harness", action="store_true", help="TEST ONLY: drive the CPU harness instead of the GPU library")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary-inclusive rates (VectorEnv.step with host numpy / device tensors) at N=1")
    ap.add_argument("--no-saturation", action="store_true", help="skip the extra short run at 4x the envs at N=1")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of BASELINE.json's other configurations (2, 3, 5) at N=1")
    ap.add_argument("--split-collision", type=int, default=None, choices=[0, 1], help="1: narrow phase as batch-wide kernels between the lane "
                    "kernels (mw_set_option split_collision); 0: the fused step kernel; default: the library's own choice")
    ap.add_argument("--fixed-goals", action="store_true", help="auto-resets inside the timed loop re-use each env's look-ahead goal "
                    "(rounds 1-3) instead of drawing a new task per reset like RandomTaskSelectWrapper")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ one measurement
def build_env(args, precision, rank, world, local_rank, lib=None):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    if args.benchmark == "MT1":
        env = MetaWorldGpuVectorEnv("MT1", "reach-v3", num_envs=args.envs, seed=42 + rank, precision=precision, device_id=local_rank,
                                    rank=rank, world_size=world, lib=lib)
    else:
        env = MetaWorldGpuVectorEnv(args.benchmark, num_envs=args.envs, seed=42 + rank, use_one_hot=True, precision=precision,
                                    device_id=local_rank, rank=rank, world_size=world, lib=lib)
    if getattr(args, "split_collision", None) is not None:
        env.ctx.set_option("split_collision", args.split_collision)
    return env


def launch_stats(env):
    """min / median / max / mean of the HIP-event times of the launches of the last resident call (mw_launch_times)"""
    t = np.asarray(env.ctx.launch_times(), dtype=np.float64)
    if not len(t):
        return None
    return {"min": float(t.min()), "median": float(np.median(t)), "max": float(t.max()), "mean": float(t.mean()), "launches": int(len(t)),
            "argmax": int(t.argmax()), "first": [float(x) for x in t[:3]]}


def extra_configs(args, lib, local_rank):
    """BASELINE.json's other single-GPU configurations, short windows of the same protocol (staggered phases, untimed pre-roll), so
    that the driver's record carries them: 2 = MT1 reach-v3 @ 4096 fp32; 3 = MT10 @ 10240 fp64; 5 = ML45-train @ 2048 fp64 under
    the device-side scripted policies (one whole closed-loop episode per env, mean success beside the rate)."""
    import torch
    from metaworld_amd import tasks as T
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    out = []
    for cfg, bm, envs, prec, steps in ((2, "MT1", 4096, "fp32", 100), (3, "MT10", 10240, "fp64", 60)):
        a = argparse.Namespace(**{**vars(args), "benchmark": bm, "envs": envs, "warmup": 5})
        tb = time.perf_counter()
        env = build_env(a, prec, 0, 1, local_rank, lib)
        setup_s = time.perf_counter() - tb
        prepare(env, a, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = resident(env, a, steps)
        torch.cuda.synchronize()
        w = time.perf_counter() - t0
        ks = launch_stats(env)
        st = check_outputs(env, args.allow_status)
        out.append({"config": cfg, "workload": ("MT1 reach-v3" if bm == "MT1" else f"{bm} sync-vector") + f", {envs} envs/GPU, {prec}, random actions",
                    "value": envs * steps / w, "unit": "env-steps/s", "steps": steps, "kernel_ms_per_launch": k / steps,
                    "kernel_ms": ks, "value_median_based": envs / (ks["median"] / 1e3) if ks else None, "setup_s": setup_s, "flags": st["flags"]})
        env.close()
    n = 2048
    env = MetaWorldGpuVectorEnv("ML45-train", num_envs=n, seed=42, precision="fp64", partially_observable=False, max_episode_steps=500,
                                device_id=local_rank, lib=lib)
    if args.split_collision is not None:
        env.ctx.set_option("split_collision", args.split_collision)
    names = np.array(env.env_task_names)
    pid = np.array([T.ALL_V3.index(t) for t in names], dtype=np.int32)
    rank_in_task = np.concatenate([np.arange((names == t).sum()) for t in env.task_list])
    sched = (rank_in_task[None, :] + np.arange(2)[:, None]) % 50
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ep, su, ms = env.ctx.policy_rollout(pid, sched, 500)
    torch.cuda.synchronize()
    w = time.perf_counter() - t0
    st = env.ctx.status(clear=True)
    rates = [su[names == t].sum() / max(1, ep[names == t].sum()) for t in env.task_list]
    out.append({"config": 5, "workload": f"ML45-train, {n} envs/GPU, fp64, device-side scripted policies, one 500-step closed-loop episode per env",
                "value": n * 500 / (ms / 1e3), "unit": "env-steps/s", "steps": 500, "kernel_ms_per_launch": ms / 500,
                "wall_value_incl_reset_and_upload": n * 500 / w, "mean_success": float(np.mean(rates)),
                "tasks_at_or_above_0.8": int(sum(r >= 0.8 for r in rates)), "flags": st["flags"]})
    env.close()
    return out


def prepare(env, args, rank):
    """reset, stagger the episode phases (untimed pre-roll of one horizon), upload the action stream, warm up"""
    N = env.num_envs
    env.reset()
    acts = np.random.default_rng(rank).uniform(-1, 1, (64, N, 4)).astype(np.float32)
    env.ctx.upload_actions(acts)
    if not args.no_stagger:
        env.ctx.set_episode_phase((np.arange(N, dtype=np.int64) * 7919 % HORIZON).astype(np.int32))
        resident(env, args, HORIZON)
    resident(env, args, args.warmup)
    st = env.ctx.status(clear=True)
    if st["flags"] and not args.allow_status:          # the timed region continues from this state: an overflow here counts too
        raise RuntimeError(f"bench: the step kernel raised status flags {st} during the untimed pre-roll / warm-up (1/2 = constraint-row / "
                           "contact capacity exceeded, 4 = non-finite state)")


def resident(env, args, nsteps, gather=False):
    """nsteps launches on the resident actions; every auto-reset inside draws a new task from the sub-env's selection stream
    (RandomTaskSelectWrapper.reset, metaworld/wrappers.py:116-119) through the device goal schedule -> kernel ms"""
    if getattr(args, "fixed_goals", False):
        return env.ctx.step_resident_gather(nsteps) if gather else env.ctx.step_resident(nsteps)
    return env.step_resident(nsteps, gather=gather)


def boundary_rates(args, lib, local_rank, steps=40):
    """env-steps/s THROUGH the boundary, same workload and batch as `value` (which leaves actions and outputs in HBM): fresh random
    actions every step, everything `VectorEnv.step` returns materialised (obs, reward, flags, the dict of infos, final_obs /
    final_info where an episode ended, the host-side task selection).  host_numpy = MetaWorldGpuVectorEnv.step (numpy in / out:
    H2D actions + D2H of every output per step); torch_device = MetaWorldTorchVectorEnv.step (tensors in / out, only the done row
    visits the host)."""
    import torch
    from metaworld_amd.torch_env import MetaWorldTorchVectorEnv
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    out = {}
    for key, cls in (("host_numpy", MetaWorldGpuVectorEnv), ("torch_device", MetaWorldTorchVectorEnv)):
        kw = dict(num_envs=args.envs, seed=42, use_one_hot=True, precision=args.precision, device_id=local_rank, lib=lib)
        env = cls("MT1", "reach-v3", **{**kw, "use_one_hot": False}) if args.benchmark == "MT1" else cls(args.benchmark, **kw)
        N = env.num_envs
        env.reset()
        env.ctx.upload_actions(np.random.default_rng(0).uniform(-1, 1, (64, N, 4)).astype(np.float32))
        if not args.no_stagger:
            env.ctx.set_episode_phase((np.arange(N, dtype=np.int64) * 7919 % HORIZON).astype(np.int32))
            env.step_resident(HORIZON)
        rng = np.random.default_rng(1)
        acts = rng.uniform(-1, 1, (8, N, 4)).astype(np.float32)
        if key == "torch_device":
            acts = torch.from_numpy(acts).to(env.device)
        for t in range(3):
            o, r, te, tr, info = env.step(acts[t % 8])
        if key == "torch_device":
            float(r.sum().item())          # (untimed: the first reduction loads its kernel, ~20 ms once per process)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stamps = []
        for t in range(steps):
            o, r, te, tr, info = env.step(acts[t % 8])
            stamps.append(time.perf_counter())          # (host time at which step t returned; the tensor path's outputs are stream-ordered)
        if key == "torch_device":
            float(r.sum().item())          # the learner reads what it got
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = env.ctx.status(clear=True)
        env.close()
        per = np.diff(np.array([t0] + stamps)) * 1e3
        out[key] = {"value": N * steps / dt, "unit": "env-steps/s", "ms_per_step": dt / steps * 1e3, "median_ms_per_step": float(np.median(per)),
                    "max_ms_per_step": float(per.max()), "steps": steps, "status_flags": st["flags"]}
        if os.environ.get("MW_BENCH_PER_STEP"):          # diagnosis: the host time of every step
            out[key]["per_step_ms"] = [round(float(x), 2) for x in per]
    out["note"] = ("the same batch through VectorEnv.step instead of the resident loop: host_numpy = numpy actions in, every output + infos dict "
                   "on the host (MetaWorldGpuVectorEnv); torch_device = CUDA tensors in / out (MetaWorldTorchVectorEnv); `value` itself is the resident loop")
    return out


def check_outputs(env, allow):
    st = env.ctx.status(clear=True)
    if st["flags"] and not allow:
        raise RuntimeError(f"bench: the step kernel raised status flags {st} (1/2 = constraint-row / contact capacity exceeded, 4 = "
                           "non-finite state): the measured steps are not the reference's computation")
    obs = env.ctx.step(np.zeros((env.num_envs, 4), dtype=np.float32), env._next_goal)[0]
    if not np.isfinite(obs).all():
        raise RuntimeError("bench: non-finite observation after the timed region")
    return st


def main(argv=None):
    args = parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # one process per GPU: re-execute under torch.distributed.run (the driver's own launch line does the same)
        cmd = launcher_command(sys.argv[1:] if argv is None else argv, args.gpus)
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop the launcher and let "
                         "bench.py spawn the ranks itself)")
    import torch
    dist = None
    on_gpu = not args.host_harness
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    lib = None
    if args.split_collision == 1 and not os.environ.get("MW_LIB"):          # the split-collision experiment lives in its own build of the library (-DMW_SPLIT_COLLISION)
        os.environ["MW_LIB"] = "libmwgpu_split.so"
    if os.environ.get("MW_LIB"):          # experiments: a variant build of the library (tools/build_variants.sh)
        from metaworld_amd import native
        lib = native.load("mw_", os.path.join(ROOT, "metaworld_amd", os.environ["MW_LIB"]))
    if args.host_harness:
        import __graft_entry__ as g
        from metaworld_amd import native
        lib = native.load("mwh_", g.build_host_harness())
        local_rank = 0
    from metaworld_amd import tasks as T
    t_build = time.perf_counter()
    env = build_env(args, args.precision, rank, world, local_rank, lib)
    setup_s = time.perf_counter() - t_build          # model upload + mw_finalize: the faithful reset of every (task, goal) once (compute_snapshots)
    N = env.num_envs
    one_hot = args.benchmark != "MT1"
    wl_task = ["reach-v3"] if args.benchmark == "MT1" else list(env.task_list)
    workload = (f"MT1 reach-v3, {N} batched envs/GPU" if args.benchmark == "MT1" else f"{args.benchmark} sync-vector, {N} envs/GPU") + \
        f", {args.precision}, random actions"

    # cross-rank bookkeeping: the communicator lives inside the library (RCCL over xGMI); its id travels over torch.distributed
    gather_mode = "none (1 rank)"
    if world > 1:
        dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
        uid = torch.from_numpy(env.ctx.comm_unique_id()).to(dev) if rank == 0 else torch.zeros(128, dtype=torch.uint8, device=dev)
        dist.broadcast(uid, 0)
        env.ctx.comm_init(uid.cpu().numpy(), rank, world)
        gather_mode = "RCCL all-gather inside the library, per step, side stream (mw_step_resident_gather)" if on_gpu else \
            "host-harness shared-memory all-gather, per step (TEST)"
    use_gather = world > 1
    if world == 1 and on_gpu and os.environ.get("MW_COMM_FORCE_RCCL"):
        # single-GPU box: a REAL one-rank RCCL communicator, so that the timed loop contains ncclAllGather per step on the side stream
        # (what every rank does at N > 1); `config.comm` shows what RCCL reports for it
        env.ctx.comm_init(env.ctx.comm_unique_id(), 0, 1)
        use_gather = True
        gather_mode = "RCCL all-gather inside the library, per step, side stream, ONE-rank communicator (MW_COMM_FORCE_RCCL)"
    if use_gather:
        # every rank must see a communicator that spans the job before anything is timed (a rank that silently fell back to a
        # one-rank communicator would still produce a number)
        ci = env.ctx.comm_info()
        expect = world if (world > 1 or os.environ.get("MW_COMM_FORCE_RCCL")) else 1
        if on_gpu and ci.get("comm_count") != expect:
            raise RuntimeError(f"bench: rank {rank}: the RCCL communicator reports {ci} but the job has {world} ranks")
    prepare(env, args, rank)

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    # K launches on the library's stream bracketed by HIP events; with > 1 rank the per-step all-gather is inside the loop
    kernel_ms = resident(env, args, args.steps, gather=use_gather)
    barrier()
    wall = time.perf_counter() - t0
    per_rank_kernel_ms = [kernel_ms / args.steps]
    kstats = launch_stats(env)
    if dist is not None:
        tw = torch.tensor([wall], device="cuda" if on_gpu else "cpu", dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        tk = torch.zeros(world, device="cuda" if on_gpu else "cpu", dtype=torch.float64)
        tk[rank] = kernel_ms / args.steps
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)          # every rank's kernel ms per launch: an imbalance shows in the line
        per_rank_kernel_ms = [float(x) for x in tk.cpu()]
    status = check_outputs(env, args.allow_status)
    book = env.ctx.gather_bookkeeping()          # [world, N] records of the last step on every rank
    assert book.shape == (world, N)
    if rank == 0:
        total_steps = N * world * args.steps
        value = total_steps / wall
        per_launch_s = kernel_ms / 1e3 / args.steps
        algo = ALGO_BYTES_PER_ENV_STEP[(args.precision, one_hot)]
        ach = algo * N / per_launch_s / 1e9
        roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                    "traffic": None, "kernel_ms_per_launch": kernel_ms / args.steps, "kernel_ms": kstats, "algorithmic_bytes_per_env_step": algo,
                    "note": "achieved = algorithmic bytes/env-step x envs / HIP-event kernel time (events on the library's own stream); "
                            "traffic = FETCH_SIZE + WRITE_SIZE bytes per launch of the committed rocprofv3 PMC profile of this command on the same sources; "
                            "the step kernel is bound by the latency of its per-environment dependency chain, not by HBM (DESIGN.md 5): "
                            "see alu_issue"}
        # counter-derived numbers are quoted only from a profile of THESE device sources (tools/profile_bench.sh writes the content
        # hash of csrc/ into the summary): a stale profile gives traffic = null instead of another kernel's counters
        from metaworld_amd import native as _native
        src_hash = _native.source_hash()
        roofline["source_hash"] = src_hash
        pj, prof_path = matching_profile(args.precision, workload, src_hash)
        if pj is not None:
            roofline["profile"] = os.path.relpath(prof_path, ROOT)
        if pj is None:
            roofline["note"] += "; NO committed PMC profile matches these sources (source_hash): traffic / alu_issue not quoted"
        if pj is not None and world == 1:
                roofline["traffic"] = pj["fetch_bytes_per_launch"] + pj["write_bytes_per_launch"]
                # the bound that actually moves: VALU issue.  Per MI355X_MICROARCH.md a wave64 VALU instruction
                # occupies its SIMD for 2 cycles, a double-precision one for 4; the SIMD-cycles the launch NEEDED for its VALU
                # instructions over the SIMD-cycles it HAD (1024 SIMDs x kernel time x 2.4 GHz) is the issue fraction.
                valu = pj["valu_wave_instr_per_launch"]
                f64 = pj.get("valu_f64_wave_instr_per_launch")
                have = N_SIMD * F_CLK_HZ * per_launch_s
                need = (f64 * VALU_CYCLES_F64 + (valu - f64) * VALU_CYCLES_OTHER) if f64 is not None else None
                roofline["alu_issue"] = {
                    "valu_wave_instr_per_env_step": valu / N, "valu_wave_instr_per_launch": valu, "valu_f64_wave_instr_per_launch": f64,
                    "simd_cycles_needed": need, "simd_cycles_available": have, "frac": (need / have) if need is not None else None,
                    "cycles_per_wave64_valu": {"f64": VALU_CYCLES_F64, "other": VALU_CYCLES_OTHER},
                    "valu_f64_share": (f64 / valu) if f64 else None,
                    "valu_active_frac_of_wave_cycles": pj.get("valu_active_frac"),
                    "lane_utilisation": pj.get("lane_utilisation"), "wave_slot_occupancy": pj.get("wave_slot_occupancy"),
                    "wait_frac": pj.get("wait_frac"),
                    "mfma": pj.get("mfma"),
                    "note": "mfma = the matrix-core counters of the same profile (Newton direction: v_mfma_f32_16x16x1_4b_f32; utilisation = "
                            "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)); frac = (f64 VALU wave-instructions x 4 + the other VALU wave-instructions x 2 cycles, MI355X_MICROARCH.md) / "
                            "(1024 SIMDs x this run's kernel time x 2.4 GHz); counts from the PMC pass of the same sources (SQ_INSTS_VALU, "
                            "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64); valu_active_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; "
                            "lane_utilisation = envs / (waves x 64); wave_slot_occupancy = SQ_WAVE_CYCLES x 4 / (1024 SIMDs x kernel cycles); "
                            "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES"}
        out = {"metric": "env-steps/sec (whole node) MT50 @4096 envs/GPU; achieved HBM GB/s vs peak", "value": value,
               "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.precision == "fp32" else "f64", "data": "synthetic",
               "value_median_based": (N * world / (kstats["median"] / 1e3)) if kstats else None,
               "config": {"workload": workload, "envs_per_gpu": N, "tasks_with_device_code": len(T.supported_tasks()),
                          "setup_s": setup_s,
                          "episode_phase": "all envs start together (early-episode window)" if args.no_stagger else
                          f"staggered uniformly over the {HORIZON}-step horizon (mw_set_episode_phase + untimed {HORIZON}-step pre-roll): "
                          "every window samples whole episodes incl. auto-resets",
                          "task_resampling": "none: auto-resets re-use the env's look-ahead goal (--fixed-goals)" if args.fixed_goals else
                          "every auto-reset inside the timed loop draws a new task from the sub-env's selection stream (RandomTaskSelectWrapper.reset, "
                          "metaworld/wrappers.py:116-119) through the device goal schedule (mw_set_goal_schedule)",
                          "parallelism": f"dp{world} (independent env shards, no data-path collective)", "bookkeeping_gather": gather_mode,
                          "comm": env.ctx.comm_info(), "per_rank_kernel_ms": per_rank_kernel_ms,
                          "collision": {None: "library default (fused step kernel)", 0: "fused step kernel", 1: "split: narrow phase as batch-wide kernels "
                                        "between the lane kernels (mw_split.inl); kernel_ms = one whole step (all its launches)"}[args.split_collision],
                          "status_flags": status},
               "roofline": roofline}
        if world == 1 and not args.no_extra_precision and on_gpu:
            other = "fp32" if args.precision == "fp64" else "fp64"
            env.close()
            env2 = build_env(args, other, rank, world, local_rank, lib)
            prepare(env2, args, rank)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            k2 = resident(env2, args, args.steps)
            torch.cuda.synchronize()
            w2 = time.perf_counter() - t1
            ks2 = launch_stats(env2)
            st2 = check_outputs(env2, args.allow_status)
            key = "throughput_mode" if other == "fp32" else "parity_mode"
            out[key] = {"dtype": "f32" if other == "fp32" else "f64", "value": N * args.steps / w2, "unit": "env-steps/s",
                        "ms_per_step": w2 / args.steps * 1e3, "kernel_ms_per_launch": k2 / args.steps, "kernel_ms": ks2,
                        "value_median_based": N / (ks2["median"] / 1e3) if ks2 else None, "status_flags": st2,
                        # line searches abandoned on a non-descent direction (single-precision factor of an ill-conditioned Hessian, or
                        # rounding at the optimum): the Newton loop of that environment ends there like the reference solver's does; each
                        # env-step holds ~6 solves x ~2 iterations
                        "solver_stalls_per_1000_env_steps": 1e3 * st2["solver_stalls"] / (N * args.steps),
                        "note": "fp32 state and arithmetic: success flags exact, obs / reward within the single-precision contact-geometry "
                                "floor (not 1e-5 on every task, DESIGN.md 6)" if other == "fp32" else "fp64 state and arithmetic"}
            env2.close()
        if world == 1 and on_gpu and not args.no_boundary:
            if not env.closed:
                env.close()
            out["boundary"] = boundary_rates(args, lib, local_rank)
        if world == 1 and on_gpu and not args.no_saturation:
            # does the chip saturate?  the same workload at 4x the batch, a short window (value stays the configuration of the metric)
            if not env.closed:
                env.close()
            sat_args = argparse.Namespace(**{**vars(args), "envs": 4 * args.envs, "warmup": 10})
            env3 = build_env(sat_args, args.precision, rank, world, local_rank, lib)
            prepare(env3, sat_args, rank)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            k3 = resident(env3, sat_args, 100)
            torch.cuda.synchronize()
            w3 = time.perf_counter() - t3
            ks3 = launch_stats(env3)
            st3 = check_outputs(env3, args.allow_status)
            out["saturation"] = {"envs": env3.num_envs, "value": env3.num_envs * 100 / w3, "unit": "env-steps/s", "steps": 100,
                                 "kernel_ms_per_launch": k3 / 100, "kernel_ms": ks3,
                                 "value_median_based": env3.num_envs / (ks3["median"] / 1e3) if ks3 else None, "status_flags": st3}
            env3.close()
        if world == 1 and on_gpu and not args.no_saturation:
            # the open-loop rollout with several steps per launch (mw_step_resident_fused): the same env-steps bit for bit, the batch
            # synchronised once per launch instead of once per step.  NOT `value`: a VectorEnv.step returns after every step.
            env4 = build_env(args, args.precision, rank, world, local_rank, lib)
            prepare(env4, args, rank)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            fsteps = max(args.steps, 150)          # its own window: at least three launches of 50 steps whatever --steps says
            k4 = env4.step_resident(fsteps, steps_per_launch=50)
            torch.cuda.synchronize()
            w4 = time.perf_counter() - t4
            st4 = check_outputs(env4, args.allow_status)
            out["fused_rollout"] = {"steps_per_launch": 50, "value": env4.num_envs * fsteps / w4, "unit": "env-steps/s", "steps": fsteps,
                                    "kernel_ms_per_step": k4 / fsteps, "status_flags": st4,
                                    "note": "open-loop rollout, 50 consecutive steps of every environment per kernel launch (same results as the per-step "
                                            "loop bit for bit: tests/test_resident_schedule.py); reported beside `value`, never as it"}
            env4.close()
        if world == 1 and on_gpu and not args.no_configs:
            if not env.closed:
                env.close()
            out["configs"] = extra_configs(args, lib, local_rank)
        if world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(wl_task)
        print(json.dumps(out), flush=True)
    if not env.closed:
        env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
