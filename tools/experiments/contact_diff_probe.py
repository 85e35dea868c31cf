#!/usr/bin/env python
"""device vs oracle contact lists at a synchronised state of a policy200 trace: tools/experiments/contact_diff_probe.py <task> <step> [lib: gpu|host]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from metaworld_amd import native, tasks as T
from tests.helpers import golden, make_env
from tests.test_gpu_fullsize import _oracle_synced_to
task, t = sys.argv[1], int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "host":
    import __graft_entry__ as g
    lib = native.load("mwh_", g.build_host_harness())
else:
    lib = native.load()
G = dict(golden(f"trace_policy200_{task}_seed42.npz"))
env = make_env(lib, task, n=1, precision="fp64")
ctx = env.ctx
ctx.reset(G["goal_idx"])
ctx.write(0, "qpos", G["qpos"][0, t - 1]); ctx.write(0, "qvel", G["qvel"][0, t - 1])
ctx.write(0, "mocap", G["mocap"][0, t - 1]); ctx.write(0, "warm", G["warm"][0, t - 1])
a = G["actions"][0, t]
lo, hi = np.array(T.TASK_CONST[task]["mocap_low"]), np.array(T.TASK_CONST[task]["mocap_high"])
moc = np.clip(G["mocap"][0, t - 1] + (np.clip(a[:3], -1, 1).astype(np.float32) * np.float32(0.01)).astype(np.float64), lo, hi)
ctx.write(0, "mocap", moc); ctx.write(0, "ctrl", [a[3], -a[3]])
om, d = _oracle_synced_to(ctx, 0, task)
for sub in range(5):
    ctx.debug("forward")
    d.forward()
    ncon = int(ctx.read_int(0, "icount")[0])
    con = ctx.read(0, "con").reshape(-1, 26)[:ncon]
    oc = d.contacts()
    print(f"substep {sub}: device ncon {ncon} nefc {int(ctx.read_int(0, 'icount')[1])} niter {int(ctx.read_int(0, 'icount')[2])} | oracle ncon {d.ncon} nefc {d.nefc} niter {d.info()['niter']}")
    n = min(ncon, len(oc))
    for i in range(n):
        dd, dp = abs(con[i, 0] - oc[i]["dist"]), np.abs(con[i, 1:4] - oc[i]["pos"]).max()
        dn = np.abs(con[i, 4:7] - oc[i]["frame"][:3]).max()
        flag = "  <<<" if max(dd, dp, dn) > 1e-9 else ""
        print(f"   c{i:2d} oracle geoms ({oc[i]['geom1']:3d},{oc[i]['geom2']:3d}) dist {oc[i]['dist']:+.3e}  |ddist| {dd:.1e} |dpos| {dp:.1e} |dnormal| {dn:.1e}{flag}")
    qa_dev, qa_or = ctx.read(0, "qacc"), d.qacc
    print("   |qacc dev - oracle| max", np.abs(qa_dev - qa_or).max(), "at", int(np.abs(qa_dev - qa_or).argmax()), " |qacc|", np.abs(qa_or).max())
    ctx.debug("substeps", 1); d.step(1)
    print("   after substep: |dqpos|", np.abs(ctx.read(0, "qpos") - d.qpos).max(), "|dqvel|", np.abs(ctx.read(0, "qvel") - d.qvel).max())
