"""The DEVICE code's own output against invariants that need no oracle: after one substep the column store of an environment
holds the poses, the contacts, the constraint rows and the solver result of one dynamics evaluation, and these must be consistent
with each other and with the packed model whatever produced them:

  * equation of motion        M qacc = qfrc_smooth + J' f   and   qfrc_constraint = J' f
  * every contact force inside its elliptic friction cone   sqrt(sum_j (f_j / mu_j)^2) <= f_n,  f_n >= 0
  * single-point contacts: dist = gap(n) = min_B b.n - max_A a.n, evaluated with numpy support functions from the geom definitions
    (tests/test_narrowphase_geometry.py) on the device's geom poses
  * unit normals, right-handed orthonormal contact frames

On the host build of the lane programs here, and through libmwgpu.so on the GPU (`-m gpu`), in both precisions (fp32: at
single-precision resolution -- the portal refinement stops at 2e-6, forces of 4e2 carry 1e-4)."""
import collections

import numpy as np
import pytest

from metaworld_amd import policies as P
from metaworld_amd import tasks as T
from tests.test_narrowphase_geometry import BOX, CAPSULE, CYLINDER, PLANE, gap

TASKS = ["hammer-v3", "box-close-v3", "basketball-v3", "assembly-v3", "stick-pull-v3", "peg-insert-side-v3", "door-lock-v3", "soccer-v3",
         "drawer-open-v3", "coffee-push-v3", "sweep-into-v3", "pick-place-wall-v3"]
# eom / qfc: relative to the largest generalized force.  The residual of the equation of motion is the gradient of the solver's cost:
# Newton stops on `tolerance` x mean inertia x nv (1e-10 x ~1e4 x 15: a frozen dof carries an armature of 1e5), measured <= 5e-8
# (host build, 12 tasks: fp64 eom 6.6e-8, qfc 2e-15, cone 2e-16, ident 9.5e-10, frame 7e-16; fp32 1.2e-4, 1.1e-6, 9e-8, 2.0e-6, 5e-7)
LIMITS = {"fp64": dict(eom=1e-6, qfc=1e-12, cone=1e-9, ident=2e-6, frame=1e-12),
          "fp32": dict(eom=2e-3, qfc=5e-5, cone=1e-4, ident=2e-5, frame=1e-5)}


def check_invariants(lib, task, precision, every=16, n_read=2):
    from metaworld_amd.vector_env import MetaWorldGpuVectorEnv
    lim = LIMITS[precision]
    env = MetaWorldGpuVectorEnv("MT1", task, num_envs=2, seed=3, precision=precision, lib=lib)
    obs, _ = env.reset()
    pk, _, _ = T.packed_model(T.TASK_CONST[task]["model"], reloc_bodies=T.model_key(task)[1])
    A = dict(pk["ints"])
    A.update(pk["reals"])
    A["geom_size"] = np.asarray(A["geom_size"]).reshape(-1, 3)
    nv = len(A["dof_bodyid"])
    seen = 0
    for t in range(160):
        obs = env.step(P.batched_actions([task] * 2, obs.astype(np.float64)).astype(np.float32))[0]
        if t % every != every - 1:
            continue
        env.ctx.debug("substeps", 1)          # poses, contacts, rows and solver output of ONE evaluation (the state before it integrates)
        for e in range(n_read):
            ic = env.ctx.read_int(e, "icount")
            ncon, nefc = int(ic[0]), int(ic[1])
            con = env.ctx.read(e, "con").reshape(-1, 26)[:ncon]
            icon = env.ctx.read_int(e, "icon").reshape(-1, 4)[:ncon]
            J = env.ctx.read(e, "efcJ").reshape(-1, nv)[:nefc]
            f = env.ctx.read(e, "efcX").reshape(-1, 11)[:nefc, 5]
            M = env.ctx.read(e, "qM").reshape(nv, nv)
            M = np.tril(M) + np.tril(M, -1).T
            qacc, smooth, qfc = env.ctx.read(e, "qacc"), env.ctx.read(e, "smooth"), env.ctx.read(e, "qfrc_constraint")
            ctx = (task, precision, t, e)
            scale = max(1.0, np.abs(smooth).max(), np.abs(J.T @ f).max())
            assert np.abs(M @ qacc - smooth - J.T @ f).max() < lim["eom"] * scale, ctx
            assert np.abs(qfc - J.T @ f).max() < lim["qfc"] * scale, ctx
            xpos, xmat = env.ctx.read(e, "geom_xpos").reshape(-1, 3), env.ctx.read(e, "geom_xmat").reshape(-1, 9)
            per_pair = collections.Counter((g1, g2) for g1, g2, _, _ in icon)
            for c in range(ncon):
                g1, g2, dim, a = (int(x) for x in icon[c])
                F = con[c, 4:13].reshape(3, 3)
                assert np.abs(F @ F.T - np.eye(3)).max() < lim["frame"] and np.linalg.det(F) > 0.99, ctx
                if a >= 0:
                    mus = np.array([con[c, 14], con[c, 14], con[c, 15], con[c, 16], con[c, 16]])[:dim - 1]
                    fn = f[a]
                    assert fn >= -lim["cone"], (ctx, fn)
                    assert np.sqrt(((f[a + 1:a + dim] / mus) ** 2).sum()) <= fn + lim["cone"] * (1 + abs(fn)), (ctx, fn)
                t1, t2 = A["geom_type"][g1], A["geom_type"][g2]
                single = per_pair[(g1, g2)] == 1 and t1 != PLANE and not (t2 == BOX and t1 in (CAPSULE, CYLINDER, BOX))
                if single:          # (multi-point routines and face contacts: tests/test_narrowphase_geometry.py)
                    assert abs(con[c, 0] - gap(A, g1, g2, xpos, xmat, F[0])) < lim["ident"], (ctx, g1, g2)
                seen += 1
    env.close()
    assert seen > 0, (task, seen)


@pytest.mark.parametrize("task,precision", [(t, "fp64") for t in T.supported_tasks()] + [(t, "fp32") for t in TASKS])
def test_host_build_output_is_self_consistent(hostsim, task, precision):
    check_invariants(hostsim, task, precision)


@pytest.mark.gpu
@pytest.mark.parametrize("task,precision", [(t, "fp64") for t in T.supported_tasks()] + [(t, "fp32") for t in TASKS])          # (round 2: 12 tasks)
def test_gpu_output_is_self_consistent(gpulib, task, precision):
    check_invariants(gpulib, task, precision, every=32, n_read=1)          # (column reads cross the bus element by element)
