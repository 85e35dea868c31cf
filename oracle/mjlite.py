"""ctypes binding for the CPU oracle engine (oracle/mjl_core.c, mjl_collide.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product path (metaworld_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmjlite.so")
_LIB_PATH_FAST = os.path.join(_HERE, "_build", "libmjlite_fast.so")
_SRCS = ["mjl_core.c", "mjl_collide.c"]
# the checker: strict IEEE, no contraction (bit-reproducible against the host build of the lane programs)
_FLAGS = ["-O2", "-std=gnu99", "-ffp-contract=off"]
# bench.py's cpu_baseline only: the same sources built for speed on the host they run on (the stand-in was timed at -O2
# without -march=native)
_FLAGS_FAST = ["-O3", "-march=native", "-std=gnu99"]


def _fast_path():
    """-march=native code only runs on the CPU it was built for: the file name carries a hash of this host's CPU flags, so a
    library built in another container is never loaded on the GPU box (it is rebuilt there, gcc is in the image)"""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((l for l in f if l.startswith("flags")), "")
    except OSError:
        flags = ""
    return _LIB_PATH_FAST.replace(".so", "_" + hashlib.sha1(flags.encode()).hexdigest()[:8] + ".so")


def build(force=False, fast=False):
    """compile the C restatement; safe under concurrent callers (pytest-xdist workers): one builder at a time, the compiler writes
    a temporary file that is renamed into place.  fast=True: the timing build (-O3 -march=native), a second library"""
    import fcntl
    _LIB_PATH = _fast_path() if fast else globals()["_LIB_PATH"]
    srcs = [os.path.join(_HERE, s) for s in _SRCS]
    deps = srcs + [os.path.join(_HERE, "mjl_core.h")]

    def fresh():
        return os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in deps if os.path.exists(s))
    if not force and fresh():
        return _LIB_PATH
    os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
    with open(_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not fresh():
                tmp = f"{_LIB_PATH}.tmp.{os.getpid()}"
                subprocess.check_call(["gcc"] + (_FLAGS_FAST if fast else _FLAGS) + ["-shared", "-fPIC", "-o", tmp] + srcs + ["-lm"])
                os.replace(tmp, _LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB_PATH


_lib = {}


def lib(fast=False):
    if fast not in _lib:
        L = C.CDLL(build(fast=fast))
        L.mjl_model_new.restype = C.c_void_p
        L.mjl_data_new.restype = C.c_void_p
        L.mjl_data_new.argtypes = [C.c_void_p]
        L.mjl_model_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.mjl_model_set_real.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.mjl_model_finalize.argtypes = [C.c_void_p]
        L.mjl_model_real_ptr.restype = C.POINTER(C.c_double)
        L.mjl_model_real_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.mjl_data_real_ptr.restype = C.POINTER(C.c_double)
        L.mjl_data_real_ptr.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        for fn in ("mjl_reset_data", "mjl_forward", "mjl_step", "mjl_kinematics", "mjl_crb", "mjl_rne_bias",
                   "mjl_collision"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, fn).restype = None
        L.mjl_step_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.mjl_step_n.restype = None
        L.mjl_model_free.argtypes = [C.c_void_p]
        L.mjl_data_free.argtypes = [C.c_void_p]
        L.mjl_data_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.mjl_data_contact.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.mjl_data_efc_int.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mjl_bench_env_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_void_p, C.c_void_p]
        L.mjl_bench_env_steps.restype = None
        _lib[fast] = L
    return _lib[fast]


# acceleration tables of the product's hull support function (support cells: metaworld_amd/hullcells.py, mjcf.py add_mesh_cells;
# the hull graph / start cube map of model files written before round 4).  The oracle scans every hull vertex
# (mjl_collide.c support()) and takes none of them -- except in bench.py's timing runs (OracleModel(timing=True)), which hand it
# the support cells so that the CPU stand-in is not handicapped by the scan.
PRODUCT_ONLY_ARRAYS = {"mesh_nbradr", "mesh_nbr", "mesh_start", "mesh_hill", "mesh_celladr", "mesh_cellid"}


class OracleModel:
    """Holds a C MjlModel built from a metaworld_amd.mjcf.Model."""

    def __init__(self, model, timing=False):
        """timing=True (bench.py's cpu_baseline ONLY): the -O3 -march=native build + the hulls' support cells (same answers as the
        scan by construction of the cells, tests/test_support_cells.py; the parity chain never uses it)"""
        L = self.L = lib(fast=timing)
        self.src = model
        self.ptr = L.mjl_model_new()
        for k, v in model.arrays.items():
            if k in PRODUCT_ONLY_ARRAYS and not (timing and k in ("mesh_celladr", "mesh_cellid")):
                continue
            if v.dtype.kind in "iu":
                a = np.ascontiguousarray(v, dtype=np.int32)
                rc = L.mjl_model_set_int(self.ptr, k.encode(), a.ctypes.data, a.size)
            else:
                a = np.ascontiguousarray(v, dtype=np.float64)
                rc = L.mjl_model_set_real(self.ptr, k.encode(), a.ctypes.data, a.size)
            assert rc == 0, f"oracle does not know model field {k}"
        for k, v in (("opt_timestep", model.opt_timestep), ("opt_tolerance", model.opt_tolerance)):
            a = np.array([v], dtype=np.float64)
            L.mjl_model_set_real(self.ptr, k.encode(), a.ctypes.data, 1)
        a = np.array([model.opt_iterations], dtype=np.int32)
        L.mjl_model_set_int(self.ptr, b"opt_iterations", a.ctypes.data, 1)
        g = np.ascontiguousarray(model.gravity, dtype=np.float64)
        L.mjl_model_set_real(self.ptr, b"gravity", g.ctypes.data, 3)
        rc = L.mjl_model_finalize(self.ptr)
        assert rc == 0, rc
        A = model.arrays
        self.nq, self.nv = len(A["qpos0"]), len(A["dof_bodyid"])
        self.nbody, self.ngeom, self.nsite = len(A["body_parentid"]), len(A["geom_type"]), len(A["site_bodyid"])

    def view(self, name, shape=None):
        """writable numpy view of a real-valued model array living in C memory."""
        n = C.c_int(0)
        p = self.L.mjl_model_real_ptr(self.ptr, name.encode(), C.byref(n))
        assert p, name
        a = np.ctypeslib.as_array(p, shape=(n.value,))
        return a.reshape(shape) if shape else a

    def __del__(self):
        try:
            self.L.mjl_model_free(self.ptr)
        except Exception:
            pass


class OracleData:
    def __init__(self, om: OracleModel):
        self.om = om
        self.L = om.L
        self.ptr = self.L.mjl_data_new(om.ptr)
        self._views = {}
        m = om
        self.qpos = self.view("qpos")
        self.qvel = self.view("qvel")
        self.ctrl = self.view("ctrl")
        self.qacc = self.view("qacc")
        self.qacc_warmstart = self.view("qacc_warmstart")
        self.mocap_pos = self.view("mocap_pos")
        self.mocap_quat = self.view("mocap_quat")
        self.xpos = self.view("xpos", (m.nbody, 3))
        self.xquat = self.view("xquat", (m.nbody, 4))
        self.xmat = self.view("xmat", (m.nbody, 9))
        self.xipos = self.view("xipos", (m.nbody, 3))
        self.geom_xpos = self.view("geom_xpos", (m.ngeom, 3))
        self.geom_xmat = self.view("geom_xmat", (m.ngeom, 9))
        self.site_xpos = self.view("site_xpos", (m.nsite, 3))
        self.site_xmat = self.view("site_xmat", (m.nsite, 9))
        self.qM = self.view("qM", (m.nv, m.nv))
        self.qfrc_bias = self.view("qfrc_bias")
        self.qfrc_constraint = self.view("qfrc_constraint")
        self.qfrc_smooth = self.view("qfrc_smooth")
        self.qacc_smooth = self.view("qacc_smooth")
        self.efc_force_buf = self.view("efc_force")
        self.time_buf = self.view("time")

    def view(self, name, shape=None):
        n = C.c_int(0)
        p = self.L.mjl_data_real_ptr(self.om.ptr, self.ptr, name.encode(), C.byref(n))
        assert p, name
        a = np.ctypeslib.as_array(p, shape=(n.value,))
        return a.reshape(shape) if shape else a

    @property
    def time(self):
        return float(self.time_buf[0])

    def info(self):
        out = (C.c_int * 8)()
        self.L.mjl_data_info(self.ptr, out)
        return dict(ncon=out[0], nefc=out[1], ne=out[2], nl=out[3], niter=out[4], overflow=out[5], max_ncon=out[6], max_nefc=out[7])

    @property
    def ncon(self):
        return self.info()["ncon"]

    @property
    def nefc(self):
        return self.info()["nefc"]

    @property
    def efc_force(self):
        return self.efc_force_buf[: self.nefc]

    def contacts(self):
        res = []
        iv = (C.c_int * 4)()
        rv = (C.c_double * 16)()
        for i in range(self.ncon):
            self.L.mjl_data_contact(self.ptr, i, iv, rv)
            res.append(dict(geom1=iv[0], geom2=iv[1], dim=iv[2], efc_address=iv[3], dist=rv[0],
                            pos=np.array(rv[1:4]), frame=np.array(rv[4:13]), mu=rv[13]))
        return res

    def reset(self):
        self.L.mjl_reset_data(self.om.ptr, self.ptr)

    def forward(self):
        self.L.mjl_forward(self.om.ptr, self.ptr)

    def step(self, n=1):
        self.L.mjl_step_n(self.om.ptr, self.ptr, n)

    def bench_env_steps(self, n, rng_state, lo, hi):
        """mjl_bench_env_steps: n random-action env-steps (5 substeps + forward each) in one C call; rng_state = c_ulonglong"""
        lo = np.ascontiguousarray(lo, dtype=np.float64); hi = np.ascontiguousarray(hi, dtype=np.float64)
        self.L.mjl_bench_env_steps(self.om.ptr, self.ptr, int(n), C.byref(rng_state), lo.ctypes.data, hi.ctypes.data)

    def kinematics(self):
        self.L.mjl_kinematics(self.om.ptr, self.ptr)

    def __del__(self):
        try:
            self.L.mjl_data_free(self.ptr)
        except Exception:
            pass
