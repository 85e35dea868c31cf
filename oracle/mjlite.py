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
_SRCS = ["mjl_core.c", "mjl_collide.c"]


def build(force=False):
    """compile the C restatement; safe under concurrent callers (pytest-xdist workers): one builder at a time, the compiler writes
    a temporary file that is renamed into place"""
    import fcntl
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
                subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-shared", "-fPIC", "-ffp-contract=off", "-o", tmp] + srcs + ["-lm"])
                os.replace(tmp, _LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
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
        _lib = L
    return _lib


# acceleration tables of the product's hull support function (support cells: metaworld_amd/hullcells.py, mjcf.py add_mesh_cells;
# the hull graph / start cube map of model files written before round 4).  The oracle scans every hull vertex
# (mjl_collide.c support()) and takes none of them.
PRODUCT_ONLY_ARRAYS = {"mesh_nbradr", "mesh_nbr", "mesh_start", "mesh_hill", "mesh_celladr", "mesh_cellid"}


class OracleModel:
    """Holds a C MjlModel built from a metaworld_amd.mjcf.Model."""

    def __init__(self, model):
        L = lib()
        self.src = model
        self.ptr = L.mjl_model_new()
        for k, v in model.arrays.items():
            if k in PRODUCT_ONLY_ARRAYS:
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
        p = lib().mjl_model_real_ptr(self.ptr, name.encode(), C.byref(n))
        assert p, name
        a = np.ctypeslib.as_array(p, shape=(n.value,))
        return a.reshape(shape) if shape else a

    def __del__(self):
        try:
            lib().mjl_model_free(self.ptr)
        except Exception:
            pass


class OracleData:
    def __init__(self, om: OracleModel):
        self.om = om
        self.ptr = lib().mjl_data_new(om.ptr)
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
        p = lib().mjl_data_real_ptr(self.om.ptr, self.ptr, name.encode(), C.byref(n))
        assert p, name
        a = np.ctypeslib.as_array(p, shape=(n.value,))
        return a.reshape(shape) if shape else a

    @property
    def time(self):
        return float(self.time_buf[0])

    def info(self):
        out = (C.c_int * 8)()
        lib().mjl_data_info(self.ptr, out)
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
            lib().mjl_data_contact(self.ptr, i, iv, rv)
            res.append(dict(geom1=iv[0], geom2=iv[1], dim=iv[2], efc_address=iv[3], dist=rv[0],
                            pos=np.array(rv[1:4]), frame=np.array(rv[4:13]), mu=rv[13]))
        return res

    def reset(self):
        lib().mjl_reset_data(self.om.ptr, self.ptr)

    def forward(self):
        lib().mjl_forward(self.om.ptr, self.ptr)

    def step(self, n=1):
        lib().mjl_step_n(self.om.ptr, self.ptr, n)

    def kinematics(self):
        lib().mjl_kinematics(self.om.ptr, self.ptr)

    def __del__(self):
        try:
            lib().mjl_data_free(self.ptr)
        except Exception:
            pass
