"""TEST INFRASTRUCTURE - ctypes loader of the C restatement (roboy_oracle.c).

Used by tests/ (second witness of the spec) and by bench.py's cpu_baseline leg
("port": timed on the GPU box's host cores).  Never imported by the product.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


class COracle:
    def __init__(self, desc, precision="f64"):
        path = os.path.join(_HERE, "liboracle_%s.so" % precision)
        if not os.path.exists(path):
            build()
        self.lib = ctypes.CDLL(path)
        self.real = np.float64 if precision == "f64" else np.float32
        assert self.lib.orc_real_size() == np.dtype(self.real).itemsize
        self._desc = desc
        self.n_q, self.n_t = desc.n_q, desc.n_t
        self._m = ctypes.c_void_p()
        rc = self.lib.orc_create(ctypes.byref(desc.as_c_struct()), ctypes.byref(self._m))
        if rc:
            raise RuntimeError("orc_create failed: %d" % rc)
        self.lib.orc_destroy.restype = None

    def __del__(self):
        try:
            if self._m:
                self.lib.orc_destroy(self._m)
                self._m = None
        except Exception:
            pass

    def _p(self, a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def rest_lengths(self):
        l0 = np.empty(self.n_t, self.real)
        self.lib.orc_rest_lengths(self._m, self._p(l0))
        return l0

    def tendon_geometry(self, q):
        q = np.ascontiguousarray(q, dtype=self.real).reshape(-1, self.n_q)
        n = q.shape[0]
        length = np.empty((n, self.n_t), self.real)
        L = np.empty((n, self.n_t, self.n_q), self.real)
        self.lib.orc_geometry(self._m, ctypes.c_long(n), self._p(q), self._p(length), self._p(L))
        return length, L

    def step(self, q, qd, setpoint, step_size=0.1, integrator=0, n_substeps=1, threads=1):
        q = np.array(q, dtype=self.real).reshape(-1, self.n_q)
        qd = np.array(qd, dtype=self.real).reshape(-1, self.n_q)
        sp = np.ascontiguousarray(setpoint, dtype=self.real).reshape(-1, self.n_t)
        n = q.shape[0]
        feas = np.empty(n, np.uint8)
        rc = self.lib.orc_step(self._m, ctypes.c_long(n), self._p(q), self._p(qd), self._p(sp),
                               self._p(feas), ctypes.c_double(step_size), int(integrator),
                               int(n_substeps), int(threads))
        if rc:
            raise RuntimeError("orc_step failed: %d" % rc)
        return q, qd, feas.astype(bool)

    def step_inplace(self, q, qd, sp, feas, step_size=0.1, integrator=0, n_substeps=1, threads=1):
        """Timing form: caller-owned contiguous arrays of this oracle's dtype, updated in place."""
        return self.lib.orc_step(self._m, ctypes.c_long(q.shape[0]), self._p(q), self._p(qd), self._p(sp),
                                 self._p(feas), ctypes.c_double(step_size), int(integrator),
                                 int(n_substeps), int(threads))
