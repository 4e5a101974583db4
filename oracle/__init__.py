"""CPU oracle: ctypes front end to oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package drstencil_amd never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, so], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        L.drso_spec_size.restype = ctypes.c_size_t
        L.drso_parse_stc.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
        L.drso_fuse.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.drso_launches.argtypes = [ctypes.c_int, ctypes.c_int]
        L.drso_fill_random_f64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.drso_fill_random_f32.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        for n in ("drso_sweep_f64", "drso_sweep_f32"):
            getattr(L, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        for n in ("drso_run_f64", "drso_run_f32"):
            getattr(L, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        for n in ("drso_check_f64", "drso_check_f32"):
            f = getattr(L, n)
            f.restype = ctypes.c_double
            f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]
        L.drso_point.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3 + [ctypes.POINTER(ctypes.c_double)]
        L.drso_dims.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
        L.drso_set_dims.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
        L.drso_set_iterations.argtypes = [ctypes.c_void_p, ctypes.c_int]
        for n in ("drso_halo", "drso_iterations", "drso_npts"):
            getattr(L, n).argtypes = [ctypes.c_void_p]
        L.drso_srand.argtypes = [ctypes.c_uint]
        for n in ("drso_first_touch_f64", "drso_first_touch_f32"):
            getattr(L, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        for n in ("drso_copy_f64", "drso_copy_f32"):
            getattr(L, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.drso_isa.restype = ctypes.c_char_p
        L.drso_set_threads.argtypes = [ctypes.c_int]
        L.drso_set_avx512_path.argtypes = [ctypes.c_int]
        _LIB = L
    return _LIB


class Spec:
    """A parsed + fused stencil spec (drso_spec)."""

    def __init__(self, stc_path, ndim, step=1):
        L = lib()
        self._buf = ctypes.create_string_buffer(L.drso_spec_size())
        self.p = ctypes.addressof(self._buf)
        rc = L.drso_parse_stc(os.fsencode(stc_path), ndim, self.p)
        if rc != 0:
            raise IOError("Error opening stencil file.")
        L.drso_fuse(self.p, step)
        self.ndim, self.step = ndim, step

    @property
    def halo(self):
        return lib().drso_halo(self.p)

    @property
    def iterations(self):
        return lib().drso_iterations(self.p)

    @iterations.setter
    def iterations(self, v):
        lib().drso_set_iterations(self.p, int(v))

    @property
    def dims(self):
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().drso_dims(self.p, a, b, c)
        return (a.value, b.value, c.value)

    def set_dims(self, L_, M, N):
        lib().drso_set_dims(self.p, L_, M, N)

    @property
    def shape(self):
        d = self.dims
        return d if self.ndim == 3 else d[1:]

    @property
    def points(self):
        out = []
        k, j, i, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        for p in range(lib().drso_npts(self.p)):
            lib().drso_point(self.p, p, k, j, i, c)
            out.append(((k.value, j.value, i.value), c.value))
        return out

    @property
    def launches(self):
        return lib().drso_launches(self.iterations, self.step)

    def interior(self, arr):
        h = self.halo
        sl = tuple(slice(h, n - h) for n in arr.shape)
        return arr[sl]


def _sfx(a):
    assert a.flags.c_contiguous
    return {np.dtype("float64"): "f64", np.dtype("float32"): "f32"}[a.dtype]


def fill_random(shape, dtype=np.float64, seed=1):
    """common.hpp:9-32 input stream (glibc rand(), seed 1 == unseeded)."""
    a = np.empty(shape, dtype=dtype)
    lib().drso_srand(seed)
    getattr(lib(), "drso_fill_random_" + _sfx(a))(a.ctypes.data, a.size)
    return a


def sweep(spec, src, dst, contract=1):
    getattr(lib(), "drso_sweep_" + _sfx(src))(spec.p, src.ctypes.data, dst.ctypes.data, int(contract))


def run(spec, A, B, contract=1):
    """Whole ping-pong run in place; returns the number of launches."""
    assert A.dtype == B.dtype and tuple(A.shape) == tuple(spec.shape)
    return getattr(lib(), "drso_run_" + _sfx(A))(spec.p, A.ctypes.data, B.ctypes.data, int(contract))


def check(spec, out, ref):
    """checkError2D/3D metrics: dict(rms, max_abs, max_idx, max_rel)."""
    ma, mi, mr = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
    rms = getattr(lib(), "drso_check_" + _sfx(out))(spec.p, out.ctypes.data, ref.ctypes.data, ma, mi, mr)
    return dict(rms=rms, max_abs=ma.value, max_idx=mi.value, max_rel=mr.value)


def threads():
    return lib().drso_threads()


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container limited to 16 CPUs on a
    128-thread host still reports 128 to OpenMP)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def set_threads(n):
    lib().drso_set_threads(int(n))


def set_avx512_path(on):
    """Switch the register-blocked AVX-512 path of the contracted sweep on / off (tests compare the two bit for bit)."""
    lib().drso_set_avx512_path(1 if on else 0)


def isa():
    """The clone of the sweep the loader picked on this host: avx512f | avx2+fma | baseline."""
    return lib().drso_isa().decode()


def empty_first_touched(spec, dtype):
    """A zeroed array of the spec's shape whose pages were first written by the OpenMP threads that will sweep them (same static
    (k, j) schedule as the sweep): on a two-socket host every thread's rows then live on its own NUMA node."""
    a = np.empty(spec.shape, dtype=dtype)       # untouched pages (large allocations come straight from mmap)
    getattr(lib(), "drso_first_touch_" + _sfx(a))(spec.p, a.ctypes.data)
    return a


def copy_into(spec, dst, src):
    """dst[:] = src with the sweep's thread schedule."""
    assert dst.dtype == src.dtype and dst.shape == src.shape and src.flags.c_contiguous
    getattr(lib(), "drso_copy_" + _sfx(dst))(spec.p, dst.ctypes.data, src.ctypes.data)
