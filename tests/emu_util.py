"""Run drstencil-emitted HIP kernels on the CPU (tests/emu/hip/hip_runtime.h fibers) and
compare with the oracle.  TEST INFRASTRUCTURE: lets the not-gpu suite check the emitted
index math, guards, LDS exchange and register rotation without a GPU."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRSTENCIL = os.path.join(ROOT, "bin", "drstencil")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
EMU_INC = os.path.join(ROOT, "tests", "emu")
SUPPORT = os.path.join(ROOT, "drstencil_amd", "csrc", "support")


def build_emulated(workdir, stc_path, options):
    """drstencil <options> -> emitted source -> host shared object. Returns ctypes lib."""
    stc_dir, stc = os.path.split(os.path.abspath(stc_path))
    tag = hashlib.md5((" ".join(options) + open(stc_path).read()).encode()).hexdigest()[:12]
    src = os.path.join(str(workdir), "k_%s.hip" % tag)
    so = os.path.join(str(workdir), "k_%s_emu.so" % tag)
    p = subprocess.run([DRSTENCIL] + list(options) + ["-o", src, stc], cwd=stc_dir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and os.path.exists(src), (p.returncode, p.stdout)
    subprocess.check_call([CLANG, "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-DDRS_EMULATE", "-DDRS_PLUGIN",
                           "-I" + EMU_INC, "-I" + SUPPORT, "-x", "c++", src, "-o", so])
    lib = ctypes.CDLL(so)
    for n in ("drs_plugin_launch", "drs_plugin_launch_gold"):
        getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.drs_plugin_info.restype = ctypes.c_char_p
    return lib


def run_emulated(lib, A, B, iterations, step, gold=False):
    fn = lib.drs_plugin_launch_gold if gold else lib.drs_plugin_launch
    n = 0
    t = 0
    while t < iterations:
        assert fn(A.ctypes.data, B.ctypes.data, None) == 0
        assert fn(B.ctypes.data, A.ctypes.data, None) == 0
        n += 2
        t += 2 * step
    return n
