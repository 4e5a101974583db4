import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _build_kernel(args):
    import drstencil_amd as drs
    try:
        drs.Kernel(args)
        return None
    except Exception as e:
        return "%s: %s" % (" ".join(args), str(e)[-300:])


def pytest_sessionstart(session):
    """GPU sessions: make sure every kernel the gpu tests use is built (normally a cache hit:
    __graft_entry__.build() prebuilds them) BEFORE anything initialises HIP -- a process that
    has touched the GPU must not fork/exec hipcc."""
    import subprocess
    import drstencil_amd as drs
    if not os.path.exists(drs.LIB_PATH) or not os.path.exists(drs.CLI_PATH):
        # a checkout without built artefacts: build the host code first (g++, seconds) -- CPU and GPU sessions alike
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drstencil_amd", "csrc")], stdout=subprocess.DEVNULL)
    markexpr = session.config.getoption("-m") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_cases import all_build_args, golden_args
    from helpers import golden_cases, load_golden
    jobs = all_build_args()
    for c in golden_cases():
        meta, *_ = load_golden(c)
        opts, stc = golden_args(c, meta)
        jobs.append(opts + [stc])
    from gpu_cases import stc as stcp
    jobs.append(["--dtype", "fp32", "--streaming", "--xrim", "lds", stcp("t2_box25")])
    jobs.append(["--dtype", "fp32", "--streaming", "--xrim", "dpp", stcp("t2_box25")])
    # cache hits return at once; misses are compiled in parallel worker processes (hipcc), all of it
    # before this process initialises HIP
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        bad = [r for r in ex.map(_build_kernel, jobs) if r]
    assert not bad, "kernel builds failed:\n" + "\n".join(bad)
    # slab-view kernels of test_slab_decomposition_on_one_gpu (compiled here, before HIP is up)
    import tempfile
    from drstencil_amd.multigpu import HipSweep, SlabPlan
    global SLAB_CACHE
    SLAB_CACHE = os.path.join(ROOT, "drstencil_amd", "_kcache")
    for world, opts, halo in [(2, ["--3d", "--dtype", "fp32", "--sn", "8"], 1), (3, ["--3d", "--dtype", "fp32", "--step", "2", "--sn", "16"], 2),
                              (2, ["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4", "--sn", "16", "--prefetch"], 2)]:
        sw = HipSweep(stcp("t3_star"), opts, SLAB_CACHE)
        drs.Kernel(opts + [stcp("t3_star")])
        for r in range(world):
            for every in (1, 2):
                for lv in SlabPlan(70, halo, world, r, every).views():
                    sw.kernel(lv)
