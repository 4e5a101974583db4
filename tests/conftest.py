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


def pytest_sessionstart(session):
    """GPU sessions: make sure every kernel the gpu tests use is built (normally a cache hit:
    __graft_entry__.build() prebuilds them) BEFORE anything initialises HIP -- a process that
    has touched the GPU must not fork/exec hipcc."""
    markexpr = session.config.getoption("-m") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import drstencil_amd as drs
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
    for j in jobs:
        drs.Kernel(j)
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
            sp = SlabPlan(70, halo, world, r)
            for v in (sp.top, sp.bot, sp.interior):
                if v is not None and v[1] - v[0] > 2 * halo:
                    sw.kernel(v[1] - v[0])
