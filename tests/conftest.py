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


EMITTED_PROGRAMS = {
    # id: (stencil file under tests/stc, generator options) -- the reference's process contract, SURVEY.md 8b:
    # drstencil [options] --check -o <name>.hip <stc>; hipcc; run; read the program's stdout
    "3d_step2_fp32": ("t3_star", ["--3d", "--dtype", "fp32", "--step", "2", "--check"]),
    "2d_box25_fp64": ("t2_box25", ["--dtype", "fp64", "--check"]),
    "2d_stream_fp32_step2": ("t2_star", ["--dtype", "fp32", "--streaming", "--step", "2", "--prefetch", "--check"]),
    # round 3: the default emission of a fused 63-point stencil (rows order, pinned sums), and loader wavefronts behind LDS-DMA staging
    "3d_step3_fp32_default_rows": ("t3_star", ["--3d", "--dtype", "fp32", "--step", "3", "--check"]),
    "3d_step2_fp64_loader_waves": ("t3_star", ["--3d", "--dtype", "fp64", "--step", "2", "--stage", "dma", "--loader-waves", "2", "--prefetch-depth", "2", "--sn", "16", "--check"]),
}


def _run_emitted_programs(drs, run_them=True):
    """Generate, compile and RUN the standalone emitted programs in child processes, before this process touches
    HIP (a process that has initialised the GPU must not start other programs); tests read the saved stdout."""
    import shutil
    import subprocess
    out = os.path.join(ROOT, "drstencil_amd", "_kcache", "emitted_programs")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(drs.SUPPORT_DIR, "common.hpp"), out)
    for pid, (stc, opts) in EMITTED_PROGRAMS.items():
        src = os.path.join(ROOT, "tests", "stc", stc + ".stc")
        log = os.path.join(out, pid + ".out")
        try:
            # the kernel name is the .stc path minus 4 characters (main.cpp:243-244): run from the spec's directory
            hip, new = os.path.join(out, pid + ".hip"), os.path.join(out, pid + ".new.hip")
            gen = subprocess.run([drs.CLI_PATH] + opts + ["-o", new, stc + ".stc"],
                                 cwd=os.path.dirname(src), capture_output=True, text=True, timeout=60)
            text = "[generator rc=%d]\n%s" % (gen.returncode, gen.stdout)
            if gen.returncode == 0:
                exe = os.path.join(out, pid)
                import hashlib
                body = lambda path: [ln for ln in open(path) if not ln.startswith(("// spec:", "// options:"))]   # banner holds paths
                digest = hashlib.sha256("".join(body(new)).encode()).hexdigest()
                sha = os.path.join(out, pid + ".sha256")       # travels with the binary (the cached .hip sources do not: .gpurunignore)
                fresh = os.path.exists(exe) and os.path.exists(sha) and open(sha).read().strip() == digest
                os.replace(new, hip)
                if fresh:       # same source as the binary that travelled with the tree (built by __graft_entry__.build())
                    cc = subprocess.CompletedProcess([], 0, "", "")
                else:
                    cc = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                                         "-o", exe, hip], capture_output=True, text=True, timeout=300)
                text += "[hipcc rc=%d]\n%s" % (cc.returncode, cc.stderr[-2000:])
                if cc.returncode == 0:
                    with open(sha, "w") as f:
                        f.write(digest + "\n")
                if cc.returncode == 0 and run_them:
                    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
                    text += "[program rc=%d]\n%s%s" % (run.returncode, run.stdout, run.stderr[-1000:])
        except Exception as e:      # recorded, so that the test fails with the reason instead of the session
            text = "[exception] %r" % (e,)
        with open(log, "w") as f:
            f.write(text)


def _run_tuner_smoke():
    """A six-configuration tuner search on a small grid, in a child process started before this process touches HIP
    (the tuner builds with hipcc in its own worker pool and measures with HIP events): generate -> compile -> measure ->
    results.jsonl + duration.log, the flow of the reference's benchmarks/<stencil>/tuning.py:102-142."""
    import shutil
    import subprocess
    out = os.path.join(ROOT, "drstencil_amd", "_kcache", "tuner_smoke")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out, exist_ok=True)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "drstencil_amd", "tuner", "tuning.py"), os.path.join(ROOT, "tests", "stc", "t3_star.stc"),
                            "--3d", "--dtype", "fp32", "--steps", "1,2", "--max-configs", "6", "--seed", "3", "--out", out, "--profile-top", "1",
                            "--write-defaults", "--defaults-table", os.path.join(out, "tuned_defaults.tsv")],      # the tuner -> generator loop, into a scratch table
                           capture_output=True, text=True, timeout=600)
        text = "[tuner rc=%d]\n%s%s" % (r.returncode, r.stdout[-4000:], r.stderr[-2000:])
    except Exception as e:
        text = "[exception] %r" % (e,)
    with open(os.path.join(out, "stdout.txt"), "w") as f:
        f.write(text)


def _run_reference_flow(drs):
    """The reference's tuner <-> generator flow for ONE configuration (benchmarks/3d7pt_star/tuning.py:132-137,
    compile_run.sh:1-5, getGpuMetrics.py:4-38) in child processes: ./drstencil --3d <cfg> --check -o ./cu/<name>.hip x.stc;
    ./compile_run.sh <name> (hipcc, then the emitted program under rocprofv3: trace, FETCH_SIZE and WRITE_SIZE runs);
    getGpuMetrics.py <name> -> gpuMetrics.csv + duration.log."""
    import shutil
    import subprocess
    tdir = os.path.join(ROOT, "drstencil_amd", "tuner")
    out = os.path.join(ROOT, "drstencil_amd", "_kcache", "reference_flow")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "cu"))
    shutil.copy(os.path.join(drs.SUPPORT_DIR, "common.hpp"), os.path.join(out, "cu"))
    shutil.copy(os.path.join(ROOT, "tests", "stc", "t3_star.stc"), out)
    name = "fu2d2bx64y4sn16u4bmx4bmy2mf5pxdm2r"       # --dist 2 on the command line: the reuse schedule
    log = []
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for cmd in ([drs.CLI_PATH, "--3d", "--dtype", "fp32", "--bx", "64", "--by", "4", "--sn", "16", "--stream-unroll", "4", "--step", "2", "--dist", "2",
                     "--block-merge-x", "4", "--block-merge-y", "2", "--merge-forward", "5", "--prefetch", "--xrim", "dpp", "--xcd-remap", "2",
                     "--check", "-o", "./cu/%s.hip" % name, "t3_star.stc"],
                    ["bash", os.path.join(tdir, "compile_run.sh"), name],
                    ["bash", os.path.join(tdir, "getGpuMetrics.sh")]):
            r = subprocess.run(cmd, cwd=out, env=env, capture_output=True, text=True, timeout=600)
            log.append("[%s rc=%d]\n%s%s" % (os.path.basename(cmd[0] if cmd[0] != "bash" else cmd[1]), r.returncode, r.stdout[-1500:], r.stderr[-1500:]))
            if r.returncode != 0:
                break
    except Exception as e:
        log.append("[exception] %r" % (e,))
    with open(os.path.join(out, "flow.txt"), "w") as f:
        f.write("\n".join(log))


def _run_c_host(drs):
    """tests/native/capi_gpu_host.c (the INTEGRATION.md C example as a program) compiled with gcc and run as a child
    process before this process touches HIP; the test reads its stdout."""
    import subprocess
    out = os.path.join(ROOT, "drstencil_amd", "_kcache", "c_host")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "capi_gpu_host")
    libdir = os.path.dirname(drs.LIB_PATH)
    try:
        cc = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                             os.path.join(ROOT, "tests", "native", "capi_gpu_host.c"), "-o", exe, "-L", libdir, "-ldrstencil_amd", "-Wl,-rpath," + libdir,
                             "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True, timeout=120)
        text = "[gcc rc=%d]\n%s" % (cc.returncode, cc.stderr[-1500:])
        if cc.returncode == 0:
            r = subprocess.run([exe, os.path.join(ROOT, "tests", "stc", "t3_star.stc"), "70", "45", "530", "2"], capture_output=True, text=True, timeout=300)
            text += "[host rc=%d]\n%s%s" % (r.returncode, r.stdout, r.stderr[-800:])
        # the N > 1 entry points from plain C (tests/native/capi_slab_host.c): a middle rank of 3 on this GPU, neighbours = itself
        exe2 = os.path.join(out, "capi_slab_host")
        cc = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                             os.path.join(ROOT, "tests", "native", "capi_slab_host.c"), "-o", exe2, "-L", libdir, "-ldrstencil_amd", "-Wl,-rpath," + libdir,
                             "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True, timeout=120)
        text += "[slab gcc rc=%d]\n%s" % (cc.returncode, cc.stderr[-1500:])
        if cc.returncode == 0:
            r = subprocess.run([exe2, os.path.join(ROOT, "tests", "stc", "t3_star.stc"), "45", "530"], capture_output=True, text=True, timeout=300)
            text += "[slab host rc=%d]\n%s%s" % (r.returncode, r.stdout, r.stderr[-800:])
    except Exception as e:
        text = "[exception] %r" % (e,)
    with open(os.path.join(out, "stdout.txt"), "w") as f:
        f.write(text)


def _run_gpus_host():
    """`drstencil --gpus N`: the emitted N-GPU host generated, compiled and run on this one GPU (scripts/try_gpus_host.sh) before this
    process touches HIP: a rehearsed middle rank, the forking form (which must give up cleanly: ranks 1..3 have no GPU), a one-rank
    world with --check, and the 2D y-slab forms."""
    import subprocess
    out = os.path.join(ROOT, "drstencil_amd", "_kcache", "gpus_host")
    os.makedirs(out, exist_ok=True)
    try:
        r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "try_gpus_host.sh"), out], capture_output=True, text=True, timeout=600, cwd=ROOT,
                           env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
        text = "[rc=%d]\n%s\n[stderr]\n%s" % (r.returncode, r.stdout, r.stderr[-1500:])
    except Exception as e:
        text = "[exception] %r" % (e,)
    with open(os.path.join(out, "stdout.txt"), "w") as f:
        f.write(text)


def _run_bench_rehearsal():
    """bench.py's N > 1 branch as a child process (rank 1 of 4 on this one GPU, self-neighbour exchange through a real RCCL
    process group), before this process touches HIP: the test reads what arrived on its stdout."""
    import subprocess
    out = os.path.join(ROOT, "drstencil_amd", "_kcache", "bench_rehearsal")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, DRS_REHEARSE="1/4", MASTER_ADDR="127.0.0.1")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=300)
        stdout, text = r.stdout, "[rc=%d]\n%s" % (r.returncode, r.stderr[-1500:])
    except Exception as e:
        stdout, text = "", "[exception] %r" % (e,)
    with open(os.path.join(out, "stdout.txt"), "w") as f:
        f.write(stdout)
    with open(os.path.join(out, "stderr.txt"), "w") as f:
        f.write(text)
    # ... and the REAL multi-process form, as the driver launches it: two rank processes under torch.distributed.run, both computing on this
    # one GPU, the halo planes through a gloo group staged in host memory (DRS_BENCH_BACKEND=gloo DRS_BENCH_ONE_GPU=1): the launcher
    # environment, a first and a last rank with their own kernels, the exchange choreography and the decomposed-vs-single-domain check
    # across real processes -- everything of an N = 2 run but the RCCL transport
    env2 = dict(os.environ, DRS_BENCH_BACKEND="gloo", DRS_BENCH_ONE_GPU="1")
    try:
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                            os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                           env=env2, capture_output=True, text=True, timeout=420, cwd=ROOT)
        stdout, text = r.stdout, "[rc=%d]\n%s" % (r.returncode, r.stderr[-1500:])
    except Exception as e:
        stdout, text = "", "[exception] %r" % (e,)
    # ... and the single-GPU line with a step count beyond the overflow horizon of the data (coefficients sum to 1.5: all inf after ~218 time
    # steps): the timed loop must run in chunks, each from the restored pristine input
    try:
        r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "1", "--no-cpu-baseline", "--headline-only"],
                            capture_output=True, text=True, timeout=300, cwd=ROOT)
        with open(os.path.join(out, "n1_steps60_stdout.txt"), "w") as f:
            f.write(r1.stdout)
        with open(os.path.join(out, "n1_steps60_stderr.txt"), "w") as f:
            f.write("[rc=%d]\n%s" % (r1.returncode, r1.stderr[-1500:]))
    except Exception as e:
        with open(os.path.join(out, "n1_steps60_stderr.txt"), "w") as f:
            f.write("[exception] %r" % (e,))
    with open(os.path.join(out, "two_ranks_stdout.txt"), "w") as f:
        f.write(stdout)
    with open(os.path.join(out, "two_ranks_stderr.txt"), "w") as f:
        f.write(text)


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
    _run_emitted_programs(drs)
    _run_tuner_smoke()
    _run_reference_flow(drs)
    _run_c_host(drs)
    _run_gpus_host()
    _run_bench_rehearsal()
    from gpu_cases import all_build_args, golden_args
    from helpers import golden_cases, load_golden
    jobs = all_build_args()
    for c in golden_cases():
        meta, *_ = load_golden(c)
        for rows in (False, True):
            opts, stc = golden_args(c, meta, rows=rows)
            jobs.append(opts + [stc])
    from gpu_cases import stc as stcp, fuzz_sample_jobs
    fuzz_args = [j[3] for j in fuzz_sample_jobs()]     # some of these are refused (scratch spills): not an error
    jobs.append(["--dtype", "fp32", "--streaming", "--xrim", "lds", stcp("t2_box25")])
    jobs.append(["--dtype", "fp32", "--streaming", "--xrim", "dpp", stcp("t2_box25")])
    # cache hits return at once; misses are compiled in parallel worker processes (hipcc), all of it
    # before this process initialises HIP
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        bad = [r for r in ex.map(_build_kernel, jobs) if r]
        list(ex.map(_build_kernel, fuzz_args))
    # a kernel that cannot be built fails the tests that use it (their drs.Kernel call raises: after HIP is up a cache miss is an
    # error, not a compiler run) -- not the whole session
    if bad:
        sys.stderr.write("conftest: %d kernel build(s) failed at session start:\n%s\n" % (len(bad), "\n".join(bad)))
    # slab-view kernels of test_slab_decomposition_on_one_gpu (compiled here, before HIP is up)
    from drstencil_amd.multigpu import HipSweep, SlabPlan
    global SLAB_CACHE
    SLAB_CACHE = os.path.join(ROOT, "drstencil_amd", "_kcache")
    from gpu_cases import SLAB_CASES
    import bench
    from gpu_cases import C4_SLAB_WORLDS

    def prebuild(what, fn):
        try:
            fn()
        except Exception as e:       # the tests that need these kernels fail on their own drs.Kernel call; the session goes on
            sys.stderr.write("conftest: prebuilding %s failed: %s\n" % (what, str(e)[-300:]))

    for cid, world, stencil, ndim, opts in SLAB_CASES:
        def slab_case(world=world, stencil=stencil, ndim=ndim, opts=opts):
            step = int(opts[opts.index("--step") + 1]) if "--step" in opts else 1
            spec = drs.Spec(stcp(stencil), ndim, step)
            cut = spec.dims[0] if ndim == 3 else spec.dims[1]
            sw = HipSweep(stcp(stencil), opts, SLAB_CACHE)
            drs.Kernel(opts + [stcp(stencil)])
            for r in range(world):
                for every in (1, 2):
                    sw.prebuild(SlabPlan(cut, spec.halo, world, r, every))
        prebuild("slab case " + cid, slab_case)
    # the one-rank world of test_rccl_exchange_choreography_on_one_gpu's bench.verify_slab_run call: the whole-grid view
    prebuild("one-rank view", lambda: HipSweep(stcp("t3_star"), ["--3d", "--dtype", "fp32", "--step", "2", "--sn", "16"], SLAB_CACHE).kernel(drs.Spec(stcp("t3_star"), 3, 2).dims[0]))
    # ... and of test_c4_slab_views_at_full_size: what bench.py --gpus 2/4/8 launches
    for world in C4_SLAB_WORLDS:
        def c4(world=world):
            sw = HipSweep(bench.WORKLOADS["c4"]["stc"], bench.slab_options("c4", world), SLAB_CACHE, alone_opts=bench.slab_alone_options("c4", world))
            for r in range(world):
                for every in (1, 2):
                    sw.prebuild(SlabPlan(1024, 2, world, r, every))
        prebuild("C4 slab views, world %d" % world, c4)
    for world in (2, 4):        # test_c2_yslab_views_at_full_size
        def c2(world=world):
            sw = HipSweep(bench.WORKLOADS["c2"]["stc"], bench.slab_options("c2", world), SLAB_CACHE)
            for r in range(world):
                for every in (1, 2):
                    sw.prebuild(SlabPlan(8192, 1, world, r, every))
        prebuild("C2 y-slab views, world %d" % world, c2)
