"""C-ABI surface (loads, exports every symbol include/drstencil_amd.h declares; no compute
without a GPU), host helpers, and the tuner's space/naming logic."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest

import drstencil_amd as drs
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "drstencil_amd.h")).read()
    declared = set(re.findall(r"\b(drs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"drs_spec", "drs_kernel"}
    L = ctypes.CDLL(drs.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(L, sym), sym
    assert declared == set(drs.EXPORTS)
    assert drs.lib().drs_version().startswith(b"drstencil-amd")


def test_plain_c_host_links_and_uses_the_abi(tmp_path):
    """tests/native/capi_host.c, compiled with gcc against include/drstencil_amd.h and the shared library: generator as
    a function, error path, stencil IR getters, input fill -- the host-side surface, no GPU."""
    import subprocess
    exe = str(tmp_path / "capi_host")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "capi_host.c"),
                           "-o", exe, "-L", os.path.dirname(drs.LIB_PATH), "-ldrstencil_amd", "-Wl,-rpath," + os.path.dirname(drs.LIB_PATH)])
    out = subprocess.run([exe, os.path.join(ROOT, "benchmarks", "3d7pt_star", "3d7pt_star.stc")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = dict(ln.split(" ", 1) for ln in out.stdout.splitlines())
    assert lines["version"].startswith("drstencil-amd")
    assert lines["generate"].startswith("rc 0 ") and "has_kernel 1 has_gold 1" in lines["generate"]
    assert lines["illegal"].startswith("rc 255 message Illegal input.")              # main.cpp:129-131
    # 3d7pt_star fused twice: 25 points, Halo 2, Dist 2 (auto), Range 3, iterations 4 -> 2 launches
    assert lines["spec"].startswith("status 0 dims 512 512 512 halo 2 dist 2 range 3 points 25 iterations 4 launches 2 ")
    assert lines["point0"] == "-2 0 0 0.04"
    assert lines["rand0"] == "0.84018773"                                            # first rand()/(RAND_MAX-1), common.hpp:9-45


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(drs, "_lib", None)
    monkeypatch.setattr(drs, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(drs.NativeLibraryMissing):
        drs.lib()


def test_fill_random_and_check_error_match_oracle():
    for dt in (np.float64, np.float32):
        a = drs.fill_random(np.empty((7, 9, 11), dt))
        b = oracle.fill_random((7, 9, 11), dt)
        assert np.array_equal(a, b)
    assert a.dtype == np.float32
    a64 = drs.fill_random(np.empty(4, np.float64))
    assert a64[0] == 0.84018771754595234
    ref = np.random.default_rng(0).random((9, 10, 11))
    out = ref.copy(); out[4, 5, 6] += 2e-3; out[0, 0, 0] = 9
    m = drs.check_error(out, ref, 1)
    assert m["max_abs"] == pytest.approx(2e-3) and m["max_idx"] == (4 * 10 + 5) * 11 + 6
    assert m["rms"] == pytest.approx(np.sqrt(4e-6 / (7 * 8 * 9)))
    m2 = drs.check_error(ref[0], ref[0], 2)
    assert m2["max_abs"] == 1e-13 and m2["rms"] == 0


def test_kernel_build_reports_generator_errors(tmp_path):
    with pytest.raises(drs.KernelBuildError) as e:
        drs.Kernel(["--3d", str(tmp_path / "missing.stc")])
    assert "Error opening stencil file." in str(e.value)
    stc = os.path.join(ROOT, "benchmarks", "3d9pt_cross", "3d9pt_cross.stc")
    with pytest.raises(drs.KernelBuildError) as e:
        drs.Kernel(["--3d", stc])
    assert "No data to reuse" in str(e.value)


def test_kernel_info_and_work_model():
    stc = os.path.join(ROOT, "tests", "stc", "smoke3.stc")
    k = drs.Kernel(["--3d", "--dtype", "fp32", stc])   # built by build(); cross-compiles if not
    i = k.info
    assert (i["L"], i["M"], i["N"], i["halo"], i["step"], i["dtype"]) == (40, 36, 256, 1, 1, "fp32")
    assert k.updates_per_launch() == 38 * 34 * 254
    assert k.bytes_per_launch() == 2 * 4 * 40 * 36 * 256
    assert os.path.exists(k.path) and k.path.endswith(".so")


def test_kernel_resources_and_scratch_refusal(monkeypatch):
    """The runtime reads the compiler's resource report for every kernel it builds and refuses one that spills to
    scratch (it has outgrown the register file: slow, and the only place the round-1 parity fuzz found miscompiled
    kernels); the automatic geometry for the same heavy stencil fits and loads."""
    stc = os.path.join(ROOT, "tests", "stc", "t3_star.stc")
    k = drs.Kernel(["--3d", "--dtype", "fp32", "--step", "3", stc])                      # 63-point fused stencil, automatic geometry and emission
    # round 3: the rows order with pinned sums, 64 x 8 lanes: two workgroups per CU fit (<= 128 VGPRs); rounds 1-2 needed a 256-lane workgroup
    assert k.info["threads"] == 512 and k.info["order"] == "rows" and k.resources["scratch_bytes_per_lane"] == 0 and 0 < k.resources["vgprs"] <= 128
    # the round-2 emission of the same stencil at 512 lanes: the compiler sinks the FMA chains, keeps 7 planes of windows alive and spills
    spilling = ["--3d", "--dtype", "fp32", "--step", "3", "--order", "taps", "--bx", "32", "--by", "16", "--block-merge-y", "2", "--prefetch-depth", "3", stc]
    monkeypatch.delenv("DRS_ALLOW_SCRATCH", raising=False)
    with pytest.raises(RuntimeError) as e:
        drs.Kernel(spilling)
    assert "spills" in str(e.value) and "scratch" in str(e.value)
    monkeypatch.setenv("DRS_ALLOW_SCRATCH", "1")
    k2 = drs.Kernel(spilling)
    assert k2.resources["scratch_bytes_per_lane"] > 0


def test_scalar_register_spills_are_refused(monkeypatch):
    """A kernel that ran out of scalar registers (plane pointers, descriptors and fp64 coefficient pairs behind v_writelane /
    v_readlane) but not of vector registers: the round-2 fuzz found one of these miscompiled at -O3 (right at -O1, in the
    emulator and with unrelated backend passes off, profiles/r02_fuzz_sgpr_spill_miscompile.md), so the runtime refuses them
    like kernels that spill to scratch."""
    stc = os.path.join(ROOT, "tests", "stc", "t3_odd.stc")
    args = ["--3d", "--dtype", "fp64", "--bx", "16", "--by", "16", "--sn", "64", "--step", "3", "--dist", "3", "--block-merge-x", "2", "--block-merge-y", "4",
            "--prefetch", "--prefetch-depth", "4", "--xcd-remap", "0", "--schedule", "scatter", "--uniform-loads", "2", "--stage", "dma", stc]
    monkeypatch.setenv("DRS_ALLOW_SCRATCH", "1")
    res = drs.Kernel(args).resources
    if not (res["sgpr_spill"] > 0 and res["scratch_bytes_per_lane"] == 0):
        pytest.skip("this compiler fits the kernel's scalar state: %r" % res)
    monkeypatch.delenv("DRS_ALLOW_SCRATCH")
    with pytest.raises(RuntimeError) as e:
        drs.Kernel(args)
    assert "scalar registers" in str(e.value) and "sgpr_spill" in str(e.value)


def test_cache_key_keeps_the_whole_hash_for_long_names(tmp_path):
    """A .stc whose base name is longer than the readable part of the cache key: two option sets (or dtypes) must still
    get two different cached kernels (the round-1 key was a 64-byte buffer, so a long name truncated the hash away and
    an fp32 plugin was handed back for an fp64 request)."""
    import shutil
    long_stc = str(tmp_path / ("a_very_long_stencil_specification_name_that_goes_on_and_on_for_seventy_chars" + ".stc"))
    shutil.copy(os.path.join(ROOT, "tests", "stc", "t2_star.stc"), long_stc)
    cache = str(tmp_path / "cache")
    k32 = drs.Kernel(["--dtype", "fp32", long_stc], cache_dir=cache)
    k64 = drs.Kernel(["--dtype", "fp64", "--bx", "32", long_stc], cache_dir=cache)
    assert k32.path != k64.path
    assert k32.info["dtype"] == "fp32" and k64.info["dtype"] == "fp64" and k64.info["threads"] != k32.info["threads"]
    for k in (k32, k64):
        stem = os.path.basename(k.path)[:-3]
        assert re.fullmatch(r"[A-Za-z0-9_]{1,40}_[0-9a-f]{16}", stem), stem


def test_unverified_and_uncompilable_kernels_are_refused(monkeypatch, tmp_path):
    """Fail closed: (1) a compiler whose resource report cannot be read (here: a DRS_HIPCC wrapper that drops stderr) gives
    a kernel that was never checked for AGPR spilling / scratch, so it is not loaded unless DRS_ALLOW_UNVERIFIED=1;
    (2) DRS_NO_COMPILE=1 turns a cache miss into an error instead of a hipcc child process; (3) --pair-launch kernels
    report the maximum over dr_ and dr2_; (4) --debug-drop-barrier kernels (wrong results by design) need DRS_EXPERIMENTS=1."""
    stc = os.path.join(ROOT, "tests", "stc", "t2_star.stc")
    wrapper = tmp_path / "quiet_hipcc.sh"
    wrapper.write_text("#!/bin/sh\nexec /opt/rocm/bin/hipcc \"$@\" 2>/dev/null\n")
    wrapper.chmod(0o755)
    monkeypatch.setenv("DRS_HIPCC", str(wrapper))
    monkeypatch.delenv("DRS_ALLOW_UNVERIFIED", raising=False)
    with pytest.raises(drs.KernelBuildError) as e:
        drs.Kernel(["--dtype", "fp32", stc], cache_dir=str(tmp_path / "c1"))
    assert "no compiler resource report" in str(e.value) and "DRS_ALLOW_UNVERIFIED" in str(e.value)
    monkeypatch.setenv("DRS_ALLOW_UNVERIFIED", "1")
    k = drs.Kernel(["--dtype", "fp32", stc], cache_dir=str(tmp_path / "c1"))
    assert k.resources["verified"] == 0 and k.resources["vgprs"] == -1
    monkeypatch.delenv("DRS_ALLOW_UNVERIFIED")
    monkeypatch.delenv("DRS_HIPCC")
    # (2)
    monkeypatch.setenv("DRS_NO_COMPILE", "1")
    with pytest.raises(drs.KernelBuildError) as e:
        drs.Kernel(["--dtype", "fp32", stc], cache_dir=str(tmp_path / "c2"))
    assert "not in the cache" in str(e.value) and "DRS_NO_COMPILE" in str(e.value)
    monkeypatch.delenv("DRS_NO_COMPILE")
    # (3)
    t3 = os.path.join(ROOT, "tests", "stc", "t3_star.stc")
    kp = drs.Kernel(["--3d", "--dtype", "fp32", "--pair-launch", "1", t3], cache_dir=str(tmp_path / "c3"))
    assert kp.resources["kernels_reported"] == "dr_+dr2_" and kp.resources["verified"] == 1 and kp.resources["vgprs"] > 0
    # (4)
    monkeypatch.delenv("DRS_EXPERIMENTS", raising=False)
    with pytest.raises(drs.KernelBuildError) as e:
        drs.Kernel(["--3d", "--dtype", "fp32", "--debug-drop-barrier", "2", t3], cache_dir=str(tmp_path / "c3"))
    assert "DRS_EXPERIMENTS" in str(e.value)


def test_every_bench_kernel_is_a_full_size_parity_case():
    """No number bench.py prints may come from a kernel without a full-size parity case: every (workload, options) pair of
    bench.kernels() -- headline, step-1 and temporal kernels, fp64 workloads included -- is in gpu_cases.FULL, and the
    slab-view kernels of the N > 1 branch are covered by test_c4_slab_views_at_full_size for the same world sizes
    __graft_entry__.build() prebuilds."""
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_cases
    full = {(c[2], tuple(c[3])) for c in gpu_cases.FULL}
    ks = bench.kernels()
    assert len(ks) >= 11 and {w for _, w, _ in ks} == set(bench.WORKLOADS)
    for kid, w, opts in ks:
        assert (bench.WORKLOADS[w]["stc"], tuple(opts)) in full, kid
    assert [o + [bench.WORKLOADS[w]["stc"]] for _, w, o in ks] == bench.kernel_arg_sets()
    assert gpu_cases.C4_SLAB_WORLDS == (2, 4, 8)


def test_tuner_space_and_naming():
    from drstencil_amd.tuner import tuning as t
    t.order, t.ndim, t.elem_bytes = 1, 3, 4
    v = (2, 1, (16, 8), 8, 4, False, 1, False, 1, 5, False, "lds", False, 2, False)
    # reference naming/command-line scheme (benchmarks/3d7pt_star/tuning.py:40-78) + our suffix
    assert t.cfgToString(v).startswith("fu2d1bx16y8sn8u4cmx1cmy1mf5")
    assert t.cfgToCommandLine(v).startswith(" --bx 16 --by 8 --sn 8 --stream-unroll 4 --step 2 --dist 1 --cyclic-merge-x 1 --cyclic-merge-y 1 --merge-forward 5")
    vp = v[:10] + (True, "dpp", True, 0, False)
    assert t.cfgToString(vp).startswith("fu2d1bx16y8sn8u4cmx1cmy1mf5p")
    v3 = v[:10] + (3, "dpp", False, 2, False)
    assert t.cfgToString(v3) == "fu2d1bx16y8sn8u4cmx1cmy1mf5p3xdm2" and "--prefetch --prefetch-depth 3" in t.cfgToCommandLine(v3)
    space = t.enumerate_space((1, 2))
    assert len(space) > 100
    assert all(t.FilterParams(s) for s in space)
    names = [t.cfgToString(s) for s in space]
    assert len(set(names)) == len(names)
    # filter rules: dist range, LDS budget, wavefront multiple
    assert not t.FilterParams((2, 3, (64, 4), 64, 4, True, 4, True, 4, 5, False, "lds", False, 2, False))   # dist > step*order
    assert not t.FilterParams((1, 1, (16, 2), 64, 4, True, 4, True, 4, 5, False, "lds", False, 2, False))   # 32 lanes: half a wave
    assert not t.FilterParams((1, 1, (256, 4), 64, 4, True, 4, True, 8, 5, False, "lds", False, 2, False))  # LDS over 160 KiB
    assert not t.FilterParams((1, 1, (66, 15), 32, 4, True, 4, True, 2, 5, False, "dpp", False, 0, False))  # odd lane count without temporal
    assert t.FilterParams((2, 2, (66, 15), 32, 4, True, 4, True, 2, 5, True, "dpp", True, 0, False))        # the bench configuration
    assert t.FilterParams((1, 1, (64, 4), 64, 4, True, 4, True, 4, 5, False, "lds", False, 2, False))
    # 2D space: tile and --streaming kernels
    t.order, t.ndim, t.elem_bytes = 1, 2, 4
    sp2 = t.enumerate_space((1, 2))
    assert any(v[14] for v in sp2) and any(not v[14] for v in sp2)
    assert all(("--streaming" in t.cfgToCommandLine(v)) == v[14] for v in sp2)
    stc2 = os.path.join(ROOT, "benchmarks", "2d5pt_star", "2d5pt_star.stc")
    import random
    random.seed(1)
    for s2 in random.sample(sp2, 20):
        rc, msg, src = drs.generate(["--dtype", "fp32"] + t.cfgToCommandLine(s2).split() + [stc2])
        assert rc == 0 and src, (t.cfgToString(s2), msg)
    t.order, t.ndim, t.elem_bytes = 1, 3, 4
    # every configuration of the space is accepted by the generator
    stc = os.path.join(ROOT, "benchmarks", "3d7pt_star", "3d7pt_star.stc")
    import random
    random.seed(0)
    for s in random.sample(space, 25):
        rc, msg, src = drs.generate(["--3d", "--dtype", "fp32"] + t.cfgToCommandLine(s).split() + [stc])
        assert rc == 0 and src, (t.cfgToString(s), msg)


def test_native_slab_plan_equals_the_python_plan():
    """drs_slab_open (the plan of the N > 1 entry points, C++) against multigpu.SlabPlan for every rank of worlds 2-4 and both
    exchange modes; the view kernels are the ones build() made for the one-GPU slab tests, so nothing is compiled here."""
    from drstencil_amd.multigpu import SlabPlan
    stc = os.path.join(ROOT, "tests", "stc", "t3_star.stc")
    opts = ["--3d", "--dtype", "fp32", "--step", "2", "--sn", "16"]
    spec = drs.Spec(stc, 3, 2)
    for every in (1, 2):
        for rank in range(3):
            s = drs.Slab(opts + [stc], world=3, rank=rank, every=every)
            p = SlabPlan(spec.dims[0], spec.halo, 3, rank, every)
            assert (s.lo, s.hi, s.z0, s.z1, s.Lloc, s.G, s.H, s.every) == (p.lo, p.hi, p.z0, p.z1, p.Lloc, p.G, p.H, p.every)
            i = s.info
            assert i["graph"] == 0 and (i["kernel_pair"] != "") == (p.pair_view() is not None) and (i["kernel_full"] != "") == (every == 2)
            s.close()
    with pytest.raises(drs.KernelBuildError):
        drs.Slab(opts + [stc], world=1, rank=0, every=1, rehearse_world=3)      # only a middle rank can be its own two neighbours


def test_round3_emission_knobs_and_tuner_dimension(tmp_path):
    """Round 3: the emitter bounds live ranges itself.  The knobs are validated by the generator, show up in the kernel info, never
    change the gold kernel, and form the tuner's 17th dimension (`emit`: taps | pin | rows | rowspk) with its own name fragments."""
    import json
    from drstencil_amd.tuner import tuning as t
    stc = os.path.join(ROOT, "tests", "stc", "smoke3.stc")          # 40 x 36 x 256: 16-byte vectors

    def info(args):
        rc, msg, src = drs.generate(args + [stc])
        assert rc == 0 and src, msg
        return json.loads(re.search(r'return "(\{.*\})";', src).group(1).replace('\\"', '"')), src

    base = ["--3d", "--dtype", "fp32", "--step", "3", "--prefetch", "--prefetch-depth", "1", "--bx", "32", "--by", "8", "--block-merge-y", "2"]
    i0, s0 = info(base + ["--order", "taps"])
    assert (i0["order"], i0["packed"], i0["pinned"], i0["unroll"]) == ("taps", 0, 0, 14) and "DRS_PIN" not in s0
    assert info(base)[0]["order"] == "rows" and info(base)[0]["packed"] == 0, "a fused stencil beyond 25 taps takes the rows order by default (unpacked)"
    assert info(["--3d", "--dtype", "fp32", "--step", "2"])[0]["order"] == "taps", "the memory-bound step-2 kernel keeps round 2's emission"
    i1, s1 = info(base + ["--order", "rows"])
    assert (i1["order"], i1["packed"], i1["pinned"]) == ("rows", 1, 1)
    assert i1["unroll"] == 8, "Range 7 rotates through 8 sets with --order rows: 8 plane bodies instead of 14"
    assert "DRS_ROW_FENCE();" in s1 and "DRS_PIN(cq0_" in s1 and "__builtin_elementwise_fma((vec2_t)" in s1 and "typedef real_t vec2_t" in s1
    assert "bound_ctrl" not in s1 and "0x138, 0xf, 0xf, true)" in s1            # DPP moves without an `old` operand
    i2, s2 = info(base + ["--order", "rows", "--pack", "0", "--rot-mod", "-1"])
    assert (i2["packed"], i2["unroll"]) == (0, 14) and "vec2_t" not in s2
    i3, s3 = info(base + ["--order", "taps", "--pin", "1", "--rot-mod", "9"])
    assert (i3["order"], i3["pinned"], i3["unroll"]) == ("taps", 1, 18) and "DRS_PIN(c0_" in s3
    # the gold kernel and the arithmetic never depend on the emission
    gold = lambda src: src[src.index("// naive reference kernel"):src.index("// ---- launch entry points")]
    assert gold(s0) == gold(s1) == gold(s2) == gold(s3)
    # fp64 is never packed; a 2D tile kernel takes the rows order too
    assert info(["--3d", "--dtype", "fp64", "--step", "2", "--order", "rows"])[0]["packed"] == 0
    assert info(["--3d", "--dtype", "fp64", "--step", "2", "--order", "rows"])[0]["pinned"] == 1
    # what the rows order cannot be combined with is refused like any invalid configuration (exit 255, the reference's message)
    for bad in (["--order", "rows", "--dist", "2"], ["--order", "rows", "--schedule", "window"], ["--order", "rows", "--stage", "dma"],
                ["--order", "rows", "--cyclic-merge-y", "2"], ["--order", "columns"], ["--loader-waves", "2"],
                ["--stage", "dma", "--loader-waves", "9"], ["--stage", "dma", "--loader-waves", "2", "--schedule", "window"]):
        rc, msg, src = drs.generate(["--3d", "--dtype", "fp32", "--step", "2"] + bad + [stc])
        assert rc == 255 and not src and ("Invalid configuration!" in msg or "Illegal input." in msg), (bad, rc, msg)
    # loader wavefronts: 2 extra wavefronts, a ring of prefetch-depth + 1 slots, every plane request unconditional
    i4, s4 = info(["--3d", "--dtype", "fp32", "--step", "2", "--stage", "dma", "--loader-waves", "2", "--prefetch-depth", "3", "--bx", "32", "--by", "8", "--block-merge-y", "2"])
    assert i4["threads"] == 256 and i4["lds_slots"] == 4 and "#define DRS_NTL 384" in s4 and "__launch_bounds__(384)" in s4
    assert "if (tid >= DRS_NT)" in s4 and "DRS_VMWAIT(" in s4 and "dim3(DRS_NTL)" in s4
    # the tuner's emission dimension
    t.order, t.ndim, t.elem_bytes = 1, 3, 4
    v = (3, 3, (64, 8), 64, 4, True, 4, True, 2, 5, 1, "dpp", False, 2, False, "scatter")
    assert t.cfgToString(v) == t.cfgToString(v + ("taps",)) == "fu3d3bx64y8sn64u4bmx4bmy2mf5pxdm2"
    assert t.cfgToString(v + ("rows",)) == "fu3d3bx64y8sn64u4bmx4bmy2mf5pxdm2o" and t.cfgToCommandLine(v + ("rows",)).endswith("--schedule scatter --order rows --pack 0")
    assert t.cfgToString(v + ("rowspk",)).endswith("ok") and t.cfgToString(v + ("pin",)).endswith("k") and t.cfgToCommandLine(v + ("pin",)).endswith("--pin 1")
    assert t.FilterParams(v + ("rows",)) and not t.FilterParams(v[:15] + ("reuse", "rows")) and not t.FilterParams(v + ("columns",))
    sp = t.enumerate_space((3,), emits=("rows", "rowspk"))
    assert sp and all(len(x) == 17 and x[16] in ("rows", "rowspk") and x[15] == "scatter" for x in sp)
    names = [t.cfgToString(x) for x in sp]
    assert len(set(names)) == len(names)
    # a fenced --temporal configuration is a duplicate of its fused twin: dropped before anything is compiled
    c4 = os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc")
    assert not t.toleranceFilter(["--3d", "--dtype", "fp32", "--step", "3", "--temporal", "1", c4])
    assert t.toleranceFilter(["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", c4]) and t.toleranceFilter(["--3d", "--dtype", "fp32", "--step", "3", c4])
    # pinned kernels: the register rule is the generator's named state, not round 2's fitted model
    assert t.registerFilter(["--3d", "--dtype", "fp32", "--step", "3", "--order", "rows", "--pack", "0", "--prefetch", "--prefetch-depth", "1", "--bx", "64", "--by", "8", "--block-merge-y", "2", c4])
    assert not t.registerFilter(["--3d", "--dtype", "fp32", "--step", "3", "--order", "rows", "--bx", "32", "--by", "16", "--block-merge-y", "8", c4])


def test_output_array_placement_is_published_and_laid_out():
    """--out-skew (round 3): launch time of a z-streaming kernel depends on (out - in) mod 64 MiB, so the generator publishes a
    recommended position of the output array (kernel info), the emitted program carves both arrays out of one allocation, and the
    C ABI / the binding compute the same layout.  Nothing of the kernel text depends on it."""
    import ctypes
    import json
    import re
    cfg = os.path.join(ROOT, "benchmarks", "configs")

    def gen(args):
        rc, msg, src = drs.generate(args)
        assert rc == 0 and src, msg
        return json.loads(re.search(r'return "(\{.*\})";', src).group(1).replace('\\"', '"')), src

    c4 = os.path.join(cfg, "c4_3d7pt_star_1024.stc")
    head = ["--3d", "--dtype", "fp32", "--step", "2", "--prefetch", "--prefetch-depth", "3", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32"]
    i, src = gen(head + [c4])
    assert (i["out_skew_bytes"], i["placement_period_bytes"]) == (32 << 20, 64 << 20)     # reads 5 planes (20 MiB) ahead of its writes: 32 MiB is clear of both bad windows
    assert "hipMalloc (&arena, out_at + nbytes)" in src and "real_t *in = (real_t*)arena, *out = (real_t*)(arena + out_at);" in src
    assert "const size_t out_at = (nbytes + 67108863UL) / 67108864UL * 67108864UL + 33554432UL;" in src
    i8, src8 = gen(head + ["--out-skew", "8", c4])
    assert i8["out_skew_bytes"] == 8 << 20
    strip = lambda t: re.sub(r"out_skew_bytes[^,]*,|//.*|const size_t out_at.*", "", t)
    assert strip(src8.split("int main")[0]).split("dr_c4")[1:] == strip(src.split("int main")[0]).split("dr_c4")[1:]      # same kernels
    assert gen(head + ["--out-skew", "72", c4])[0]["out_skew_bytes"] == 8 << 20           # taken modulo the period
    # small planes (512^3: 1 MiB) and 2D kernels are flat: no skew
    assert gen(["--3d", "--dtype", "fp32", "--step", "2", "--prefetch", os.path.join(cfg, "c3_3d7pt_star_512.stc")])[0]["out_skew_bytes"] == 0
    assert gen(["--dtype", "fp32", os.path.join(cfg, "c2_2d5pt_star_8192.stc")])[0]["out_skew_bytes"] == 0
    # layout arithmetic: binding == C ABI
    k = drs.Kernel(head + ["--cc-opt", "-fno-slp-vectorize", "--xcd-remap", "2", c4])
    arena, off = k.pair_layout()
    assert off == (4 << 30) + (32 << 20) and arena == off + (4 << 30) and k.pair_layout(skew=0) == (8 << 30, 4 << 30)
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    fn = drs.lib().drs_kernel_pair_layout
    fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    assert fn(k.h, ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value, b.value) == (arena, off)


def test_coefficient_register_emission():
    """--coef sgpr|vgpr: the distinct coefficient values are named scalar / vector constants made opaque by an empty asm, the FMAs
    refer to them, and the gold kernel is untouched."""
    stc = os.path.join(ROOT, "tests", "stc", "smoke3.stc")
    base = ["--3d", "--dtype", "fp32", "--step", "3", "--prefetch", "--prefetch-depth", "1", "--bx", "32", "--by", "8", "--block-merge-y", "2", "--order", "rows", "--pack", "0"]
    rc, _, lit = drs.generate(base + [stc])
    rc1, _, sg = drs.generate(base + ["--coef", "sgpr", stc])
    rc2, _, vg = drs.generate(base + ["--coef", "vgpr", stc])
    assert rc == rc1 == rc2 == 0
    assert "DRS_SREG" not in lit and "kc0" not in lit
    assert 'asm volatile("" : "+s"(x))' in sg and 'asm volatile("" : "+v"(x))' in vg
    for t in (sg, vg):
        assert "real_t kc0 = (real_t)(" in t and "DRS_SREG(kc0);" in t and "__builtin_fmaf(kc" in t and "__builtin_fmaf((real_t)(" not in t.split("// naive reference kernel")[0]
    gold = lambda src: src[src.index("// naive reference kernel"):src.index("// ---- launch entry points")]
    assert gold(lit) == gold(sg) == gold(vg)
    assert drs.generate(base + ["--coef", "mmx", stc])[0] == 255
    # fp64 keeps its literals (the compiler holds them in scalar pairs already)
    assert "kc0" not in drs.generate(["--3d", "--dtype", "fp64", "--step", "2", "--order", "rows", "--coef", "sgpr", stc])[2]


def test_gpus_option_emits_an_n_gpu_host():
    """--gpus N > 1 replaces the emitted program's main() by the N-GPU host (launcher + ranks, drs_slab_* through the C ABI) and leaves every
    kernel, the gold kernel and the plugin entry points as they are."""
    stc = os.path.join(ROOT, "tests", "stc", "t3_star.stc")
    base = ["--3d", "--dtype", "fp32", "--step", "2", "--sn", "16"]
    rc1, _, one = drs.generate(base + ["--check", "-o", "x.hip", stc])
    rc4, _, four = drs.generate(base + ["--gpus", "4", "--check", "-o", "x.hip", stc])
    assert rc1 == rc4 == 0
    cut = lambda t: t[t.index("#include <hip/hip_runtime.h>"):t.index("#ifndef DRS_PLUGIN\n", t.index("// ---- launch entry points"))]
    assert cut(one) == cut(four)                                             # same kernels, same plugin API
    host = four[four.index("#ifndef DRS_PLUGIN\n", four.index("// ---- launch entry points")):]
    assert "#define DRS_WORLD 4" in host and 'static const char *drs_opts[] = { "--3d", "--dtype", "fp32", "--step", "2", "--sn", "16", NULL };' in host
    assert '"L 70\\nM 45\\nN 530\\n' in host and "fork ()" in host and "drs_slab_open (" in host and "drs_slab_connect (" in host and "#pragma push_macro(\"L\")" in host
    assert "#if 1\n    if (!rh) {" in host and "checkError3D (M, N, h_own, h_gold" in host           # --check: own planes vs the gold kernel on a wider slab
    assert "drs_slab_open" not in one and "#define DRS_WORLD" not in one
    rc2d, _, two = drs.generate(["--dtype", "fp64", "--gpus", "2", os.path.join(ROOT, "tests", "stc", "t2_star.stc")])
    assert rc2d == 0 and "const long DIM0 = M;" in two and "const size_t plane = (size_t)N;" in two and "#if 0\n    if (!rh) {" in two
    assert drs.generate(base + ["--gpus", "0", stc])[0] == 255 and drs.generate(base + ["--gpus", "65", stc])[0] == 255


def test_metrics_launch_limit_family():
    """getGpuMetrics.py's launch-limit columns (the reference's "Block Limit ...", "Theoretical / Achieved Occupancy": getGpuMetrics.py:9)
    on the committed counters of three bench kernels (profiles/r04h_*_counters.json: VGPRs from the compiler report)."""
    import json
    from drstencil_amd.tuner import getGpuMetrics as g
    def counters(name):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        return {k: d[k] for k in ("SQ_WAVE_CYCLES", "SQ_WAVES")}, d["GRBM_GUI_ACTIVE"] / 8.0, d
    # C4 headline: 1024 lanes, 95 VGPRs, 76 KiB of LDS: one workgroup per CU by registers (5 waves per SIMD < 2 x 4), two by LDS and waves
    vals, cycles, d = counters("r04h_c4_headline_counters.json")
    out = g.launch_limits(d["wg"], "95", "0", d["lds"], vals, cycles)
    assert out[:6] == [16, 1, 2, 2, 16, 0.5] and 15.0 < out[6] <= 16.0 and 0.028 < out[8] < 0.034, out      # 8192 workgroups, one per CU: 32 rounds
    # C3 headline: 512 lanes, 204 VGPRs: 8 resident waves per CU, and the 512 workgroups run as two rounds of 256 (a wave lives half the launch)
    vals, cycles, d = counters("r04h_c3_headline_counters.json")
    out = g.launch_limits(d["wg"], "204", "0", d["lds"], vals, cycles)
    assert out[:6] == [8, 1, 4, 4, 8, 0.25] and 7.0 < out[6] <= 8.0 and 0.4 < out[8] < 0.55, out
    # 4-stage pipeline: 748 lanes = 12 waves, 163 VGPRs (3 waves per SIMD)
    vals, cycles, d = counters("r04h_c4f64_temporal4_counters.json")
    out = g.launch_limits(d["wg"], "163", "0", d["lds"], vals, cycles)
    assert out[:5] == [12, 1, 1, 2, 12] and 11.0 < out[6] <= 12.0 and 0.11 < out[8] < 0.13, out      # 2048 workgroups: 8 rounds
    assert g.launch_limits("", "", "", "", {}, None) == [""] * 9 and len(g.HEADER) == len(g.UNITS) == 53
