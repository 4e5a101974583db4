"""Host logic against the reference's recorded behaviour: CLI messages / exit codes /
macros (tests/golden/ref_cli.json, produced by running the reference generator), the
partition table of SURVEY.md section 2, fused coefficients and gold term order."""
import json
import os
import subprocess

import pytest

import drstencil_amd as drs
from helpers import GOLDEN, golden_cases, load_golden, stc_from_meta, write_stc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "drstencil")


def _mg():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mg", os.path.join(ROOT, "oracle", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def cli_dir(tmp_path_factory):
    mg = _mg()
    d = tmp_path_factory.mktemp("cli")
    write_stc(str(d / "s3.stc"), 3, (512, 512, 512), 4, mg.STAR3)
    write_stc(str(d / "c3.stc"), 3, (512, 512, 512), 4, mg.CROSS3)
    write_stc(str(d / "s2.stc"), 2, (1, 8192, 8192), 4, mg.STAR2)
    write_stc(str(d / "b25.stc"), 2, (1, 8192, 8192), 4, mg.BOX25)
    write_stc(str(d / "typo.stc"), 2, (1, 64, 64), 4, mg.CROSS9, iter_token="iteratioins")
    return d


REF_CLI = json.load(open(os.path.join(GOLDEN, "ref_cli.json")))


@pytest.mark.parametrize("rec", REF_CLI, ids=[" ".join(r["args"]) or "<none>" for r in REF_CLI])
def test_cli_matches_reference(rec, cli_dir):
    args = rec["args"]
    if rec["rc"] == "hang":
        pytest.skip("the reference spins forever on this input (3D spec read in 2D mode or vice versa); ours terminates")
    if "--3d" in args and "s2.stc" in args:
        pytest.skip("2D spec read in 3D mode: the reference emits from uninitialised sizes")
    if args == ["--step", "2", "--by", "8", "b25.stc"]:
        pytest.skip("2*Halo == by: the reference emits a program that divides by zero (ceil(M, By-Halo*2)); we reject it")
    out = cli_dir / "out.cu"
    if out.exists():
        out.unlink()
    # --ref-defaults keeps the 16x16x16 geometry so the Bx/By/Sn macros are comparable
    ours = list(args)
    if rec.get("emitted"):
        ours = ["--ref-defaults"] + ours if ours and ours[0] not in ("--help", "-h") else ours
    p = subprocess.run([CLI] + ours, cwd=str(cli_dir), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == rec["rc"], (p.stdout, p.stderr)
    if args and args[0] in ("--help", "-h"):
        assert "Usage: drstencil [options] <input_stcfile>" in p.stdout
        return
    assert p.stdout == rec["stdout"]
    assert out.exists() == rec["emitted"]
    if rec["emitted"]:
        src = out.read_text()
        import re
        macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define (L|M|N|Iterations|Range|Halo|Dist|Bx|By|Sn) (-?\d+)\s*$", src, re.M)}
        for k, v in rec["macros"].items():
            if k in ("By",) and "--streaming" in args and "--3d" not in args:
                continue  # 2D streaming ignores --by (codegen_2d.hpp:125); we report the effective 1
            assert macros[k] == v, (k, macros, rec["macros"])
        assert ("__global__ void gold_" in src or "void gold_" in src)
        assert rec["kernel"] in src


def test_partition_table_from_survey():
    """(step, dist) -> Halo, Range, fk/fj/fi/bw."""
    b = os.path.join(ROOT, "benchmarks")
    rows = [
        (3, "3d7pt_star", 1, 0, 1, 2, (2, 0, 0, 5)),
        (3, "3d7pt_star", 2, 0, 2, 3, (7, 5, 0, 13)),
        (3, "3d7pt_star", 2, 1, 2, 4, (12, 0, 0, 13)),
        (3, "3d7pt_star", 3, 0, 3, 5, (12, 10, 9, 32)),
        (3, "3d9pt_cross", 2, 2, 2, 3, (22, 8, 0, 5)),
        (2, "2d5pt_star", 1, 0, 1, 2, (0, 2, 0, 3)),
        (2, "2d5pt_star", 2, 0, 2, 3, (0, 5, 0, 8)),
        (2, "2d25pt_box", 1, 0, 2, 3, (0, 15, 6, 4)),
        (2, "2d25pt_box", 2, 0, 4, 5, (0, 45, 20, 16)),
    ]
    for ndim, name, step, dist, halo, rng, part in rows:
        s = drs.Spec(os.path.join(b, name, name + ".stc"), ndim, step, dist)
        assert s.status == 0
        assert (s.halo, s.range, s.partition) == (halo, rng, part), (name, step, dist)
    s = drs.Spec(os.path.join(b, "3d9pt_cross", "3d9pt_cross.stc"), 3, 1, 0)
    assert s.status == 2   # "No data to reuse" with the automatic dist


@pytest.mark.parametrize("case", golden_cases())
def test_gold_terms_match_reference_emission(case, tmp_path):
    """Fused coefficient literals and term order == what the reference generator printed."""
    meta, *_ = load_golden(case)
    s = drs.Spec(stc_from_meta(tmp_path, meta), meta["ndim"], meta["step"], meta["macros"]["Dist"])
    pts = s.points
    assert len(pts) == len(meta["terms"])
    for (off, c, text), t in zip(pts, meta["terms"]):
        assert list(off[3 - meta["ndim"]:]) == t["off"]
        assert text == t["coef"]
        assert c == float(t["coef"])
    assert s.halo == meta["macros"]["Halo"] and s.dist == meta["macros"]["Dist"]
    if "Range" in meta["macros"]:
        assert s.range == meta["macros"]["Range"]
    assert s.launches == meta["launches"]


def test_generate_function_equals_cli(tmp_path):
    mg = _mg()
    write_stc(str(tmp_path / "k.stc"), 3, (64, 64, 64), 4, mg.STAR3)
    rc, msg, src = drs.generate(["--3d", "--dtype", "fp32", "--check", str(tmp_path / "k.stc")])
    assert rc == 0 and msg == "" and "dr_k" in src and "gold_k" in src
    p = subprocess.run([CLI, "--3d", "--dtype", "fp32", "--check", "-o", str(tmp_path / "o.hip"), str(tmp_path / "k.stc")], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0
    body = lambda t: t[t.index("#include"):]
    assert body((tmp_path / "o.hip").read_text()) == body(src)
    rc, msg, src = drs.generate(["--3d", "--dist", str(tmp_path / "k.stc")])
    assert (rc, msg, src) == (255, "Illegal input.\n", None)


def test_emitted_stdout_protocol_strings(tmp_path):
    mg = _mg()
    write_stc(str(tmp_path / "k.stc"), 2, (1, 64, 64), 4, mg.STAR2)
    rc, msg, src = drs.generate(["--check", str(tmp_path / "k.stc")])
    for s in ("Initiating ...", "GPU computing ...", "GPU finished computing.", "GPU computation time: %f ms",
              "Checking error ...", "[Test] RMS Error: %e", "#include \"common.hpp\"", "hipcc" if False else "hip_runtime"):
        assert s in src
    assert "atomicAdd" not in src and "__HIP_PLATFORM" not in src


def test_temporal_fallback_is_reported(tmp_path):
    """--temporal 1 on a stencil whose fused coefficients do not survive the 6-digit rounding (drstencil.hpp:192) cannot equal
    the reference's fused arithmetic, so the fused single-pass kernel is emitted -- and the user is told (stdout note +
    banner), not left to find stages == 1 in the kernel info."""
    import drstencil_amd as drs
    stc = os.path.join(ROOT, "tests", "stc", "t3_odd.stc")
    rc, msg, src = drs.generate(["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", stc])
    assert rc == 0 and src
    assert "note: --temporal 1 ignored" in msg and "// note: --temporal 1 ignored" in src
    assert '\\"stages\\":1' in src
    rc, msg, src = drs.generate(["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", os.path.join(ROOT, "tests", "stc", "t3_star.stc")])
    # (a note about the tuned-defaults table may appear for this stencil and size; none about --temporal)
    assert rc == 0 and "--temporal 1 ignored" not in msg and "not honoured" not in msg and '\\"stages\\":2' in src


def test_temporal_blocking_is_fenced_to_the_tolerance(tmp_path):
    """On-chip time steps re-associate the reference's fused sum (drstencil.hpp:262-282).  `--temporal 1` emits them only where
    the generator's drift estimate (planner.hpp: temporal_drift_per_launch, calibrated: profiles/r03_temporal_drift_calibration.txt)
    keeps 1e-6 relative (fp32) / 1e-12 (fp64) for the spec's iterations, and the fused kernel -- exact reference arithmetic --
    otherwise, with a note; `--temporal force` emits them regardless and the plugin says so."""
    import json
    import re
    import drstencil_amd as drs

    def info(src):
        m = re.search(r'return "(\{.*\})";', src)
        return json.loads(m.group(1).replace('\\"', '"'))

    it4 = os.path.join(ROOT, "tests", "stc", "t3_star.stc")            # iterations 4
    it100 = os.path.join(ROOT, "tests", "stc", "t3_star_it100.stc")    # iterations 100
    base = ["--3d", "--dtype", "fp32", "--step", "2"]
    rc, msg, src = drs.generate(base + ["--temporal", "1", it4])
    i = info(src)
    assert rc == 0 and i["stages"] == 2 and i["arithmetic"] == "reassociated" and i["temporal_forced"] == 0
    assert 0 < i["drift_estimate"] <= 1e-6 and i["iterations"] <= i["tolerance_horizon_iterations"] < 100
    assert "// arithmetic: reassociated" in src
    # 100 iterations: beyond the horizon -> the fused kernel, and the user is told
    rc, msg, src = drs.generate(base + ["--temporal", "1", it100])
    i = info(src)
    assert rc == 0 and i["stages"] == 1 and i["arithmetic"] == "gold-order" and i["tolerance_horizon_iterations"] == -1 and i["drift_estimate"] == 0
    assert "note: --temporal 1 not honoured" in msg and "// note: --temporal 1 not honoured" in src and "// arithmetic: gold order" in src
    # ... unless forced
    rc, msg, src = drs.generate(base + ["--temporal", "force", it100])
    i = info(src)
    assert rc == 0 and i["stages"] == 2 and i["arithmetic"] == "reassociated" and i["temporal_forced"] == 1 and i["drift_estimate"] > 1e-6
    assert "emitted by --temporal force" in src and "note:" not in msg
    # three stages on the full C4 grid are beyond the bar even at iterations 4 (measured 9.65e-7 at 1024^3: no margin); two stages are within
    c4 = os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc")
    assert info(drs.generate(["--3d", "--dtype", "fp32", "--step", "3", "--temporal", "1", c4])[2])["stages"] == 1
    assert info(drs.generate(["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", c4])[2])["stages"] == 2
    # fp64 has 29 bits more: the same pipelines are far inside 1e-12
    i = info(drs.generate(["--3d", "--dtype", "fp64", "--step", "3", "--temporal", "1", it100])[2])
    assert i["stages"] == 3 and i["drift_estimate"] < 1e-13 and i["tolerance_horizon_iterations"] > 1000
    # the estimate grows with the stages, the taps and the launches
    e = lambda step, stc: info(drs.generate(["--3d", "--dtype", "fp32", "--step", str(step), "--temporal", "force", stc])[2])["drift_estimate"]
    assert e(2, it4) < e(3, it4) and e(2, it4) < e(2, it100)
    # a fused kernel is gold order whatever the step
    assert info(drs.generate(base + [it100])[2])["arithmetic"] == "gold-order"
    # the option takes 0 | 1 | force
    rc, msg, src = drs.generate(base + ["--temporal", "maybe", it4])
    assert rc == 255 and "Illegal input." in msg and not src


def test_temporal_kernel_refuses_iterations_beyond_its_horizon():
    """drs_kernel_run answers -3 before anything is launched when a reassociated kernel (not forced) is asked for more iterations than
    its tolerance horizon (include/drstencil_amd.h); no GPU needed to see that."""
    import drstencil_amd as drs
    from gpu_cases import SMALL
    cid, ndim, stc, opts = next(c for c in SMALL if c[0] == "3d7_fp32_t2")
    kern = drs.Kernel(opts + [stc])                                   # built by build(); cross-compiles if not
    assert kern.info["arithmetic"] == "reassociated"
    with pytest.raises(drs.ToleranceHorizonExceeded):
        kern.run(0, 0, iterations=kern.info["tolerance_horizon_iterations"] + 1)
    with pytest.raises(drs.ToleranceHorizonExceeded):
        kern.run_timed(0, 0, iterations=100)


def test_tuned_defaults_table_feeds_generator_and_bench(tmp_path):
    """The tuner -> generator loop (reference: benchmarks/3d7pt_star/tuning.py:125-131 leaves the winner in duration.log): the table
    drstencil_amd/tuned_defaults.tsv, its generated C++ twin, the generator's behaviour without geometry options and bench.py's TUNED
    lists all say the same thing."""
    import bench
    import drstencil_amd as drs
    from drstencil_amd import tuned_defaults as td
    rows = td.load()
    assert td.header_text(rows) == open(td.HEADER).read(), "csrc/tuned_defaults.hpp is stale: python3 -m drstencil_amd.tuned_defaults; make -C drstencil_amd/csrc"
    body = lambda src: src[src.index("#include"):]
    for w in ("c4", "c3", "c4f64", "c5", "c2", "s_2d25pt_box", "s_2d5pt_cross", "s_3d9pt_cross"):
        wl, h = bench.WORKLOADS[w], bench.HEADLINE[w]
        naming = td.problem_options(wl["ndim"], wl["dtype"], **h)
        assert bench.TUNED[w][:len(naming)] == naming and len(bench.TUNED[w]) > len(naming)
        rc0, msg0, bare = drs.generate(naming + [wl["stc"]])
        rc1, msg1, full = drs.generate(bench.TUNED[w] + [wl["stc"]])
        assert rc0 == rc1 == 0 and body(bare) == body(full), w
        assert "the tuner's configuration" in msg0 and "the tuner's configuration" not in msg1
        rc2, _, generic = drs.generate(naming + ["--tuned-defaults", "0", wl["stc"]])
        assert rc2 == 0 and (body(generic) != body(full) or w not in ("c4", "c4f64")), w      # (several winners ARE the generic geometry; C4's is not)
    # any geometry option of the user's switches the table off; --ref-defaults too
    stc4 = bench.WORKLOADS["c4"]["stc"]
    _, m, s_sn = drs.generate(["--3d", "--dtype", "fp32", "--step", "2", "--sn", "16", stc4])
    assert "the tuner's configuration" not in m and "#define Bx 32\n" in s_sn
    # a size far from every row (256^3) keeps the generic defaults; 1000^3 is within sqrt 2 of the 1024 row
    from helpers import write_stc
    star = [(0, 0, 0, 0.3), (1, 0, 0, 0.2), (-1, 0, 0, 0.2), (0, 1, 0, 0.2), (0, -1, 0, 0.2), (0, 0, 1, 0.2), (0, 0, -1, 0.2)]
    small, near = str(tmp_path / "s256.stc"), str(tmp_path / "s1000.stc")
    write_stc(small, 3, (256, 256, 256), 4, star)
    write_stc(near, 3, (1000, 1000, 1000), 4, star)
    assert "the tuner's configuration" not in drs.generate(["--3d", "--dtype", "fp32", "--step", "2", small])[1]
    assert "--bx 64 --by 16" in drs.generate(["--3d", "--dtype", "fp32", "--step", "2", near])[1]


def test_table_note_goes_to_stderr_not_stdout(tmp_path):
    """A reference command line (`drstencil --3d --step 2 -o k.cu 3d7pt_star.stc`, main.cpp:10-280) prints nothing on stdout when it succeeds; ours
    takes the tuner's row for the shipped spec and says so -- on stderr, so that stdout stays the reference's byte for byte (compared with the
    reference binary itself in oracle/fuzz_vs_reference.py on random shapes, here on the shipped specs that have table rows)."""
    import subprocess
    import drstencil_amd as drs
    for name, flags in (("3d7pt_star", ["--3d"]), ("2d5pt_star", []), ("2d9pt_box", []), ("2d9pt_star", [])):
        stc = os.path.join(ROOT, "benchmarks", name, name + ".stc")
        r = subprocess.run([drs.CLI_PATH] + flags + ["--step", "2", "-o", str(tmp_path / (name + ".hip")), stc], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout == "", (name, r.stdout[:200])
        assert "the tuner's configuration" in r.stderr and "--tuned-defaults 0" in r.stderr, (name, r.stderr[:200])
        r0 = subprocess.run([drs.CLI_PATH] + flags + ["--step", "2", "--tuned-defaults", "0", "-o", str(tmp_path / (name + "_g.hip")), stc], capture_output=True, text=True)
        assert r0.returncode == 0 and r0.stdout == "" and r0.stderr == "", (name, r0.stderr[:200])


def test_tuner_writes_the_defaults_table(tmp_path):
    """tuning.py --write-defaults: the fastest VERIFIED record per (step, temporal, streaming) class becomes the row; naming options are
    stripped; an existing row of the same class and size is replaced, others stay."""
    import shutil
    import bench
    from drstencil_amd import tuned_defaults as td
    from drstencil_amd.tuner import tuning as t
    table = str(tmp_path / "t.tsv")
    shutil.copy(td.TABLE, table)
    before = td.load(table)
    stc4 = bench.WORKLOADS["c4"]["stc"]
    recs = [dict(name="a", args="--3d --dtype fp32 --step 2 --bx 32 --by 16 --sn 32", duration_ns=1500.0, GStencil=1400.0, step=2, arithmetic="gold-order", verified=True),
            dict(name="b", args="--3d --dtype fp32 --step 2 --bx 64 --by 8 --sn 8 --pin 1", duration_ns=1400.0, GStencil=1500.0, step=2, arithmetic="gold-order", verified=True),
            dict(name="wrong", args="--3d --dtype fp32 --step 2 --bx 16 --by 16", duration_ns=100.0, GStencil=9999.0, step=2, arithmetic="gold-order", verified=False),
            dict(name="t", args="--3d --dtype fp32 --step 2 --temporal 1 --bx 66 --by 15", duration_ns=1450.0, GStencil=1450.0, step=2, arithmetic="reassociated", verified=True)]
    rows = t.write_defaults(stc4, True, "fp32", recs, "unit test", table=table)
    assert sorted((r["temporal"], r["options"]) for r in rows) == [(0, "--bx 64 --by 8 --sn 8 --pin 1"), (1, "--bx 66 --by 15")]
    after = td.load(table)
    key = lambda r: tuple(r[f] for f in td.FIELDS)
    assert {key(r) for r in after} == {key(r) for r in before} | {key(r) for r in rows}          # rows of the same class and size are replaced, the others stay
    mode, shape, points, order, N = td.key_of(stc4, 3)
    assert td.lookup(mode, shape, 2, "fp32", 0, N, after) == "--bx 64 --by 8 --sn 8 --pin 1".split()
    assert td.lookup(mode, shape, 2, "fp32", 1, N, after) == ["--bx", "66", "--by", "15"]
    assert td.lookup(mode, shape, 2, "fp64", 0, N, after) == td.lookup(mode, shape, 2, "fp64", 0, N, before)
