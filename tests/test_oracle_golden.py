"""Pin the CPU oracle to the reference: bit-exact against arrays produced by the
reference-emitted gold statement (oracle/make_golden.py) and against the
known-answer values recorded in SURVEY.md section 8(c)."""
import numpy as np
import pytest

import oracle
from helpers import golden_cases, load_golden, stc_from_meta


@pytest.mark.parametrize("case", golden_cases())
def test_oracle_bit_exact_vs_reference_gold(case, tmp_path):
    meta, a0, a_ref, b_ref = load_golden(case)
    spec = oracle.Spec(stc_from_meta(tmp_path, meta), meta["ndim"], meta["step"])
    assert spec.halo == meta["macros"]["Halo"]
    assert spec.iterations == meta["macros"]["Iterations"]
    # gold terms: same offsets in the same order, same printed coefficients
    pts = spec.points
    assert len(pts) == len(meta["terms"])
    for (off, c), t in zip(pts, meta["terms"]):
        assert list(off[3 - meta["ndim"]:]) == t["off"]
        assert c == float(t["coef"])
    # input stream == common.hpp rand() fill
    A = oracle.fill_random(a0.shape, np.float64)
    assert np.array_equal(A, a0)
    B = np.zeros_like(A)
    n = oracle.run(spec, A, B, contract=0)
    assert n == meta["launches"] == spec.launches
    assert np.array_equal(A, a_ref), "result buffer differs from reference gold"
    assert np.array_equal(B, b_ref), "scratch buffer differs from reference gold"


# values captured by the surveyor from the reference-emitted gold
# (fp64, g++ -O0, glibc rand() seed 1), flat row-major indices.
KNOWN = {
    "tiny3d_s1": dict(idx=589, val=2.4920680248106066, sum_a=1401.2182949042365, sum_b=828.22812807166076),
    "tiny3d_s2": dict(idx=589, val=2.4920680248106062, sum_a=906.5531129951529, sum_b=274.10139048942506),
    "tiny2d_s1": dict(idx=71, val=0.66945895966050128, sum_a=89.047173596230323, sum_b=66.172632255245603),
    "tiny2d_s2": dict(idx=71, val=0.66945895966050117, sum_a=83.186017546907522, sum_b=41.368072661662524),
    "tiny25_s1": dict(idx=161, val=17.769162208746852, sum_a=2589.1001999063042, sum_b=1156.785215912282),
    "tiny25_s2": dict(idx=161, val=17.769162208746845, sum_a=1435.9490754437454, sum_b=302.17625767997231),
}


@pytest.mark.parametrize("case", sorted(KNOWN))
def test_oracle_vs_survey_known_answers(case, tmp_path):
    meta, a0, _, _ = load_golden(case)
    spec = oracle.Spec(stc_from_meta(tmp_path, meta), meta["ndim"], meta["step"])
    A = oracle.fill_random(a0.shape, np.float64)
    assert A.flat[0] == 0.84018771754595234
    B = np.zeros_like(A)
    oracle.run(spec, A, B, contract=0)
    k = KNOWN[case]
    assert A.flat[k["idx"]] == k["val"]
    # the surveyor summed sequentially; allow last-digit summation-order noise
    assert sum(A.flat) == pytest.approx(k["sum_a"], rel=1e-14)
    assert sum(B.flat) == pytest.approx(k["sum_b"], rel=1e-14)


def test_tiny2d_s1_second_pin(tmp_path):
    meta, a0, _, _ = load_golden("tiny2d_s1")
    spec = oracle.Spec(stc_from_meta(tmp_path, meta), 2, 1)
    A = oracle.fill_random(a0.shape, np.float64)
    B = np.zeros_like(A)
    oracle.run(spec, A, B, contract=0)
    assert A.flat[72] == 0.7234308752012728


def test_fused_coefficient_multisets(tmp_path):
    """SURVEY 8(c) 'Other pins': 3d7pt s=3 multiset and 2d25pt_box s=2 centre."""
    from collections import Counter
    meta, *_ = load_golden("tiny3d_s3")
    spec = oracle.Spec(stc_from_meta(tmp_path, meta), 3, 3)
    cnt = Counter(c for _, c in spec.points)
    assert cnt == Counter({0.008: 6, 0.024: 24, 0.036: 6, 0.048: 8, 0.072: 12, 0.174: 6, 0.243: 1})
    meta, *_ = load_golden("tiny25_s2")
    spec = oracle.Spec(stc_from_meta(tmp_path, meta), 2, 2)
    assert dict(spec.points)[(0, 0, 0)] == 0.3516


def test_contract_modes_agree_to_rounding(tmp_path):
    """fma-contracted order (what the HIP kernels compute) vs uncontracted: ~1 ulp/op."""
    meta, a0, a_ref, _ = load_golden("tiny3d_s1")
    spec = oracle.Spec(stc_from_meta(tmp_path, meta), 3, 1)
    A = a0.copy(); B = np.zeros_like(A)
    oracle.run(spec, A, B, contract=1)
    m = oracle.check(spec, A, a_ref)
    assert m["max_rel"] < 1e-14
    A32 = a0.astype(np.float32); B32 = np.zeros_like(A32)
    oracle.run(spec, A32, B32, contract=1)
    m = oracle.check(spec, A32.astype(np.float64), a_ref)
    assert m["max_rel"] < 1e-6


def test_check_error_metric_matches_definition(tmp_path):
    meta, a0, a_ref, _ = load_golden("tiny2d_s1")
    spec = oracle.Spec(stc_from_meta(tmp_path, meta), 2, 1)
    out = a_ref.copy()
    out[5, 6] += 1e-3
    out[0, 0] += 5.0   # ring: must be ignored (common.hpp:74-75 loops over the interior only)
    m = oracle.check(spec, out, a_ref)
    h = spec.halo
    n_int = (a_ref.shape[0] - 2 * h) * (a_ref.shape[1] - 2 * h)
    assert m["max_abs"] == pytest.approx(1e-3, rel=1e-9)
    assert m["max_idx"] == 5 * a_ref.shape[1] + 6
    assert m["rms"] == pytest.approx(np.sqrt(1e-6 / n_int), rel=1e-9)
    m0 = oracle.check(spec, a_ref, a_ref)
    assert m0["max_abs"] == 1e-13 and m0["rms"] == 0.0   # the reference's 1e-13 floor


def test_c1_reference_cpu_path_plumbing():
    """BASELINE config C1: 2d5pt_star 4096^2 fp32, 100 iterations on the CPU path (plumbing + correctness):
    launch count, frozen rings of both buffers, contraction modes within the fp32 bar of each other."""
    import os
    stc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks", "configs", "c1_2d5pt_star_4096.stc")
    spec = oracle.Spec(stc, 2, 1)
    assert spec.shape == (4096, 4096) and spec.iterations == 100 and spec.halo == 1
    A0 = oracle.fill_random(spec.shape, np.float32)
    A, B = A0.copy(), np.zeros_like(A0)
    assert oracle.run(spec, A, B, contract=1) == 100
    assert np.isfinite(A).all()
    assert np.array_equal(A[0], A0[0]) and np.array_equal(A[-1], A0[-1]) and np.array_equal(A[:, 0], A0[:, 0]) and np.array_equal(A[:, -1], A0[:, -1])
    assert not B[0].any() and not B[-1].any() and not B[:, 0].any() and not B[:, -1].any()
    # FMA-contracted vs uncontracted fp32 arithmetic drift apart by ~1e-8 per sweep: 1.1e-6 after 100
    # sweeps (the 1e-6 gate only makes sense against an oracle with the SAME
    # arithmetic as the kernel -- the HIP kernels are compared with contract=1, bit for bit)
    A2, B2 = A0.copy(), np.zeros_like(A0)
    oracle.run(spec, A2, B2, contract=0)
    assert oracle.check(spec, A, A2)["max_rel"] <= 2e-6


def test_random_shapes_against_the_reference_binary():
    """Where the reference's sources are present (the authoring container; oracle/Makefile builds its generator from them
    where they lie), a small slice of oracle/fuzz_vs_reference.py: random stencil shapes through the reference binary and
    through bin/drstencil (exit code, stdout, macros, gold term order and coefficient literals identical), and the oracle
    against the reference-emitted gold statement, bit for bit.  profiles/r02_fuzz_vs_reference.txt is the 3 000-shape run."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (os.path.exists(os.path.join(root, "oracle", "_ref", "drstencil_ref")) and os.path.isdir("/root/reference")):
        pytest.skip("the reference is not present on this box")
    p = subprocess.run([sys.executable, os.path.join(root, "oracle", "fuzz_vs_reference.py"), "40", "7", "8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and " 0 DIFFERENCES" in p.stdout, p.stdout[-2000:]
    assert "bit for bit on 8 of 8 sampled cases" in p.stdout, p.stdout[-600:]


def test_avx512_fast_path_equals_the_generic_sweep(tmp_path):
    """The contracted sweep has a register-blocked AVX-512 path (oracle/drs_oracle.c: eight zmm accumulators per block of 128 / 64 points,
    masked tail) next to the generic, auto-vectorised one that the golden fixtures pin.  Same chain per point -- t = c0*a0, t = fma(ci, ai, t)
    in table order -- so the two must agree BIT FOR BIT, on ragged sizes (tails of 1 ... 15 points), both dtypes, 2D and 3D, steps 1-3."""
    from helpers import write_stc
    if "register-blocked" not in oracle.isa():
        pytest.skip("no avx512f on this host: the generic path is the only one")
    rng = np.random.default_rng(3)
    star3 = [(0, 0, 0, 0.3), (1, 0, 0, 0.2), (-1, 0, 0, 0.15), (0, 1, 0, 0.2), (0, -1, 0, 0.1), (0, 0, 1, 0.25), (0, 0, -1, 0.2)]
    box2 = [(j, i, round(0.01 + 0.01 * ((j + 2) * 5 + i + 2), 2)) for j in range(-2, 3) for i in range(-2, 3)]
    cases = [(3, star3, (9, 11, 130 + 7), 1), (3, star3, (11, 13, 257), 2), (3, star3, (13, 15, 129 + 6 + 15), 3), (2, box2, (1, 23, 64 + 4 + 1), 1),
             (2, box2, (1, 19, 300), 2), (3, star3, (7, 9, 7), 1)]
    try:
        for ndim, pts, dims, step in cases:
            stc = str(tmp_path / ("f%d_%d.stc" % (ndim, dims[2])))
            write_stc(stc, ndim, dims, 4, pts)
            spec = oracle.Spec(stc, ndim, step)
            for dt in (np.float32, np.float64):
                a = rng.random(spec.shape).astype(dt)
                out = []
                for on in (1, 0):
                    oracle.set_avx512_path(on)
                    b = np.full_like(a, 7.0)             # the ring must stay untouched by both paths
                    oracle.sweep(spec, a, b, contract=1)
                    out.append(b)
                assert np.array_equal(out[0], out[1]), (ndim, dims, step, dt)
    finally:
        oracle.set_avx512_path(1)
