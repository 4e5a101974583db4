"""GPU parity tests (run with -m gpu on a MI355X): the HIP path, called through the C ABI
(libdrstencil_amd.so -> generated plugin), against the CPU oracle on the same seeded inputs,
against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through
size-independent properties (dr == gold kernel, frozen ring, warm-up idempotence).

Bar: fp32 within 1e-6 relative (BASELINE.json north_star), fp64 within 1e-12; every kernel
the generator emits keeps the gold summation order as an FMA chain, so the tests also
require BIT-EXACT agreement with the oracle's contracted mode."""
import json

import numpy as np
import pytest

import oracle
from gpu_cases import FULL, SMALL, golden_args
from helpers import golden_cases, load_golden

pytestmark = pytest.mark.gpu

REL_TOL = {"fp32": 1e-6, "fp64": 1e-12}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _dtype(opts):
    return "fp32" if "fp32" in opts else "fp64"


def _np_dtype(dt):
    return np.float32 if dt == "fp32" else np.float64


def _step(opts):
    return int(opts[opts.index("--step") + 1]) if "--step" in opts else 1


def run_hip(torch, kern, A_host, B_host, gold=False):
    dA = torch.from_numpy(A_host).cuda()
    dB = torch.from_numpy(B_host).cuda()
    n = kern.run(dA.data_ptr(), dB.data_ptr(), gold=gold)
    torch.cuda.synchronize()
    return n, dA.cpu().numpy(), dB.cpu().numpy()


@pytest.mark.parametrize("cid,ndim,stc,opts", SMALL, ids=[c[0] for c in SMALL])
def test_hip_vs_oracle_seeded(torch_cuda, cid, ndim, stc, opts):
    import drstencil_amd as drs
    torch = torch_cuda
    dt = _dtype(opts)
    kern = drs.Kernel(opts + [stc])
    spec = oracle.Spec(stc, ndim, _step(opts))
    assert kern.info["halo"] == spec.halo and kern.info["iterations"] == spec.iterations
    A0 = oracle.fill_random(spec.shape, _np_dtype(dt))
    B0 = np.zeros_like(A0)
    A_ref, B_ref = A0.copy(), B0.copy()
    n_ref = oracle.run(spec, A_ref, B_ref, contract=1)
    n, A, B = run_hip(torch, kern, A0, B0)
    assert n == n_ref
    for got, ref in ((A, A_ref), (B, B_ref)):
        m = oracle.check(spec, got, ref)
        assert m["max_rel"] <= REL_TOL[dt], (cid, m)
        # ring (never written by the reference) must be untouched, interior bit-exact
        assert np.array_equal(got, ref), (cid, "not bit-exact", m)
    # the emitted gold kernel agrees too (the reference's own check path)
    n, Ag, Bg = run_hip(torch, kern, A0, B0, gold=True)
    assert np.array_equal(Ag, A_ref) and np.array_equal(Bg, B_ref)


@pytest.mark.parametrize("case", golden_cases())
def test_hip_vs_reference_golden_fixture(torch_cuda, case):
    """HIP path (fp64) against arrays produced by the reference-emitted gold statement.
    The fixtures were computed without FMA contraction, the kernels contract: 1e-12."""
    import drstencil_amd as drs
    torch = torch_cuda
    meta, a0, a_ref, b_ref = load_golden(case)
    opts, stc = golden_args(case, meta)
    kern = drs.Kernel(opts + [stc])
    assert kern.info["halo"] == meta["macros"]["Halo"]
    n, A, B = run_hip(torch, kern, np.ascontiguousarray(a0), np.zeros_like(a0))
    assert n == meta["launches"]
    spec = oracle.Spec(stc, meta["ndim"], meta["step"])
    for got, ref in ((A, a_ref), (B, b_ref)):
        m = oracle.check(spec, got, ref)
        assert m["max_rel"] <= 1e-12, (case, m)
        h = spec.halo
        ring = np.ones(got.shape, bool)
        ring[tuple(slice(h, s - h) for s in got.shape)] = False
        assert np.array_equal(got[ring], ref[ring])


@pytest.mark.parametrize("cid,ndim,stc,opts", FULL, ids=[c[0] for c in FULL])
def test_full_size_properties(torch_cuda, cid, ndim, stc, opts):
    """BASELINE.json sizes: dr kernel == gold kernel bit for bit, ring frozen, warm-up
    launches idempotent, and a checksum against the oracle on a bounded slab."""
    import drstencil_amd as drs
    torch = torch_cuda
    dt = _dtype(opts)
    tdt = torch.float32 if dt == "fp32" else torch.float64
    kern = drs.Kernel(opts + [stc])
    i = kern.info
    shape = (i["L"], i["M"], i["N"]) if ndim == 3 else (i["M"], i["N"])
    g = torch.Generator(device="cuda").manual_seed(1234)
    A0 = torch.rand(shape, dtype=tdt, device="cuda", generator=g)
    A = A0.clone(); B = torch.zeros_like(A)
    Ag = A0.clone(); Bg = torch.zeros_like(A)
    # warm-up launches (codegen.hpp:575-578) must be idempotent
    kern.launch(A.data_ptr(), B.data_ptr())
    torch.cuda.synchronize()
    B1 = B.clone()
    kern.launch(A.data_ptr(), B.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(B, B1)
    B.zero_()
    n = kern.run(A.data_ptr(), B.data_ptr())
    ng = kern.run(Ag.data_ptr(), Bg.data_ptr(), gold=True)
    torch.cuda.synchronize()
    assert n == ng == i["iterations"] // (2 * i["step"]) * 2 + (2 if i["iterations"] % (2 * i["step"]) else 0)
    assert torch.equal(A, Ag), cid
    assert torch.equal(B, Bg), cid
    h = i["halo"]
    inner = tuple(slice(h, s - h) for s in shape)
    ringA = A.clone(); ringA[inner] = 0
    ring0 = A0.clone(); ring0[inner] = 0
    assert torch.equal(ringA, ring0), "A's ring must keep the initial values"
    ringB = B.clone(); ringB[inner] = 0
    assert not ringB.any(), "B's ring must stay zero"
    del Ag, Bg, ringA, ring0, ringB, B1
    # oracle on a bounded sub-domain: one sweep of the first 2h+8 outermost slices
    nsl = 2 * h + 8
    sub = A0[:nsl].contiguous().cpu().numpy()
    dst = np.zeros_like(sub)
    spec = oracle.Spec(stc, ndim, _step(opts))
    if ndim == 3:
        spec.set_dims(nsl, shape[1], shape[2])
    else:
        spec.set_dims(1, nsl, shape[1])
    oracle.sweep(spec, sub, dst, contract=1)
    B.zero_()
    kern.launch(A0.data_ptr(), B.data_ptr())
    torch.cuda.synchronize()
    got = B[h:nsl - h].cpu().numpy()
    assert np.array_equal(got, dst[h:nsl - h]), cid


def test_dpp_wave_shift_semantics(torch_cuda):
    """--xrim dpp relies on wave_shr:1 / wave_shl:1 moving data by one lane across the whole
    64-lane wavefront on gfx950: a dpp kernel and an lds kernel must agree bit for bit."""
    import drstencil_amd as drs
    from gpu_cases import stc as stcp
    torch = torch_cuda
    s = stcp("t2_box25")
    k1 = drs.Kernel(["--dtype", "fp32", "--streaming", "--xrim", "lds", s])
    k2 = drs.Kernel(["--dtype", "fp32", "--streaming", "--xrim", "dpp", s])
    A0 = torch.rand((k1.info["M"], k1.info["N"]), dtype=torch.float32, device="cuda")
    o1 = torch.zeros_like(A0); o2 = torch.zeros_like(A0)
    k1.launch(A0.data_ptr(), o1.data_ptr()); k2.launch(A0.data_ptr(), o2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)


def test_native_library_is_the_path():
    import drstencil_amd as drs
    assert drs.lib() is not None
    maps = open("/proc/self/maps").read()
    assert "libdrstencil_amd.so" in maps
