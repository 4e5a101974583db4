#!/usr/bin/env python3
"""bench.py -- the north-star measurement (BASELINE.json): GStencil/s and achieved HBM GB/s
vs the MI355X roofline on 3d7pt_star, fp32.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): C4 = 3d7pt_star 1024^3 fp32, `iterations 4` (the configuration
the metric is quoted on; 8 GiB for the two buffers, fits one GPU).  One "step" = the
reference's timed region (codegen.hpp:581-584): the ping-pong loop over the .stc's 4 time
steps = 2*ceil(4/(2*step)) kernel launches.  N > 1: the same 1024^3 grid cut into z slabs,
one process per GPU, RCCL halo exchange overlapped with the interior sweep (strong scaling).
`python bench.py --gpus N` without a launcher starts its own N rank processes (spawn_ranks: children started
before anything in this process touches HIP or imports torch); under `python -m torch.distributed.run` (RANK /
WORLD_SIZE in the environment) it is one of the ranks.

value      = grid-point updates of all ranks / max-over-ranks wall time, in GStencil/s
roofline   = algorithmic bytes (2*sizeof(T) per grid point per launch)
             / average launch duration measured with HIP events on the launch stream
cpu_baseline = the CPU oracle (oracle/, OpenMP) on a bounded z-slab sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = os.path.join(ROOT, "benchmarks", "configs")
WORKLOADS = {
    "c4": dict(stc=os.path.join(CFG, "c4_3d7pt_star_1024.stc"), ndim=3, dtype="fp32", name="3d7pt_star 1024^3 fp32 (BASELINE C4), iterations 4"),
    "c3": dict(stc=os.path.join(CFG, "c3_3d7pt_star_512.stc"), ndim=3, dtype="fp32", name="3d7pt_star 512^3 fp32 (BASELINE C3), iterations 4"),
    "c2": dict(stc=os.path.join(CFG, "c2_2d5pt_star_8192.stc"), ndim=2, dtype="fp32", name="2d5pt_star 8192^2 fp32 (BASELINE C2), iterations 4"),
    "c5": dict(stc=os.path.join(CFG, "c5_2d25pt_box_16384.stc"), ndim=2, dtype="fp64", name="2d25pt_box 16384^2 fp64 (BASELINE C5), iterations 4"),
    # the reference's native precision (codegen.hpp:148: double only) on the BASELINE grids; 8192^2 and 512^3 are its own shipped sizes
    # (benchmarks/2d5pt_star/2d5pt_star.stc:1-4, benchmarks/3d7pt_star/3d7pt_star.stc:1-5) -- SURVEY.md 8(f) rank 4
    "c2f64": dict(stc=os.path.join(CFG, "c2_2d5pt_star_8192.stc"), ndim=2, dtype="fp64", name="2d5pt_star 8192^2 fp64 (reference precision and size), iterations 4"),
    "c3f64": dict(stc=os.path.join(CFG, "c3_3d7pt_star_512.stc"), ndim=3, dtype="fp64", name="3d7pt_star 512^3 fp64 (reference precision and size), iterations 4"),
    "c4f64": dict(stc=os.path.join(CFG, "c4_3d7pt_star_1024.stc"), ndim=3, dtype="fp64", name="3d7pt_star 1024^3 fp64 (reference precision), iterations 4"),
    # the reference's other shipped benchmark directories as they stand (benchmarks/<name>/<name>.stc: 8192^2 / 512^3, fp64, iterations 4);
    # 3d7pt_star is c3f64 above.  2d9pt_cross: its spec's `iteratioins` typo (benchmarks/2d9pt_cross/2d9pt_cross.stc:4) leaves Iterations
    # unset in the reference (its emitted program times ZERO launches), so the workload runs the spec as shipped -- the
    # generated kernel says Iterations 0 like the reference's -- with the iteration count the other seven specs use given explicitly
    # to drs_kernel_run (`iterations`: 4)
    "s_2d5pt_star": dict(stc=os.path.join(ROOT, "benchmarks", "2d5pt_star", "2d5pt_star.stc"), ndim=2, dtype="fp64", name="2d5pt_star 8192^2 fp64 (shipped spec), iterations 4"),
    "s_2d5pt_cross": dict(stc=os.path.join(ROOT, "benchmarks", "2d5pt_cross", "2d5pt_cross.stc"), ndim=2, dtype="fp64", name="2d5pt_cross 8192^2 fp64 (shipped spec), iterations 4"),
    "s_2d9pt_box": dict(stc=os.path.join(ROOT, "benchmarks", "2d9pt_box", "2d9pt_box.stc"), ndim=2, dtype="fp64", name="2d9pt_box 8192^2 fp64 (shipped spec), iterations 4"),
    "s_2d9pt_star": dict(stc=os.path.join(ROOT, "benchmarks", "2d9pt_star", "2d9pt_star.stc"), ndim=2, dtype="fp64", name="2d9pt_star 8192^2 fp64 (shipped spec), iterations 4"),
    "s_2d25pt_box": dict(stc=os.path.join(ROOT, "benchmarks", "2d25pt_box", "2d25pt_box.stc"), ndim=2, dtype="fp64", name="2d25pt_box 8192^2 fp64 (shipped spec), iterations 4"),
    "s_3d9pt_cross": dict(stc=os.path.join(ROOT, "benchmarks", "3d9pt_cross", "3d9pt_cross.stc"), ndim=3, dtype="fp64", name="3d9pt_cross 512^3 fp64 (shipped spec), iterations 4"),
    "s_2d9pt_cross": dict(stc=os.path.join(ROOT, "benchmarks", "2d9pt_cross", "2d9pt_cross.stc"), ndim=2, dtype="fp64", iterations=4,
                          name="2d9pt_cross 8192^2 fp64 (shipped spec: `iteratioins` typo, Iterations unset), run with iterations 4 given explicitly"),
}
# tuned generator options per workload (found with drstencil_amd/tuner; logs under profiles/).
# Headline for the 3D workloads: two time steps per launch with the reference's own --step 2 arithmetic
# (algebraically fused 25-point stencil, one pass) -- bit-identical to the oracle.  32 lanes x 4 points per row,
# 16 lane rows x 2 rows (32 x 128 tile, halo fetched by the halo loader lanes), 32-plane stream blocks, three
# planes in flight per lane (software prefetch depth 3: the 200+ VGPRs of the fused kernel leave one workgroup
# per CU, so the bytes in flight have to come from depth -- profiles/r01_exp_r1z*_prefetch_depth.log), one x-y
# band per XCD.  The exhaustive 1780-configuration search
# (profiles/r01_tune_c4_s2_exhaustive.txt) puts the 32x16-lane and 64x8-lane fused kernels and the 66x15-lane
# temporal pipeline within a few per cent of each other; their order changes from device to device.
# Round 4: the option lists are no longer copied by hand from the tuner's logs.  HEADLINE names the PROBLEM each workload's headline kernel
# solves; geometry and emission options come from the tuner -> generator table (drstencil_amd/tuned_defaults.tsv, maintained by
# `tuning.py --write-defaults`), the same rows the generator applies when `drstencil` gets no geometry option -- so
# `drstencil --3d --dtype fp32 --step 2 c4_3d7pt_star_1024.stc` emits exactly TUNED["c4"]'s kernel (tests/test_cli_and_ir.py).
# Where each row came from (its `source` column) and what was measured against it -- history of the hand-kept dict this replaces:
    # round 2: + `-fno-slp-vectorize` for this kernel (the SLP vectoriser packs the 200 FMAs per lane and plane into v_pk_fma_f32,
    # whose register-pair operands cost 935 v_mov per 24 planes; without it 202 VGPRs instead of 226 and 0.4-1 % less time in four
    # interleaved comparisons, profiles/r02_exp_r2[a-d]*.log)
    # round 3, second half: with finite data and the output array's position tuned per configuration, two sweeps (165 + 403 configurations,
    # profiles/r03_tune_c4_s2_{top165,random700}_placement.txt) and an interleaved comparison (r03_exp_r3q.log: 1.452 ms against 1.488) put a
    # PINNED 1024-lane kernel first: 64 x 16 lanes (256 x 32 tile), 16-plane blocks, no prefetch, 95 VGPRs.  Rounds 1-3's headline (32 x 16
    # lanes, 32-plane blocks, prefetch depth 3, 204 VGPRs) stays as PREV_HEADLINE: a side measurement (the N > 1 runs take the new kernel too: -3...4 %
    # per ping-pong pair in the rehearsals of ranks of 8 and of 4, profiles/r03_bench_rehearse5_*.json)
    # 2D one-shot LDS tiles (BASELINE C2 "no temporal blocking (baseline LDS tile)", C5 "wide-halo LDS staging"): best of the
    # exhaustive 2D searches, profiles/r01_tune_c2_exhaustive.txt / r01_tune_c5_exhaustive.txt (0.78 of the HBM peak each)
    # round 3: the tile consumed by source row (--order rows: the row's own vector + DPP neighbours just before its FMAs, sums pinned), 8 rows per
    # lane: 0.694 ms against 0.717 for round 2's kernel in the same process (profiles/r03_exp_r3d.log), bit-identical
    # fp64 (profiles/r01_tune_shipped.md: exhaustive searches at the reference sizes): the 2D tile of 2d5pt_star step 1; fused step 2 in 3D
    # round 2: a tuner run over the space with the reference's dist dimension put a reuse-schedule kernel (32x8 lanes, --dist 2) 2 % ahead on
    # its box (profiles/r02_tune_c3f64_s2.txt); the interleaved comparison on another box has round 1's scatter kernel 4-5 % ahead at both sizes
    # (profiles/r02_exp_r2i_diagnostics.log: 0.374 vs 0.388 ms at 512^3, 3.04 vs 3.22 ms at 1024^3), so it stays, with -fno-slp-vectorize (+0.3 %)
    # round 3, second half: the pinned 1024-lane shape of the fp32 headline pays in fp64 too at 1024^3: 2.897 ms against 3.087 for round 1's kernel
    # (128 x 4 lanes, 32-plane blocks) in one process, finite data, arrays placed (profiles/r03_exp_r3s.log); at 512^3 it does not (c3f64 stays)
    # shipped specs: the best configurations of profiles/r01_tune_shipped.md at --step 2, the only step the reference's tuner sweeps
    # (benchmarks/*/tuning.py:110): fused kernels (bit-exact) for the order-1 stencils
    # 2d9pt_cross (diagonal cross of order 2; the reference's tuner sweeps it at --step 2 --dist 2, benchmarks/2d9pt_cross/tuning.py:127): fused, bit-exact
    # (rows order: 746 GStencil/s against 522 for the taps-order kernel of the same geometry, profiles/r03_exp_r3g.log)
    # the wide stencils (order 2 / 35 fused taps) are fastest as on-chip temporal pipelines; in fp64 those stay within 1e-12 of the fused
    # arithmetic (measured 1.7e-15), the bar the tests and bench.py's verification hold fp64 temporal kernels to
    # round 3: with the emitter in charge of the registers the FUSED (bit-exact) step-2 kernels of 2d9pt_star and 3d9pt_cross are faster than
    # round 2's temporal pipelines (749 vs 722 and 665 vs 611 GStencil/s in one process, profiles/r03_exp_r3d.log); 2d25pt_box step 2 (81 fp64
    # taps) stays a pipeline (574 vs 519)
HEADLINE = {
    "c4": dict(step=2), "c3": dict(step=2), "c2": dict(), "c5": dict(),
    "c2f64": dict(), "c3f64": dict(step=2), "c4f64": dict(step=2),
    "s_2d5pt_star": dict(step=2), "s_2d5pt_cross": dict(step=2, dist=2), "s_2d9pt_box": dict(step=2), "s_2d9pt_cross": dict(step=2),
    "s_2d9pt_star": dict(step=2), "s_2d25pt_box": dict(step=2, dist=4, temporal=1, streaming=True), "s_3d9pt_cross": dict(step=2),
}


def _tuned():
    from drstencil_amd import tuned_defaults as td
    rows = td.load()
    return {w: td.options_for(WORKLOADS[w]["stc"], WORKLOADS[w]["ndim"], WORKLOADS[w]["dtype"], rows=rows, **h) for w, h in HEADLINE.items()}


TUNED = _tuned()
# the same workloads with one time step per launch: highest roofline fraction.  Full-row tiles (256 lanes x 4
# points = N), 2 lane rows x 4 rows, 4-plane stream blocks, prefetch: the optimum of the exhaustive 1520-configuration
# search (profiles/r01_tune_c4_s1_exhaustive.txt) -- 80 % of the HBM peak, the chip's measured copy ceiling
def _step1(workloads):
    from drstencil_amd import tuned_defaults as td
    rows = td.load()
    return {w: td.options_for(WORKLOADS[w]["stc"], WORKLOADS[w]["ndim"], WORKLOADS[w]["dtype"], rows=rows) for w in workloads}


STEP1 = _step1(("c4", "c3"))      # the table's step-1 rows
# on-chip temporal blocking (two applications of the one-step stencil per launch; equal to the fused stencil up to
# rounding, 6.8e-7 relative at full size): 66 lanes x 4 = 264 columns own 256, so 4 tiles cover N = 1024 exactly
def _temporal(step, workloads):
    from drstencil_amd import tuned_defaults as td
    rows = td.load()
    return {w: td.options_for(WORKLOADS[w]["stc"], WORKLOADS[w]["ndim"], WORKLOADS[w]["dtype"], step=step, temporal=1, rows=rows) for w in workloads}


TEMPORAL2 = _temporal(2, ("c4", "c3"))
# round 4: three on-chip stages in the reference's precision, where they pay: 21 fp64 FMAs per point instead of the fused stencil's 63.  The
# SKEWED pipeline (--skew 1: stage t consumes what stage t-1 completed one iteration earlier, so an iteration's stages are independent and
# share two barriers), exact source-plane halo (--exact-y 1), pinned sums, 256-plane blocks, XCD work units (--xcd-remap 4):
# 3.21 ms per launch at 1024^3 = 986 GStencil/s against 868 for the fused --step 3 kernel and 730 for the step-2 headline
# (profiles/r04_exp_r4b/d.log); 1.6e-15 from the fused arithmetic (bar 1e-12, tolerance horizon 112 386 iterations)
TEMPORAL3 = _temporal(3, ("c4f64", "c3f64"))
# ... and FOUR stages: 726-lane workgroups (66 x 11 lanes, 12 wavefronts = 3 per SIMD: 151-163 VGPRs hold the 12 planes of sums), 132 x 22 tiles
# that own 128 x 16, two planes of prefetch: 4.08-4.17 ms per launch of four time steps = 1005-1028 GStencil/s at 1024^3 (profiles/r04_exp_r4p/q.log)
TEMPORAL4 = _temporal(4, ("c4f64", "c3f64"))
TEMPORALS = {w: [(3, TEMPORAL3[w]), (4, TEMPORAL4[w])] for w in TEMPORAL3}
# round 2: the same fused stencil with rotating register windows instead of carried partial sums (--schedule window): 126 VGPRs with
# -fno-slp-vectorize and --waves-per-eu 4, no scratch, so TWO 512-lane workgroups share a CU (one reads while the other writes) and 8-plane
# stream blocks cost nothing: +0.8 % over the headline in interleaved runs on one box (profiles/r02_exp_r2j/k_*.log) -- a side measurement
WINDOW2WG = {
    "c4": ["--3d", "--dtype", "fp32", "--step", "2", "--schedule", "window", "--prefetch", "--prefetch-depth", "1", "--waves-per-eu", "4", "--bx", "32", "--by", "16",
           "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "8", "--xcd-remap", "2", "--cc-opt", "-fno-slp-vectorize"],
}
# round 3: three time steps per launch with the reference's own --step 3 arithmetic (algebraically fused 63-point stencil, one pass), BIT-IDENTICAL
# to the oracle.  It fits because the emitter now owns its register allocation (--order rows: the arriving plane is consumed by source row;
# partial sums pinned where they are updated so that LLVM cannot sink the FMA chains and keep 7 planes of source windows alive -- 256 VGPRs +
# scratch before, 98-104 now, profiles/r03_sinking.md): two 512-lane workgroups per CU.  Best of a 1248-configuration grid and a 900-configuration
# random sweep (profiles/r03_tune_c4_s3_rows_grid.txt: 1.584 ms = 1998 GStencil/s on that box; 1.547 ms on another, 1.92 on the slowest one seen:
# these VALU-dense kernels vary far more from device to device than the memory-bound step-2 headline), so a few candidates are timed on the
# device at hand and the fastest one is reported -- what the tuner does, in miniature.  Side measurement: the headline and the roofline stay
# with the step-2 kernel (>= 0.70 of the HBM peak).
_S3 = ["--3d", "--dtype", "fp32", "--step", "3", "--prefetch", "--prefetch-depth", "1", "--block-merge-x", "4", "--order", "rows", "--pack", "0", "--cc-opt", "-fno-slp-vectorize"]
FUSED3 = {
    # candidates timed on the device at hand, on FINITE data (profiles/r03_exp_r3m.log: 1.84-1.90 ms; the 1.55-1.60 ms of earlier logs were
    # measured after the arrays had overflowed to inf, where the chip clocks higher)
    "c4": [_S3 + ["--bx", "64", "--by", "8", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"],
           _S3 + ["--bx", "64", "--by", "8", "--block-merge-y", "2", "--sn", "128", "--xcd-remap", "2"],
           _S3 + ["--xrim", "lds", "--bx", "64", "--by", "8", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"],      # x rim through LDS instead of DPP moves: the fastest on finite data
           _S3 + ["--xrim", "lds", "--bx", "64", "--by", "8", "--block-merge-y", "2", "--sn", "128", "--xcd-remap", "2"],
           _S3 + ["--bx", "32", "--by", "16", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"],
           [x for x in _S3 if x not in ("--pack", "0", "--cc-opt", "-fno-slp-vectorize")] + ["--bx", "64", "--by", "8", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"]],   # packed pairs
    # C2 (2d5pt_star 8192^2 fp32, one-shot LDS tiles): the fused 25-point (step 3) and 41-point (step 4) stencils, 2166 / 2732 GStencil/s bit-exact
    "c2": [["--dtype", "fp32", "--step", "4", "--bx", "128", "--by", "4", "--block-merge-x", "4", "--block-merge-y", "4", "--xcd-remap", "0", "--order", "rows", "--pack", "0"],
           ["--dtype", "fp32", "--step", "3", "--bx", "128", "--by", "4", "--block-merge-x", "4", "--block-merge-y", "4", "--xcd-remap", "0", "--order", "rows", "--pack", "0"]],
    # the reference's precision: fused --step 3 in fp64, 899 / 835 GStencil/s against 702 / 704 for the step-2 kernels (profiles/r03_exp_r3f.log)
    "c4f64": [["--3d", "--dtype", "fp64", "--step", "3", "--prefetch", "--prefetch-depth", "1", "--order", "rows", "--bx", "64", "--by", "8", "--block-merge-x", "2", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"],
              ["--3d", "--dtype", "fp64", "--step", "3", "--prefetch", "--prefetch-depth", "1", "--order", "rows", "--bx", "128", "--by", "4", "--block-merge-x", "2", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"],
              ["--3d", "--dtype", "fp64", "--step", "3", "--xrim", "lds", "--order", "rows", "--bx", "64", "--by", "8", "--block-merge-x", "2", "--block-merge-y", "2", "--sn", "128", "--xcd-remap", "2"],
              ["--3d", "--dtype", "fp64", "--step", "3", "--order", "rows", "--bx", "64", "--by", "8", "--block-merge-x", "2", "--block-merge-y", "2", "--sn", "128", "--xcd-remap", "2"]],
    "c3f64": [["--3d", "--dtype", "fp64", "--step", "3", "--prefetch", "--prefetch-depth", "1", "--order", "rows", "--bx", "64", "--by", "8", "--block-merge-x", "2", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "1"],
              ["--3d", "--dtype", "fp64", "--step", "3", "--prefetch", "--prefetch-depth", "1", "--order", "rows", "--bx", "128", "--by", "4", "--block-merge-x", "2", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2"],
              ["--3d", "--dtype", "fp64", "--step", "3", "--xrim", "lds", "--order", "rows", "--bx", "64", "--by", "8", "--block-merge-x", "2", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"]],
    "c3": [_S3 + ["--bx", "64", "--by", "8", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "1"],
           _S3 + ["--bx", "32", "--by", "16", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2"],
           _S3 + ["--xrim", "lds", "--bx", "64", "--by", "8", "--block-merge-y", "2", "--sn", "64", "--xcd-remap", "2"]],      # the fastest on finite data (r03_exp_r3n.log)
}
# N > 1 (z slabs of C4): the same fused kernel; slabs of 256 planes or fewer get 16-plane stream blocks.  One stream block
# per tile (256 tiles = one workgroup per CU) is the fastest way to sweep a slab ALONE (128-plane view 0.186 ms vs 0.199,
# 256: 0.380 vs 0.424, 512: 0.758 vs 0.804 -- profiles/r01_exp_r1zj_one_block_per_tile.log, r01_exp_r1zk_...), but its
# workgroups hold every CU until the launch ends, so the RCCL send/recv kernel cannot start beside it: in the rehearsal
# a rank of the 8-GPU run drops from 1200 to 1130 GStencil/s.  Short blocks retire every ~20 us and let it in.
PREV_HEADLINE = {
    # rounds 1-3's single-GPU headline (and, until the second half of round 3, the kernel the N > 1 runs cut into slab views): side measurement
    "c4": ["--3d", "--dtype", "fp32", "--step", "2", "--prefetch", "--prefetch-depth", "3", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2", "--cc-opt", "-fno-slp-vectorize"],
}


def slab_options(workload, world, weak=False):
    """Generator options of the slab-view kernels of an N > 1 run: the workload's single-GPU headline.  (C4: the pinned 64 x 16-lane kernel with
    16-plane blocks -- short blocks retire every ~10 us and let the RCCL send/recv kernel in beside the interior launch; rounds 1-3 used the
    32 x 16-lane kernel with 16-plane blocks at 4+ ranks.)"""
    return list(TUNED[workload])


def slab_alone_options(workload, world, weak=False):
    """Options of the launches that have the GPU to themselves (the whole-slab launch of a one-exchange-per-pair run), or None: the same kernels as
    the exchanging launches.  Rounds 1-3 used one stream block per tile there (0.186 vs 0.199 ms at 128 planes for the 32 x 16-lane kernel); the
    pinned 64 x 16-lane kernel with its short blocks is faster than that on every slab length (0.181-0.188 ms; profiles/r03_exp_r3r.log), and ONE
    block per tile is its worst case (0.231), so since the second half of round 3 there is no separate kernel.  DRS_SLAB_ALONE_SN=<n>: experiments."""
    sn = os.environ.get("DRS_SLAB_ALONE_SN")
    if sn and workload == "c4" and world >= 2 and not weak:
        opts = list(TUNED[workload])
        opts[opts.index("--sn") + 1] = sn
        return opts
    return None


MIN_WARM_S = 0.25     # untimed warm-up continues (beyond --warmup steps) until the GPU has been busy this long
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def kernels():
    """Every kernel bench.py can time: (id, workload, generator options).  __graft_entry__.build() prebuilds them and
    tests/gpu_cases.py::FULL holds a full-size parity case for each (tests/test_capi_and_tuner.py checks that)."""
    out = [("bench_%s_headline" % w, w, TUNED[w]) for w in sorted(TUNED)]
    out += [("bench_%s_prev_headline" % w, w, PREV_HEADLINE[w]) for w in sorted(PREV_HEADLINE)]
    out += [("bench_%s_step1" % w, w, STEP1[w]) for w in sorted(STEP1)]
    out += [("bench_%s_temporal2" % w, w, TEMPORAL2[w]) for w in sorted(TEMPORAL2)]
    out += [("bench_%s_temporal3" % w, w, TEMPORAL3[w]) for w in sorted(TEMPORAL3)]
    out += [("bench_%s_temporal4" % w, w, TEMPORAL4[w]) for w in sorted(TEMPORAL4)]
    out += [("bench_%s_window_two_workgroups" % w, w, WINDOW2WG[w]) for w in sorted(WINDOW2WG)]
    out += [("bench_%s_fused3_%d" % (w, i), w, o) for w in sorted(FUSED3) for i, o in enumerate(FUSED3[w])]
    return out


def kernel_arg_sets():
    """Argument vectors of kernels(); prebuilt by __graft_entry__.build()."""
    return [o + [WORKLOADS[w]["stc"]] for _, w, o in kernels()]


def pmc_traffic(workload, option_string):
    """HBM bytes per launch of this exact workload + kernel configuration from the committed rocprofv3 PMC
    passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/traffic_by_options.json): (bytes, where the number comes from), or
    (None, why not) when this configuration was not profiled -- the counters cannot be read inside a timed run."""
    path = os.path.join("profiles", "traffic_by_options.json")
    try:
        m = json.load(open(os.path.join(ROOT, path)))
        e = m[workload][option_string]
        return e["traffic_bytes_per_launch"], "%s [%s][%s] <- %s" % (path, workload, option_string, e.get("source", "?"))
    except Exception:
        return None, "no PMC passes committed for [%s][%s] in %s" % (workload, option_string, path)


def oracle_check_slabs(workload, step, slabs, temporal):
    """The checker half of the CPU leg: one oracle sweep per slab of the run's own input that verify_timed_kernel kept -- bottom of the grid,
    across a stream-block / tile-row boundary in the middle, top of the grid (byte offsets beyond 2^32 at 1024^3) -- against what the
    timed kernel wrote there in ONE launch: bit-exact, except temporal blocking (1e-6 relative fp32 / 1e-12 fp64)."""
    import numpy as np
    import oracle
    w = WORKLOADS[workload]
    if oracle.usable_cpus() < oracle.threads():       # as many OpenMP threads as this process may really run on (see cpu_baseline)
        oracle.set_threads(oracle.usable_cpus())
    spec = oracle.Spec(w["stc"], w["ndim"], step)
    L, M, N = spec.dims
    dt = np.float32 if w["dtype"] == "fp32" else np.float64
    h = spec.halo
    tol = 1e-6 if w["dtype"] == "fp32" else 1e-12
    check = {"ok": True, "max_rel": 0.0, "bit_exact_required": not temporal, "slabs": []}
    for sl in slabs:
        sub = np.ascontiguousarray(sl["input"], dtype=dt)
        nsl = sub.shape[0]
        dst = np.zeros_like(sub)
        cs = oracle.Spec(w["stc"], w["ndim"], step)
        if w["ndim"] == 3:
            cs.set_dims(nsl, M, N)
        else:
            cs.set_dims(1, nsl, N)
        oracle.sweep(cs, sub, dst, contract=1)
        ref, got = dst[h:nsl - h], sl["output"]
        def max_rel(sel):        # in chunks of 64 slices: a whole 1024^3 fp64 launch is 8.6 GB per array
            m = 0.0
            for z in range(0, ref.shape[0], 64):
                g_, r_ = got[z:z + 64][sel].astype(np.float64), ref[z:z + 64][sel].astype(np.float64)
                m = max(m, float(np.max(np.abs(g_ - r_) / np.maximum(np.abs(r_), 1e-30))))
            return m
        if temporal:
            rel = max_rel((slice(None),) + tuple(slice(h, d - h) for d in ref.shape[1:]))
            ok = rel <= tol
        else:
            ok = bool(np.array_equal(got, ref))
            rel = 0.0 if ok else max_rel((slice(None),) * ref.ndim)
        check["slabs"].append({"position": sl["label"], "first_slice": int(sl["z0"]), "slices": int(nsl), "ok": bool(ok), "max_rel": rel})
        check["ok"] = bool(check["ok"] and ok)
        check["max_rel"] = max(check["max_rel"], rel)
    return check


def cpu_baseline(workload, step, budget_s=4.0, host_slab=None, gpu_first_launch=None, temporal=False):
    """The CPU leg (the only place bench.py touches oracle/): the oracle (port) timed on the host cores on a bounded z/y-slab
    sample of the workload -- the first slices of the very array the GPU loop started from (host_slab), or seeded random
    numbers of the same shape -- and, as the checker, one oracle sweep of each slab of 2*Halo+12 slices verify_timed_kernel kept (bottom,
    across a block boundary, top: gpu_first_launch) against what the timed GPU kernel wrote there in one launch: bit-exact, except
    temporal blocking (1e-6 relative fp32 / 1e-12 fp64)."""
    import numpy as np
    import oracle
    w = WORKLOADS[workload]
    # as many OpenMP threads as this process may really run on: the GPU box's host reports 128 hardware threads, the container's
    # CPU quota may be 16 -- 128 threads on that quota spend most of their time throttled (round 3, first run: 2.3 GStencil/s)
    host_threads = max(oracle.threads(), os.cpu_count() or 1)      # (the checker may already have lowered OpenMP's count to the usable CPUs)
    usable = oracle.usable_cpus()
    if usable < host_threads:
        oracle.set_threads(usable)
    spec = oracle.Spec(w["stc"], w["ndim"], step)
    L, M, N = spec.dims
    dt = np.float32 if w["dtype"] == "fp32" else np.float64
    check = oracle_check_slabs(workload, step, gpu_first_launch, temporal) if gpu_first_launch else None
    # bounded sample: a slab of the outermost dim (same plane size, same stencil, same dtype)
    if w["ndim"] == 3:
        Ls = min(L, 128)
        spec.set_dims(Ls, M, N)
        sample = "SAMPLE: %d x %d x %d z-slab of the %d^3 grid" % (Ls, M, N, L)
    else:
        Ms = min(M, 4096)
        spec.set_dims(1, Ms, N)
        sample = "SAMPLE: %d x %d y-slab of the %d x %d grid" % (Ms, N, M, N)
    # both arrays are first touched by the OpenMP team that sweeps them (the GPU box's host has two sockets), then filled
    A = oracle.empty_first_touched(spec, dt)
    B = oracle.empty_first_touched(spec, dt)
    if host_slab is not None and tuple(host_slab.shape) == tuple(spec.shape):
        oracle.copy_into(spec, A, np.ascontiguousarray(host_slab, dtype=dt))
        sample += " (the GPU run's own input)"
    else:
        oracle.copy_into(spec, A, np.random.default_rng(1).random(spec.shape, dtype=dt))
    h = spec.halo
    interior = 1
    for d in spec.shape:
        interior *= d - 2 * h
    oracle.sweep(spec, A, B, 1)  # warm (page faults)
    t0 = time.perf_counter()
    sweeps = 0
    while True:
        oracle.sweep(spec, A, B, 1)
        oracle.sweep(spec, B, A, 1)
        sweeps += 2
        el = time.perf_counter() - t0
        if el > budget_s or sweeps >= 400:
            break
    gst = sweeps * step * interior / el / 1e9
    return dict(value=gst, unit="GStencil/s", cores=oracle.threads(), kind="port", isa=oracle.isa(),
                host_hardware_threads=host_threads,
                sample="%s, %d sweeps in %.1f s (OpenMP, %d threads = the CPUs this process may use of the host's %d, %s clone of the sweep, arrays first-touched by the team)"
                       % (sample, sweeps, el, oracle.threads(), host_threads, oracle.isa())), check


def verify_timed_kernel(torch, kern, workload, A, B, temporal):
    """What bench.py timed is what the parity tests check: ONE launch of the timed kernel on the run's own input against the
    emitted gold kernel on the whole grid (the reference's --check path, codegen.hpp:591-627), plus the untouched ring.
    Bit-exact, except temporal blocking (re-associated: 1e-6 relative fp32 / 1e-12 fp64).  A holds the input, B receives the
    output.  Returns (ok, details, host copy of the first <= 128 slices of A, [input / output slabs at Kernel.check_slabs positions]) --
    the last two feed the CPU leg (cpu_baseline), which times the oracle on the first and compares the slabs with it."""
    w = WORKLOADS[workload]
    h = kern.info["halo"]
    tol = 1e-6 if w["dtype"] == "fp32" else 1e-12
    G = torch.zeros_like(A)
    B.zero_()
    kern.launch(A.data_ptr(), B.data_ptr())
    kern.launch_gold(A.data_ptr(), G.data_ptr())
    torch.cuda.synchronize()
    inner = tuple(slice(h, d - h) for d in A.shape)
    if temporal:
        rel = float(((B[inner] - G[inner]).abs() / G[inner].abs().clamp_min(1e-30)).max())
        gold_ok = rel <= tol
    else:
        gold_ok = torch.equal(B, G)
        rel = 0.0 if gold_ok else float(((B[inner] - G[inner]).abs() / G[inner].abs().clamp_min(1e-30)).max())
    ring_ok = int(torch.count_nonzero(B)) == int(torch.count_nonzero(B[inner]))      # the ring of the output is never written
    del G
    nsl = min(A.shape[0], 2 * h + 12)
    keep = min(A.shape[0], 128 if w["ndim"] == 3 else 4096)
    host = A[:keep].cpu().numpy()
    slabs = [{"label": label, "z0": z0, "input": A[z0:z0 + nsl].cpu().numpy(), "output": B[z0 + h:z0 + nsl - h].cpu().numpy()}
             for label, z0 in kern.check_slabs(nsl)]
    # ... and the WHOLE grid of that launch (the oracle sweeps 1024^3 in a second or two on the box's cores): no plane is left to the gold
    # kernel alone.  DRS_BENCH_WHOLE_GRID_MAX_GB (input + output on the host; default 20) bounds it; 0 keeps the three slabs only
    if 2.0 * A.numel() * A.element_size() <= float(os.environ.get("DRS_BENCH_WHOLE_GRID_MAX_GB", "20")) * 2 ** 30:
        slabs.append({"label": "whole_grid", "z0": 0, "input": A.cpu().numpy(), "output": B[h:A.shape[0] - h].cpu().numpy()})
    return bool(gold_ok and ring_ok), {
        "vs_gold_kernel_full_grid": {"ok": bool(gold_ok), "max_rel": rel, "bit_exact_required": not temporal},
        "vs_cpu_oracle_slab": None,       # filled by the CPU leg (cpu_baseline) unless --no-cpu-baseline
        "ring_untouched": bool(ring_ok), "tolerance": 0.0 if not temporal else tol}, host, slabs


def _seeded_planes(torch, lo, hi, rest, dtype, device):
    """Planes [lo, hi) of a grid whose plane z is torch.rand with seed 1000 + z: any rank can produce any plane of the global grid."""
    out = torch.empty((hi - lo,) + tuple(rest), dtype=dtype, device=device)
    g = torch.Generator(device=device)
    for z in range(lo, hi):
        g.manual_seed(1000 + z)
        out[z - lo].copy_(torch.rand(tuple(rest), dtype=dtype, device=device, generator=g))
    return out


def slab_verify_view(dim0, H, launches, world, rank):
    """Global range [lo, hi) of the slab a rank recomputes WITHOUT exchange to check its exchanged run: its own planes plus
    launches*H planes per interior face (each launch invalidates H more planes from the cut faces inwards)."""
    from drstencil_amd.multigpu import slab_bounds
    z0, z1 = slab_bounds(dim0, world, rank)
    return max(0, z0 - launches * H), min(dim0, z1 + launches * H)


def oracle_check_slab_run(torch, run, workload, step, dims, H, launches, n, rank, world, tdt):
    """The CPU oracle as the checker of an N > 1 run (the checker half of the CPU leg; nothing is timed here): this rank's own planes after the
    exchanged run against `n` oracle sweeps of the same seeded global grid on [z0 - launches*H, z1 + launches*H) -- wide enough that the planes
    the rank owns cannot be reached from the cut faces.  Bit for bit (the slab kernels are the fused gold-order headline).  136 planes of 1024^2
    per rank at N = 8: a fraction of a second on the node's cores, shared out between the ranks."""
    import numpy as np
    import oracle
    w = WORKLOADS[workload]
    p = run.plan
    lo, hi = slab_verify_view(dims[0], H, launches, world, rank)
    A = _seeded_planes(torch, lo, hi, tuple(dims[1:]), tdt, run.A.device).cpu().numpy()
    B = np.zeros_like(A)
    oracle.set_threads(max(1, min(oracle.threads(), oracle.usable_cpus() // max(1, world))))
    cs = oracle.Spec(w["stc"], w["ndim"], step)
    if w["ndim"] == 3:
        cs.set_dims(hi - lo, dims[1], dims[2])
    else:
        cs.set_dims(1, hi - lo, dims[1])
    for _ in range(n // 2):
        oracle.sweep(cs, A, B, contract=1)
        oracle.sweep(cs, B, A, contract=1)
    okA = bool(np.array_equal(run.owned(run.A).cpu().numpy(), A[p.z0 - lo:p.z1 - lo]))
    okB = bool(np.array_equal(run.owned(run.B).cpu().numpy(), B[p.z0 - lo:p.z1 - lo]))
    return okA and okB


def verify_slab_run(torch, dist, run, sweep, dims, H, launches, iters, rank, world, dev, tdt, workload=None, step=None):
    """N > 1: is the decomposed run (slab views, boundary / interior / pair kernels, RCCL halo exchange over xGMI) the
    single-domain run?  Every rank fills its slab with seeded planes of the GLOBAL grid, runs the reference's loop through
    SlabRun (with exchange), then recomputes the same launches on a wider slab of the same global grid with plain launches
    and NO exchange -- wide enough that its own planes cannot be reached by the missing neighbours -- and compares its own
    planes bit for bit.  No rank needs another rank's result; the verdicts are AND-ed over the ranks."""
    gpu = dev.type == "cuda"          # (the gloo test runs this on CPU tensors with the oracle as the sweep)
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    p = run.plan
    rest = tuple(dims[1:])
    run.A.copy_(_seeded_planes(torch, p.lo, p.hi, rest, tdt, dev))
    run.B.zero_()
    sync()
    dist.barrier()
    n = run.run(iterations=iters)
    sync()
    lo, hi = slab_verify_view(dims[0], H, launches, world, rank)
    A = _seeded_planes(torch, lo, hi, rest, tdt, dev)
    B = torch.zeros_like(A)
    stream = torch.cuda.current_stream(dev).cuda_stream if gpu else 0
    for _ in range(n // 2):
        sweep(A, B, stream)
        sweep(B, A, stream)
    sync()
    okA = torch.equal(run.owned(run.A), A[p.z0 - lo:p.z1 - lo])
    okB = torch.equal(run.owned(run.B), B[p.z0 - lo:p.z1 - lo])
    del A, B
    # ... and against the CPU oracle (GPU runs of a named workload): the same planes, the same seeded grid, `n` oracle sweeps.  A rank whose
    # oracle cannot be loaded or runs out of host memory reports that (ok: None) and the verdict rests on the comparison above
    ok_oracle, oracle_note = None, "not run (CPU tensors / no workload named)"
    if gpu and workload is not None and not os.environ.get("DRS_BENCH_NO_SLAB_ORACLE"):
        try:
            sys.path.insert(0, ROOT)
            ok_oracle, oracle_note = oracle_check_slab_run(torch, run, workload, step, dims, H, launches, n, rank, world, tdt), "own planes == %d oracle sweeps of the same slab" % n
        except Exception as e:       # the checker must not take the run down
            ok_oracle, oracle_note = None, "oracle check failed to run on rank %d: %r" % (rank, e)
    from drstencil_amd.multigpu import coll_device
    mine = bool(okA and okB and ok_oracle is not False)
    flag = torch.tensor([1 if mine else 0, 1 if ok_oracle else 0], dtype=torch.int32, device=coll_device(torch, dist, dev))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag[0])), {"decomposed_vs_single_domain": {"ok": bool(int(flag[0])), "this_rank_ok": bool(okA and okB), "launches": n, "bit_exact_required": True,
                                                                "how": "own planes of the exchanged run == plain launches on [z0 - %d, z1 + %d) of the same seeded global grid" % (launches * H, launches * H)},
                                "vs_cpu_oracle_own_planes": {"ok_on_every_rank": bool(int(flag[1])), "this_rank_ok": ok_oracle, "note": oracle_note, "bit_exact_required": True}}


def device_info(torch, dev):
    """Which MI355X this was (boxes differ by a few per cent in memory clocks): name, uuid, CU count -- from the HIP device
    properties, no child process."""
    try:
        p = torch.cuda.get_device_properties(dev)
        return {"name": p.name, "arch": getattr(p, "gcnArchName", None), "uuid": str(getattr(p, "uuid", "")), "compute_units": p.multi_processor_count,
                "total_memory_GiB": round(p.total_memory / 2**30, 1)}
    except Exception as e:       # diagnostics only
        return {"error": repr(e)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--trial-steps", type=int, default=3, help="N > 1 with --exchange-every 0: steps of the run itself timed in each exchange mode before the timed loop (the faster mode is taken)")
    ap.add_argument("--kernel-args", default=None, help="override the generator options (space separated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the workload's grid cut into N slabs (default, SURVEY 8e); weak = every rank holds a full-size slab (grid L*N planes)")
    ap.add_argument("--exchange-every", type=int, default=0, choices=[0, 1, 2],
                    help="N > 1: launches per halo exchange (2: ghost planes twice as wide, one exchange per ping-pong pair; "
                         "0 = measured during warm-up on this machine, multigpu.measure_exchange_every)")
    ap.add_argument("--slab-runtime", default="torch", choices=["torch", "native"],
                    help="N > 1: torch = drstencil_amd.multigpu.SlabRun (torch.distributed send/recv; the reference implementation); native = the "
                         "C ABI's drs_slab_* entry points (RCCL called directly, one ping-pong pair captured into a HIP graph)")
    ap.add_argument("--placement", default="measured", choices=["measured", "kernel", "separate"],
                    help="N = 1: where the output array sits relative to the input array (launch time depends on (out - in) mod 64 MiB): measured on "
                         "this device before the timed region (default), the kernel's own recommendation, or two separate allocations")
    ap.add_argument("--headline-only", action="store_true", help="skip the side measurements (profiling runs: one dr_ kernel in the trace)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-loop comparison of the timed kernel with the gold kernel and the oracle")
    ap.add_argument("--n1-value", type=float, default=None, help="N > 1: the N = 1 value of the same workload; the line then carries value / (N * n1) as efficiency_vs_n1")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="N > 1 as a plain command: wall-clock limit for the rank processes this launcher starts (then terminate, kill, exit 124)")
    ap.add_argument("--prebuild-only", action="store_true", help="build (or find cached) every kernel the run needs for all --gpus ranks, then exit; no GPU is touched")
    return ap.parse_args(argv)


def child_command(argv):
    """Command line of one rank process.  DRS_BENCH_CHILD (a JSON list) replaces `python bench.py` -- the CPU test uses it
    to record what would be started."""
    stub = os.environ.get("DRS_BENCH_CHILD")
    return (json.loads(stub) if stub else [sys.executable, os.path.abspath(__file__)]) + list(argv)


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               DRS_NO_COMPILE="1")      # every kernel was prebuilt by the parent's prebuild child: a rank never starts hipcc
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` as a plain command: start N rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set), relay rank 0's JSON line, return the worst exit code.  This process imports neither torch nor the HIP
    runtime: a launcher hop is only legal before any GPU call.  Kernels are built once, by a child, before the ranks start."""
    import socket
    import subprocess
    n = args.gpus
    rc = subprocess.call(child_command(list(argv) + ["--prebuild-only"]))
    if rc != 0:
        print("bench.py: prebuild failed (rc %d)" % rc, file=sys.stderr)
        return rc or 1
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [subprocess.Popen(child_command(argv), env=rank_env(r, n, port), stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)) for r in range(n)]
    worst = 0
    # a rank that hangs (RCCL rendezvous, a collective whose partner never arrives) would keep this loop alive for ever: after
    # --rank-timeout seconds of wall clock the ranks are terminated, then killed, and the launcher exits 124
    deadline = time.time() + max(1.0, args.rank_timeout)
    try:
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                worst = worst or code
                if code != 0:            # a rank died: the others would wait for it in RCCL for ever
                    for q in alive:
                        procs[q].terminate()
            if alive and time.time() > deadline:
                print("bench.py: rank(s) %s still running after %.0f s: terminating them" % (sorted(alive), args.rank_timeout), file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
                t_kill = time.time() + 10.0
                while time.time() < t_kill and any(procs[q].poll() is None for q in alive):
                    time.sleep(0.1)
                worst = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    out = procs[0].stdout.read() if procs[0].stdout else ""
    sys.stdout.write(out)
    sys.stdout.flush()
    return worst


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    rehearse = os.environ.get("DRS_REHEARSE")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not rehearse and not args.prebuild_only:
        raise SystemExit(spawn_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not args.prebuild_only and not rehearse and world != args.gpus:       # decided before anything touches the GPU
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    if world > 1 and not args.prebuild_only:
        # a rank that is still here after --rank-timeout seconds (a collective that never completes on a fabric this code has not met) says where
        # it is stuck and exits, instead of hanging until somebody else's clock runs out: a C-level watchdog thread, so it fires inside a blocked
        # RCCL / HIP call too.  (Started as a plain command, the launcher's own deadline in spawn_ranks fires first.)
        import faulthandler
        faulthandler.dump_traceback_later(max(30.0, args.rank_timeout + 30.0), exit=True, file=sys.stderr)
    # stdout carries exactly ONE line, the result: RCCL prints a version banner on stdout when its first communicator comes up
    # (rank 0, five lines), and libraries may print more -- so from here on file descriptor 1 is stderr, and the JSON line goes
    # to the saved descriptor at the end
    result_fd = 1
    if not args.prebuild_only:
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)

    import drstencil_amd as drs

    if not os.path.exists(drs.LIB_PATH):
        # a checkout without built artefacts: build the host library (g++ only, seconds); ranks take turns on a lock
        import fcntl
        import subprocess
        with open(os.path.join(ROOT, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if not os.path.exists(drs.LIB_PATH):
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "drstencil_amd", "csrc")], stdout=subprocess.DEVNULL)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DRS_REHEARSE="r/R": run the N > 1 code path as middle rank r of R on ONE GPU (process group of size 1, the rank is
    # its own neighbour) -- a rehearsal of the multi-GPU branch where only one GPU exists; the line says so
    prank, pworld = (int(x) for x in rehearse.split("/")) if rehearse else (rank, world)
    if rehearse and world != 1:
        raise SystemExit("bench.py: DRS_REHEARSE runs in ONE process")
    if args.prebuild_only:
        pworld = max(args.gpus, pworld)
    # Generate + compile (or find cached) every kernel BEFORE HIP is initialised: a process
    # that has touched the GPU must not fork/exec the compiler.
    w = WORKLOADS[args.workload]
    opts = args.kernel_args.split() if args.kernel_args else (slab_options(args.workload, pworld, args.scaling == "weak") if pworld > 1 else TUNED[args.workload])
    spec = drs.Spec(w["stc"], w["ndim"], int(opts[opts.index("--step") + 1]) if "--step" in opts else 1)
    L, M, N = spec.dims
    H, step, iters = spec.halo, spec.step, spec.iterations
    if w.get("iterations") and iters <= 0:      # a spec without a (readable) iteration count: the workload names one
        iters = int(w["iterations"])
    weak = args.scaling == "weak" and pworld > 1
    if weak:       # the outermost dim grows with the number of ranks: fixed work per GPU
        if w["ndim"] == 3:
            L *= pworld
        else:
            M *= pworld
    kern1 = kernf = kernw = kernp = None
    kernts = []
    kern3 = []
    if pworld == 1:
        kern = drs.Kernel(opts + [w["stc"]])
        if args.workload in STEP1 and not args.kernel_args and not args.headline_only:
            kern1 = drs.Kernel(STEP1[args.workload] + [w["stc"]])
            kernf = drs.Kernel(TEMPORAL2[args.workload] + [w["stc"]])
        if args.workload in TEMPORALS and not args.kernel_args and not args.headline_only:
            kernts = [(st_, drs.Kernel(o_ + [w["stc"]]), o_) for st_, o_ in TEMPORALS[args.workload]]
        if not args.kernel_args and not args.headline_only:
            for o3 in FUSED3.get(args.workload, []):
                try:
                    kern3.append((drs.Kernel(o3 + [w["stc"]]), o3))
                except drs.KernelBuildError:          # a compiler that needs scratch for it: not a candidate
                    pass
            if args.workload in PREV_HEADLINE and not args.kernel_args:
                kernp = drs.Kernel(PREV_HEADLINE[args.workload] + [w["stc"]])      # rounds 1-3's headline: side measurement (continuity)
            if args.workload in WINDOW2WG:
                try:
                    kernw = drs.Kernel(WINDOW2WG[args.workload] + [w["stc"]])
                except drs.KernelBuildError:          # a compiler that does not reach 126 registers without scratch: no side measurement
                    kernw = None
    else:
        from drstencil_amd.multigpu import HipSweep, SelfNeighbourRun, SlabPlan, SlabRun, measure_exchange_every
        auto_every = args.exchange_every == 0      # decided after the process group is up, from measured sweep / exchange times
        sweep = HipSweep(w["stc"], opts, os.path.join(ROOT, "drstencil_amd", "_kcache"),
                         alone_opts=None if args.kernel_args else slab_alone_options(args.workload, pworld, args.scaling == "weak"))
        kern_n1 = None
        if not args.kernel_args and args.scaling != "weak":
            kern_n1 = drs.Kernel(TUNED[args.workload] + [w["stc"]])      # rank 0 measures the same box's single-GPU rate before the slab run
        if args.slab_runtime == "native" and args.scaling == "weak":
            raise SystemExit("bench.py: --slab-runtime native runs the spec's own grid (strong scaling)")
        for r in (range(pworld) if args.prebuild_only else (prank,)):
            for ev in ((1, 2) if auto_every else (args.exchange_every,)):      # both modes' kernels: built (cache hits) before HIP is up
                sweep.prebuild(SlabPlan(L if w["ndim"] == 3 else M, H, pworld, r, ev))
                if os.environ.get("DRS_EXP_SLAB") == "folded":        # timing experiment (multigpu.SlabRun.launch): boundary + interior as one view
                    sp_ = SlabPlan(L if w["ndim"] == 3 else M, H, pworld, r, ev)
                    sweep.kernel((sp_.bot[1] if sp_.bot else sp_.interior[1]) - (sp_.top[0] if sp_.top else sp_.interior[0]))
            if not args.no_verify and not rehearse:       # the wider no-exchange slab of verify_slab_run
                vlo, vhi = slab_verify_view(L if w["ndim"] == 3 else M, H, spec.launches, pworld, r)
                sweep.kernel(vhi - vlo)
    if args.prebuild_only:
        print("bench.py: kernels for %d rank(s) of %s are in the cache" % (pworld, args.workload), file=sys.stderr)
        return
    os.environ["DRS_NO_COMPILE"] = "1"      # from here on a cache miss is an error, not a hipcc child of a GPU process
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # DRS_BENCH_BACKEND=gloo DRS_BENCH_ONE_GPU=1: a rehearsal of the real multi-process N > 1 path where only one GPU exists -- every rank
    # process computes on cuda:0 and the halo planes travel through a gloo group staged in host memory (multigpu.batch_p2p).  Everything
    # but the RCCL transport is the code of a real run: the launcher, the ranks' own kernels (first, middle, last), the exchange
    # choreography, the self-check.  The line says so, and its value is meaningless as a rate (<= 6 rank processes per GPU on the pool).
    backend = os.environ.get("DRS_BENCH_BACKEND", "nccl")
    one_gpu = bool(os.environ.get("DRS_BENCH_ONE_GPU"))
    if (backend != "nccl" or one_gpu) and args.slab_runtime == "native":
        raise SystemExit("bench.py: the native slab runtime calls RCCL directly (no gloo / one-GPU rehearsal)")
    gpu_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    dist = None
    if pworld > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        from drstencil_amd.multigpu import nccl_options
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=nccl_options(dist))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev         # where the few-element tensors of the collectives below live (the CPU under gloo)
    if dist is not None:
        from drstencil_amd.multigpu import coll_device
        cdev = coll_device(torch, dist, dev)

    tdt = torch.float32 if w["dtype"] == "fp32" else torch.float64
    esz = 4 if w["dtype"] == "fp32" else 8
    launches_per_step = 2 * (-(-iters // (2 * step))) if iters > 0 else 0      # codegen.hpp:581-584 (== spec.launches when the spec names the count)
    interior = (M - 2 * H) * (N - 2 * H) * ((L - 2 * H) if w["ndim"] == 3 else 1)
    # time steps after which an array started from U[0, 1) has overflowed: the values grow by the sum of |coefficients| per step
    import math
    growth = sum(abs(c) for _, c, _ in drs.Spec(w["stc"], w["ndim"], 1).points)
    finite_steps = int(math.log(3.0e38 if w["dtype"] == "fp32" else 1.0e308) / math.log(growth)) if growth > 1.0 else 10 ** 9
    npoints = M * N * (L if w["ndim"] == 3 else 1)

    ev_ms = 0.0
    warm_extra = 0
    calibration = None
    if pworld == 1:
        g = torch.Generator(device=dev).manual_seed(1)
        shape = (L, M, N) if w["ndim"] == 3 else (M, N)
        # Both arrays in ONE allocation, the output placed relative to the input: a z-streaming kernel's launch time depends on
        # (out - in) mod 64 MiB (profiles/r03_probe_skew4.log: C4 headline 1.47 ms in the good half of the period, 1.65-1.68 in the worst
        # eighth).  --placement measured (default): the position is measured on this device before anything is timed; kernel: the
        # generator's recommendation (kernel info out_skew_bytes); separate: two independent allocations, as rounds 1-2 did
        placement = {"mode": args.placement}
        if args.placement == "separate":
            A = torch.rand(shape, dtype=tdt, device=dev, generator=g)
            B = torch.zeros_like(A)
        else:
            A, B, _arena = kern.alloc_pair(torch, dev, dtype=tdt, calibrate=(args.placement == "measured"))
            A.copy_(torch.rand(shape, dtype=tdt, device=dev, generator=g))
            B.zero_()
            placement.update({"out_minus_in_mod_period_bytes": kern.pair_skew_bytes, "period_bytes": kern.info.get("placement_period_bytes"),
                              "kernel_recommendation_bytes": kern.info.get("out_skew_bytes")})
            if args.placement == "measured":
                placement["measured_ms_fwd_bwd_by_skew_MiB"] = {str(sk >> 20): [round(f, 4), round(b, 4)] for sk, f, b in kern.skew_calibration}
        R = A.clone()         # the pristine input: restored before every timed loop (below)
        stream = torch.cuda.current_stream(dev)
        for _ in range(args.warmup):
            kern.run(A.data_ptr(), B.data_ptr(), iterations=iters, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        # a few milliseconds of launches do not bring the GPU to its steady clocks (C3: 221 us per launch right after 2 ms of
        # warm-up, 193 us from then on -- scripts/archive/c3_loop_probe.py): keep running untimed steps until MIN_WARM_S have gone by
        tw = time.perf_counter()
        while args.warmup > 0 and time.perf_counter() - tw < MIN_WARM_S:
            for _ in range(8):
                kern.run(A.data_ptr(), B.data_ptr(), iterations=iters, stream=stream.cuda_stream)
                warm_extra += 1
            torch.cuda.synchronize()
        # FINITE DATA in every timed launch.  The shipped coefficients sum to 1.5 (0.3 + 6 x 0.2), so the values grow 1.5 x per time step and
        # a float array started from U[0, 1) is all inf after ~218 time steps -- which the warm-up above has long passed.  inf / NaN operands
        # cost the VALU less power and the chip clocks higher on them: the VALU-dense fused step-3 kernel takes 1.80-1.87 ms on finite data
        # and 1.53-1.60 once the arrays have overflowed (scripts/archive/probe_cold2.py, profiles/r03_probe_cold2.log; the memory-bound step-2
        # headline: 1.50 either way).  So the input is restored from a pristine copy right before every timed loop, and a loop longer than
        # the overflow horizon is timed in chunks with the restore between them (outside the events).
        def reseed():
            A.copy_(R)
            B.zero_()
        A0_, B0_ = A, B
        chunk = max(1, min(args.steps, int(0.9 * finite_steps) // max(1, launches_per_step * step)))
        ev_ms, el, n, done = 0.0, 0.0, 0, 0
        while done < args.steps:
            reseed()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(min(chunk, args.steps - done)):
                n += kern.run(A.data_ptr(), B.data_ptr(), iterations=iters, stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            el += time.perf_counter() - t0
            ev_ms += e0.elapsed_time(e1)
            done += min(chunk, args.steps - done)
        assert n == launches_per_step * args.steps
        kinfo = kern.info
        kres = kern.resources
        parallelism = "1 GPU"
        def side(k, o, iters, pair=None):
            # side measurement on the same grid (reference protocol: warm-up launches, then the timed
            # ping-pong loop bracketed by HIP events).  pair = (A, B) of the kernel's own arena (its own measured placement) instead of the headline's
            hz = k.info.get("tolerance_horizon_iterations", -1)
            reseed()                                          # finite data (see above); warm-up 4 launches + the loop stay inside the overflow horizon
            A, B = (A0_, B0_) if pair is None else pair
            if pair is not None:
                A.copy_(R); B.zero_()
            iters = max(2 * k.info["step"], min(iters, int(0.8 * finite_steps) - 4 * k.info["step"]))
            if k.info.get("arithmetic") == "reassociated" and not k.info.get("temporal_forced") and 0 < hz < iters:
                # a temporal pipeline keeps the tolerance up to its horizon only (drs_kernel_run refuses more): timed in loops of that length
                n1, ms1 = 0, 0.0
                for rep in range(-(-iters // hz)):
                    n_, ms_ = k.run_timed(A.data_ptr(), B.data_ptr(), iterations=hz, warmup=4 if rep == 0 else 0, stream=stream.cuda_stream)
                    n1, ms1 = n1 + n_, ms1 + ms_
            else:
                n1, ms1 = k.run_timed(A.data_ptr(), B.data_ptr(), iterations=iters, warmup=4, stream=stream.cuda_stream)
            by = k.bytes_per_launch()
            out_ = {"generator_options": " ".join(o), "step": k.info["step"], "arithmetic": k.info.get("arithmetic"), "tolerance_horizon_iterations": k.info.get("tolerance_horizon_iterations"),
                    "GStencil_per_s": k.updates_per_launch() * n1 / (ms1 * 1e-3) / 1e9,
                    "avg_launch_ms": ms1 / n1, "achieved_GBps": by * n1 / (ms1 * 1e-3) / 1e9,
                    "roofline_frac": by * n1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if not args.no_verify:
                # no number without a check in the same run: ONE launch on the pristine input against the emitted gold kernel, whole grid -- bit for
                # bit, or within the tolerance for a re-associated pipeline (the oracle-slab check is the headline's and the 3- / 4-stage pipelines';
                # every side kernel is also a full-size parity case of the GPU suite, tests/gpu_cases.py FULL)
                G_ = torch.zeros_like(A)
                A.copy_(R); B.zero_()
                k.launch(A.data_ptr(), B.data_ptr())
                k.launch_gold(A.data_ptr(), G_.data_ptr())
                torch.cuda.synchronize()
                if k.info.get("arithmetic") == "reassociated":
                    h_ = k.info["halo"]
                    inner_ = tuple(slice(h_, d_ - h_) for d_ in A.shape)
                    rel_ = float(((B[inner_] - G_[inner_]).abs() / G_[inner_].abs().clamp_min(1e-30)).max())
                    out_["verified_vs_gold_kernel_full_grid"] = {"ok": bool(rel_ <= (1e-6 if w["dtype"] == "fp32" else 1e-12)), "max_rel": rel_, "bit_exact_required": False}
                else:
                    out_["verified_vs_gold_kernel_full_grid"] = {"ok": bool(torch.equal(B, G_)), "bit_exact_required": True}
                del G_
            return out_
        step1 = side(kern1, STEP1[args.workload], 16) if kern1 is not None else None
        fused2 = side(kernf, TEMPORAL2[args.workload], 32) if kernf is not None else None
        window2 = side(kernw, WINDOW2WG[args.workload], 32) if kernw is not None else None
        prev_headline = side(kernp, PREV_HEADLINE[args.workload], 32) if kernp is not None else None
        temporals = {}
        for st_, kt, ot in kernts:
            # its own arena: the pipeline's read front runs several planes further ahead of its write front than the headline's, so its good
            # (out - in) mod 64 MiB differs (profiles/r04_exp_r4c.log: 3.36 ms at 0 MiB, 3.26 at 32) -- measured on this device like the headline's
            t_pair = None
            if args.placement != "separate":
                At_, Bt_, _arena_t = kt.alloc_pair(torch, dev, dtype=tdt, calibrate=(args.placement == "measured"))
                t_pair = (At_, Bt_)
            tk = side(kt, ot, 8 * st_, pair=t_pair)
            if t_pair is not None:
                tk["placement_out_minus_in_mod_period_bytes"] = kt.pair_skew_bytes
                del At_, Bt_, _arena_t, t_pair
                torch.cuda.empty_cache()
            tk["vgprs"], tk["lds_bytes"], tk["stages"], tk["threads"] = kt.resources.get("vgprs"), kt.info["lds_bytes"], kt.info.get("stages"), kt.info["threads"]
            tk["drift_estimate"] = kt.info.get("drift_estimate")
            tk["traffic"], tk["traffic_source"] = pmc_traffic(args.workload, " ".join(ot))
            if not args.no_verify:      # like the headline: one launch against the gold kernel on the whole grid (tolerance) + three oracle slabs (CPU leg)
                g3 = torch.Generator(device=dev).manual_seed(1)
                A.copy_(torch.rand(shape, dtype=tdt, device=dev, generator=g3))
                tk["verified"], tk["verification"], _, tk["_slabs"] = verify_timed_kernel(torch, kt, args.workload, A, B, True)
                if not args.no_cpu_baseline:      # checked right away: the host copies of a whole 1024^3 fp64 launch are 17 GB per pipeline
                    sys.path.insert(0, ROOT)
                    tc = oracle_check_slabs(args.workload, tk["step"], tk.pop("_slabs"), True)
                    tk["verification"]["vs_cpu_oracle_slab"] = tc
                    tk["verified"] = bool(tk["verified"] and tc["ok"])
            temporals[st_] = tk
        fused3 = None
        if kern3:
            # every candidate timed on THIS device, on finite data (side() restores the pristine input first), the fastest reported in full
            cands = [(side(k3, o3, 24), k3) for k3, o3 in kern3]
            best3, kbest = max(cands, key=lambda c: c[0]["GStencil_per_s"])
            fused3 = dict(best3)
            fused3["candidates_timed_on_this_device"] = [{"generator_options": c["generator_options"], "avg_launch_ms": c["avg_launch_ms"], "GStencil_per_s": c["GStencil_per_s"]} for c, _ in cands]
            fused3["vgprs"], fused3["scratch_bytes_per_lane"] = kbest.resources.get("vgprs"), kbest.resources.get("scratch_bytes_per_lane")
            if not args.no_verify:      # one launch against the emitted gold kernel on the whole grid: bit for bit
                G3 = torch.zeros_like(A)
                g3 = torch.Generator(device=dev).manual_seed(1)
                A.copy_(torch.rand(shape, dtype=tdt, device=dev, generator=g3))      # fresh input: the timing loops have long overflowed the old one
                B.zero_()
                kbest.launch(A.data_ptr(), B.data_ptr())
                kbest.launch_gold(A.data_ptr(), G3.data_ptr())
                torch.cuda.synchronize()
                fused3["verified_vs_gold_kernel_full_grid_bit_exact"] = bool(torch.equal(B, G3))
                del G3
        verified, verification, host_slab, first_out = None, None, None, None
        if not args.no_verify:
            g = torch.Generator(device=dev).manual_seed(1)
            A.copy_(torch.rand(shape, dtype=tdt, device=dev, generator=g))      # the input the timed loop started from
            verified, verification, host_slab, first_out = verify_timed_kernel(torch, kern, args.workload, A, B, kinfo.get("stages", 1) > 1)
    else:
        # the N = 1 reference of THIS machine, measured by rank 0 before the slab run (VERDICT r02 item 4 iii: the driver's `bench.py
        # --gpus N` passes no --n1-value): the single-GPU headline kernel on a scratch copy of the whole grid, same protocol as N = 1
        n1_value = args.n1_value
        if n1_value is None and kern_n1 is not None:
            n1 = torch.zeros(2, dtype=torch.float64, device=cdev)
            if rank == 0:
                shape1 = (L, M, N) if w["ndim"] == 3 else (M, N)
                # the same protocol as the N = 1 line: both arrays in one arena at the measured position, finite data in the timed loops
                if args.placement == "separate":
                    A1 = torch.rand(shape1, dtype=tdt, device=dev)
                    B1 = torch.zeros_like(A1)
                    arena1 = None
                else:
                    A1, B1, arena1 = kern_n1.alloc_pair(torch, dev, dtype=tdt, calibrate=(args.placement == "measured"))
                R1 = torch.rand(shape1, dtype=tdt, device=dev)
                A1.copy_(R1); B1.zero_()
                st1 = torch.cuda.current_stream(dev)
                tw = time.perf_counter()
                while time.perf_counter() - tw < MIN_WARM_S:
                    for _ in range(8):
                        kern_n1.run(A1.data_ptr(), B1.data_ptr(), iterations=iters, stream=st1.cuda_stream)
                    torch.cuda.synchronize()
                n_1, ms_1 = 0, 0.0
                for _ in range(3):
                    A1.copy_(R1); B1.zero_()
                    a_, b_ = kern_n1.run_timed(A1.data_ptr(), B1.data_ptr(), iterations=4 * iters, warmup=0, stream=st1.cuda_stream)
                    n_1, ms_1 = n_1 + a_, ms_1 + b_
                n1[0] = kern_n1.updates_per_launch() * n_1 / (ms_1 * 1e-3) / 1e9
                n1[1] = ms_1 / n_1
                del A1, B1, R1, arena1
                torch.cuda.empty_cache()
            dist.all_reduce(n1, op=dist.ReduceOp.MAX)
            n1_value = float(n1[0])
            n1_launch_ms = float(n1[1])
        else:
            n1_launch_ms = None
        dims_ = (L, M, N) if w["ndim"] == 3 else (M, N)

        def make_run(ev):
            """One rank of the slab run with an exchange every `ev` launches: buffers, streams, (native: communicator and graph), placement."""
            if args.slab_runtime == "native":
                from drstencil_amd.multigpu import NativeSlabRun
                r_ = NativeSlabRun(torch, dist, w["stc"], opts, dims_, H, step, iters, prank, world, dev, tdt, every=ev,
                                   alone_opts=None if args.kernel_args else slab_alone_options(args.workload, pworld, args.scaling == "weak"),
                                   rehearse_world=pworld if rehearse else 0, cache_dir=os.path.join(ROOT, "drstencil_amd", "_kcache"))
            else:
                r_ = (SelfNeighbourRun if rehearse else SlabRun)(torch, dist, dims_, H, step, iters, prank, pworld, sweep, dev, tdt, every=ev)
            r_.bench_placement = None
            if args.placement == "measured" and w["ndim"] == 3:
                # where this rank's output slab sits relative to its input slab, measured with LOCAL launches of the run's longest-lived kernel
                # (no exchange, nothing collective; DESIGN.md section 3 "Placement of the two arrays")
                try:
                    from drstencil_amd.multigpu import calibrate_slab_placement
                    p_ = r_.plan
                    kcal = sweep.kernel(p_.Lloc, alone=True) if (p_.every == 2 and sweep.alone_opts is not None) else sweep.kernel(max(p_.views()))
                    r_.bench_placement = calibrate_slab_placement(torch, r_, kcal)
                except Exception as e:       # a kernel that is not in the cache, a CPU run: the arrays stay where slab_pair put them
                    r_.bench_placement = {"skipped": repr(e)}
            return r_

        def fill_finite(r_, seed):
            g_ = torch.Generator(device=dev).manual_seed(seed)
            r_.A.copy_(torch.rand(r_.A.shape, dtype=tdt, device=dev, generator=g_))      # finite data: a few hundred time steps overflow the slab (N = 1 branch)
            r_.B.zero_()
            torch.cuda.synchronize()     # the run's streams are not the one that filled A

        def timed_steps(r_, nsteps):
            """nsteps runs of the reference's loop on the slab between barriers: (wall seconds, event ms, launches), MAX over the ranks."""
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0_ = time.perf_counter()
            e0_.record(r_.main)
            n_ = 0
            for _ in range(nsteps):
                n_ += r_.run()
            e1_.record(r_.main)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            el_ = time.perf_counter() - t0_
            return el_, e0_.elapsed_time(e1_), n_

        run_alt = None
        if auto_every:
            # (1) the model: sweep / exchange / boundary times measured piecewise, MAX over ranks (multigpu.decide_exchange_every);
            # (2) the trial: the run itself in BOTH modes for a few steps between barriers -- first contact with a real link is not the
            #     moment to trust a model fitted on one-GPU rehearsals.  The trial decides; the model's choice is reported beside it.
            model_every, calibration = measure_exchange_every(torch, dist, dims_, H, prank, pworld, sweep, dev, tdt, self_neighbour=bool(rehearse))
            runs, trial = {}, {}
            from drstencil_amd.multigpu import slab_bounds
            thinnest = min(slab_bounds(dims_[0], pworld, r_)[1] - slab_bounds(dims_[0], pworld, r_)[0] for r_ in range(pworld))
            for ev in (1, 2):
                if thinnest < 2 * ev * H:    # some rank's slab is too thin for ghosts this wide: decided from ALL ranks' sizes, so every rank skips alike
                    continue
                runs[ev] = make_run(ev)
                fill_finite(runs[ev], 1 + prank)
                for _ in range(2):
                    runs[ev].run()
                el_, _, _ = timed_steps(runs[ev], args.trial_steps)
                tt = torch.tensor([el_], dtype=torch.float64, device=cdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                trial[ev] = float(tt[0]) * 1e3 / args.trial_steps
            chosen = min(trial, key=lambda e_: trial[e_])
            calibration = dict(calibration, model_every=model_every, trial_steps=args.trial_steps,
                               trial_ms_per_step={"every_%d" % e_: round(v_, 4) for e_, v_ in sorted(trial.items())}, chosen_every=chosen,
                               decided_by="trial of the run itself in both modes (MAX over ranks); the piecewise model is reported beside it")
            args.exchange_every = chosen
            run = runs.pop(chosen)
            run_alt = next(iter(runs.values()), None)
        else:
            run = make_run(args.exchange_every)
        slab_placement = run.bench_placement
        fill_finite(run, 1 + prank)
        for _ in range(args.warmup):
            run.run()
        torch.cuda.synchronize()
        nwarm = max(0, int(MIN_WARM_S / 0.003) * (1 if weak else pworld) - args.warmup) if args.warmup > 0 else 0   # ~MIN_WARM_S of runs; the same count on every rank (exchanges pair up)
        for _ in range(nwarm):
            run.run()
            warm_extra += 1
        fill_finite(run, 101 + prank)      # finite data in the timed loop: the warm-up has overflowed the slab (N = 1 branch)
        el, ev_ms, n = timed_steps(run, args.steps)
        mine = torch.tensor([el * 1e3 / max(args.steps, 1)], dtype=torch.float64, device=cdev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        rank_ms_per_step = [float(x[0]) for x in per_rank]
        t = torch.tensor([el, ev_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, ev_ms = float(t[0]), float(t[1])
        # A/B in the same run: the mode that was NOT chosen, same protocol, same number of steps (side measurement: `value` is the chosen mode's)
        exchange_ab = None
        if run_alt is not None:
            fill_finite(run_alt, 101 + prank)
            for _ in range(2):
                run_alt.run()
            fill_finite(run_alt, 101 + prank)
            el_a, _, _ = timed_steps(run_alt, args.steps)
            ta = torch.tensor([el_a], dtype=torch.float64, device=cdev)
            dist.all_reduce(ta, op=dist.ReduceOp.MAX)
            exchange_ab = {"every_%d_ms_per_step" % run.plan.every: el * 1e3 / args.steps, "every_%d_ms_per_step" % run_alt.plan.every: float(ta[0]) * 1e3 / args.steps,
                           "value_is": "every_%d" % run.plan.every, "steps_each": args.steps}
        # what one exchanging launch is made of, per rank (collective: the exchanges pair up), and who the ranks are
        tl = run.timeline(4) if hasattr(run, "timeline") else None
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else backend
        except Exception as e:
            rccl = repr(e)
        me = {"rank": prank, "device_ordinal": gpu_index, "device": device_info(torch, dev), "rccl": rccl, "every": run.plan.every,
              "planes_owned": run.plan.z1 - run.plan.z0, "ms_per_step": rank_ms_per_step[rank],
              "calibration": (calibration or {}).get("this_rank"), "launch_timeline_us": tl, "slab_placement": run.bench_placement}
        rank_reports = [None] * world
        dist.all_gather_object(rank_reports, me)
        kinfo = sweep.kernel(run.plan.interior[1] - run.plan.interior[0]).info
        kres = sweep.kernel(run.plan.interior[1] - run.plan.interior[0]).resources
        if args.slab_runtime == "native":
            calibration = dict(calibration or {}, native_runtime=run.slab.info)
        parallelism = "%s-slab x%d%s, RCCL send/recv halo every %d launch(es), overlapped" % ("z" if w["ndim"] == 3 else "y", pworld, " (weak: grid %s)" % "x".join(str(d) for d in ((L, M, N) if w["ndim"] == 3 else (M, N))) if weak else "", args.exchange_every)
        step1 = fused2 = window2 = fused3 = prev_headline = None
        temporals = {}
        verified, verification, host_slab, first_out = None, None, None, None     # the slab kernels' parity is tests/test_gpu_parity.py::test_c4_slab_views_at_full_size
        if not args.no_verify and not rehearse:      # (a rehearsal's self-neighbour exchange is not the physical one)
            verified, verification = verify_slab_run(torch, dist, run, sweep, (L, M, N) if w["ndim"] == 3 else (M, N), H, launches_per_step, iters, prank, pworld, dev, tdt,
                                                     workload=None if args.kernel_args else args.workload, step=step)

    if rank == 0:
        total_launches = launches_per_step * args.steps
        updates = total_launches * step * interior
        if rehearse:    # one rank's share of the work only: the other ranks do not exist
            updates /= pworld
            parallelism = "REHEARSAL on one GPU of rank %d of %d (self-neighbour exchange through RCCL): %s" % (prank, pworld, parallelism)
        if pworld > 1 and (backend != "nccl" or one_gpu):
            parallelism = "REHEARSAL: %d rank processes on %s, halo planes through a %s group%s -- not a rate: %s" % (
                pworld, "ONE GPU" if one_gpu else "their GPUs", backend, " staged in host memory" if backend == "gloo" else "", parallelism)
        value = updates / el / 1e9
        # roofline of the dominant kernel dr_<name>: algorithmic bytes per launch / average
        # launch duration from the HIP events around the timed launches
        alg_bytes = 2.0 * esz * npoints / max(pworld, 1)    # per launch per GPU
        avg_launch_s = (ev_ms * 1e-3) / total_launches
        achieved = alg_bytes / avg_launch_s / 1e9
        traffic, traffic_source = pmc_traffic(args.workload, " ".join(opts)) if pworld == 1 else (None, "single-GPU PMC passes only")
        out = {
            "metric": "GStencil/s (grid-point updates/s), 3d7pt_star" if args.workload in ("c3", "c4") else "GStencil/s (grid-point updates/s)",
            "value": value, "unit": "GStencil/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": args.warmup + warm_extra,
            "ms_per_step": el * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f32" if w["dtype"] == "fp32" else "f64", "data": "synthetic", "verified": verified,
            "config": {"workload": w["name"], "generator_options": " ".join(opts), "step": step,
                       "launches_per_step": launches_per_step, "parallelism": parallelism,
                       "kernel": "dr_" + kinfo["name"], "threads": kinfo["threads"], "lds_bytes": kinfo["lds_bytes"],
                       "vgprs": kres.get("vgprs"), "agprs": kres.get("agprs"), "scratch_bytes_per_lane": kres.get("scratch_bytes_per_lane"),
                       "occupancy_waves_per_simd": kres.get("occupancy_waves_per_simd")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_launch_s * 1e3},
            "verification": verification,
            "device": device_info(torch, dev),
        }
        if pworld == 1:
            out["config"]["placement"] = placement
            out["config"]["finite_data"] = {"overflow_horizon_time_steps": finite_steps, "time_steps_per_step": launches_per_step * step, "steps_per_chunk": chunk,
                                            "chunks": -(-args.steps // chunk), "note": "the pristine U[0, 1) input is restored before every timed loop (outside the events)"}
        elif slab_placement is not None:
            out["config"]["placement"] = dict(slab_placement, mode="measured, rank %d" % prank)
        if pworld > 1:
            out["rank_ms_per_step"] = rank_ms_per_step
            out["ranks"] = rank_reports          # per rank: device, RCCL version, exchange mode, its own calibration numbers, the launch timeline
            out["exchange_ab"] = exchange_ab      # both exchange modes timed in this run (None when --exchange-every fixed one)
            if n1_value:
                # strong and weak alike: N GPUs against N times one GPU's rate (a rehearsal's value is ONE rank's share of the work)
                out["efficiency_vs_n1"] = value / n1_value if rehearse else value / (pworld * n1_value)
                out["n1_reference"] = {"value": n1_value, "avg_launch_ms": n1_launch_ms,
                                       "source": "--n1-value" if args.n1_value else "measured by rank 0 on this machine before the slab run: the single-GPU headline kernel on the whole grid"}
        if calibration:
            out["config"]["exchange_calibration"] = calibration      # measured on this machine during warm-up (multigpu.measure_exchange_every)
        out["step1_kernel"] = step1                 # one time step per launch: highest roofline fraction
        out["temporal_step2_kernel"] = fused2       # on-chip temporal blocking (2 stages): equal to the headline up to rounding
        # round 4: skewed 3- and 4-stage pipelines (fp64 workloads): equal to the fused --step 3 / 4 arithmetic within 1e-12
        out["temporal_step3_kernel"] = temporals.get(3)
        out["temporal_step4_kernel"] = temporals.get(4)
        # round 3: several time steps per launch with the reference's own fused --step n arithmetic, bit for bit (C3 / C4: --step 3, C2: --step 4)
        out["fused_multistep_kernel"] = fused3
        out["fused_step3_kernel"] = fused3 if (fused3 and fused3["step"] == 3) else None
        if fused3:
            out["best_bit_exact_GStencil_per_s"] = max(value, fused3["GStencil_per_s"])
        if prev_headline is not None:
            out["rounds_1_to_3_headline_kernel"] = prev_headline    # 32 x 16 lanes, 32-plane blocks, prefetch depth 3 (204 VGPRs)
        out["two_workgroups_per_cu_kernel"] = window2   # the same fused arithmetic from rotating register windows at 126 VGPRs: two workgroups per CU
        sides_ = [v_ for v_ in (step1, fused2, window2, prev_headline, fused3) + tuple(temporals.values()) if v_ and "verified_vs_gold_kernel_full_grid" in v_]
        if sides_:
            out["side_kernels_verified"] = all(v_["verified_vs_gold_kernel_full_grid"]["ok"] for v_ in sides_)
        if not args.no_cpu_baseline and pworld == 1:
            sys.path.insert(0, ROOT)
            out["cpu_baseline"], oracle_check = cpu_baseline(args.workload, step, host_slab=host_slab, gpu_first_launch=first_out, temporal=kinfo.get("stages", 1) > 1)
            if oracle_check is not None:       # the CPU leg is also the checker of what was timed
                out["verification"]["vs_cpu_oracle_slab"] = oracle_check
                out["verified"] = bool(out["verified"] and oracle_check["ok"])
            for tk in temporals.values():      # ... and of the pipelines' launches (their own step: the fused --step n arithmetic)
                if tk.get("_slabs"):
                    tc = oracle_check_slabs(args.workload, tk["step"], tk.pop("_slabs"), True)
                    tk["verification"]["vs_cpu_oracle_slab"] = tc
                    tk["verified"] = bool(tk["verified"] and tc["ok"])
        else:
            out["cpu_baseline"] = None
        for tk in temporals.values():
            tk.pop("_slabs", None)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if world > 1:
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
    # a number from a kernel whose result is wrong is not a measurement: the line says "verified": false AND the exit code says so
    # (every rank holds the same AND-ed verdict of an N > 1 run; at N = 1 rank 0 folds the oracle's verdict in above)
    if (out["verified"] if rank == 0 else verified) is False:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
