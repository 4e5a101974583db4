"""Kernel configurations exercised on the GPU (tests -m gpu) and prebuilt by
__graft_entry__.build() so that the GPU box finds them in drstencil_amd/_kcache."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STC = os.path.join(ROOT, "tests", "stc")
CFG = os.path.join(ROOT, "benchmarks", "configs")


def stc(name):
    return os.path.join(STC, name + ".stc")


# (id, ndim, stc, options)
SMALL = []


def _add(cid, ndim, name, *opts):
    SMALL.append((cid, ndim, stc(name), (["--3d"] if ndim == 3 else []) + list(opts)))


for dt in ("fp32", "fp64"):
    _add("3d7_%s_default" % dt, 3, "t3_star", "--dtype", dt)
    _add("3d7_%s_prefetch_dpp" % dt, 3, "t3_star", "--dtype", dt, "--prefetch", "--xrim", "dpp", "--sn", "16")
    _add("3d7_%s_step2" % dt, 3, "t3_star", "--dtype", dt, "--step", "2", "--sn", "32")
    _add("2d5_%s_tile" % dt, 2, "t2_star", "--dtype", dt)
    _add("2d5_%s_stream" % dt, 2, "t2_star", "--dtype", dt, "--streaming", "--sn", "40")
    _add("2d25_%s_tile" % dt, 2, "t2_box25", "--dtype", dt)
    _add("2d25_%s_stream_dpp" % dt, 2, "t2_box25", "--dtype", dt, "--streaming", "--xrim", "dpp", "--prefetch")
_add("3d7_fp32_step3", 3, "t3_star", "--dtype", "fp32", "--step", "3", "--sn", "16")
# every tap ahead along the streamed dimension (none on the output's own plane / row): found by tests/fuzz_shapes.py, seed 31
_add("2d_ahead_fp64_stream_scatter", 2, "t2_ahead", "--dtype", "fp64", "--streaming", "--sn", "32", "--prefetch")
_add("2d_ahead_fp32_stream_step2", 2, "t2_ahead", "--dtype", "fp32", "--streaming", "--sn", "128", "--step", "2", "--dist", "2", "--schedule", "scatter")
_add("2d_ahead_fp64_tile_step2", 2, "t2_ahead", "--dtype", "fp64", "--step", "2")
_add("3d_ahead_fp32_scatter", 3, "t3_ahead", "--dtype", "fp32", "--sn", "8", "--schedule", "scatter", "--prefetch")
_add("3d_ahead_fp64_step2_reuse", 3, "t3_ahead", "--dtype", "fp64", "--sn", "8", "--step", "2", "--dist", "2")
_add("3d_ahead_fp64_window_dma", 3, "t3_ahead", "--dtype", "fp64", "--sn", "8", "--schedule", "window", "--stage", "dma")
_add("3d7_fp32_eager", 3, "t3_star", "--dtype", "fp32", "--lazy-rims", "0", "--prefetch")
_add("3d7_fp32_cyclicy", 3, "t3_star", "--dtype", "fp32", "--cyclic-merge-y", "3", "--by", "2", "--bx", "32", "--sn", "9")
_add("3d7_fp32_step2_xcd_units", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--sn", "8", "--xcd-remap", "4", "--bx", "16", "--by", "4", "--block-merge-y", "2")
_add("3d7_fp64_t3_skew", 3, "t3_star", "--dtype", "fp64", "--step", "3", "--temporal", "1", "--skew", "1", "--pin", "1", "--exact-y", "1", "--bx", "34", "--by", "8", "--block-merge-y", "2", "--sn", "16", "--xcd-remap", "4")
_add("3d7_fp64_t4_auto", 3, "t3_star", "--dtype", "fp64", "--step", "4", "--temporal", "1", "--prefetch", "--prefetch-depth", "2", "--bx", "36", "--by", "11", "--block-merge-y", "2", "--sn", "16", "--xcd-remap", "4")
_add("3d7_fp32_t2_skew_rows", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--skew", "1", "--order", "rows", "--prefetch", "--bx", "34", "--by", "8", "--block-merge-y", "2", "--sn", "16")
# round 4: --cyclic-merge-x is the reference's strided layout (a lane's points Bx columns apart; codegen.hpp:116-141)
_add("3d7_fp32_cyclicx", 3, "t3_star", "--dtype", "fp32", "--cyclic-merge-x", "4", "--bx", "32", "--by", "4", "--block-merge-y", "2", "--sn", "9")
_add("3d7_fp64_cyclicx_step2", 3, "t3_star", "--dtype", "fp64", "--step", "2", "--cyclic-merge-x", "2", "--bx", "64", "--by", "4", "--block-merge-y", "2", "--sn", "8", "--prefetch")
_add("2d25_fp64_cyclicx_tile", 2, "t2_box25", "--dtype", "fp64", "--cyclic-merge-x", "4", "--bx", "32", "--by", "4", "--block-merge-y", "2")
_add("3d7_fp32_bx128", 3, "t3_star", "--dtype", "fp32", "--bx", "128", "--by", "2", "--block-merge-y", "2", "--xrim", "dpp")
_add("3d7_fp32_refdefaults", 3, "t3_star", "--dtype", "fp32", "--ref-defaults")
_add("3d7_fp32_nt", 3, "t3_star", "--dtype", "fp32", "--nt-store", "1", "--nt-load", "1", "--xcd-remap", "0")
_add("3d7_fp32_oddN", 3, "t3_star_odd", "--dtype", "fp32")
_add("3d7_fp64_oddN_dpp", 3, "t3_star_odd", "--dtype", "fp64", "--xrim", "dpp", "--block-merge-x", "2", "--bx", "16", "--by", "4")
_add("3d9x_fp32", 3, "t3_cross", "--dtype", "fp32", "--dist", "2")
_add("3d9x_fp64_step2_dpp", 3, "t3_cross", "--dtype", "fp64", "--dist", "2", "--step", "2", "--xrim", "dpp")
_add("3dodd_fp32", 3, "t3_odd", "--dtype", "fp32")
_add("3dodd_fp64_step2", 3, "t3_odd", "--dtype", "fp64", "--step", "2")
_add("2d5x_fp32", 2, "t2_cross", "--dtype", "fp32", "--dist", "2")
_add("2d5x_fp64_stream", 2, "t2_cross", "--dtype", "fp64", "--dist", "2", "--streaming")
_add("2d9b_fp32", 2, "t2_box9", "--dtype", "fp32", "--xrim", "dpp")
_add("2d9s_fp32_stream", 2, "t2_star9", "--dtype", "fp32", "--streaming", "--prefetch")
_add("2d9x_fp64", 2, "t2_cross9", "--dtype", "fp64", "--dist", "2")
_add("2d5_fp32_step2_stream", 2, "t2_star", "--dtype", "fp32", "--step", "2", "--streaming", "--xrim", "dpp")
_add("2d25_fp64_step2", 2, "t2_box25", "--dtype", "fp64", "--step", "2")
_add("2dodd_fp32_it5", 2, "t2_odd", "--dtype", "fp32")
_add("2dodd_fp64_step2_stream", 2, "t2_odd", "--dtype", "fp64", "--step", "2", "--streaming")

# --dist / --merge-forward as real knobs: an explicit --dist selects --schedule reuse (`Range` source planes in register
# windows, partial sums carried over the rest); one case per legal value (step-1)*order <= dist <= step*order
# (benchmarks/3d7pt_star/tuning.py:20), --merge-forward on both sides of the retained planes' in-plane tap counts
_add("3d7_fp32_s1_dist1", 3, "t3_star", "--dtype", "fp32", "--dist", "1", "--sn", "8")
_add("3d7_fp32_s2_dist1", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--dist", "1", "--sn", "16", "--prefetch")
_add("3d7_fp32_s2_dist2", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--dist", "2", "--sn", "16", "--prefetch")
_add("3d7_fp64_s2_dist1_mf0", 3, "t3_star", "--dtype", "fp64", "--step", "2", "--dist", "1", "--merge-forward", "0", "--sn", "9")
_add("3d7_fp32_s2_dist2_mf2_lds", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--dist", "2", "--merge-forward", "2", "--xrim", "lds", "--sn", "32")
_add("3d7_fp32_s3_dist2", 3, "t3_star", "--dtype", "fp32", "--step", "3", "--dist", "2", "--sn", "16")
_add("3d7_fp32_s3_dist3", 3, "t3_star", "--dtype", "fp32", "--step", "3", "--dist", "3", "--sn", "16")
_add("3dodd_fp64_s2_dist2", 3, "t3_odd", "--dtype", "fp64", "--step", "2", "--dist", "2")
_add("2d5_fp32_s2_stream_dist1", 2, "t2_star", "--dtype", "fp32", "--step", "2", "--dist", "1", "--streaming", "--prefetch")
_add("2d25_fp64_stream_dist2_mf0", 2, "t2_box25", "--dtype", "fp64", "--streaming", "--dist", "2", "--merge-forward", "0")
_add("3d7_fp32_window", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--schedule", "window", "--sn", "16", "--prefetch")
# memory-path knobs (round 2): exact vmcnt pipeline -- unconditional loads (re-read / closed window), buffer-masked stores, drains
_add("3d7_fp32_s2_ul1_buf", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--sn", "16", "--prefetch", "--prefetch-depth", "3", "--uniform-loads", "1", "--store-mask", "buffer", "--cc-opt", "-fno-slp-vectorize")
_add("3d7_fp64_ul2_buf_drain1", 3, "t3_star_odd", "--dtype", "fp64", "--sn", "5", "--prefetch", "--prefetch-depth", "2", "--uniform-loads", "2", "--store-mask", "buffer", "--drain", "1")
_add("3d7_fp32_t2_ul2_drain2", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4", "--sn", "16", "--prefetch", "--uniform-loads", "2", "--drain", "2")
_add("2d25_fp32_stream_buf", 2, "t2_box25", "--dtype", "fp32", "--streaming", "--prefetch", "--uniform-loads", "2", "--store-mask", "buffer")
_add("2d5_fp64_tile_buf", 2, "t2_star", "--dtype", "fp64", "--store-mask", "buffer")
# overlapped tiles in x (--exact-x 0), deferred stores
_add("3d7_fp32_s2_overlap_x", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--exact-x", "0", "--bx", "34", "--by", "7", "--block-merge-y", "2", "--sn", "16", "--prefetch")
_add("2d25_fp64_stream_overlap_x", 2, "t2_box25", "--dtype", "fp64", "--streaming", "--exact-x", "0", "--bx", "34")
_add("3d7_fp32_s2_defer_stores", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--sn", "16", "--bx", "32", "--by", "8", "--block-merge-y", "2", "--prefetch", "--prefetch-depth", "1", "--defer-stores", "1")
_add("3dodd_fp64_defer_stores_dma", 3, "t3_odd", "--dtype", "fp64", "--stage", "dma", "--defer-stores", "1", "--sn", "7")

# round 3: --order rows (the arriving plane consumed by source row, row groups fenced), --pack (v_pk_fma_f32 on pairs of x points, odd pairs
# assembled from DPP moves) and --pin (partial sums used where they are updated, so that the compiler cannot sink the FMA chains)
_add("3d7_fp32_s2_rows_packed", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--sn", "16", "--prefetch", "--order", "rows")
_add("3d7_fp32_s3_rows_packed_pd2", 3, "t3_star", "--dtype", "fp32", "--step", "3", "--sn", "16", "--prefetch", "--prefetch-depth", "2", "--order", "rows", "--bx", "32", "--by", "16", "--block-merge-y", "2")
_add("3d7_fp32_s3_rows_unpacked", 3, "t3_star", "--dtype", "fp32", "--step", "3", "--sn", "16", "--prefetch", "--order", "rows", "--pack", "0", "--bx", "32", "--by", "16", "--block-merge-y", "2")
_add("3d7_fp32_s3_rows_coef_sgpr", 3, "t3_star", "--dtype", "fp32", "--step", "3", "--sn", "16", "--prefetch", "--order", "rows", "--pack", "0", "--coef", "sgpr", "--bx", "32", "--by", "16", "--block-merge-y", "2")
_add("3d7_fp32_s2_taps_coef_vgpr", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--sn", "16", "--prefetch", "--coef", "vgpr")
_add("3d7_fp64_s2_rows", 3, "t3_star", "--dtype", "fp64", "--step", "2", "--sn", "16", "--order", "rows")
_add("3d7_fp32_s1_rows_oddN", 3, "t3_star_odd", "--dtype", "fp32", "--order", "rows")
_add("3d7_fp32_s2_rows_lds_nofence", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--sn", "9", "--xrim", "lds", "--order", "rows", "--row-fence", "-1")
_add("3d9x_fp32_rows", 3, "t3_cross", "--dtype", "fp32", "--dist", "2", "--schedule", "scatter", "--order", "rows")
_add("3dodd_fp32_s2_rows", 3, "t3_odd", "--dtype", "fp32", "--step", "2", "--order", "rows")
_add("3d_ahead_fp32_rows", 3, "t3_ahead", "--dtype", "fp32", "--sn", "8", "--order", "rows", "--prefetch")
_add("3d7_fp32_t2_rows", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4", "--sn", "16", "--prefetch", "--order", "rows")
_add("2d25_fp32_tile_rows", 2, "t2_box25", "--dtype", "fp32", "--order", "rows")
_add("2d25_fp64_stream_rows", 2, "t2_box25", "--dtype", "fp64", "--streaming", "--prefetch", "--order", "rows")
_add("2d9b_fp32_s2_rows", 2, "t2_box9", "--dtype", "fp32", "--step", "2", "--order", "rows")
_add("2d_ahead_fp32_stream_rows", 2, "t2_ahead", "--dtype", "fp32", "--streaming", "--sn", "32", "--order", "rows")
_add("3d7_fp32_s2_pinned_taps", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--sn", "16", "--prefetch", "--prefetch-depth", "3", "--pin", "1", "--cc-opt", "-fno-slp-vectorize")
_add("3d7_fp64_s3_pinned_taps", 3, "t3_star", "--dtype", "fp64", "--step", "3", "--sn", "16", "--pin", "1")

# round 3: wave specialisation (--stage dma --loader-waves n: n extra wavefronts request the planes by LDS-DMA into a ring of LDS slots and
# count their own vmcnt; the consumer lanes never load)
_add("3d7_fp32_s2_loader_waves2", 3, "smoke3", "--dtype", "fp32", "--step", "2", "--sn", "16", "--stage", "dma", "--loader-waves", "2", "--prefetch-depth", "3", "--pin", "1")
_add("3d7_fp64_s2_loader_waves3_depth2", 3, "t3_star", "--dtype", "fp64", "--step", "2", "--sn", "9", "--stage", "dma", "--loader-waves", "3", "--prefetch-depth", "2", "--bx", "32", "--by", "8", "--block-merge-y", "2")
_add("3dodd_fp32_loader_waves1_depth1_lds", 3, "t3_odd", "--dtype", "fp32", "--stage", "dma", "--loader-waves", "1", "--prefetch-depth", "1", "--xrim", "lds", "--sn", "7")
_add("3d7_fp32_s3_loader_waves2", 3, "smoke3", "--dtype", "fp32", "--step", "3", "--sn", "16", "--stage", "dma", "--loader-waves", "2", "--prefetch-depth", "2", "--pin", "1", "--bx", "32", "--by", "16", "--block-merge-y", "2")
_add("2d25_fp32_stream_loader_waves2", 2, "t2_box25", "--dtype", "fp32", "--streaming", "--stage", "dma", "--loader-waves", "2")

# --stage dma (round 2): planes staged by LDS-DMA (global_load_lds_dwordx4) into the per-lane-dense LDS image, tile-edge lanes re-reading
# from the halo regions; every schedule, both x-rim paths, box corners, wide halos (hx > points per lane), ragged grids
_add("3d7_fp64_dma", 3, "t3_star", "--dtype", "fp64", "--stage", "dma", "--sn", "8")
_add("3d7_fp64_dma_s2", 3, "t3_star", "--dtype", "fp64", "--step", "2", "--stage", "dma", "--sn", "16", "--bx", "32", "--by", "8", "--block-merge-y", "2")
_add("3dodd_fp32_dma_s2", 3, "t3_odd", "--dtype", "fp32", "--step", "2", "--stage", "dma", "--sn", "9")
_add("3d9x_fp32_dma_lds", 3, "t3_cross", "--dtype", "fp32", "--dist", "2", "--schedule", "scatter", "--stage", "dma", "--xrim", "lds")
_add("3dodd_fp32_dma_reuse_d1", 3, "t3_odd", "--dtype", "fp32", "--step", "2", "--dist", "1", "--stage", "dma")
_add("3d7_fp64_dma_reuse_d2_mf0", 3, "t3_star", "--dtype", "fp64", "--step", "2", "--dist", "2", "--merge-forward", "0", "--stage", "dma", "--sn", "5")
_add("3dodd_fp32_dma_window", 3, "t3_odd", "--dtype", "fp32", "--step", "2", "--schedule", "window", "--stage", "dma", "--xrim", "lds")
_add("3dodd_fp32_dma_bx128", 3, "t3_odd", "--dtype", "fp32", "--stage", "dma", "--bx", "128", "--by", "2", "--block-merge-y", "2")
_add("smoke_fp32_dma_s3", 3, "smoke3", "--dtype", "fp32", "--step", "3", "--stage", "dma", "--sn", "12")
_add("2d25_fp32_dma_stream", 2, "t2_box25", "--dtype", "fp32", "--streaming", "--stage", "dma")
_add("2d25_fp64_dma_stream_s2", 2, "t2_box25", "--dtype", "fp64", "--streaming", "--step", "2", "--stage", "dma", "--sn", "20")
_add("2d9s_fp32_dma_stream_d1", 2, "t2_star9", "--dtype", "fp32", "--streaming", "--dist", "1", "--stage", "dma", "--xrim", "lds")

# temporal blocking (on-chip multi-step): equal to the fused stencil up to rounding
_add("3d7_fp32_t2", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4", "--sn", "16", "--prefetch")
_add("3d7_fp64_t2_lds", 3, "t3_star", "--dtype", "fp64", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4", "--sn", "9", "--xrim", "lds")
_add("3d7_fp32_t3", 3, "t3_star", "--dtype", "fp32", "--step", "3", "--temporal", "1", "--bx", "34", "--by", "15", "--block-merge-y", "2", "--sn", "16", "--prefetch")
_add("3d7_fp32_t2_b66", 3, "t3_star", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--bx", "66", "--by", "15", "--block-merge-y", "2", "--sn", "32", "--prefetch", "--xcd-remap", "0")
_add("3d9x_fp64_t2", 3, "t3_cross", "--dtype", "fp64", "--dist", "2", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4")
_add("3dodd_fp32_t2_falls_back_to_fused", 3, "t3_odd", "--dtype", "fp32", "--step", "2", "--temporal", "1")
_add("2d5_fp32_t2_tile", 2, "t2_star", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4")
_add("2d5_fp64_t3_stream", 2, "t2_star", "--dtype", "fp64", "--step", "3", "--temporal", "1", "--streaming", "--sn", "24", "--prefetch")
_add("2d25_fp64_t2_tile", 2, "t2_box25", "--dtype", "fp64", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4")

# Temporal blocking and the tolerance.  On-chip time steps re-associate the fused sum; `--temporal 1` emits them only
# where the generator's drift estimate stays within the bar for the spec's iterations and the fused kernel otherwise, `--temporal force`
# emits them regardless.  t3_star_it100 asks for 100 iterations: the forced pipelines measure the drift (they are where 1e-6 breaks), the
# unforced one must come back as the fused kernel and stay bit-exact for all 100.
_TM2 = ["--3d", "--dtype", "fp32", "--step", "2", "--by", "8", "--block-merge-y", "2", "--sn", "16", "--prefetch"]
_TM3 = ["--3d", "--dtype", "fp32", "--step", "3", "--bx", "34", "--by", "15", "--block-merge-y", "2", "--sn", "16", "--prefetch"]
TEMPORAL_MARGIN = [
    ("t2_fp32_it100_forced", 3, stc("t3_star_it100"), _TM2 + ["--temporal", "force"]),
    ("t3_fp32_it100_forced", 3, stc("t3_star_it100"), _TM3 + ["--temporal", "force"]),
    ("t2_fp32_it100_fenced", 3, stc("t3_star_it100"), _TM2 + ["--temporal", "1"]),
    ("fused2_fp32_it100", 3, stc("t3_star_it100"), ["--3d", "--dtype", "fp32", "--step", "2", "--sn", "16"]),
]

# Stencil shapes whose temporal pipelines are BEYOND 1e-6 at their own (small) iteration counts: dense boxes at step 2-3, found by the CPU
# calibration (tests/calibrate_temporal_drift.py, seed 33: chained oracle sweeps, 1.0e-6 .. 1.6e-6).  Round 2's GPU fuzz reported 13 such
# shapes as "DRIFT" (profiles/r02_fuzz_shapes_seed3*.txt); their .stc files were scratch and the shape generator has changed since, so these
# are the same kind of shape regenerated and committed.  (id, ndim, stc, options without --temporal, unforced kernel builds?)
_D3 = ["--bx", "16", "--by", "8", "--block-merge-x", "2", "--block-merge-y", "1", "--sn", "8"]
_D2 = ["--by", "8", "--block-merge-y", "2"]
DRIFT = [
    ("c23_3d_o1_s2", 3, stc("drift_c23_3d_o1"), ["--3d", "--dtype", "fp32", "--step", "2"] + _D3, True),
    ("c27_3d_o1_s3", 3, stc("drift_c27_3d_o1"), ["--3d", "--dtype", "fp32", "--step", "3"] + _D3, True),
    ("c22_3d_o1_s3", 3, stc("drift_c22_3d_o1"), ["--3d", "--dtype", "fp32", "--step", "3"] + _D3, False),   # its fused 3-step kernel spills: forced pipeline only
    ("c34_3d_o1_s2", 3, stc("drift_c34_3d_o1"), ["--3d", "--dtype", "fp32", "--step", "2"] + _D3, False),
    ("c26_2d_o2_s3", 2, stc("drift_c26_2d_o2"), ["--dtype", "fp32", "--step", "3"] + _D2, True),
    ("c30_2d_o2_s3", 2, stc("drift_c30_2d_o2"), ["--dtype", "fp32", "--step", "3"] + _D2, True),
    ("c21_2d_o3_s2", 2, stc("drift_c21_2d_o3"), ["--dtype", "fp32", "--step", "2"] + _D2, True),
]


def drift_build_args():
    out = []
    for _, _, s, opts, unforced in DRIFT:
        out.append(opts + ["--temporal", "force", s])
        if unforced:
            out.append(opts + ["--temporal", "1", s])
    return out


SMOKE = ("smoke3", 3, stc("smoke3"), ["--3d", "--dtype", "fp32"])

# BASELINE.json configs at full size: (id, ndim, stc, options).  Default-option kernels first, then EVERY kernel bench.py can
# time (bench.kernels(): headline, step-1 and temporal kernels of every workload, the fp64 workloads included), so that no
# number bench.py prints comes from a kernel without a full-size parity case.
FULL = [
    ("C2_2d5pt_8192_fp32_tile", 2, os.path.join(CFG, "c2_2d5pt_star_8192.stc"), ["--dtype", "fp32"]),
    ("C2_2d5pt_8192_fp32_stream", 2, os.path.join(CFG, "c2_2d5pt_star_8192.stc"), ["--dtype", "fp32", "--streaming"]),
    ("C3_3d7pt_512_fp32", 3, os.path.join(CFG, "c3_3d7pt_star_512.stc"), ["--3d", "--dtype", "fp32"]),
    ("C3_3d7pt_512_fp32_step2", 3, os.path.join(CFG, "c3_3d7pt_star_512.stc"), ["--3d", "--dtype", "fp32", "--step", "2"]),
    ("C4_3d7pt_1024_fp32", 3, os.path.join(CFG, "c4_3d7pt_star_1024.stc"), ["--3d", "--dtype", "fp32"]),
    ("C5_2d25pt_16384_fp64", 2, os.path.join(CFG, "c5_2d25pt_box_16384.stc"), ["--dtype", "fp64"]),
]


def _bench_kernels():
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    seen = {(c[2], tuple(c[3])) for c in FULL}
    for kid, w, opts in bench.kernels():
        wl = bench.WORKLOADS[w]
        if (wl["stc"], tuple(opts)) not in seen:
            seen.add((wl["stc"], tuple(opts)))
            FULL.append((kid, wl["ndim"], wl["stc"], list(opts)))


_bench_kernels()
# the reference's legal --dist range on the headline geometry at full size (different kernels, identical results)
FULL.append(("C4_3d7pt_1024_fp32_fused2_dma", 3, os.path.join(CFG, "c4_3d7pt_star_1024.stc"),
             ["--3d", "--dtype", "fp32", "--step", "2", "--stage", "dma", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2"]))
FULL.append(("C2_2d5pt_8192_fp32_stream_dma", 2, os.path.join(CFG, "c2_2d5pt_star_8192.stc"), ["--dtype", "fp32", "--streaming", "--stage", "dma"]))
for _d in ("1", "2"):
    FULL.append(("C4_3d7pt_1024_fp32_fused2_dist%s" % _d, 3, os.path.join(CFG, "c4_3d7pt_star_1024.stc"),
                 ["--3d", "--dtype", "fp32", "--step", "2", "--dist", _d, "--prefetch", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2"]))

# BASELINE config C4 AS WRITTEN ("z-slab decomposition across 8 x MI355X"): the slab-view and pair-launch kernels that
# bench.py --gpus 2/4/8 launches, at 1024^3, every rank in turn on one GPU (tests/test_gpu_parity.py::test_c4_slab_views_at_full_size)
C4_SLAB_WORLDS = (2, 4, 8)


# BASELINE config C1: 2d5pt_star 4096^2 fp32, 100 iterations (the reference-CPU-path config): HIP vs oracle at full size
C1 = ("C1_2d5pt_4096_fp32_it100", 2, os.path.join(CFG, "c1_2d5pt_star_4096.stc"), ["--dtype", "fp32"])


def all_build_args():
    out = [c[3] + [c[2]] for c in SMALL] + [SMOKE[3] + [SMOKE[2]]] + [c[3] + [c[2]] for c in FULL] + [C1[3] + [C1[2]]] + [c[3] + [c[2]] for c in TEMPORAL_MARGIN] + drift_build_args()
    return out


def golden_args(case, meta, rows=False):
    """Options for running a golden fixture case through the HIP path (fp64, like the reference).  rows: round 3's emission (the
    plane consumed by source row, partial sums pinned) instead of the reuse schedule the fixture's --dist selects."""
    # 3D fused multi-step stencils (27 - 63 taps in fp64) keep 2 rows per lane: with 4 the partial sums spill to scratch and the
    # runtime refuses the kernel; the tile still covers the halo (by * my = 8 > 2 * Halo = 6 at step 3)
    my = "2" if meta["ndim"] == 3 and meta["step"] > 1 else "4"
    opts = (["--3d"] if meta["ndim"] == 3 else []) + ["--dtype", "fp64", "--step", str(meta["step"])] + \
        ["--dist", str(meta["macros"]["Dist"])] + (["--schedule", "scatter", "--order", "rows"] if rows else []) + ["--bx", "16", "--by", "4", "--block-merge-x", "2", "--block-merge-y", my, "--sn", "4"]
    return opts, stc("gold_" + case)


FUZZ_SAMPLE = (56, 5)   # (configurations, seed) of the sampled parity fuzz run by every GPU test session


def fuzz_sample_jobs():
    import fuzz_parity
    return fuzz_parity.make_jobs(*FUZZ_SAMPLE)


# slab decomposition on one GPU (tests/test_gpu_parity.py::test_slab_decomposition_on_one_gpu): (id, world, stencil, ndim, options)
SLAB_CASES = [
    ("w2_step1", 2, "t3_star", 3, ["--3d", "--dtype", "fp32", "--sn", "8"]),
    ("w3_fused2", 3, "t3_star", 3, ["--3d", "--dtype", "fp32", "--step", "2", "--sn", "16"]),
    ("w2_temporal2", 2, "t3_star", 3, ["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--by", "8", "--block-merge-y", "4", "--sn", "16", "--prefetch"]),
    ("w2_2d_box25_tile", 2, "t2_box25", 2, ["--dtype", "fp64"]),
    ("w3_2d_star_stream_step2", 3, "t2_star", 2, ["--dtype", "fp32", "--streaming", "--step", "2", "--sn", "16"]),
]
