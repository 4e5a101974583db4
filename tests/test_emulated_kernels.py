"""Emitted HIP kernels executed on the CPU (fiber emulation of workgroups, tests/emu) --
checks the generator's index math, guards, halo loaders, LDS exchange, register rotation,
DPP path and prefetch pipeline without a GPU.  The GPU run of the same kernels is in
test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

import oracle
from emu_util import build_emulated, run_emulated
from gpu_cases import golden_args
from helpers import golden_cases, load_golden, write_stc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rows", [False, True], ids=["reuse_schedule", "rows_order"])
@pytest.mark.parametrize("case", golden_cases())
def test_emulated_kernel_vs_reference_golden(case, rows, tmp_path):
    meta, a0, a_ref, b_ref = load_golden(case)
    opts, stc = golden_args(case, meta, rows=rows)
    lib = build_emulated(tmp_path, stc, opts)
    A = np.ascontiguousarray(a0.copy()); B = np.zeros_like(A)
    n = run_emulated(lib, A, B, meta["iterations"], meta["step"])
    assert n == meta["launches"]
    spec = oracle.Spec(stc, meta["ndim"], meta["step"])
    # fixtures are uncontracted fp64, the kernel is an FMA chain
    assert oracle.check(spec, A, a_ref)["max_rel"] < 1e-12
    assert oracle.check(spec, B, b_ref)["max_rel"] < 1e-12
    h = spec.halo
    ring = np.ones(A.shape, bool)
    ring[tuple(slice(h, s - h) for s in A.shape)] = False
    assert np.array_equal(A[ring], a_ref[ring]) and np.array_equal(B[ring], b_ref[ring])
    # and bit-exact against the oracle's contracted mode
    A2 = a0.copy(); B2 = np.zeros_like(A2)
    oracle.run(spec, A2, B2, contract=1)
    assert np.array_equal(A, A2) and np.array_equal(B, B2)


def _mg():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mg", os.path.join(ROOT, "oracle", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


VARIANTS = [
    ("3d_default_fp32", 3, "STAR3", (19, 23, 520), ["--3d", "--dtype", "fp32", "--sn", "8"]),
    ("3d_prefetch_dpp_eager", 3, "STAR3", (19, 23, 520), ["--3d", "--dtype", "fp32", "--sn", "7", "--prefetch", "--xrim", "dpp", "--lazy-rims", "0"]),
    ("3d_cyclicy_fp64", 3, "STAR3", (15, 29, 140), ["--3d", "--dtype", "fp64", "--sn", "5", "--cyclic-merge-y", "3", "--by", "2", "--bx", "32"]),
    ("3d_step2_dpp_prefetch", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--prefetch", "--xrim", "dpp"]),
    ("3d_oddN_scalar", 3, "STAR3", (12, 17, 263), ["--3d", "--dtype", "fp32", "--sn", "8"]),
    ("3d_bx128_dpp", 3, "STAR3", (12, 19, 1030), ["--3d", "--dtype", "fp32", "--bx", "128", "--by", "2", "--block-merge-y", "2", "--xrim", "dpp"]),
    ("3d_cross_step2", 3, "CROSS3", (14, 19, 136), ["--3d", "--dtype", "fp64", "--dist", "2", "--step", "2", "--xrim", "dpp"]),
    ("2d_tile_fp32", 2, "STAR2", (1, 75, 530), ["--dtype", "fp32"]),
    ("2d_stream_step3_dpp", 2, "STAR2", (1, 75, 530), ["--dtype", "fp64", "--streaming", "--prefetch", "--xrim", "dpp", "--sn", "16", "--step", "3"]),
    ("2d25_tile_fp64", 2, "BOX25", (1, 61, 268), ["--dtype", "fp64"]),
    ("2d25_stream_step2", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--step", "2", "--prefetch"]),
    # round 3: --order rows (plane consumed by source row), packed pairs (float2 halves of the accumulator vectors), pinned sums
    ("3d_s2_rows_packed", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--prefetch", "--order", "rows"]),
    ("3d_s3_rows_packed_pd2", 3, "STAR3", (19, 23, 260), ["--3d", "--dtype", "fp32", "--sn", "8", "--step", "3", "--prefetch", "--prefetch-depth", "2", "--order", "rows", "--bx", "32", "--by", "4", "--block-merge-y", "2"]),
    ("3d_s1_rows_oddN_scalar", 3, "STAR3", (12, 17, 263), ["--3d", "--dtype", "fp32", "--sn", "8", "--order", "rows"]),
    ("3d_s2_rows_fp64", 3, "STAR3", (15, 19, 140), ["--3d", "--dtype", "fp64", "--sn", "5", "--step", "2", "--order", "rows"]),
    ("3d_cross_rows", 3, "CROSS3", (14, 19, 136), ["--3d", "--dtype", "fp32", "--dist", "2", "--schedule", "scatter", "--order", "rows"]),
    ("2d25_tile_rows", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--order", "rows"]),
    ("2d25_stream_rows_s2", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--step", "2", "--prefetch", "--order", "rows"]),
    ("3d_s2_rows_lds_rim", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--xrim", "lds", "--order", "rows"]),
    # round 3: coefficients in scalar / vector registers instead of FMA literals (--coef; the values are the same, DRS_SREG is empty here)
    ("3d_s3_rows_coef_sgpr", 3, "STAR3", (19, 23, 260), ["--3d", "--dtype", "fp32", "--sn", "8", "--step", "3", "--prefetch", "--order", "rows", "--pack", "0", "--coef", "sgpr", "--bx", "32", "--by", "4", "--block-merge-y", "2"]),
    ("3d_s2_taps_coef_vgpr", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--prefetch", "--coef", "vgpr"]),
    ("2d25_tile_rows_packed_coef_sgpr", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--order", "rows", "--coef", "sgpr"]),
    ("3d_s2_rows_two_points_per_lane", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--order", "rows", "--bx", "16", "--block-merge-x", "2", "--by", "4", "--block-merge-y", "3"]),
    ("3d_s2_taps_pinned", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--prefetch", "--pin", "1"]),
    # round 3: wave specialisation -- loader wavefronts request planes by LDS-DMA into a ring, consumers never load (the fibers' barrier counts
    # arrivals since this round: loaders and consumers run different code between barriers)
    ("3d_s2_loader_waves2_depth3", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--stage", "dma", "--loader-waves", "2", "--prefetch-depth", "3"]),
    ("3d_s1_loader_waves1_depth1", 3, "STAR3", (12, 17, 264), ["--3d", "--dtype", "fp32", "--sn", "8", "--stage", "dma", "--loader-waves", "1", "--prefetch-depth", "1"]),
    ("3d_s2_fp64_loader_waves3", 3, "STAR3", (15, 19, 140), ["--3d", "--dtype", "fp64", "--sn", "5", "--step", "2", "--stage", "dma", "--loader-waves", "3", "--prefetch-depth", "2", "--bx", "32", "--by", "8", "--block-merge-y", "2"]),
    ("2d25_stream_loader_waves2", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--stage", "dma", "--loader-waves", "2"]),
    ("3d_cross_loader_waves2_lds", 3, "CROSS3", (14, 19, 136), ["--3d", "--dtype", "fp32", "--dist", "2", "--schedule", "scatter", "--stage", "dma", "--loader-waves", "2", "--xrim", "lds"]),
    ("2d_refdefaults", 2, "BOX9", (1, 41, 70), ["--dtype", "fp64", "--ref-defaults"]),
    ("3d_step2_prefetch_depth2", 3, "STAR3", (23, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--prefetch", "--prefetch-depth", "2"]),
    ("3d_step1_prefetch_depth3", 3, "STAR3", (19, 23, 270), ["--3d", "--dtype", "fp64", "--sn", "5", "--prefetch", "--prefetch-depth", "3", "--xrim", "lds"]),
    ("2d_stream_prefetch_depth2", 2, "STAR2", (1, 75, 530), ["--dtype", "fp32", "--streaming", "--prefetch", "--prefetch-depth", "2", "--sn", "3"]),
    # memory-pipeline variants (round 2): guarded loads / stores (round-1 form), re-read tail loads, window loads that close past the
    # block's last plane, explicit drains; ragged grids so that out-of-grid lanes, partial vectors and short blocks all occur
    ("3d_r1_memory_path", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--prefetch", "--uniform-loads", "0", "--store-mask", "branch"]),
    ("3d_window_loads_depth2", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--prefetch", "--prefetch-depth", "2", "--uniform-loads", "2", "--drain", "1"]),
    ("3d_window_loads_oddN", 3, "STAR3", (12, 17, 263), ["--3d", "--dtype", "fp64", "--sn", "3", "--prefetch", "--prefetch-depth", "3", "--uniform-loads", "2", "--drain", "2"]),
    ("3d_window_loads_short_blocks", 3, "STAR3", (9, 11, 140), ["--3d", "--dtype", "fp32", "--sn", "1", "--prefetch", "--prefetch-depth", "3", "--uniform-loads", "2"]),
    ("2d_stream_window_loads", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--prefetch", "--uniform-loads", "2", "--store-mask", "branch"]),
    ("2d_tile_branch_stores", 2, "BOX9", (1, 41, 70), ["--dtype", "fp64", "--store-mask", "branch"]),
    ("3d_buffer_stores_uniform_loads", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--prefetch", "--uniform-loads", "1", "--store-mask", "buffer"]),
    ("2d_tile_buffer_stores", 2, "BOX9", (1, 41, 70), ["--dtype", "fp64", "--store-mask", "buffer"]),
    # --exact-x 0: overlapped tiles in x (the tile's outermost lanes load the halo columns with the row and own nothing)
    ("3d_overlap_x_bx34", 3, "STAR3", (19, 23, 300), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--prefetch", "--exact-x", "0", "--bx", "34", "--by", "7", "--block-merge-y", "2"]),
    ("3d_overlap_x_fp64_lds", 3, "STAR3", (12, 17, 263), ["--3d", "--dtype", "fp64", "--sn", "3", "--exact-x", "0", "--xrim", "lds", "--bx", "18", "--by", "8"]),
    ("3d_overlap_x_reuse", 3, "STAR3", (19, 23, 300), ["--3d", "--dtype", "fp32", "--sn", "5", "--step", "2", "--dist", "1", "--exact-x", "0", "--bx", "34", "--by", "7", "--block-merge-y", "2"]),
    ("2d_overlap_x_box25_stream", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--prefetch", "--exact-x", "0", "--bx", "34"]),
    ("2d_overlap_x_tile", 2, "BOX9", (1, 41, 140), ["--dtype", "fp64", "--exact-x", "0", "--bx", "18", "--by", "8"]),
    # --defer-stores: a completed plane leaves one iteration later (blocks of 1, 2 and many planes; every staging / store path)
    ("3d_defer_stores", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--prefetch", "--defer-stores", "1"]),
    ("3d_defer_stores_sn1", 3, "STAR3", (9, 11, 140), ["--3d", "--dtype", "fp64", "--sn", "1", "--prefetch", "--prefetch-depth", "2", "--defer-stores", "1"]),
    ("3d_defer_stores_sn2_noprefetch_buf", 3, "STAR3", (12, 17, 263), ["--3d", "--dtype", "fp32", "--sn", "2", "--defer-stores", "1", "--store-mask", "buffer"]),
    ("3d_defer_stores_reuse_dma", 3, "STAR3", (19, 23, 264), ["--3d", "--dtype", "fp32", "--sn", "5", "--step", "2", "--dist", "1", "--stage", "dma", "--defer-stores", "1"]),
    ("3d_defer_stores_window", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "2", "--schedule", "window", "--prefetch", "--defer-stores", "1", "--uniform-loads", "2"]),
    ("2d_stream_defer_stores", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--prefetch", "--defer-stores", "1"]),
    # round 4: --xcd-remap 4 -- units of 32 consecutive tiles of one stream block dealt round-robin to the XCDs (ragged: 3 x 5 tiles, 3 blocks)
    ("3d_xcd_units_s2", 3, "STAR3", (23, 37, 300), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--prefetch", "--xcd-remap", "4", "--bx", "32", "--by", "4", "--block-merge-y", "2"]),
    ("3d_xcd_units_many_tiles", 3, "STAR3", (12, 70, 530), ["--3d", "--dtype", "fp64", "--sn", "5", "--xcd-remap", "4", "--bx", "16", "--by", "2", "--block-merge-y", "2"]),
    ("2d_stream_xcd_units", 2, "BOX25", (1, 61, 1068), ["--dtype", "fp32", "--streaming", "--sn", "9", "--xcd-remap", "4", "--bx", "16"]),
    ("2d_tile_xcd_chunks", 2, "BOX25", (1, 61, 1068), ["--dtype", "fp32", "--xcd-remap", "5", "--xcd-chunk", "3", "--bx", "16", "--by", "4", "--block-merge-y", "2"]),
    ("3d_xcd_chunks_s2", 3, "STAR3", (23, 37, 300), ["--3d", "--dtype", "fp64", "--sn", "7", "--step", "2", "--xcd-remap", "5", "--bx", "16", "--by", "4", "--block-merge-y", "2"]),
    # round 4: --cyclic-merge-x is the reference's strided layout (codegen.hpp:116-141, `mi += blockDim.x`): a lane's points are Bx columns
    # apart, element-wide accesses, the x rim by DPP from the neighbouring lane's point of the same index
    ("3d_cyclicx_fp32", 3, "STAR3", (12, 17, 263), ["--3d", "--dtype", "fp32", "--sn", "5", "--cyclic-merge-x", "4", "--bx", "32", "--by", "4", "--block-merge-y", "2"]),
    ("3d_cyclicx_s2_prefetch_fp64", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--prefetch", "--cyclic-merge-x", "2", "--bx", "64", "--by", "4", "--block-merge-y", "2"]),
    ("3d_cyclicx_cyclicy_lds", 3, "STAR3", (15, 29, 140), ["--3d", "--dtype", "fp32", "--sn", "5", "--cyclic-merge-x", "3", "--cyclic-merge-y", "3", "--by", "2", "--bx", "16", "--xrim", "lds"]),
    ("3d_cyclicx_reuse_dist1", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--dist", "1", "--cyclic-merge-x", "2", "--bx", "32", "--by", "4"]),
    ("3d_cyclicx_window_buffer_stores", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "2", "--schedule", "window", "--cyclic-merge-x", "2", "--bx", "32", "--by", "8", "--store-mask", "buffer"]),
    ("3d_cross_cyclicx", 3, "CROSS3", (14, 19, 136), ["--3d", "--dtype", "fp64", "--dist", "2", "--step", "2", "--cyclic-merge-x", "2", "--bx", "32", "--by", "8", "--block-merge-y", "2"]),
    ("2d25_tile_cyclicx", 2, "BOX25", (1, 61, 268), ["--dtype", "fp64", "--cyclic-merge-x", "4", "--bx", "32", "--by", "4", "--block-merge-y", "2"]),
    ("2d25_stream_cyclicx_s2", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--step", "2", "--prefetch", "--cyclic-merge-x", "2", "--bx", "64"]),
    ("2d_refdefaults_cyclicx", 2, "BOX9", (1, 41, 70), ["--dtype", "fp64", "--ref-defaults", "--cyclic-merge-x", "2"]),
]

# --stage dma: planes staged by LDS-DMA into the per-lane-dense LDS image (own region [row][vector][lane], halo pieces by loader task);
# tile-edge lanes re-read from the halo regions.  Ragged grids: edge tiles, out-of-grid lanes, short blocks; box stencils: corners.
DMA = [
    ("3d_dma_s1", 3, "STAR3", (19, 23, 268), ["--3d", "--dtype", "fp32", "--sn", "7", "--stage", "dma"]),
    ("3d_dma_s2_dpp", 3, "STAR3", (19, 23, 264), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--stage", "dma", "--bx", "32", "--by", "4", "--block-merge-y", "2"]),
    ("3d_dma_s2_lds_fp64", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--stage", "dma", "--xrim", "lds", "--bx", "16", "--by", "8", "--block-merge-x", "2", "--block-merge-y", "2"]),
    ("3d_dma_bx128_wave_edges", 3, "STAR3", (12, 19, 1032), ["--3d", "--dtype", "fp32", "--bx", "128", "--by", "2", "--block-merge-y", "2", "--stage", "dma", "--sn", "5"]),
    ("3d_dma_cross_s2", 3, "CROSS3", (14, 19, 136), ["--3d", "--dtype", "fp64", "--dist", "2", "--step", "2", "--stage", "dma", "--schedule", "scatter"]),
    ("3d_dma_reuse_dist1", 3, "STAR3", (19, 23, 264), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--dist", "1", "--stage", "dma"]),
    ("3d_dma_reuse_dist2_mf0", 3, "STAR3", (19, 23, 264), ["--3d", "--dtype", "fp32", "--sn", "4", "--step", "2", "--dist", "2", "--merge-forward", "0", "--stage", "dma", "--xrim", "lds"]),
    ("3d_dma_window", 3, "STAR3", (19, 23, 264), ["--3d", "--dtype", "fp64", "--sn", "3", "--step", "2", "--schedule", "window", "--stage", "dma"]),
    ("3d_dma_short_blocks", 3, "STAR3", (9, 11, 140), ["--3d", "--dtype", "fp32", "--sn", "1", "--stage", "dma", "--bx", "16", "--by", "4"]),
    ("3d_dma_my4", 3, "STAR3", (13, 37, 264), ["--3d", "--dtype", "fp32", "--sn", "5", "--step", "2", "--stage", "dma", "--bx", "32", "--by", "2", "--block-merge-y", "4"]),
    ("2d_stream_dma_box25", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--sn", "9", "--stage", "dma"]),
    ("2d_stream_dma_box25_s2_fp64", 2, "BOX25", (1, 61, 268), ["--dtype", "fp64", "--streaming", "--sn", "9", "--step", "2", "--stage", "dma", "--xrim", "lds"]),
    ("2d_stream_dma_star_dist1", 2, "STAR2", (1, 75, 532), ["--dtype", "fp32", "--streaming", "--step", "2", "--dist", "1", "--sn", "16", "--stage", "dma"]),
]

# --dist / --merge-forward as real knobs: an explicit --dist selects --schedule reuse -- `Range` source planes in
# register windows, partial sums carried over the rest -- over the reference's legal range (step-1)*order <= dist <= step*order
# (benchmarks/3d7pt_star/tuning.py:20), --merge-forward deciding whether a retained plane's neighbours are carried or re-read.
# Every one of them must equal the oracle bit for bit, like every other schedule.
REUSE = [
    ("3d_s1_dist1", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--dist", "1"]),
    ("3d_s1_dist1_prefetch_lds", 3, "STAR3", (12, 17, 263), ["--3d", "--dtype", "fp64", "--sn", "3", "--dist", "1", "--prefetch", "--xrim", "lds"]),
    ("3d_s2_dist1", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--dist", "1", "--prefetch"]),
    ("3d_s2_dist2", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--dist", "2", "--prefetch"]),
    ("3d_s2_dist1_mf0", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp64", "--sn", "5", "--step", "2", "--dist", "1", "--merge-forward", "0"]),
    ("3d_s2_dist2_mf2_lds", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "6", "--step", "2", "--dist", "2", "--merge-forward", "2", "--xrim", "lds", "--prefetch", "--prefetch-depth", "2"]),
    ("3d_s2_dist2_mf100", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp32", "--sn", "1", "--step", "2", "--dist", "2", "--merge-forward", "100"]),
    ("3d_s3_dist2", 3, "STAR3", (23, 21, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "3", "--dist", "2", "--bx", "16", "--by", "4", "--block-merge-x", "2", "--block-merge-y", "2"]),
    ("3d_s3_dist3", 3, "STAR3", (23, 21, 140), ["--3d", "--dtype", "fp64", "--sn", "9", "--step", "3", "--dist", "3", "--bx", "16", "--by", "4", "--block-merge-x", "2", "--block-merge-y", "2", "--prefetch"]),
    ("3d_cross_s1_dist2", 3, "CROSS3", (14, 19, 136), ["--3d", "--dtype", "fp32", "--dist", "2", "--sn", "4"]),
    ("3d_cross_s2_dist4_cyclicy", 3, "CROSS3", (17, 29, 140), ["--3d", "--dtype", "fp64", "--step", "2", "--dist", "4", "--sn", "5", "--cyclic-merge-y", "2", "--by", "4", "--bx", "32"]),
    ("2d_stream_s2_dist1", 2, "STAR2", (1, 75, 530), ["--dtype", "fp32", "--streaming", "--step", "2", "--dist", "1", "--sn", "16", "--prefetch"]),
    ("2d_stream_box25_dist2_mf0", 2, "BOX25", (1, 61, 268), ["--dtype", "fp64", "--streaming", "--dist", "2", "--merge-forward", "0", "--sn", "9"]),
    ("2d_stream_box25_s2_dist3", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--streaming", "--step", "2", "--dist", "3", "--sn", "9", "--prefetch"]),
    ("2d_tile_dist_is_a_noop", 2, "BOX25", (1, 61, 268), ["--dtype", "fp32", "--dist", "2"]),
    ("3d_window_schedule", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp32", "--sn", "7", "--step", "2", "--schedule", "window", "--prefetch"]),
    ("3d_window_schedule_eager", 3, "STAR3", (19, 23, 262), ["--3d", "--dtype", "fp64", "--sn", "3", "--schedule", "window", "--lazy-rims", "0"]),
]


@pytest.mark.parametrize("vid,ndim,pts,dims,opts", VARIANTS + REUSE + DMA, ids=[v[0] for v in VARIANTS + REUSE + DMA])
def test_emulated_variants_bit_exact(vid, ndim, pts, dims, opts, tmp_path):
    mg = _mg()
    stc = str(tmp_path / "v.stc")
    write_stc(stc, ndim, dims, 4, getattr(mg, pts))
    step = int(opts[opts.index("--step") + 1]) if "--step" in opts else 1
    lib = build_emulated(tmp_path, stc, opts)
    spec = oracle.Spec(stc, ndim, step)
    dt = np.float32 if "fp32" in opts else np.float64
    A = oracle.fill_random(spec.shape, dt); B = np.zeros_like(A)
    A2, B2 = A.copy(), B.copy()
    oracle.run(spec, A2, B2, contract=1)
    n = run_emulated(lib, A, B, spec.iterations, step)
    assert n == spec.launches
    assert np.array_equal(A, A2) and np.array_equal(B, B2)
    # the emitted gold kernel too
    A3 = oracle.fill_random(spec.shape, dt); B3 = np.zeros_like(A3)
    run_emulated(lib, A3, B3, spec.iterations, step, gold=True)
    assert np.array_equal(A3, A2) and np.array_equal(B3, B2)


EDGE = [
    ("3d_min_interior", 3, "STAR3", (3, 3, 3), ["--3d", "--dtype", "fp64"]),
    ("3d_thin", 3, "STAR3", (5, 4, 7), ["--3d", "--dtype", "fp32", "--sn", "1"]),
    ("3d_one_tile_row", 3, "STAR3", (6, 3, 260), ["--3d", "--dtype", "fp32", "--sn", "3", "--prefetch"]),
    ("3d_step2_min", 3, "STAR3", (5, 5, 5), ["--3d", "--dtype", "fp64", "--step", "2"]),
    ("3d_temporal2_min", 3, "STAR3", (5, 5, 6), ["--3d", "--dtype", "fp64", "--step", "2", "--temporal", "1", "--by", "4", "--block-merge-y", "2"]),
    ("3d_temporal3_ragged", 3, "STAR3", (13, 17, 70), ["--3d", "--dtype", "fp64", "--step", "3", "--temporal", "1", "--bx", "18", "--by", "8", "--block-merge-y", "2", "--sn", "2", "--prefetch"]),
    ("2d_min_interior", 2, "STAR2", (1, 3, 3), ["--dtype", "fp64"]),
    ("2d_box25_min", 2, "BOX25", (1, 5, 5), ["--dtype", "fp32"]),
    ("2d_stream_min", 2, "STAR2", (1, 3, 9), ["--dtype", "fp32", "--streaming", "--sn", "1"]),
    ("2d_temporal2_tile_small", 2, "BOX9", (1, 7, 9), ["--dtype", "fp64", "--step", "2", "--temporal", "1"]),
    ("2d_oddN_box25_stream", 2, "BOX25", (1, 23, 31), ["--dtype", "fp64", "--streaming", "--sn", "4", "--xrim", "lds"]),
    ("3d_temporal2_prefetch_depth2", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--prefetch", "--prefetch-depth", "2", "--by", "8", "--block-merge-y", "2"]),
    ("3d_cyclicx_temporal2_fwd", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--cyclic-merge-x", "4", "--bx", "32", "--by", "6", "--block-merge-y", "2"]),
    ("3d_temporal3_skew_rows_fwd", 3, "STAR3", (15, 19, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "3", "--temporal", "1", "--skew", "1", "--prefetch", "--order", "rows", "--exact-y", "1", "--bx", "34", "--by", "8", "--block-merge-y", "2"]),
    ("3d_temporal2_skew_taps_pd2_fwd", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--skew", "1", "--prefetch", "--prefetch-depth", "2", "--by", "8", "--block-merge-y", "2"]),
    ("3d_temporal3_skew_ragged_fwd", 3, "STAR3", (13, 17, 70), ["--3d", "--dtype", "fp64", "--step", "3", "--temporal", "1", "--skew", "1", "--bx", "18", "--by", "8", "--block-merge-y", "2", "--sn", "2", "--prefetch"]),
    ("2d_stream_temporal4_skew_fwd", 2, "STAR2", (1, 40, 270), ["--dtype", "fp64", "--streaming", "--sn", "7", "--step", "4", "--temporal", "1", "--skew", "1", "--prefetch"]),
    ("3d_temporal3_skew2_taps_fwd", 3, "STAR3", (15, 19, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "3", "--temporal", "1", "--skew", "2", "--prefetch", "--pin", "1", "--exact-y", "1", "--bx", "34", "--by", "8", "--block-merge-y", "2"]),
    ("3d_temporal3_skew2_sn1_fwd", 3, "STAR3", (13, 17, 70), ["--3d", "--dtype", "fp64", "--step", "3", "--temporal", "1", "--skew", "2", "--bx", "18", "--by", "8", "--block-merge-y", "2", "--sn", "1", "--prefetch"]),
    ("3d_temporal4_auto_pd2_fwd", 3, "STAR3", (21, 30, 150), ["--3d", "--dtype", "fp64", "--sn", "8", "--step", "4", "--temporal", "1", "--prefetch", "--prefetch-depth", "2", "--bx", "20", "--by", "11", "--block-merge-y", "2"]),
    ("2d_stream_temporal4_skew2_fwd", 2, "STAR2", (1, 40, 270), ["--dtype", "fp64", "--streaming", "--sn", "7", "--step", "4", "--temporal", "1", "--skew", "2", "--prefetch", "--order", "rows"]),
    ("3d_temporal2_window_loads", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--prefetch", "--prefetch-depth", "2", "--by", "8", "--block-merge-y", "2", "--uniform-loads", "2", "--drain", "1"]),
]


@pytest.mark.parametrize("vid,ndim,pts,dims,opts", EDGE, ids=[v[0] for v in EDGE])
def test_emulated_edge_geometries(vid, ndim, pts, dims, opts, tmp_path):
    """Grids barely larger than the halo, grids smaller than a tile, single-plane stream blocks."""
    mg = _mg()
    stc = str(tmp_path / "e.stc")
    write_stc(stc, ndim, dims, 4, getattr(mg, pts))
    step = int(opts[opts.index("--step") + 1]) if "--step" in opts else 1
    lib = build_emulated(tmp_path, stc, opts)
    spec = oracle.Spec(stc, ndim, step)
    dt = np.float32 if "fp32" in opts else np.float64
    A = oracle.fill_random(spec.shape, dt); B = np.zeros_like(A)
    A2, B2 = A.copy(), B.copy()
    oracle.run(spec, A2, B2, contract=1)
    n = run_emulated(lib, A, B, spec.iterations, step)
    assert n == spec.launches
    if "--temporal" in opts:
        assert oracle.check(spec, A, A2)["max_rel"] < 1e-12 and oracle.check(spec, B, B2)["max_rel"] < 1e-12
        h = spec.halo
        ring = np.ones(A.shape, bool)
        ring[tuple(slice(h, s - h) for s in A.shape)] = False
        assert np.array_equal(A[ring], A2[ring]) and np.array_equal(B[ring], B2[ring])
    else:
        assert np.array_equal(A, A2) and np.array_equal(B, B2)


RACE = [
    ("3d_temporal2_prefetch", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--prefetch", "--by", "8", "--block-merge-y", "2"]),
    ("3d_cyclicx_temporal2", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--cyclic-merge-x", "4", "--bx", "32", "--by", "6", "--block-merge-y", "2"]),
    # round 4: --skew 1 -- stage t consumes what stage t-1 completed one iteration earlier; two barriers per iteration for all stages
    ("3d_temporal3_skew_rows", 3, "STAR3", (15, 19, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "3", "--temporal", "1", "--skew", "1", "--prefetch", "--order", "rows", "--exact-y", "1", "--bx", "34", "--by", "8", "--block-merge-y", "2"]),
    ("3d_temporal2_skew_taps_pd2", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--skew", "1", "--prefetch", "--prefetch-depth", "2", "--xrim", "lds", "--by", "8", "--block-merge-y", "2"]),
    ("3d_temporal3_skew_sn1", 3, "STAR3", (13, 17, 70), ["--3d", "--dtype", "fp64", "--step", "3", "--temporal", "1", "--skew", "1", "--bx", "18", "--by", "8", "--block-merge-y", "2", "--sn", "1", "--prefetch", "--pin", "1"]),
    ("2d_stream_temporal4_skew", 2, "STAR2", (1, 40, 270), ["--dtype", "fp64", "--streaming", "--sn", "7", "--step", "4", "--temporal", "1", "--skew", "1", "--prefetch", "--order", "rows"]),
    ("2d25_stream_temporal2_skew", 2, "BOX25", (1, 61, 268), ["--dtype", "fp64", "--streaming", "--sn", "9", "--step", "2", "--temporal", "1", "--skew", "1", "--prefetch", "--xrim", "lds"]),
    ("3d_temporal3_skew2_taps", 3, "STAR3", (15, 19, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "3", "--temporal", "1", "--skew", "2", "--prefetch", "--pin", "1", "--exact-y", "1", "--bx", "34", "--by", "8", "--block-merge-y", "2"]),
    ("3d_temporal3_skew2_rows", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "3", "--temporal", "1", "--skew", "2", "--prefetch", "--order", "rows", "--bx", "18", "--by", "8", "--block-merge-y", "2"]),
    ("3d_temporal2_skew2_pd2_lds", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--skew", "2", "--prefetch", "--prefetch-depth", "2", "--xrim", "lds", "--by", "8", "--block-merge-y", "2"]),
    # the shipped 4-stage geometry in small: 11-row workgroups, depth-2 prefetch, skew / pin / exact-y left to the generator (on for >= 3 stages)
    ("3d_temporal4_auto_pd2", 3, "STAR3", (21, 30, 150), ["--3d", "--dtype", "fp64", "--sn", "8", "--step", "4", "--temporal", "1", "--prefetch", "--prefetch-depth", "2", "--bx", "20", "--by", "11", "--block-merge-y", "2"]),
    ("3d_temporal4_default_geometry", 3, "STAR3", (9, 40, 290), ["--3d", "--dtype", "fp64", "--sn", "5", "--step", "4", "--temporal", "1"]),
    ("2d_stream_temporal4_skew2", 2, "STAR2", (1, 40, 270), ["--dtype", "fp64", "--streaming", "--sn", "7", "--step", "4", "--temporal", "1", "--skew", "2", "--prefetch"]),
    ("2d25_stream_temporal2_skew2", 2, "BOX25", (1, 61, 268), ["--dtype", "fp64", "--streaming", "--sn", "9", "--step", "2", "--temporal", "1", "--skew", "2", "--prefetch", "--xrim", "lds"]),
    ("3d_temporal3_lds", 3, "STAR3", (15, 19, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "3", "--temporal", "1", "--xrim", "lds", "--bx", "34", "--by", "8", "--block-merge-y", "2"]),
    ("3d_step1_window_lazy", 3, "STAR3", (15, 19, 300), ["--3d", "--dtype", "fp64", "--sn", "5", "--schedule", "window", "--lazy-rims", "1", "--xrim", "lds"]),
    ("3d_step1_scatter_prefetch", 3, "STAR3", (15, 19, 300), ["--3d", "--dtype", "fp64", "--sn", "5", "--prefetch", "--xrim", "lds"]),
    ("2d_stream_temporal4", 2, "STAR2", (1, 40, 270), ["--dtype", "fp64", "--streaming", "--sn", "7", "--step", "4", "--temporal", "1", "--xrim", "lds"]),
    ("2d_tile_temporal3", 2, "BOX9", (1, 40, 140), ["--dtype", "fp64", "--step", "3", "--temporal", "1", "--xrim", "lds"]),
    ("3d_step2_prefetch_depth2", 3, "STAR3", (23, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "7", "--step", "2", "--prefetch", "--prefetch-depth", "2", "--xrim", "lds"]),
    ("3d_temporal2_prefetch_depth2", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--temporal", "1", "--prefetch", "--prefetch-depth", "2", "--xrim", "lds", "--by", "8", "--block-merge-y", "2"]),
    # LDS-DMA staging: plane n+1 is written into the slot of plane n-1 (or older, with late re-reads) while plane n is read
    ("3d_dma_scatter", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--stage", "dma", "--xrim", "lds"]),
    ("3d_dma_reuse_late_rereads", 3, "STAR3", (17, 21, 300), ["--3d", "--dtype", "fp64", "--sn", "6", "--step", "2", "--dist", "1", "--merge-forward", "100", "--stage", "dma", "--xrim", "lds"]),
    ("3d_dma_window_lazy", 3, "STAR3", (15, 19, 300), ["--3d", "--dtype", "fp64", "--sn", "5", "--schedule", "window", "--lazy-rims", "1", "--stage", "dma"]),
    ("2d_stream_dma", 2, "BOX25", (1, 61, 268), ["--dtype", "fp64", "--streaming", "--sn", "9", "--step", "2", "--stage", "dma"]),
]


@pytest.mark.parametrize("vid,ndim,pts,dims,opts", RACE, ids=[v[0] for v in RACE])
def test_emulated_reverse_fiber_order(vid, ndim, pts, dims, opts, tmp_path, monkeypatch):
    """LDS hand-off hazards: run the same kernels with the fibers of every barrier phase executed
    last-to-first (tests/emu EMU_ORDER=reverse); together with the forward order this exposes any
    writer/reader pair that lacks a barrier."""
    monkeypatch.setenv("EMU_ORDER", "reverse")
    mg = _mg()
    stc = str(tmp_path / "r.stc")
    write_stc(stc, ndim, dims, 4, getattr(mg, pts))
    step = int(opts[opts.index("--step") + 1]) if "--step" in opts else 1
    lib = build_emulated(tmp_path, stc, opts)
    spec = oracle.Spec(stc, ndim, step)
    A = oracle.fill_random(spec.shape, np.float64); B = np.zeros_like(A)
    A2, B2 = A.copy(), B.copy()
    oracle.run(spec, A2, B2, contract=1)
    run_emulated(lib, A, B, spec.iterations, step)
    assert oracle.check(spec, A, A2)["max_rel"] < 1e-12 and oracle.check(spec, B, B2)["max_rel"] < 1e-12


ASYM = [
    # forward-only taps (zl = 0, no -x/-y halo): PADL = 0, no rows above
    ("3d_forward_only", 3, [(0, 0, 0, 0.4), (1, 0, 0, 0.1), (2, 0, 0, 0.05), (0, 1, 0, 0.2), (0, 0, 1, 0.15), (0, 0, 2, 0.1)], (14, 13, 140), ["--3d", "--dtype", "fp64", "--sn", "5", "--dist", "1"]),
    # different reach per dimension (hx = 2, hy = 1, hz = 2), not symmetric in x
    ("3d_mixed_reach", 3, [(0, 0, 0, 0.3), (-2, 0, 0, 0.1), (2, 0, 0, 0.1), (1, 0, 0, 0.05), (0, -1, 0, 0.2), (0, 1, 0, 0.1), (0, 0, -2, 0.05), (0, 0, 1, 0.1)], (15, 12, 150), ["--3d", "--dtype", "fp32", "--sn", "4", "--prefetch", "--dist", "2"]),
    ("3d_mixed_reach_temporal", 3, [(0, 0, 0, 0.3), (-1, 0, 0, 0.1), (1, 0, 0, 0.1), (0, -1, 0, 0.2), (0, 1, 0, 0.1), (0, 0, -1, 0.05), (0, 0, 1, 0.15)], (15, 17, 150), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "2", "--temporal", "1", "--by", "4", "--block-merge-y", "2"]),
    ("2d_upwind_rows", 2, [(0, 0, 0.5), (1, 0, 0.2), (2, 0, 0.1), (0, 1, 0.1), (0, 2, 0.1)], (1, 33, 70), ["--dtype", "fp64", "--dist", "1"]),
    # every tap AHEAD along the streamed dimension (nothing on the output's own plane or row): the partial sum starts on a later
    # plane than the first one it is carried over (tests/fuzz_shapes.py seed 31 found the scatter schedule starting it nowhere)
    ("2d_taps_ahead_stream_scatter", 2, [(1, 0, 0.14), (1, 2, 0.13), (2, 0, 0.21), (1, 1, 0.2), (1, -2, 0.2)], (1, 41, 90), ["--dtype", "fp64", "--streaming", "--sn", "6"]),
    ("2d_taps_ahead_stream_step2", 2, [(1, 0, 0.14), (1, 2, 0.13), (2, 0, 0.21), (1, 1, 0.2), (1, -2, 0.2)], (1, 41, 90), ["--dtype", "fp32", "--streaming", "--sn", "6", "--step", "2", "--prefetch"]),
    ("2d_taps_ahead_stream_temporal2", 2, [(1, 0, 0.15), (1, 2, 0.1), (2, 0, 0.2), (1, 1, 0.2), (1, -2, 0.2)], (1, 41, 90), ["--dtype", "fp64", "--streaming", "--sn", "7", "--step", "2", "--temporal", "1"]),
    ("2d_taps_ahead_tile", 2, [(1, 0, 0.14), (1, 2, 0.13), (2, 0, 0.21), (1, 1, 0.2), (1, -2, 0.2)], (1, 41, 90), ["--dtype", "fp64", "--step", "2"]),
    ("3d_taps_ahead_scatter", 3, [(1, 0, 0, 0.3), (2, 0, 0, 0.1), (1, 1, 0, 0.2), (1, 0, -1, 0.15), (2, -1, 1, 0.1)], (15, 14, 140), ["--3d", "--dtype", "fp32", "--sn", "4", "--schedule", "scatter", "--prefetch"]),
    ("3d_taps_ahead_step2", 3, [(1, 0, 0, 0.3), (2, 0, 0, 0.1), (1, 1, 0, 0.2), (1, 0, -1, 0.15), (2, -1, 1, 0.1)], (17, 16, 140), ["--3d", "--dtype", "fp64", "--sn", "5", "--step", "2"]),
    ("3d_taps_ahead_reuse_dma", 3, [(1, 0, 0, 0.3), (2, 0, 0, 0.1), (1, 1, 0, 0.2), (1, 0, -1, 0.15), (2, -1, 1, 0.1)], (15, 14, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--dist", "1", "--stage", "dma"]),
    ("3d_taps_ahead_temporal2", 3, [(1, 0, 0, 0.3), (2, 0, 0, 0.1), (1, 1, 0, 0.2), (1, 0, -1, 0.15), (2, -1, 1, 0.1)], (17, 20, 140), ["--3d", "--dtype", "fp64", "--sn", "4", "--step", "2", "--temporal", "1", "--by", "4", "--block-merge-y", "2"]),
    ("2d_upwind_rows_stream", 2, [(0, 0, 0.5), (1, 0, 0.2), (2, 0, 0.1), (0, 1, 0.1), (0, 2, 0.1)], (1, 33, 70), ["--dtype", "fp32", "--streaming", "--sn", "6", "--dist", "1", "--xrim", "lds"]),
]


@pytest.mark.parametrize("vid,ndim,pts,dims,opts", ASYM, ids=[v[0] for v in ASYM])
def test_emulated_asymmetric_stencils(vid, ndim, pts, dims, opts, tmp_path):
    """Stencils that are not symmetric: one-sided taps, different reach per dimension."""
    stc = str(tmp_path / "a.stc")
    write_stc(stc, ndim, dims, 4, pts)
    step = int(opts[opts.index("--step") + 1]) if "--step" in opts else 1
    lib = build_emulated(tmp_path, stc, opts)
    spec = oracle.Spec(stc, ndim, step)
    dt = np.float32 if "fp32" in opts else np.float64
    A = oracle.fill_random(spec.shape, dt); B = np.zeros_like(A)
    A2, B2 = A.copy(), B.copy()
    oracle.run(spec, A2, B2, contract=1)
    assert run_emulated(lib, A, B, spec.iterations, step) == spec.launches
    if "--temporal" in opts:
        assert oracle.check(spec, A, A2)["max_rel"] < 1e-12 and oracle.check(spec, B, B2)["max_rel"] < 1e-12
    else:
        assert np.array_equal(A, A2) and np.array_equal(B, B2)


def _emulated_fuzz_jobs(n=14, seed=9):
    """A fixed random sample of the tuner's space on tiny ragged grids (the emulator runs one fiber per lane: grids stay
    small and workgroups below 512 lanes)."""
    import random
    from drstencil_amd.tuner import tuning as t
    mg = _mg()
    rnd = random.Random(seed)
    jobs = []
    for ndim, pts, dims, order in [(3, "STAR3", (13, 21, 300), 1), (2, "STAR2", (1, 37, 300), 1), (2, "BOX25", (1, 29, 280), 2)]:
        for dtype in ("fp32", "fp64"):
            t.order, t.ndim, t.elem_bytes = order, ndim, 4 if dtype == "fp32" else 8
            # fuzz_emulated.py with FUZZ_STEPS="2,3,4" FUZZ_SPACE_R4=1: deep pipelines and the round-4 space (the pytest sample keeps steps 1-2)
            steps_ = tuple(int(x) for x in os.environ.get("FUZZ_STEPS", "1,2").split(","))
            if order > 1:
                steps_ = tuple(x for x in steps_ if x <= 2) or (2,)
            lanes_ = 768 if os.environ.get("FUZZ_SPACE_R4") else 256
            space = [v for v in t.enumerate_space(steps_, round4=bool(os.environ.get("FUZZ_SPACE_R4"))) if v[2][0] * v[2][1] <= lanes_ and v[2][0] <= 68 and v[3] <= 16]
            for v in rnd.sample(space, min(len(space), max(1, n // 6))):
                cl = t.cfgToCommandLine(v).split()
                if "--prefetch-depth" in cl:
                    cl[cl.index("--prefetch-depth") + 1] = str(rnd.choice([1, 2, 3]))
                # round-2 knobs: --merge-forward on both sides of the retained planes' tap counts (reuse schedule), memory path
                if "--schedule" not in cl and rnd.random() < 0.6:
                    cl[cl.index("--merge-forward") + 1] = str(rnd.choice([0, 2, 3, 100]))
                if rnd.random() < 0.3:
                    cl += ["--uniform-loads", str(rnd.choice([1, 2]))]
                if rnd.random() < 0.3:
                    cl += ["--store-mask", "buffer"]
                if rnd.random() < 0.2:
                    cl += ["--drain", str(rnd.choice([1, 2]))]
                if rnd.random() < 0.35 and "--temporal" not in cl and "--cyclic-merge-y" not in cl and (ndim == 3 or "--streaming" in cl):
                    cl += ["--stage", "dma"]
                if rnd.random() < 0.3:
                    cl += ["--defer-stores", "1"]          # LDS-DMA staging (16-byte vectors: the generator rejects the others)
                import fuzz_parity
                fuzz_parity.round3_knobs(rnd, cl)          # round 3: rows order / packed pairs / pinned sums / rotation modulus / loader wavefronts
                fuzz_parity.round4_knobs(rnd, cl)          # round 4: skewed pipelines, XCD unit / chunk maps, strided x merge
                if "--skew" in cl and ndim == 2 and "--streaming" not in cl:
                    del cl[cl.index("--skew"):cl.index("--skew") + 2]
                jobs.append((t.cfgToString(v) + "_" + dtype + "_%dd" % ndim + pts.lower(), ndim, pts, dims, (["--3d"] if ndim == 3 else []) + ["--dtype", dtype] + cl, v[0]))
    return jobs


@pytest.mark.parametrize("vid,ndim,pts,dims,opts,step", _emulated_fuzz_jobs(), ids=[j[0] for j in _emulated_fuzz_jobs()])
def test_emulated_sampled_fuzz(vid, ndim, pts, dims, opts, step, tmp_path):
    """Random tuner-space configurations through the CPU emulator against the oracle: the emitter's logic without the
    GPU compiler in the loop (the GPU-side sweeps are tests/fuzz_parity.py, tests/fuzz_gold.py)."""
    mg = _mg()
    stc = str(tmp_path / "f.stc")
    write_stc(stc, ndim, dims, 4, getattr(mg, pts))
    try:
        lib = build_emulated(tmp_path, stc, opts)
    except AssertionError as e:
        assert "Invalid configuration" in str(e) or "tile" in str(e) or "halo" in str(e), str(e)[-300:]
        pytest.skip("rejected by the generator")
    spec = oracle.Spec(stc, ndim, step)
    dt = np.float32 if "fp32" in opts else np.float64
    A = oracle.fill_random(spec.shape, dt); B = np.zeros_like(A)
    A2, B2 = A.copy(), B.copy()
    oracle.run(spec, A2, B2, contract=1)
    assert run_emulated(lib, A, B, spec.iterations, step) == spec.launches
    if "--temporal" in opts and json.loads(lib.drs_plugin_info().decode()).get("stages", 1) > 1:
        bar = 1e-6 if dt == np.float32 else 1e-12
        assert oracle.check(spec, A, A2)["max_rel"] < bar and oracle.check(spec, B, B2)["max_rel"] < bar
    else:
        assert np.array_equal(A, A2) and np.array_equal(B, B2)


def test_emulated_pair_launch(tmp_path):
    """--pair-launch 1: dr2_<name> over two (in, out) pairs in one launch == two launches of dr_<name>."""
    import ctypes
    mg = _mg()
    stc = str(tmp_path / "p.stc")
    write_stc(stc, 3, (9, 21, 140), 4, mg.STAR3)
    lib = build_emulated(tmp_path, stc, ["--3d", "--dtype", "fp32", "--step", "2", "--sn", "4", "--pair-launch", "1", "--bx", "16", "--by", "4", "--block-merge-y", "2"])
    lib.drs_plugin_launch_pair.argtypes = [ctypes.c_void_p] * 5
    spec = oracle.Spec(stc, 3, 2)
    a0 = oracle.fill_random(spec.shape, np.float32)
    a1 = (a0[::-1] * np.float32(0.5)).copy()
    o0, o1, r0, r1 = (np.zeros_like(a0) for _ in range(4))
    assert lib.drs_plugin_launch_pair(a0.ctypes.data, o0.ctypes.data, a1.ctypes.data, o1.ctypes.data, None) == 0
    assert lib.drs_plugin_launch(a0.ctypes.data, r0.ctypes.data, None) == 0 and lib.drs_plugin_launch(a1.ctypes.data, r1.ctypes.data, None) == 0
    assert np.array_equal(o0, r0) and np.array_equal(o1, r1) and o0.any() and not np.array_equal(o0, o1)


def _random_shape_jobs(n=int(os.environ.get("EMU_SHAPES", "24")), seed=int(os.environ.get("EMU_SHAPE_SEED", "5"))):
    """Random point sets from tests/fuzz_shapes.py (sparse to dense, one-sided, no centre, duplicate offsets, mixed signs)
    with a rotating set of schedules; tiny ragged grids."""
    import random
    import fuzz_shapes as fs
    rnd = random.Random(seed)
    schedules = [["--schedule", "scatter"], ["--dist", "{d}"], ["--dist", "{d}", "--merge-forward", "100"], ["--schedule", "window"],
                 ["--dist", "{d}", "--merge-forward", "0", "--prefetch"], ["--step", "2", "--dist", "{d2}"], ["--step", "2", "--schedule", "scatter", "--prefetch"],
                 ["--step", "2", "--temporal", "1"], ["--step", "2", "--temporal", "1", "--prefetch", "--xrim", "lds"]]
    jobs = []
    for s in range(n):
        ndim = 3 if s % 2 else 2
        h = rnd.choice([1, 2])
        pts, mixed = fs.random_shape(rnd, ndim, h)
        dims = (rnd.randint(6 + 4 * h, 14), rnd.randint(9 + 4 * h, 24), rnd.randint(40, 150)) if ndim == 3 else (1, rnd.randint(20, 50), rnd.randint(40, 280))
        d1, d2 = fs.legal_dists(pts, 1), fs.legal_dists(pts, 2)
        if not d1:
            continue               # nothing behind anything along the streamed dimension: "No data to reuse" for every --dist
        sched = [x.format(d=rnd.choice(d1), d2=rnd.choice(d2)) for x in schedules[s % len(schedules)]]
        if "--temporal" in sched and mixed:
            sched = sched[:2]      # a relative bar means nothing where the sum cancels: the fused kernel instead
        if "--dist" not in sched:
            sched += ["--dist", str(rnd.choice(d2 if "--step" in sched else d1))]
        dtype = "fp32" if rnd.random() < 0.5 else "fp64"
        opts = (["--3d"] if ndim == 3 else ["--streaming"] if rnd.random() < 0.6 else []) + ["--dtype", dtype, "--sn", str(rnd.choice([3, 5, 8])), "--bx", "16", "--by", "4", "--block-merge-y", "2"] + sched
        jobs.append(("shape%d_%dd_o%d_%s" % (s, ndim, h, "_".join(x.strip("-") for x in sched)), ndim, pts, dims, opts))
    return jobs


@pytest.mark.parametrize("vid,ndim,pts,dims,opts", _random_shape_jobs(), ids=[j[0] for j in _random_shape_jobs()])
def test_emulated_random_shapes(vid, ndim, pts, dims, opts, tmp_path):
    """Stencil shapes nobody drew by hand, through every schedule: the emitter's per-plane bookkeeping (which planes stay
    resident, which taps are carried, where the rims come from) must not depend on the shape being a star, box or cross.
    The GPU-side sweep over shapes is tests/fuzz_shapes.py."""
    stc = str(tmp_path / "s.stc")
    write_stc(stc, ndim, dims, 4, pts)
    step = int(opts[opts.index("--step") + 1]) if "--step" in opts else 1
    try:
        lib = build_emulated(tmp_path, stc, opts)
    except AssertionError as e:
        assert "No data to reuse" in str(e) or "Invalid configuration" in str(e), str(e)[-300:]
        pytest.skip("rejected by the generator like the reference would: " + str(e).strip().splitlines()[-1][:120])
    spec = oracle.Spec(stc, ndim, step)
    dt = np.float32 if "fp32" in opts else np.float64
    A = oracle.fill_random(spec.shape, dt); B = np.zeros_like(A)
    A2, B2 = A.copy(), B.copy()
    oracle.run(spec, A2, B2, contract=1)
    assert run_emulated(lib, A, B, spec.iterations, step) == spec.launches
    if "--temporal" in opts and json.loads(lib.drs_plugin_info().decode()).get("stages", 1) > 1:
        bar = 1e-6 if dt == np.float32 else 1e-12
        assert oracle.check(spec, A, A2)["max_rel"] < bar and oracle.check(spec, B, B2)["max_rel"] < bar
        ring = np.ones(A.shape, bool)
        ring[tuple(slice(spec.halo, n - spec.halo) for n in A.shape)] = False
        assert np.array_equal(A[ring], A2[ring]) and np.array_equal(B[ring], B2[ring])
    else:
        assert np.array_equal(A, A2) and np.array_equal(B, B2)
