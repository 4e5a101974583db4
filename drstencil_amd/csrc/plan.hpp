// plan.hpp -- generator options (the reference CLI surface, main.cpp:12-56, plus
// additive MI355X knobs) and the kernel plan derived from them.
//
// How the reference's tuning knobs map onto CDNA4 (all are pure performance knobs,
// results never depend on them -- SURVEY.md section 4):
//   --bx/--by           workgroup shape in lanes (x) and rows (y); bx*by threads,
//                       best as a multiple of 64 (one wavefront = 64 lanes of x)
//   --block-merge-x n   n CONTIGUOUS points per lane in x -> one 16-byte
//                       global_load_dwordx4 per lane for n=4 fp32 / n=2 fp64
//   --cyclic-merge-x n  same tile width (n*bx); on a 64-wide wavefront strided
//                       points would turn every row into n dword loads, so the
//                       points are laid out contiguously as well
//   --block-merge-y n   n adjacent rows per lane (y taps between them stay in VGPRs)
//   --cyclic-merge-y n  n rows per lane, `by` apart
//   --sn                planes (3D) / rows (2D --streaming) streamed per workgroup
//   --stream-unroll     lower bound for the unroll of the streaming loop (the loop is
//                       unrolled by a multiple of Range so register rotation is pure
//                       renaming)
//   --prefetch          software prefetch of the next plane into VGPRs across the barrier
//   --dist/--merge-forward  accepted and validated like the reference (they select the
//                       forward/backward partition reported as Range/Dist); the CDNA4
//                       "scatter" schedule is that data-reuse idea with every partial sum
//                       carried in registers, so they do not change the emitted arithmetic
//   --step n            fused stencil (exact reference arithmetic) or, with --temporal 1,
//                       n on-chip applications of the one-step stencil (temporal blocking)
#pragma once
#include <string>
#include <vector>
#include "stencil_ir.hpp"

namespace drs {

struct GenOptions {
    // reference options (defaults: main.cpp:12-56)
    std::string out_name = "out.cu";
    bool is3d = false;
    int step = 1;
    int dist = 0;
    bool streaming = false;
    int bx = 16, by = 16, sn = 16;
    int stream_unroll = 4;
    int bmx = 1, bmy = 1, cmx = 1, cmy = 1;
    bool prefetch = false;
    int merge_forward = 5;
    bool gold = false;   // parsed, never used (as in the reference)
    bool check = false;
    // which geometry options the user actually gave (others take MI355X defaults)
    bool bx_set = false, by_set = false, sn_set = false, mx_set = false, my_set = false;
    // additive MI355X options
    std::string dtype = "fp64";  // fp32 | fp64
    bool out_set = false;
    int lazy_rims = 1;           // read LDS rims when first needed (1) or on arrival (0)
    int xcd_remap = -1;          // workgroup -> tile mapping: 0 dispatch order, 1 contiguous chunk per XCD, 2 x-y band per XCD, -1 auto
    std::string xrim = "dpp";    // lds | dpp  (x halo inside a wavefront via DPP wave shifts)
    int nt_store = 1;            // non-temporal stores of the output (+8 % measured on MI355X)
    int nt_load = 0;             // non-temporal loads of the input
    int waves_per_eu = 0;        // __launch_bounds__ second argument (0 = unset)
    int lds_pad = 0;             // extra dwords of padding per LDS row
    int ref_defaults = 0;        // 1: keep the reference's 16x16x16 defaults instead of MI355X ones
    int exact_y = -1;            // 1: halo loaders also fetch the source plane's y halo rows (every tile row owned);
                                 // 0: overlapped tiles; -1 auto: 1 for single-stage kernels, 0 for temporal pipelines
                                 // (measured: +3 % at step 1, -12 % on the 2-stage pipeline whose lanes own only 2 rows)
    int debug_drop_barrier = 0;  // TIMING EXPERIMENTS ONLY: 1 drops the barriers of stages >= 1, 2 also the per-plane barrier of single-stage kernels (results are wrong)
    int clamp_loads = 1;         // branch-free loads (out-of-grid lanes read the plane origin) and uniform-guarded scalar stores
    int halo_spread = 0;         // spread the halo loader tasks over all wavefronts of the workgroup (no gain measured)
    int xedge_select = 0;        // --xrim dpp: wavefront-edge lanes take the LDS value by select instead of a branch
    int pair_launch = 0;         // also emit dr2_<name>(in0, out0, in1, out1): one launch over two (in, out) pairs, chosen by blockIdx.y
                                 // (the two boundary views of a slab-decomposed run)
    int zgroup = 4;              // --xcd-remap 3: stream blocks of one tile taken by consecutive workgroups
    int prefetch_auto = 1;       // 3D kernels with step > 1 (fused or temporal) prefetch unless --prefetch-auto 0 (+28 % measured)
    int prefetch_depth = -1;     // planes in flight ahead of the one being summed (register sets = depth + 1); -1 auto:
                                 // 3 (fp32) / 2 (fp64) for fused multi-step 3D kernels (their wide halo leaves one resident workgroup per CU and
                                 // too few bytes in flight: 1.66 -> 1.56 ms on a slow-memory device, +1 % on a fast one), else 1
    int temporal = 0;            // step > 1: apply the one-step stencil `step` times on chip instead of the fused stencil
    std::string schedule = "scatter";  // scatter: partial sums carried in VGPRs; window: rotating register windows
};

struct Tap {
    int ds, dy, dx;     // offsets in (streamed, tile-row, tile-col) roles
    std::string coef;   // as printed: 6 significant digits
};

struct KernelPlan {
    std::string name;        // kernel base name (symbols dr_<name>, gold_<name>)
    int ndim = 3;
    bool has_s = true;       // a streamed dimension exists
    bool has_y = true;       // the tile has a row dimension
    bool fp32 = false;
    int L = 1, M = 1, N = 1; // grid as in the .stc
    int DS = 1, DY = 1, DX = 1;          // dims in role order
    long stride_s = 0, stride_y = 0;     // element strides (x is contiguous)
    int iterations = 0, step = 1, halo = 0, dist = 0, range = 0;
    std::vector<Tap> taps;   // taps the kernel applies per stage, gold order (lexicographic k, j, i)
    std::vector<Tap> gtaps;  // taps of the fused stencil (gold kernel, reference semantics)
    int stages = 1;          // on-chip time steps per launch (temporal blocking); 1 = apply `taps` once
    int oym = 0, oyp = 0;    // rows at the tile's y edges that are not owned (halo of all stages)
    bool exact_y = true;     // y halo rows of the source plane come from the halo loaders
    int AL = 0;              // columns at each x edge of the lane tile that are not owned (stages > 1)
    int OX = 0;              // columns owned per tile
    int zl = 0, zh = 0, hym = 0, hyp = 0, hxm = 0, hxp = 0;
    // geometry
    int BX = 64, BY = 4, VX = 4, RY = 1, SN = 16;
    bool cyclic_y = false;
    int VL = 4;              // elements per vector memory access
    int NV = 1;              // vectors per row per lane
    int TX = 256, TY = 4;    // lanes' footprint (tile without the exchanged x halo)
    int PADL = 0, PADR = 0;  // x halo columns loaded by the halo loaders
    int OY = 1;              // rows owned (stored) per tile
    int NBX = 1, NBY = 1, NBS = 1;
    int NT = 256;
    int SROW = 0, SROWS = 0, NSLOT = 2;  // LDS row length, rows per plane, plane slots
    int UN = 1;              // unroll of the streaming loop
    bool prefetch = false;
    int PD = 1;              // prefetch depth (planes in flight)
    std::string error;       // non-empty: invalid configuration
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace drs
