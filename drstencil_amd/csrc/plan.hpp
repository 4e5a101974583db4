// plan.hpp -- generator options (the reference CLI surface, main.cpp:12-56, plus
// additive MI355X knobs) and the kernel plan derived from them.
//
// How the reference's tuning knobs map onto CDNA4 (all are pure performance knobs,
// results never depend on them -- SURVEY.md section 4):
//   --bx/--by           workgroup shape in lanes (x) and rows (y); bx*by threads,
//                       best as a multiple of 64 (one wavefront = 64 lanes of x)
//   --block-merge-x n   n CONTIGUOUS points per lane in x -> one 16-byte
//                       global_load_dwordx4 per lane for n=4 fp32 / n=2 fp64
//   --cyclic-merge-x n  the reference's cyclic merge (codegen.hpp:116-141: `mi += blockDim.x`): n points
//                       per lane, bx columns apart -- every row becomes n element-wide accesses per
//                       lane (a wavefront instruction = 64 consecutive elements); measured slower
//                       than block merging on MI355X, kept as the reference's knob
//   --block-merge-y n   n adjacent rows per lane (y taps between them stay in VGPRs)
//   --cyclic-merge-y n  n rows per lane, `by` apart
//   --sn                planes (3D) / rows (2D --streaming) streamed per workgroup
//   --stream-unroll     lower bound for the unroll of the streaming loop (the loop is
//                       unrolled by a multiple of Range so register rotation is pure
//                       renaming)
//   --prefetch          software prefetch of the next plane into VGPRs across the barrier
//   --dist d            the reference's data-reuse distance (drstencil.hpp:198-259): it fixes `Range`, the number of source
//                       planes the reference keeps in shared memory, the rest of the reuse going through partial sums in
//                       `out` (atomicAdd).  Here an explicit --dist selects --schedule reuse: `Range` planes stay resident in
//                       register windows and partial sums are carried in VGPRs over the other R - Range planes
//                       (emit_hip.hpp: carry()); without --dist everything is carried (scatter).  Different legal --dist
//                       values give different kernels and identical results.
//   --merge-forward t   reference: forward sets smaller than t are folded back into the backward set.  Here, per retained
//                       plane: fewer in-plane taps than t -> its neighbours are re-read from LDS when due ("folded into the
//                       window read"), else read on arrival and carried in registers.
//   --step n            fused stencil (exact reference arithmetic) or, with --temporal 1,
//                       n on-chip applications of the one-step stencil (temporal blocking)
// MI355X-only knobs (round 2; each measured, see DESIGN.md section 3 -- the defaults are what measured fastest):
//   --schedule          scatter | reuse | window: how reuse along the streamed dimension is split between resident planes and carried sums
//   --stage reg|dma     arriving planes staged through VGPRs (software prefetch) or by LDS-DMA
//   --uniform-loads, --store-mask, --drain, --defer-stores   the memory pipeline of the streaming loop (s_waitcnt vmcnt bookkeeping)
//   --exact-x/--exact-y halo columns / rows from loader lanes (1) or overlapped tiles (0)
//   --cc-opt            per-kernel compiler flags (the bench kernels use -fno-slp-vectorize)
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
    int exact_x = 1;             // 1: the x halo columns of the source plane come from the halo loader lanes (16-byte pieces, one per row and side);
                                 // 0: overlapped tiles in x -- the tile's outermost lanes load the halo columns as part of the row (one longer
                                 // contiguous run per row, e.g. 34 lanes x 4 = 136 columns own 128) and store nothing
    int debug_skip = 0;          // TIMING EXPERIMENTS ONLY (results are wrong; such kernels load only with DRS_EXPERIMENTS=1), a bit mask of what the
                                 // kernel leaves out so that the parts of a plane iteration can be timed separately: 1 the x-halo loader tasks,
                                 // 2 the y-halo loader tasks, 4 every tap that is not on the lane's own point (5 of 25 FMAs left for the fused
                                 // 7-point star, no rim reads), 8 every store except the block's last plane (the sums become dead code: what
                                 // remains is loads + LDS staging + barrier)
    int debug_drop_barrier = 0;  // TIMING EXPERIMENTS ONLY: 1 drops the barriers of stages >= 1, 2 also the per-plane barrier of single-stage kernels (results are wrong)
    int clamp_loads = 1;         // branch-free loads (out-of-grid lanes read the plane origin) and uniform-guarded scalar stores
    int halo_spread = 0;         // spread the halo loader tasks over all wavefronts of the workgroup (no gain measured)
    int xedge_select = 0;        // --xrim dpp: wavefront-edge lanes take the LDS value by select instead of a branch
    int pair_launch = 0;         // also emit dr2_<name>(in0, out0, in1, out1): one launch over two (in, out) pairs, chosen by blockIdx.y
                                 // (the two boundary views of a slab-decomposed run)
    int zigzag = 0;              // --zigzag 1: the launch whose output array lies below its input array (every second launch of a ping-pong)
                                 // takes its stream blocks in REVERSE order, so that it starts on the planes the previous launch wrote last
                                 // (still in the memory-side cache).  Which workgroup sweeps which block never changes a result
    int xcd_chunk = 4;           // --xcd-remap 5: consecutive tiles an XCD takes per round
    int zgroup = 4;              // --xcd-remap 3: stream blocks of one tile taken by consecutive workgroups
    int prefetch_auto = 1;       // 3D kernels with step > 1 (fused or temporal) prefetch unless --prefetch-auto 0 (+28 % measured)
    int prefetch_depth = -1;     // planes in flight ahead of the one being summed (register sets = depth + 1); -1 auto:
                                 // 3 (fp32) / 2 (fp64) for fused multi-step 3D kernels (their wide halo leaves one resident workgroup per CU and
                                 // too few bytes in flight: 1.66 -> 1.56 ms on a slow-memory device, +1 % on a fast one), else 1
    int temporal = 0;            // step > 1: apply the one-step stencil `step` times on chip instead of the fused stencil (--temporal 1);
                                 // on-chip stages re-associate the fused sum, so the generator only emits them where its a-priori drift
                                 // estimate (planner.hpp: temporal_drift) stays within the dtype's bar (1e-6 relative fp32, 1e-12 fp64) for the
                                 // spec's `iterations`, and emits the fused kernel (exact reference arithmetic) otherwise; 2 = --temporal force:
                                 // emit the pipeline regardless (measurements, experiments; the plugin says so in drs_plugin_info)
    // Memory-path knobs of the streaming loop.  Defaults = the round-1 form, which the round-2 measurements could not beat
    // (profiles/r02_exp_r2a_vmcnt_pipeline.log, r02_exp_r2b_memory_path.log; DESIGN.md section 3 "vmcnt"): with guarded loads and
    // stores the compiler cannot count the vector-memory operations in flight and drains them (s_waitcnt vmcnt(0)) every plane;
    // an exactly counted pipeline (uniform loads + buffer-masked stores) keeps 3-4 planes in flight and is 1-10 % SLOWER, because
    // buffer stores cost 6 % (50 % where a quarter of the lanes is masked) and the per-plane drain itself is worth 3 %.
    int defer_stores = 0;        // 1: a completed output plane is held in registers (points-per-lane VGPRs) and stored one iteration later, right
                                 // after the next plane's loads have been issued: its write latency runs under that iteration's exchange and
                                 // FMAs instead of sitting in front of the per-plane drain (single-stage streaming kernels)
    int drain = 0;               // 1: s_waitcnt vmcnt(0) before the prefetch loads of every plane are issued (a workgroup's reads and writes
                                 // never overlap); 2: at the top of the iteration, before the staged plane is written to LDS; 0: none
    int uniform_loads = 0;       // --prefetch: 0: plane loads under `if (the block still needs planes)`; 1: issued unconditionally (past the
                                 // block's last plane they re-read it); 2: unconditionally through a buffer window that closes past the last
                                 // plane (the loads then move nothing).  1 and 2 leave no vector-memory instruction of the loop under a
                                 // branch, so the compiler's s_waitcnt vmcnt(N) counts are exact
    std::string stage = "reg";   // how an arriving plane gets into LDS: reg = global loads into VGPRs (software prefetch), then ds_write;
                                 // dma = LDS-DMA (global_load_lds_dwordx4): the plane lands in its LDS slot without touching VGPRs, one plane
                                 // ahead of the one being summed (look-ahead costs an LDS slot instead of prefetch register sets)
    int loader_waves = 0;        // --stage dma --loader-waves n (round 3, wave specialisation): n extra wavefronts per workgroup do nothing but request
                                 // planes by LDS-DMA into a ring of prefetch-depth + 1 LDS slots and count their own vmcnt; the bx*by "consumer"
                                 // lanes never issue a load: barrier, read the landed plane from LDS, sum, store.  Loads and stores sit in
                                 // different wavefronts' queues, no consumer drains loads, no prefetch registers (a stencil-free copy of the
                                 // headline's tile shape: 1.436 -> 1.376 ms with 8-plane blocks, profiles/r03_exp_wave_specialised_copy.log)
    std::vector<std::string> cc_opts;   // --cc-opt <flag> (repeatable): extra hipcc flags for this kernel (e.g. -fno-slp-vectorize); part of the
                                 // kernel's identity: printed in the banner's build line and applied by drs_kernel_build
    std::string store_mask = "branch";  // branch: plain global stores under per-lane guards (default: measured faster);
                                 // buffer: outputs leave through buffer_store with the lane's offset set out of range where the
                                 // lane must not store (dropped by the hardware's range check, no exec mask, no branch)
    std::string schedule = "scatter";  // scatter: every partial sum carried in VGPRs; window: rotating register windows, nothing carried;
                                 // reuse: `Range` planes in register windows + the rest carried (the reference's split for --dist)
    bool schedule_set = false;   // --schedule given (else an explicit --dist selects "reuse")
    // Round 3: the emitter bounds the live ranges itself instead of leaving a straight-line plane body to the compiler's pre-RA scheduler
    // (named state 136 words -> 256 VGPRs + scratch for the fused 63-point stencil).
    bool order_set = false;      // --order given; else the generator picks rows for fused 3D stencils beyond 25 taps and 2D tile kernels beyond 9 (measured)
    std::string order = "taps";  // scatter schedule, emission order of a plane's FMAs.  taps: every partial sum's chain in one piece, the whole rim
                                 // window read up front (rounds 1-2).  rows: by SOURCE ROW -- the arriving plane is consumed one row at a time
                                 // (the row's own-column vector from registers / one LDS read, its x neighbours by DPP), each row's tap groups
                                 // fenced from the next with __builtin_amdgcn_sched_barrier, the next row's LDS read issued one group ahead.
                                 // Gold order is kept: for a fixed output the taps still arrive sorted by (streamed offset, row, column)
    int pack = -1;               // --order rows, fp32: sums of two adjacent x points as float2 halves of the accumulator vectors and
                                 // __builtin_elementwise_fma on float2 operands -> v_pk_fma_f32 (two FMAs per lane and issue slot; each half is an
                                 // IEEE fma, so results stay bit-identical).  Pairs that start at an odd column are assembled from DPP moves.  -1 auto
    int rot_mod = 0;             // scatter schedule: register sets the partial sums rotate through (0 / < Range: Range).  A larger modulus costs
                                 // points-per-lane registers per extra set and can shrink the unroll lcm(modulus, LDS slots, prefetch sets) -- and with
                                 // it the code size: 14 -> 8 plane bodies for Range 7 with --rot-mod 8 (the instruction cache holds 64 KB)
    int pin = -1;                // 1: every partial sum a plane iteration (taps) / a row group (rows) has updated is passed through an empty
                                 // `asm volatile("" : "+v"(sum))` at its end.  Without it LLVM SINKS the FMA chains of the unrolled loop down to
                                 // the block that stores the finished plane -- the chain of an output spans `Range` unrolled iterations and has no
                                 // other use -- and keeps the SOURCE windows of all those planes alive instead of one partial sum (seen in the
                                 // ISA: one mul + 62 FMAs on one register pair right in front of the store, 370 live values for 136 named
                                 // ones, profiles/r03_sinking.md).  The pin makes each update used where it is written.  -1 auto: on with --order rows
    int gpus = 1;                // --gpus N > 1: the emitted program's main() is an N-GPU host -- launcher and ranks in one (it forks its ranks before any
                                 // HIP call, one process per GPU) on the C ABI's drs_slab_* entry points (z slabs / y slabs, RCCL send/recv halo exchange)
    std::vector<std::string> slab_args;   // the generator options of this invocation without -o / --gpus / --check / --gold and without the .stc path
    int out_skew = -1;           // --out-skew <MiB>: where the output array should sit relative to the input array, modulo 64 MiB (see
                                 // HipEmitter::out_skew_bytes; -1: chosen by the generator).  It changes no kernel text: the value is published
                                 // in the info JSON / the banner and honoured by the emitted host program, which owns its allocations
    int tuned_defaults = 1;      // --tuned-defaults 0: never consult the tuner's table (generator.hpp)
    bool tuning_given = false;   // some option other than the problem-naming ones was given
    int skew = -1;               // -1 auto: 1 for pipelines of three or more stages (measured: fp64 3 stages 3.49 -> 3.28 ms; 2 stages lose), else 0.
                                 // --skew 1 (round 4; temporal pipelines of streaming kernels, with --prefetch): stage t consumes the plane stage t-1 completed
                                 // in the PREVIOUS iteration.  All stages of one iteration are then independent of each other: they run back to back
                                 // between two barriers -- [barrier] every stage reads its arriving plane's rim from its own LDS slot, the last stage first
                                 // (its stores leave early) [barrier] every stage's completed plane and the next source plane are written to the slots --
                                 // instead of write / barrier / read once per stage; costs stages - 1 more prologue iterations per stream block
    int coef_sgpr = 0;           // --coef sgpr: fp32 coefficient values in scalar registers instead of 32-bit literals (4-byte instead of 8-byte FMAs)
    int row_fence = 0;           // mask of __builtin_amdgcn_sched_barrier between row groups (0: nothing crosses; -1: no fence)
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
    bool exact_x = true;     // x halo columns of the source plane come from the halo loaders (false: the tile's edge lanes load them)
    int AL = 0;              // columns at each x edge of the lane tile that are not owned (stages > 1)
    int OX = 0;              // columns owned per tile
    int zl = 0, zh = 0, hym = 0, hyp = 0, hxm = 0, hxp = 0;
    // geometry
    int BX = 64, BY = 4, VX = 4, RY = 1, SN = 16;
    bool cyclic_y = false;
    bool cyclic_x = false;   // --cyclic-merge-x n > 1 (and no larger --block-merge-x): a lane's n points are BX columns apart
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
    int ws = 0;              // loader wavefronts (--loader-waves): planes are requested by them, DEPTH = PD planes ahead, into NSLOT = PD + 1 slots
    bool dma = false;        // --stage dma: planes are staged by LDS-DMA into a per-lane-dense LDS image (emit_hip.hpp)
    // temporal blocking and the tolerance (planner.hpp: temporal_drift): which arithmetic the kernel computes
    bool reassociated = false;   // true: on-chip stages (equal to the reference's fused arithmetic up to rounding); false: gold order, bit-exact
    bool temporal_forced = false;    // --temporal force: emitted although the estimate may exceed the bar
    double drift_estimate = 0.0;     // predicted max relative distance from the fused arithmetic after the spec's iterations (0: gold order)
    double drift_per_launch = 0.0;   // the same after one launch (grows ~ launches^0.62)
    int horizon_iterations = -1;     // largest `iterations` for which the estimate stays within the bar (-1: unlimited, gold order)
    std::string error;       // non-empty: invalid configuration
    std::string note;        // non-empty: something the user asked for was not done (printed by the generator, kept in the banner)
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace drs
