// generator.hpp -- the `drstencil` command as a function: argv -> (messages, exit code,
// emitted source).  Option surface, defaults, messages and exit codes follow the
// reference's main.cpp:10-280 (hand-rolled scan 118-230: the last argument is always the
// .stc; a value-taking flag in the second-to-last slot is "Illegal input." exit 255; an
// unknown flag is "Illegal input." exit 0; `-o` without a value is ignored), plus
// additive MI355X options that the reference does not have.
#pragma once
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "emit_hip.hpp"
#include "planner.hpp"
#include "stencil_ir.hpp"
#include "tuned_defaults.hpp"

namespace drs {

// ---- tuner -> generator feedback (round 4; reference: benchmarks/3d7pt_star/tuning.py:125-131 ends with the best configuration in
// duration.log).  tuned_defaults.hpp is generated from drstencil_amd/tuned_defaults.tsv, which `tuning.py --write-defaults` maintains.
inline unsigned tuned_shape_hash(const CoefTable &base) {          // FNV-1a over "k,j,i;" of the one-step stencil's offsets in map order
    unsigned h = 0x811c9dc5u;
    for (auto &e : base.v) {
        char b[64];
        snprintf(b, sizeof b, "%d,%d,%d;", e.first.k, e.first.j, e.first.i);
        for (const char *c = b; *c; c++) h = (h ^ (unsigned char)*c) * 0x01000193u;
    }
    return h;
}
// the row of this problem class whose N is nearest in log2 (within a factor of sqrt 2), or nullptr
inline const TunedDefault *tuned_lookup(const char *mode, unsigned shape, int step, const std::string &dtype, int temporal, int N) {
    const TunedDefault *best = nullptr;
    double bestd = 0.0;
    for (int i = 0; i < kTunedDefaultsCount; i++) {
        const TunedDefault &r = kTunedDefaults[i];
        if (std::string(r.mode) != mode || r.shape != shape || r.step != step || dtype != r.dtype || r.temporal != temporal || N <= 0) continue;
        const double d = std::fabs(std::log2((double)N / (double)r.N));
        if (d <= 0.5 + 1e-9 && (!best || d < bestd)) { best = &r; bestd = d; }
    }
    return best;
}

struct GenResult {
    int exit_code = 0;
    std::string messages;   // what the command prints on stdout
    std::string source;     // emitted HIP source (empty when nothing was emitted)
    std::string out_name;
    bool emitted = false;
    Stencil st;
    KernelPlan plan;
    GenOptions opt;
    std::string notes;        // what the command prints on STDERR: remarks the reference has no counterpart for on a reference command line (stdout stays the reference's)
    std::string tuned_from;   // non-empty: the geometry / emission options came from the tuned-defaults table (this row's option string)
};

inline const char *help_text() {
    return R"(
    Generate data-reusing stencil kernels for AMD Instinct MI355X (gfx950, HIP).

    Usage: drstencil [options] <input_stcfile>
Options:

-o <file>               Specify the name of the output HIP source file.
                        (out.cu by default)

--3d                    Choose 3D mode.

--step <num>            Specify the number of time steps to fuse the stencil.
                        (step_num = 1 by default)

--dist <num>            Specify the number of the distance between points for data-reuse.

--streaming             Apply streaming optimization (2D; 3D always streams).

--bx <num>              Specify the workgroup size bx (lanes along x; 64 = one wavefront).

--by <num>              Specify the workgroup size by (rows of lanes along y).

--sn <num>              Specify the length of stream block sn.

--stream-unroll <num>   Specify the (minimum) unroll factor of the streaming loop.
                        (stream_unroll = 4 by default)

--block-merge-x <num>   Specify the number of contiguous points per lane along dimension x.

--block-merge-y <num>   Specify the number of adjacent rows per lane along dimension y.

--cyclic-merge-x <num>  Specify the number of points per lane along dimension x, bx columns apart.

--cyclic-merge-y <num>  Specify the number of rows per lane along dimension y, by rows apart.

--prefetch              Prefetch the next plane into registers to hide the transfer latency.

--merge-forward <num>   Specify the threshold for whether to merge the forward_j or forward_i into backward.
                        (merge_forward = 5 by default)

--check                 Check the correctness of the generated code against the gold kernel.

--gold                  Accepted for compatibility (no effect).

MI355X options:

--dtype <fp32|fp64>     Element type (fp64 by default, as the reference).
--xrim <lds|dpp>        x halo inside a wavefront by DPP wave shifts (default) or through LDS.
--schedule <scatter|reuse|window>  How reuse along the streamed dimension is split between source planes kept on chip and
                        partial sums carried in VGPRs (results never depend on it):
                        scatter (default without --dist): nothing retained, every arriving plane adds its taps to the partial
                        sums of all output planes in flight;
                        reuse (default with --dist d): the reference's split for that dist -- `Range` source planes stay
                        resident in register windows, partial sums are carried over the remaining planes; a retained
                        plane's in-plane neighbours are kept in registers when that plane has at least --merge-forward
                        in-plane taps, else re-read from LDS when they are due;
                        window: every contributing plane resident, nothing carried.
--order <taps|rows>     Emission order of a plane's FMAs (scatter schedule; results never depend on it).  taps (default): one
                        chain per partial sum, the plane's whole rim window read up front.  rows: the arriving plane is
                        consumed one source row at a time, each row's tap groups fenced from the next (sched_barrier), so
                        that the compiler cannot stretch every row's reads over the whole plane (register pressure).
--pack <0|1>            With --order rows, fp32: two adjacent x points per v_pk_fma_f32 (bit-identical results; default 1).
--pin <0|1>             Pass every partial sum through an empty asm statement where it is updated, so that the compiler cannot
                        sink the FMA chains of the unrolled streaming loop down to the store (which keeps `Range` planes of
                        source windows alive instead of the sums; default: 1 with --order rows, else 0).
--rot-mod <n>           Scatter schedule: the partial sums rotate through n >= Range register sets (default Range).  The streaming
                        loop is unrolled by lcm(n, LDS slots, prefetch sets): e.g. Range 7 -> 14 plane bodies, --rot-mod 8 -> 8
                        (code size; the instruction cache holds 64 KB).
--row-fence <mask>      sched_barrier mask between row groups (0 default: nothing crosses; -1: no fence).
--gpus <N>              N > 1: the emitted program's main() runs the spec slab-decomposed over N GPUs of one node (z slabs in 3D, y slabs in 2D): it
                        forks one rank process per GPU before any HIP call and drives the C ABI's drs_slab_* entry points (RCCL send/recv of the
                        halo planes, overlapped with the interior sweep).  Link it with -ldrstencil_amd; DRS_SLAB_REHEARSE=r/N runs rank r alone
                        on one GPU.  The kernels in the file (and its use as a plugin) are unchanged.
--out-skew <MiB>        Placement of the output array relative to the input array: (out - in) mod 64 MiB.  A z-streaming kernel reads a few
                        planes ahead of the plane it writes; when its writes land, modulo 64 MiB, 8-16 MiB behind its read front the launch
                        takes up to 14 % longer on MI355X (profiles/r03_probe_skew4.log).  Default: chosen from the kernel's read-ahead
                        distance (32 or 0).  The emitted program allocates both arrays in one arena accordingly; callers of the C ABI
                        read the recommendation from the kernel info (out_skew_bytes, placement_period_bytes).  Results never depend on it.
--zigzag <0|1>          1: a launch whose output array lies below its input array (every second launch of the reference's ping-pong loop)
                        walks its stream blocks in reverse order and so begins on the planes the previous launch wrote last.
--coef <lit|sgpr|vgpr>  fp32: coefficients as 32-bit literals of every FMA (lit, default) or held in scalar registers (sgpr: 4-byte
                        instead of 8-byte FMAs -- code size and instruction fetch of the fused multi-step kernels; same results).
--temporal <0|1|force>  With --step n > 1: run the one-step stencil n times on chip (temporal blocking,
                        intermediate planes never leave the CU) instead of the fused stencil.  On-chip stages
                        re-associate the fused sum: 1 emits them only where the estimated drift from the reference's
                        fused arithmetic stays within 1e-6 relative (fp32; 1e-12 fp64) for the spec's `iterations`
                        and emits the fused kernel otherwise (a note says so); force emits them regardless.
--skew <0|1|2>          Temporal pipelines of streaming kernels: 1 = stage t consumes what stage t-1 completed one iteration EARLIER, so the
                        stages of an iteration are independent and share two barriers instead of one write/barrier/read round per stage;
                        2 = also double-buffers the intermediate planes' LDS slots (compact, no pads), so completed planes are written
                        during the compute phase instead of in a burst between the two barriers.
--tuned-defaults <0|1>  1 (default): a command line without any geometry / emission option takes them from the tuner's table for
                        this stencil shape, step, dtype and grid size (drstencil_amd/tuned_defaults.tsv), when it has a row.
--prefetch-depth <n>    With --prefetch: planes in flight ahead of the one being summed (n+1 register sets; default 3 (fp32) /
                        2 (fp64) for fused multi-step 3D kernels, else 1).
--pair-launch <0|1>     Also emit dr2_<name>(in0, out0, in1, out1): the same sweep over two buffer pairs in one launch.
--exact-y <0|1>         1 (default for single-stage kernels): the y halo rows of the source plane are fetched by the halo loader
                        lanes, so every tile row is owned; 0: overlapped tiles (tile rows include the halo).
--uniform-loads <0|1|2> With --prefetch: 0 (default) plane loads under `if (more planes)`; 1 issued unconditionally (past the
                        block's last plane they re-read it); 2 unconditionally through a buffer window that closes past it.
                        1 and 2 leave no vector-memory instruction under a branch: exact s_waitcnt vmcnt(N) pipelining.
--store-mask <branch|buffer>  branch (default): guarded global stores; buffer: buffer stores whose per-lane offset is out of
                        range where the lane must not store (no branch).
--stage <reg|dma>       How an arriving plane reaches LDS in streaming kernels: reg (default) = global loads into VGPRs
                        (software prefetch) + LDS writes; dma = LDS-DMA (global_load_lds_dwordx4) straight into the plane's
                        LDS slot, one plane ahead, no prefetch registers (16-byte vectors, block y merging, one stage).
--loader-waves <n>      With --stage dma: n extra wavefronts per workgroup only request planes by LDS-DMA, --prefetch-depth planes
                        ahead into a ring of LDS slots, counting their own vmcnt; the bx*by consumer lanes never issue a load.
--defer-stores <0|1>    Hold a completed output plane in registers and store it one plane later, right after the next
                        plane's loads were issued (its write latency runs under that plane's work).
--drain <0|1|2>         s_waitcnt vmcnt(0) before every plane's prefetch loads (1) or before its LDS staging (2).
--cc-opt <flag>         Extra hipcc flag for this kernel (repeatable), e.g. --cc-opt -fno-slp-vectorize.
--exact-x <0|1>         1 (default): the x halo columns are fetched by the halo loader lanes; 0: overlapped tiles in x (the
                        tile's outermost lanes load them with the row and own nothing, e.g. --bx 34: 136 columns own 128).
--clamp-loads <0|1>     1 (default): branch-free loads -- lanes outside the grid read the plane origin (their
                        values never reach a stored output); 0: loads under per-lane guards.
--halo-spread <0|1>     Spread the halo loader tasks over all wavefronts (default 0: the first lanes take them).
--xedge-select <0|1>    With --xrim dpp: wavefront-edge lanes pick the LDS value by select (1) or branch (0, default).
--lazy-rims <0|1>       Read LDS rims when first needed (1) or when a plane arrives (0).
--xcd-remap <0|1|2>     workgroup to tile mapping: 0 dispatch order, 1 contiguous chunk of tiles
                        per XCD, 2 one x-y band per XCD with all XCDs on the same stream block
                        (default: 2 for 3D, 0 for 2D); 3 = like 2 with --zgroup <n> successive stream
                        blocks of a tile taken by consecutive workgroups; 4 = units of 32 consecutive tiles of one
                        stream block (the CUs of an XCD) dealt round-robin to the XCDs: neighbouring tiles run
                        together on one XCD and share their halo reads in its L2; 5 = chunks of --xcd-chunk <n> (4)
                        consecutive tiles per XCD and round (x neighbours share the lines of their x halos), one front.
--nt-store <0|1>        Non-temporal stores of the output (1 by default).
--nt-load <0|1>         Non-temporal loads of the input.
--waves-per-eu <num>    Second argument of __launch_bounds__.
--lds-pad <num>         Extra elements of padding per LDS row.
--ref-defaults          Keep the reference's 16x16x16 geometry defaults.

--help  (-h)            Print this help information on this tool.
        )";
}

inline GenResult generate(const std::vector<std::string> &args /* argv[1..] */) {
    GenResult res;
    GenOptions &o = res.opt;
    const int argc = (int)args.size() + 1;
    auto arg = [&](int i) -> const std::string & { return args[i - 1]; };
    if (argc < 2) { res.messages = "Please specify the .stc file.\n"; return res; }
    if (arg(1) == "--help" || arg(1) == "-h") { res.messages = std::string(help_text()) + "\n"; return res; }

    bool illegal_exit = false;
    for (int i = 1; i < argc - 1; i++) {
        const std::string &a = arg(i);
        auto int_opt = [&](int &dst, bool *flag) -> bool {
            if (i != argc - 2) { dst = atoi(arg(++i).c_str()); if (flag) *flag = true; return true; }
            res.messages += "Illegal input.\n"; res.exit_code = 255; illegal_exit = true; return false;
        };
        auto str_opt = [&](std::string &dst) -> bool {
            if (i != argc - 2) { dst = arg(++i); return true; }
            res.messages += "Illegal input.\n"; res.exit_code = 255; illegal_exit = true; return false;
        };
        // options that NAME the problem or the artefact; every other option is a tuning choice and switches the tuned-defaults table off
        if (!(a == "-o" || a == "--3d" || a == "--step" || a == "--streaming" || a == "--check" || a == "--gold" || a == "--dtype" || a == "--gpus" ||
              a == "--pair-launch" || a == "--temporal" || a == "--dist" || a == "--tuned-defaults" || a == "--out-skew")) o.tuning_given = true;
        if (a == "-o") { if (i != argc - 2) { o.out_name = arg(++i); o.out_set = true; } }
        else if (a == "--3d") o.is3d = true;
        else if (a == "--step") { if (!int_opt(o.step, nullptr)) break; }
        else if (a == "--dist") { if (!int_opt(o.dist, nullptr)) break; }
        else if (a == "--streaming") o.streaming = true;
        else if (a == "--bx") { if (!int_opt(o.bx, &o.bx_set)) break; }
        else if (a == "--by") { if (!int_opt(o.by, &o.by_set)) break; }
        else if (a == "--sn") { if (!int_opt(o.sn, &o.sn_set)) break; }
        else if (a == "--block-merge-x") { if (!int_opt(o.bmx, &o.mx_set)) break; }
        else if (a == "--block-merge-y") { if (!int_opt(o.bmy, &o.my_set)) break; }
        else if (a == "--cyclic-merge-x") { if (!int_opt(o.cmx, &o.mx_set)) break; }
        else if (a == "--cyclic-merge-y") { if (!int_opt(o.cmy, &o.my_set)) break; }
        else if (a == "--stream-unroll") { if (!int_opt(o.stream_unroll, nullptr)) break; }
        else if (a == "--prefetch") o.prefetch = true;
        else if (a == "--merge-forward") { if (!int_opt(o.merge_forward, nullptr)) break; }
        else if (a == "--check") o.check = true;
        else if (a == "--gold") o.gold = true;
        // ---- additive options
        else if (a == "--dtype") { if (!str_opt(o.dtype)) break; }
        else if (a == "--xrim") { if (!str_opt(o.xrim)) break; }
        else if (a == "--schedule") { if (!str_opt(o.schedule)) break; o.schedule_set = true; }
        else if (a == "--order") { if (!str_opt(o.order)) break; o.order_set = true; }
        else if (a == "--pack") { if (!int_opt(o.pack, nullptr)) break; }
        else if (a == "--row-fence") { if (!int_opt(o.row_fence, nullptr)) break; }
        else if (a == "--out-skew") { if (!int_opt(o.out_skew, nullptr)) break; }
        else if (a == "--gpus") { if (!int_opt(o.gpus, nullptr)) break; }
        else if (a == "--coef") {
            std::string v;
            if (!str_opt(v)) break;
            if (v == "sgpr") o.coef_sgpr = 1;
            else if (v == "vgpr") o.coef_sgpr = 2;
            else if (v == "lit") o.coef_sgpr = 0;
            else { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
        }
        else if (a == "--loader-waves") { if (!int_opt(o.loader_waves, nullptr)) break; }
        else if (a == "--rot-mod") { if (!int_opt(o.rot_mod, nullptr)) break; }
        else if (a == "--skew") { if (!int_opt(o.skew, nullptr)) break; }
        else if (a == "--tuned-defaults") { if (!int_opt(o.tuned_defaults, nullptr)) break; }
        else if (a == "--pin") { if (!int_opt(o.pin, nullptr)) break; }
        else if (a == "--exact-y") { if (!int_opt(o.exact_y, nullptr)) break; }
        else if (a == "--exact-x") { if (!int_opt(o.exact_x, nullptr)) break; }
        else if (a == "--xedge-select") { if (!int_opt(o.xedge_select, nullptr)) break; }
        else if (a == "--debug-skip") { if (!int_opt(o.debug_skip, nullptr)) break; }
        else if (a == "--debug-drop-barrier") { if (!int_opt(o.debug_drop_barrier, nullptr)) break; }
        else if (a == "--clamp-loads") { if (!int_opt(o.clamp_loads, nullptr)) break; }
        else if (a == "--halo-spread") { if (!int_opt(o.halo_spread, nullptr)) break; }
        else if (a == "--zgroup") { if (!int_opt(o.zgroup, nullptr)) break; }
        else if (a == "--xcd-chunk") { if (!int_opt(o.xcd_chunk, nullptr)) break; }
        else if (a == "--zigzag") { if (!int_opt(o.zigzag, nullptr)) break; }
        else if (a == "--pair-launch") { if (!int_opt(o.pair_launch, nullptr)) break; }
        else if (a == "--prefetch-auto") { if (!int_opt(o.prefetch_auto, nullptr)) break; }
        else if (a == "--prefetch-depth") { if (!int_opt(o.prefetch_depth, nullptr)) break; }
        else if (a == "--temporal") {
            std::string v;
            if (!str_opt(v)) break;
            if (v == "force") o.temporal = 2;
            else if (v == "0" || v == "1" || v == "2") o.temporal = atoi(v.c_str());
            else { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
        }
        else if (a == "--defer-stores") { if (!int_opt(o.defer_stores, nullptr)) break; }
        else if (a == "--drain") { if (!int_opt(o.drain, nullptr)) break; }
        else if (a == "--uniform-loads") { if (!int_opt(o.uniform_loads, nullptr)) break; }
        else if (a == "--store-mask") { if (!str_opt(o.store_mask)) break; }
        else if (a == "--stage") { if (!str_opt(o.stage)) break; }
        else if (a == "--cc-opt") { std::string f; if (!str_opt(f)) break; o.cc_opts.push_back(f); }
        else if (a == "--lazy-rims") { if (!int_opt(o.lazy_rims, nullptr)) break; }
        else if (a == "--xcd-remap") { if (!int_opt(o.xcd_remap, nullptr)) break; }
        else if (a == "--nt-store") { if (!int_opt(o.nt_store, nullptr)) break; }
        else if (a == "--nt-load") { if (!int_opt(o.nt_load, nullptr)) break; }
        else if (a == "--waves-per-eu") { if (!int_opt(o.waves_per_eu, nullptr)) break; }
        else if (a == "--lds-pad") { if (!int_opt(o.lds_pad, nullptr)) break; }
        else if (a == "--ref-defaults") o.ref_defaults = 1;
        else { res.messages += "Illegal input.\n"; res.exit_code = 0; return res; }
    }
    if (illegal_exit) return res;
    if (o.dtype != "fp32" && o.dtype != "fp64") { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
    // an explicit --dist selects the reference's kind of reuse: `Range` source planes resident, the rest carried as partial sums
    if (!o.schedule_set && o.dist != 0) o.schedule = "reuse";
    if (o.schedule != "scatter" && o.schedule != "window" && o.schedule != "reuse") { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
    if (o.order != "taps" && o.order != "rows") { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
    if (o.xrim != "lds" && o.xrim != "dpp") { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
    if (o.stage != "reg" && o.stage != "dma") { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
    if (o.store_mask != "buffer" && o.store_mask != "branch") { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
    if (o.step < 1) { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }

    const std::string &stcfile = arg(argc - 1);
    Stencil &st = res.st;
    st.ndim = o.is3d ? 3 : 2;
    if (st.read_stc(stcfile) != 0) { res.messages += "Error opening stencil file.\n"; res.exit_code = 255; return res; }
    // No geometry / emission option on the command line: the tuner's winner for this problem class, if it has one, supplies them
    // (drstencil_amd/tuned_defaults.tsv).  The command is re-read with the row's options in front of the .stc, so the banner's
    // `options:` line, the kernel info and the cache key are those of the explicit command line.
    if (o.tuned_defaults && !o.tuning_given && !o.ref_defaults) {
        int order = 0;
        for (auto &e : st.pts.v) order = std::max(order, std::abs(st.ndim == 3 ? e.first.k : e.first.j));
        const char *mode = st.ndim == 3 ? "3d" : (o.streaming ? "2ds" : "2d");
        if (const TunedDefault *t = tuned_lookup(mode, tuned_shape_hash(st.pts), o.step, o.dtype, (o.temporal && o.step > 1) ? 1 : 0, st.N)) {
            std::vector<std::string> again(args.begin(), args.end() - 1);
            std::istringstream is(t->options);
            for (std::string w; is >> w;) again.push_back(w);
            again.push_back("--tuned-defaults"); again.push_back("0");
            again.push_back(args.back());
            GenResult r2 = generate(again);
            r2.tuned_from = t->options;
            if (r2.emitted) r2.notes += std::string("drstencil: note: no geometry option given: the tuner's configuration for this stencil, step, dtype and grid size is used (") + t->options + "); --tuned-defaults 0 keeps the generic defaults\n";
            return r2;
        }
    }
    st.fuse(o.step);
    st.choose_halo_dist(o.dist);
    if (st.partition_reuse(o.merge_forward) != REUSE_OK) { res.messages += "No data to reuse. You can try another dist.\n"; res.exit_code = 1; return res; }
    st.stream_range();

    // Round 3 default emission (no --order given): the rows order -- plane consumed by source row, partial sums pinned -- where it measured
    // faster AND is what makes the kernel fit at all: fused 3D stencils beyond 25 taps (--step 3: 256 VGPRs + scratch in the taps order,
    // 98-114 in the rows order; 1880-2080 against 430 GStencil/s at 1024^3) and one-shot 2D tiles of more than 9 taps (2d25pt_box: +6 %,
    // profiles/r03_exp_r3d.log).  Everything else keeps round 2's emission, which measured faster there (the memory-bound step-2 headline).
    if (!o.order_set && !o.ref_defaults && !o.temporal && o.schedule == "scatter" && o.stage == "reg" && std::max(o.bmy, o.cmy) == std::max(o.bmy, 1) &&
        !(o.cmx > 1 && o.cmx >= o.bmx) &&
        ((st.ndim == 3 && st.pts.size() > 25) || (st.ndim == 2 && !o.streaming && st.pts.size() > 9))) {
        o.order = "rows";
        if (o.pack < 0) o.pack = 0;       // packed pairs buy nothing with four waves per SIMD
    }
    res.plan = make_plan(st, o, kernel_base_name(stcfile));
    if (!res.plan.error.empty()) { res.messages += "Invalid configuration!\n"; res.exit_code = 255; return res; }
    if (!res.plan.note.empty()) res.messages += "drstencil: note: " + res.plan.note + "\n";
    std::string cmdline;
    for (size_t i = 0; i + 1 < args.size(); i++) cmdline += (i ? " " : "") + args[i];
    for (size_t i = 0; i + 1 < args.size(); i++) {
        if (args[i] == "-o" || args[i] == "--gpus") { i++; continue; }
        if (args[i] == "--check" || args[i] == "--gold") continue;
        o.slab_args.push_back(args[i]);
    }
    if (o.gpus < 1 || o.gpus > 64) { res.messages += "Illegal input.\n"; res.exit_code = 255; return res; }
    HipEmitter em(res.plan, o);
    if (!em.config_error().empty()) { res.messages += "Invalid configuration!\n"; res.exit_code = 255; res.plan.error = em.config_error(); return res; }
    if (em.lds_bytes() > 160 * 1024) {   // gfx950: 160 KiB of LDS per workgroup
        res.messages += "Invalid configuration!\n"; res.exit_code = 255; res.plan.error = "tile needs more than 160 KiB of LDS"; return res;
    }
    res.source = em.source(stcfile, cmdline);
    res.out_name = o.out_name;
    res.emitted = true;
    return res;
}

inline bool write_text(const std::string &path, const std::string &text) {
    std::ofstream f(path, std::ios::out | std::ios::trunc);
    if (!f) return false;
    f << text;
    return (bool)f;
}

}  // namespace drs
