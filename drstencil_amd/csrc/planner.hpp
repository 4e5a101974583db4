// planner.hpp -- turns (Stencil, GenOptions) into a KernelPlan: role mapping of the
// dims, tile geometry, LDS layout and the validity checks.
//
// Reference behaviour kept: block-vs-cyclic resolution and merge factors
// (main.cpp:232-235), the "Invalid configuration!" test (codegen.hpp:50-55,
// codegen_2d.hpp:52-57), --streaming ignored in 3D and --by / y-merge ignored by
// the 2D streaming kernel (codegen_2d.hpp:125).
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include "plan.hpp"

namespace drs {

inline std::string kernel_base_name(const std::string &stc_path) {
    // reference: path minus its last 4 characters (main.cpp:243-244), so the .stc has
    // to sit in the cwd; we additionally drop directories and sanitise to an identifier.
    std::string s = stc_path;
    size_t slash = s.find_last_of('/');
    if (slash != std::string::npos) s = s.substr(slash + 1);
    if (s.size() >= 4) s.erase(s.size() - 4);
    for (auto &c : s) if (!(isalnum((unsigned char)c) || c == '_')) c = '_';
    if (s.empty() || isdigit((unsigned char)s[0])) s = "s" + s;
    return s;
}

// true when the reference would print "Invalid configuration!" (exit 255)
inline bool reference_invalid(const Stencil &st, const GenOptions &o, int mx, int my) {
    int halo2 = st.halo * 2;
    bool fi = st.fwd_i.size() > 0, fj = st.fwd_j.size() > 0;
    if (st.ndim == 3) return (halo2 >= o.bx * mx && fi) || (halo2 >= o.by * my && fj);
    // 2D (codegen_2d.hpp:52-56): only the x extent is tested
    (void)fj; (void)my;
    return halo2 >= o.bx * mx && fi;
}

// A-priori estimate of how far a temporal pipeline (the one-step stencil applied `step` times on chip, every stage an FMA chain
// in gold order) drifts from the reference's fused arithmetic (drstencil.hpp:262-282: ONE chain over the fused taps): the max
// relative difference over the grid after `launches` launches of the ping-pong loop (codegen.hpp:581-584).
//   * both sides are rounded evaluations of the same exact sum; their difference after one launch behaves like noise of
//     u * sqrt(step * taps + fused taps) (u = unit roundoff: one rounding per FMA of either chain), scaled by the condition
//     number of the sum for the harness's non-negative inputs, sum|c| / |sum c| (common.hpp:9-32: inputs in [0, 1]);
//   * it grows like launches^0.62 (a random walk that the stencil's own averaging damps) and the MAX over n grid points sits
//     sqrt(2 ln n) standard deviations out;
//   * 0.178 is fitted to the MI355X measurements of round 2 (profiles/r02_temporal_margin.json: 3d7pt_star, 2 and 3 stages, 2 and
//     50 / 34 launches on 96x80x264; scripts/parity_full.py at 1024^3) -- all six within 4 % -- and checked on the CPU against
//     chained oracle sweeps for the shipped stencils and random shapes (tests/calibrate_temporal_drift.py,
//     profiles/r03_temporal_drift_calibration.txt); 1.25 is the safety margin on top.
// Returns the estimate for one launch; est(launches) = est(1) * launches^0.62.
inline double temporal_drift_per_launch(const Stencil &st, bool fp32) {
    const double u = fp32 ? 5.9604644775390625e-08 : 1.1102230246251565e-16;
    double sabs = 0.0, ssum = 0.0;
    for (auto &e : st.pts.v) { sabs += std::fabs(e.second); ssum += e.second; }
    const double cond = std::fabs(ssum) > 1e-300 ? sabs / std::fabs(ssum) : 1e300;
    double n = (double)std::max(1, st.M) * (double)std::max(1, st.N) * (st.ndim == 3 ? (double)std::max(1, st.L) : 1.0);
    const double spread = std::sqrt(2.0 * std::log(std::max(16.0, n)));
    return 1.25 * 0.178 * u * std::sqrt((double)st.step * (double)st.base.size() + (double)st.pts.size()) * spread * cond;
}
inline double temporal_bar(bool fp32) { return fp32 ? 1e-6 : 1e-12; }
// largest `iterations` whose launches (2 * ceil(iterations / (2 * step))) keep the estimate within the bar; 0: not even one pair
inline int temporal_horizon(double per_launch, bool fp32, int step) {
    if (per_launch <= 0.0) return 1 << 30;
    const double nmax = std::pow(temporal_bar(fp32) / per_launch, 1.0 / 0.62);
    if (nmax >= 1e9) return 1 << 30;
    return 2 * step * (int)(std::floor(nmax + 1e-9) / 2);
}

inline KernelPlan make_plan(const Stencil &st, const GenOptions &o_in, const std::string &name) {
    GenOptions o = o_in;
    KernelPlan p;
    p.name = name;
    p.ndim = st.ndim;
    p.fp32 = (o.dtype == "fp32");
    p.L = st.L; p.M = st.M; p.N = st.N;
    p.iterations = st.iterations; p.step = st.step; p.halo = st.halo; p.dist = st.dist; p.range = st.range();
    p.prefetch = o.prefetch || (o_in.step > 1 && st.ndim == 3 && !o.ref_defaults && o.prefetch_auto);

    const bool stream2d = (st.ndim == 2 && o.streaming);
    p.has_s = (st.ndim == 3) || stream2d;
    p.has_y = (st.ndim == 3) || !stream2d;

    // Temporal blocking applies the ONE-STEP stencil `step` times on chip.  It equals the
    // reference's algebraically fused stencil (drstencil.hpp:262-282) up to rounding, provided
    // the 6-digit rounding of the fused coefficients (drstencil.hpp:192) is a no-op; otherwise
    // the fused single-pass kernel is emitted (exact reference arithmetic).
    bool temporal = o.temporal && st.step > 1;
    if (temporal)
        for (auto &e : st.pts.v)
            if (std::fabs(coef_rounded(e.second) - e.second) > 1e-12 * std::fabs(e.second)) temporal = false;
    if (o.temporal && st.step > 1 && !temporal)
        p.note = "--temporal 1 ignored: rounding the fused coefficients to 6 digits is not a no-op for this stencil, so on-chip "
                 "stages would not equal the reference's fused arithmetic; the fused single-pass kernel is emitted instead";
    // ... and provided the re-association stays within the tolerance for the spec's iteration count (--temporal force overrides)
    if (temporal) {
        p.drift_per_launch = temporal_drift_per_launch(st, p.fp32);
        const int launches = st.launches();
        p.drift_estimate = p.drift_per_launch * std::pow((double)std::max(1, launches), 0.62);
        p.horizon_iterations = temporal_horizon(p.drift_per_launch, p.fp32, st.step);
        if (o.temporal >= 2) p.temporal_forced = true;
        else if (p.drift_estimate > temporal_bar(p.fp32)) {
            char b[512];
            snprintf(b, sizeof b, "--temporal 1 not honoured: %d on-chip stages re-associate the fused sum, and for this stencil (%zu + %zu taps), grid and "
                                  "`iterations %d` (%d launches) the estimated drift from the reference's fused arithmetic is %.2g > %.0e relative "
                                  "(tolerance horizon: iterations <= %d); the fused single-pass kernel is emitted instead (exact reference arithmetic; "
                                  "--temporal force emits the pipeline anyway)",
                     st.step, st.base.size(), st.pts.size(), st.iterations, launches, p.drift_estimate, temporal_bar(p.fp32), p.horizon_iterations);
            p.note = b;
            temporal = false;
            p.drift_estimate = p.drift_per_launch = 0.0;
            p.horizon_iterations = -1;
        }
    }
    p.reassociated = temporal;
    p.stages = temporal ? st.step : 1;
    if (!temporal) o.temporal = 0;       // the geometry defaults below are those of the kernel that is really emitted

    // MI355X defaults for whatever geometry the user left unset: one wavefront (64
    // lanes) along x with 16-byte accesses per lane.
    const int vec_elems = p.fp32 ? 4 : 2;
    // Temporal pipelines (tuned on 3d7pt_star 1024^3, profiles/r01_tune_c4_s2.txt): pick the lane count
    // whose OWNED width (mx*bx - 2*roundup((step-1)*hx, mx)) tiles N with the least idle lanes -- e.g.
    // 66 lanes x 4 = 264 columns own 256, so 4 tiles cover N = 1024 where 64 lanes would need 5 --
    // many short rows per workgroup (15 x 2), 32-plane stream blocks, software prefetch, dispatch order.
    if (!o.ref_defaults && o.temporal && st.step > 1 && st.ndim == 3) {
        int hx = 0;
        for (auto &e : st.base.v) hx = std::max(hx, std::abs(e.first.i));
        if (!o.mx_set) { o.bmx = vec_elems; o.cmx = 1; o.mx_set = true; }
        const int mxv = std::max(o.bmx, o.cmx);
        if (!o.bx_set) {
            const int al = round_up((st.step - 1) * hx, mxv);
            long best = -1;
            for (int bx : {66, 64, 34, 32, 68, 36}) {   // wider rows leave too few rows per workgroup (measured slower)
                if ((bx == 68 || bx == 36) && st.step < 4) continue;      // 68 / 36: for four or more stages only (fp64: they lose 4 columns per side, 68 x 2 = 136 own 128)
                const int ox = bx * mxv - 2 * al;
                if (ox < 1) continue;
                const long cost = (long)ceil_div(st.N - st.halo, ox) * bx * mxv;   // lane-columns spent on a row
                if (best < 0 || cost < best) { best = cost; o.bx = bx; }
            }
            o.bx_set = true;
        }
        // four or more stages keep 3(n - 1) + ... planes of sums per lane: 12 wavefronts (<= 768 lanes, 3 per SIMD: 168 registers) hold what 16
        // cannot (fp64, 4 stages: 151-163 VGPRs at 726 lanes, scratch at 990; profiles/r04_exp_r4o.log)
        if (!o.by_set) { o.by = std::max(1, std::min(15, (st.step >= 4 ? 768 : 1000) / o.bx)); o.by_set = true; }
        if (!o.my_set) { o.bmy = 2; o.cmy = 1; o.my_set = true; }
        if (!o.sn_set) { o.sn = 32; o.sn_set = true; }
    }
    // Fused multi-step kernels in 3D (step > 1 without --temporal; tuned on 3d7pt_star 1024^3 step 2,
    // profiles/r01_tune_c4_s2_exhaustive.txt, r01_tune_c3_s2_depth_exhaustive.txt): the wide fused window costs
    // registers, so 2 rows per lane, 512 lanes (fp32: 32 x 16, the optimum of both searches; fp64: 64 x 8), longer
    // stream blocks (the z halo is step*order planes) and software prefetch (depth: see HipEmitter::analyse).
    if (!o.ref_defaults && !o.temporal && st.step > 1 && st.ndim == 3) {
        // beyond the 25-point fused 7-point star (63-point step 3, 27-point fused cross, ...) the partial sums and rims of
        // a 512-lane workgroup no longer fit 256 registers per lane and spill to scratch: 256 lanes (64 x 4) may use the
        // AGPR half of the register file as well and stay spill-free
        const bool heavy = st.pts.size() > 25;
        if (heavy && o.order == "rows" && !o.bx_set && !o.by_set) {
            // round 3 (rows order, pinned sums: 98-114 VGPRs, two workgroups per CU): best of the 1248-configuration grid on 3d7pt_star 1024^3
            // step 3 (profiles/r03_tune_c4_s3_rows_grid.txt) and of the fp64 runs (r03_exp_r3f.log): 64 x 8 lanes, 2 rows per lane, 64-plane blocks
            o.bx = 64; o.by = 8; o.bx_set = o.by_set = true;
            if (!o.sn_set) { o.sn = 64; o.sn_set = true; }
        } else if (heavy && !o.bx_set && !o.by_set) {
            if (p.fp32) { o.bx = 64; o.by = 4; }
            else { o.bx = 32; o.by = 8; if (!o.my_set) { o.bmy = 1; o.cmy = 1; o.my_set = true; } }   // fp64: one row per lane (2 rows spill at 512^3)
            o.bx_set = o.by_set = true;
        }
        if (p.fp32 && !o.bx_set && !o.by_set) { o.bx = 32; o.by = 16; o.bx_set = o.by_set = true; }
        if (!o.by_set) { o.by = 8; o.by_set = true; }
        if (!o.my_set) { o.bmy = 2; o.cmy = 1; o.my_set = true; }
        if (!o.sn_set) { o.sn = 32; o.sn_set = true; }
    }
    // One-step 3D kernels (tuned by the exhaustive search on 3d7pt_star 1024^3, profiles/r01_tune_c4_s1_exhaustive.txt:
    // 80 % of the HBM peak = the chip's copy ceiling): a workgroup of 512 lanes owns FULL rows when a row fits
    // 256 lanes (no x halo, every plane access is a run of whole rows), 4 rows per lane, 4-plane stream blocks,
    // software prefetch.
    if (!o.ref_defaults && st.step == 1 && st.ndim == 3 && !o.bx_set && !o.by_set && !o.my_set && !o.sn_set) {
        const int mxv = o.mx_set ? std::max(o.bmx, o.cmx) : vec_elems;
        const int lanes = st.N / mxv;
        if (st.N % mxv == 0 && lanes == 256) {   // measured only there; narrower rows keep the 64-lane default (C3: full rows lose)
            o.bx = lanes; o.by = 512 / lanes; o.bmy = 4; o.cmy = 1; o.sn = 4;
            o.bx_set = o.by_set = o.my_set = o.sn_set = true;
            if (o.prefetch_auto) o.prefetch = true;
        }
    }
    if (!o.ref_defaults) {
        if (!o.bx_set) o.bx = 64;
        if (!o.mx_set) { o.bmx = vec_elems; o.cmx = 1; }
        if (!o.by_set) o.by = p.has_y ? 4 : 1;
        if (!o.my_set && p.has_y) { o.bmy = 8; o.cmy = 1; }
        // short stream blocks keep all workgroups on one compact front through memory
        // (measured: sn 8 beats sn 64 by 9 % on 3d7pt_star 1024^3, profiles/)
        if (!o.sn_set) o.sn = 8;
    }
    p.prefetch = o.prefetch || (o_in.step > 1 && st.ndim == 3 && !o.ref_defaults && o.prefetch_auto);
    const bool bmerge_y = o.bmy > o.cmy;
    const bool bmerge_x = o.bmx > o.cmx;       // main.cpp:232-235
    const int mx = std::max(o.bmx, o.cmx), my = std::max(o.bmy, o.cmy);

    // role mapping (streamed, row, col)
    auto to_tap = [&](const std::pair<Pt, double> &e, bool full_precision) {
        Tap t;
        if (st.ndim == 3) { t.ds = e.first.k; t.dy = e.first.j; t.dx = e.first.i; }
        else if (stream2d) { t.ds = e.first.j; t.dy = 0; t.dx = e.first.i; }
        else { t.ds = 0; t.dy = e.first.j; t.dx = e.first.i; }
        if (full_precision) { char b[64]; snprintf(b, sizeof b, "%.17g", e.second); t.coef = b; }
        else t.coef = coef_text(e.second);
        return t;
    };
    for (auto &e : st.pts.v) p.gtaps.push_back(to_tap(e, false));
    if (temporal) for (auto &e : st.base.v) p.taps.push_back(to_tap(e, true));
    else p.taps = p.gtaps;
    for (auto &t : p.taps) {
        p.zl = std::min(p.zl, t.ds); p.zh = std::max(p.zh, t.ds);
        p.hym = std::max(p.hym, -t.dy); p.hyp = std::max(p.hyp, t.dy);
        p.hxm = std::max(p.hxm, -t.dx); p.hxp = std::max(p.hxp, t.dx);
    }
    if (st.ndim == 3) { p.DS = st.L; p.DY = st.M; p.DX = st.N; p.stride_s = (long)st.M * st.N; p.stride_y = st.N; }
    else if (stream2d) { p.DS = st.M; p.DY = 1; p.DX = st.N; p.stride_s = st.N; p.stride_y = 0; }
    else { p.DS = 1; p.DY = st.M; p.DX = st.N; p.stride_s = 0; p.stride_y = st.N; }

    if (p.taps.empty()) { p.error = "empty stencil"; return p; }
    if (st.M <= 0 || st.N <= 0 || (st.ndim == 3 && st.L <= 0)) { p.error = "grid size missing in the .stc"; return p; }
    // The interior guard is Halo in every dim (codegen.hpp:654); a tap reaching further
    // than Halo would read outside the arrays in the reference.
    if (p.stages * std::max({-p.zl, p.zh, p.hym, p.hyp, p.hxm, p.hxp}) > st.halo) { p.error = "a tap reaches beyond Halo (outermost-dim order)"; return p; }
    // evaluated on the options as given (reference defaults for the unset ones), so the
    // error behaviour is the reference's own
    if (reference_invalid(st, o_in, std::max(o_in.bmx, o_in.cmx), std::max(o_in.bmy, o_in.cmy))) { p.error = "tile does not cover the halo"; return p; }

    p.BX = o.bx; p.VX = mx;
    p.BY = p.has_y ? o.by : 1;
    p.RY = p.has_y ? my : 1;
    p.cyclic_y = p.has_y && !bmerge_y && my > 1;
    p.cyclic_x = !bmerge_x && mx > 1;
    p.SN = p.has_s ? std::max(1, o.sn) : 1;
    p.NT = p.BX * p.BY;
    if (p.BX < 1 || p.BY < 1 || p.NT > 1024) { p.error = "workgroup must have 1..1024 threads"; return p; }

    // vector width of memory accesses: 16 bytes when rows stay 16-byte aligned
    int vl = p.fp32 ? 4 : 2;
    while (vl > 1 && (p.VX % vl != 0 || st.N % vl != 0)) vl >>= 1;
    if (p.cyclic_x) vl = 1;                    // a lane's points are BX columns apart: element-wide accesses
    p.VL = vl; p.NV = p.VX / vl;
    p.TX = p.BX * p.VX;
    p.TY = p.BY * p.RY;
    p.PADL = p.hxm ? round_up(p.hxm, vl) : 0;
    p.PADR = p.hxp ? round_up(p.hxp, vl) : 0;
    // The source plane's y halo rows are fetched by the halo loaders (like its x halo columns), so the
    // first stage is valid on every tile row; each further on-chip stage loses hy rows per side.
    // (auto: every tile row owned for single-stage kernels; pipelines keep overlapped rows -- their lanes own 2 rows and the loader work of
    // the first wavefronts shows -- except skewed ones, where the exact source halo measured 6 % faster: 3.56 -> 3.35 ms, profiles/r04_exp_r4b.log)
    const bool skew_on = (o.skew > 0 || (o.skew < 0 && p.stages >= 3)) && p.stages > 1 && p.has_s && p.prefetch && o.stage != "dma";
    p.exact_y = o.exact_y < 0 ? (p.stages == 1 || skew_on) : (o.exact_y != 0);
    p.oym = (p.stages - (p.exact_y ? 1 : 0)) * p.hym; p.oyp = (p.stages - (p.exact_y ? 1 : 0)) * p.hyp;
    p.OY = p.has_y ? p.TY - p.oym - p.oyp : 1;
    if (p.OY < 1) { p.error = "tile has no rows left after the y halo"; return p; }
    if (p.PADL > p.TX || p.PADR > p.TX) { p.error = "x halo wider than the tile"; return p; }
    // stages > 1: intermediate planes have no x halo, so each extra stage loses hx columns per side
    p.exact_x = (o.exact_x != 0);
    p.AL = (p.stages > 1) ? round_up((p.stages - 1) * std::max(p.hxm, p.hxp), vl) : 0;
    if (!p.exact_x) p.AL += round_up(std::max(p.hxm, p.hxp), vl);      // the source plane's x halo is part of the lane tile too
    p.OX = p.TX - 2 * p.AL;
    if (p.OX < 1) { p.error = "tile has no columns left after the x halo of the fused stages"; return p; }
    const int H = st.halo;
    p.NBX = std::max(1, ceil_div(p.DX - H, p.OX));
    p.NBY = p.has_y ? std::max(1, ceil_div(p.DY - H, p.OY)) : 1;
    p.NBS = p.has_s ? std::max(1, ceil_div(p.DS - 2 * H, p.SN)) : 1;
    if (p.DX - 2 * H < 1 || (p.has_y && p.DY - 2 * H < 1) || (p.has_s && p.DS - 2 * H < 1)) { p.error = "grid has no interior"; return p; }

    if (o.loader_waves > 0 && o.stage != "dma") { p.error = "--loader-waves goes with --stage dma"; return p; }
    if (o.stage == "dma") {
        // LDS-DMA writes 64 lanes x 16 bytes of one wavefront instruction to consecutive LDS addresses: the LDS image is
        // dense per (row, vector, lane) and the halo pieces dense per loader task (emit_hip.hpp); what that needs:
        if (!p.has_s) { p.error = "--stage dma is for streaming kernels"; return p; }
        if (p.stages > 1) { p.error = "--stage dma: on-chip stages exchange through LDS writes"; return p; }
        if (p.VL * (p.fp32 ? 4 : 8) != 16) { p.error = "--stage dma needs 16-byte vectors (N and the x merge factor multiples of 4 fp32 / 2 fp64)"; return p; }
        if (p.cyclic_y) { p.error = "--stage dma needs block y merging"; return p; }
        if (p.cyclic_x) { p.error = "--stage dma needs block x merging"; return p; }
        if (!o.clamp_loads || o.halo_spread) { p.error = "--stage dma needs --clamp-loads 1 --halo-spread 0"; return p; }
        if (p.has_y && !p.exact_y) { p.error = "--stage dma needs --exact-y 1"; return p; }
        if (!p.exact_x) { p.error = "--stage dma needs --exact-x 1"; return p; }
        p.dma = true;
        p.prefetch = false;      // the look-ahead is the DMA itself
        if (o.loader_waves > 0) {
            if (p.NT % 64 != 0) { p.error = "--loader-waves needs whole consumer wavefronts (bx * by a multiple of 64)"; return p; }
            if (p.NT + 64 * o.loader_waves > 1024) { p.error = "--loader-waves: more than 1024 lanes per workgroup"; return p; }
            if (o.schedule != "scatter" || o.defer_stores) { p.error = "--loader-waves needs --schedule scatter without --defer-stores"; return p; }
            p.ws = o.loader_waves;
        }
    }
    p.SROW = p.PADL + p.TX + p.PADR + o.lds_pad;
    p.SROWS = p.has_y ? p.TY + p.hym + p.hyp : 1;
    if (p.stages > 1) p.NSLOT = p.stages;
    return p;
}

}  // namespace drs
