// stencil_ir.hpp -- stencil IR for the MI355X generator: .stc reader, algebraic
// step fusion, halo/dist selection and the data-reuse partition.
//
// Behavioural contract (file:line under the DRStencil reference tree):
//   read_stc          drstencil.hpp:52-78, drstencil_2d.hpp:48-73
//   fuse              drstencil.hpp:262-282, drstencil_2d.hpp:231-251
//   choose_halo_dist  drstencil.hpp:88-103,  drstencil_2d.hpp:82-97
//   partition_reuse   drstencil.hpp:198-259, drstencil_2d.hpp:180-228
//   stream_range      drstencil.hpp:285-304, drstencil_2d.hpp:254-269
//   coef_text         drstencil.hpp:192 (default ostream precision == "%g")
//
// One IR serves 2D and 3D: a point is always (k, j, i); 2D specs keep k == 0 and
// L == 1, and the "outermost" dim that defines Halo is j instead of k.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace drs {

struct Pt {
    int k = 0, j = 0, i = 0;
    bool operator<(const Pt &o) const {
        if (k != o.k) return k < o.k;
        if (j != o.j) return j < o.j;
        return i < o.i;
    }
    bool operator==(const Pt &o) const { return k == o.k && j == o.j && i == o.i; }
    Pt shifted(int dim, int by) const {  // dim: 0=k 1=j 2=i
        Pt p = *this;
        (dim == 0 ? p.k : dim == 1 ? p.j : p.i) += by;
        return p;
    }
    int at(int dim) const { return dim == 0 ? k : dim == 1 ? j : i; }
};

// Sorted association list: the iteration order (lexicographic k, j, i) is part of
// the contract -- it is the summation order of the gold expression.
class CoefTable {
public:
    std::vector<std::pair<Pt, double>> v;
    int find(const Pt &p) const {
        auto it = std::lower_bound(v.begin(), v.end(), p, [](const std::pair<Pt, double> &a, const Pt &b) { return a.first < b; });
        return (it != v.end() && it->first == p) ? int(it - v.begin()) : -1;
    }
    bool has(const Pt &p) const { return find(p) >= 0; }
    double get(const Pt &p) const { int x = find(p); return x < 0 ? 0.0 : v[x].second; }
    void put(const Pt &p, double c, bool accumulate) {
        auto it = std::lower_bound(v.begin(), v.end(), p, [](const std::pair<Pt, double> &a, const Pt &b) { return a.first < b; });
        if (it != v.end() && it->first == p) { if (accumulate) it->second += c; else it->second = c; }
        else v.insert(it, std::make_pair(p, c));
    }
    size_t size() const { return v.size(); }
};

struct PtSet {
    std::vector<Pt> v;  // kept sorted
    bool has(const Pt &p) const { return std::binary_search(v.begin(), v.end(), p); }
    void add(const Pt &p) { auto it = std::lower_bound(v.begin(), v.end(), p); if (it == v.end() || !(*it == p)) v.insert(it, p); }
    size_t size() const { return v.size(); }
    void clear() { v.clear(); }
};

// Coefficient as it appears in emitted source: 6 significant digits.
inline std::string coef_text(double c) {
    char buf[64];
    snprintf(buf, sizeof buf, "%g", c);
    return buf;
}
inline double coef_rounded(double c) { return strtod(coef_text(c).c_str(), nullptr); }

enum ReuseStatus { REUSE_OK = 0, REUSE_NONE = 1 };

struct Stencil {
    int ndim = 3;
    int L = 1, M = 0, N = 0;
    int iterations = 0;           // reference leaves it uninitialised when the .stc has no `iterations`
    bool iterations_set = false;
    CoefTable pts;                // after fuse(): the fused table (unrounded doubles)
    CoefTable base;               // the table as read from the .stc (one time step)
    int step = 1;
    int halo = 0;                 // "order"
    int dist = 0;
    PtSet fwd_k, fwd_j, fwd_i, bwd;
    int lo = 1, hi = -1;          // extent of the streamed offsets over all reuse sets

    int outer_dim() const { return ndim == 3 ? 0 : 1; }
    int range() const { return hi - lo + 1; }

    // 0 ok, 1 cannot open
    int read_stc(const std::string &path) {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return 1;
        char tok[256];
        while (fscanf(f, "%255s", tok) == 1) {
            std::string t(tok);
            if (ndim == 3 && t == "L") { if (fscanf(f, "%d", &L) != 1) break; }
            else if (t == "M") { if (fscanf(f, "%d", &M) != 1) break; }
            else if (t == "N") { if (fscanf(f, "%d", &N) != 1) break; }
            else if (t == "iterations") { if (fscanf(f, "%d", &iterations) != 1) break; iterations_set = true; }
            else if (t == "stencil") {
                for (;;) {
                    Pt p; double c; bool ok;
                    if (ndim == 3) ok = fscanf(f, "%d %d %d %lf", &p.k, &p.j, &p.i, &c) == 4;
                    else ok = fscanf(f, "%d %d %lf", &p.j, &p.i, &c) == 3;
                    if (!ok) break;
                    pts.put(p, c, false);   // duplicate offsets: last one wins
                }
                break;  // `stencil` is the last section (the reference spins forever otherwise)
            }
        }
        fclose(f);
        return 0;
    }

    // stencil <- stencil convolved with itself `s` times; depth-first in table order.
    void fuse(int s) {
        step = s;
        base = pts;
        CoefTable out;
        fuse_walk(out, Pt(), 1.0, s);
        pts = out;
    }

    void choose_halo_dist(int dist_opt) {
        int high = 0, low = 0, od = outer_dim();
        for (auto &e : pts.v) { high = std::max(high, e.first.at(od)); low = std::min(low, e.first.at(od)); }
        halo = high;
        dist = dist_opt != 0 ? dist_opt : ((high - low) >> 1);
    }

    // Forward/backward split used by the data-reusing schedule.  A point p is
    // "forward along d" when p - dist*e_d is also a stencil point.
    ReuseStatus partition_reuse(int merge_forward) {
        fwd_k.clear(); fwd_j.clear(); fwd_i.clear(); bwd.clear();
        PtSet ck, cj, ci, done;
        for (auto &e : pts.v) {
            const Pt &p = e.first;
            if (ndim == 3 && pts.has(p.shifted(0, -dist))) ck.add(p);
            if (pts.has(p.shifted(1, -dist))) cj.add(p);
            if (pts.has(p.shifted(2, -dist))) ci.add(p);
        }
        for (auto &p : ck.v) { fwd_k.add(p); done.add(p.shifted(0, -dist)); }
        for (auto &p : cj.v) { if (done.has(p.shifted(1, -dist))) continue; fwd_j.add(p); done.add(p.shifted(1, -dist)); }
        for (auto &p : ci.v) { if (done.has(p.shifted(2, -dist))) continue; fwd_i.add(p); done.add(p.shifted(2, -dist)); }
        for (auto &e : pts.v) if (!done.has(e.first)) { bwd.add(e.first); done.add(e.first); }
        const PtSet &primary = (ndim == 3) ? fwd_k : fwd_j;
        if (primary.size() == 0) return REUSE_NONE;
        // fold small forward sets back (the primary one is never folded)
        if (ndim == 3 && (int)fwd_j.size() < merge_forward) { for (auto &p : fwd_j.v) bwd.add(p.shifted(1, -dist)); fwd_j.clear(); }
        if ((int)fwd_i.size() < merge_forward) { for (auto &p : fwd_i.v) bwd.add(p.shifted(2, -dist)); fwd_i.clear(); }
        return REUSE_OK;
    }

    void stream_range() {
        lo = 1; hi = -1;
        int od = outer_dim();
        for (const PtSet *s : { &fwd_k, &fwd_j, &fwd_i, &bwd })
            for (auto &p : s->v) { lo = std::min(lo, p.at(od)); hi = std::max(hi, p.at(od)); }
    }

    // number of kernel launches of the timed ping-pong loop (codegen.hpp:581-584)
    int launches() const { int n = 0; for (int t = 0; t < iterations; t += 2 * step) n += 2; return n; }

private:
    void fuse_walk(CoefTable &out, Pt at, double c, int depth) const {
        if (depth == 0) { out.put(at, c, true); return; }
        for (auto &e : pts.v) {
            Pt n; n.k = at.k + e.first.k; n.j = at.j + e.first.j; n.i = at.i + e.first.i;
            fuse_walk(out, n, c * e.second, depth - 1);
        }
    }
};

}  // namespace drs
