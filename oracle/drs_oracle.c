/*
 * drs_oracle.c -- CPU restatement of the DRStencil reference semantics.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity *checker* for the HIP
 * product path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under drstencil_amd/ links, imports
 * or calls it, and the product path never falls back to it.
 *
 * Parity is PINNED: tests/golden/ holds arrays produced by the reference's own
 * generator (oracle/_ref/drstencil_ref, built from /root/reference/main.cpp by
 * oracle/Makefile) whose emitted gold_<name> statement was compiled with
 * `g++ -O0` by oracle/make_golden.py; tests/test_oracle_golden.py checks this
 * restatement bit-for-bit (fp64, contraction off) against those arrays and
 * against the known-answer sums recorded in SURVEY.md section 8(c).
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * it restates.  The code is written from the behaviour, not copied: plain C
 * arrays instead of std::map/std::set, one implementation for 2D and 3D
 * (a 2D spec is stored with k == 0 and L == 1).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DRSO_MAXP 4096

typedef struct {
    int ndim;              /* 2 or 3 */
    int L, M, N;           /* grid (L == 1 in 2D) */
    int iterations;        /* -1 when the .stc never set it (reference: uninitialised) */
    int npts;
    int off[DRSO_MAXP][3]; /* (k, j, i), sorted lexicographically == std::map order */
    double coef[DRSO_MAXP];
    int halo;              /* "order": max k (3D) / max j (2D) offset, set by drso_fuse */
    int step;
} drso_spec;

/* ---- sorted point table (stands in for std::map<tuple<int...>, double>) ---- */

static int cmp_off(const int *a, const int *b)
{
    for (int d = 0; d < 3; d++) {
        if (a[d] < b[d]) return -1;
        if (a[d] > b[d]) return 1;
    }
    return 0;
}

/* find index of key or insertion position; *found tells which */
static int tbl_find(const drso_spec *s, const int *key, int *found)
{
    int lo = 0, hi = s->npts;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        int c = cmp_off(s->off[mid], key);
        if (c == 0) { *found = 1; return mid; }
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    *found = 0;
    return lo;
}

static int tbl_insert(drso_spec *s, const int *key, double v, int accumulate)
{
    int found;
    int pos = tbl_find(s, key, &found);
    if (found) {
        if (accumulate) s->coef[pos] += v;  /* drstencil.hpp:265-266 */
        else s->coef[pos] = v;              /* drstencil.hpp:71: last one wins */
        return 0;
    }
    if (s->npts >= DRSO_MAXP) return -1;
    memmove(&s->off[pos + 1], &s->off[pos], sizeof(s->off[0]) * (size_t)(s->npts - pos));
    memmove(&s->coef[pos + 1], &s->coef[pos], sizeof(double) * (size_t)(s->npts - pos));
    memcpy(s->off[pos], key, sizeof(int) * 3);
    s->coef[pos] = v;
    s->npts++;
    return 0;
}

/* ---- .stc parser: drstencil.hpp:52-78 (3D), drstencil_2d.hpp:48-73 (2D) ----
 * Whitespace-separated tokens.  `L`/`M`/`N`/`iterations` take an int; `stencil`
 * starts the point list (k j i coef, or j i coef) which runs until a record
 * fails to parse; every other token is skipped.  The reference then spins
 * forever if anything but EOF stopped the list; we stop there instead. */
int drso_parse_stc(const char *path, int ndim, drso_spec *s)
{
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    memset(s, 0, sizeof(*s));
    s->ndim = ndim;
    s->L = 1; s->M = 0; s->N = 0;
    s->iterations = -1;
    s->step = 1;
    char tok[256];
    while (fscanf(f, "%255s", tok) == 1) {
        if (ndim == 3 && strcmp(tok, "L") == 0) { if (fscanf(f, "%d", &s->L) != 1) break; }
        else if (strcmp(tok, "M") == 0) { if (fscanf(f, "%d", &s->M) != 1) break; }
        else if (strcmp(tok, "N") == 0) { if (fscanf(f, "%d", &s->N) != 1) break; }
        else if (strcmp(tok, "iterations") == 0) { if (fscanf(f, "%d", &s->iterations) != 1) break; }
        else if (strcmp(tok, "stencil") == 0) {
            for (;;) {
                int key[3] = {0, 0, 0};
                double c;
                int ok;
                if (ndim == 3) ok = fscanf(f, "%d %d %d %lf", &key[0], &key[1], &key[2], &c) == 4;
                else ok = fscanf(f, "%d %d %lf", &key[1], &key[2], &c) == 3;
                if (!ok) break;
                if (tbl_insert(s, key, c, 0) != 0) { fclose(f); return 2; }
            }
            break;
        }
    }
    fclose(f);
    return 0;
}

/* ---- coefficient text round trip: drstencil.hpp:192 streams a double with the
 * default ostream precision (6 significant digits, %g), and the emitted source
 * is then re-read by the device compiler. */
static double round6(double c)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%g", c);
    return strtod(buf, NULL);
}

/* ---- fusion: drstencil.hpp:262-282, drstencil_2d.hpp:231-251 ----
 * Depth-first self-convolution in table order; products accumulate along the
 * path, sums accumulate in visit order.  Afterwards `order` (Halo) is the
 * largest k offset (3D) / j offset (2D): drstencil.hpp:88-98. */
static void fuse_rec(const drso_spec *base, drso_spec *acc, const int *at, double c, int depth)
{
    if (depth == 0) { tbl_insert(acc, at, c, 1); return; }
    for (int p = 0; p < base->npts; p++) {
        int nxt[3] = { at[0] + base->off[p][0], at[1] + base->off[p][1], at[2] + base->off[p][2] };
        fuse_rec(base, acc, nxt, c * base->coef[p], depth - 1);
    }
}

int drso_fuse(drso_spec *s, int step)
{
    drso_spec *acc = (drso_spec *)calloc(1, sizeof(drso_spec));
    if (!acc) return -1;
    int origin[3] = {0, 0, 0};
    fuse_rec(s, acc, origin, 1.0, step);
    s->npts = acc->npts;
    memcpy(s->off, acc->off, sizeof(s->off));
    for (int p = 0; p < acc->npts; p++) s->coef[p] = round6(acc->coef[p]);
    free(acc);
    s->step = step;
    int outer = (s->ndim == 3) ? 0 : 1;
    int high = 0;
    for (int p = 0; p < s->npts; p++) if (s->off[p][outer] > high) high = s->off[p][outer];
    s->halo = high;
    return 0;
}

/* launches in the timed ping-pong loop: codegen.hpp:581-584 */
int drso_launches(int iterations, int step)
{
    int n = 0;
    for (int t = 0; t < iterations; t += 2 * step) n += 2;
    return n;
}

/* ---- inputs: common.hpp:9-45.  rand()/(RAND_MAX-1), row-major, glibc rand(). */
void drso_srand(unsigned seed) { srand(seed); }

void drso_fill_random_f64(double *a, size_t n)
{
    for (size_t i = 0; i < n; i++) a[i] = (double)rand() / (double)(RAND_MAX - 1);
}

void drso_fill_random_f32(float *a, size_t n)
{
    for (size_t i = 0; i < n; i++) a[i] = (float)((double)rand() / (double)(RAND_MAX - 1));
}

/* ---- one launch of gold_<name>: codegen.hpp:637-660, codegen_2d.hpp:666-688,
 * body from drstencil.hpp:182-196: out[x] = (c0)*in[x+p0] + (c1)*in[x+p1] + ...
 * evaluated left to right in table order on the interior [Halo, dim-Halo) of
 * every dim; nothing outside it is written.
 *   contract == 0: every product and sum rounded separately (g++ -O0 on x86-64,
 *                  the arithmetic the golden fixtures were produced with); this
 *                  file MUST be built with -ffp-contract=off (oracle/Makefile);
 *   contract == 1: t = c0*a0; t = fma(ci, ai, t) -- what a device compiler with
 *                  FMA contraction makes of the same left-to-right expression,
 *                  and what the HIP kernels compute. */
/* Implementation notes (round 3: the sweep is also bench.py's cpu_baseline, so it should not be a strawman):
 *   * rows are processed in blocks of DRSO_BLK points; inside a block the TAP loop is the outer one and the point loop the
 *     inner one, so the compiler vectorises ACROSS points (AVX-512 / AVX2 FMA) while every point still runs its own chain
 *     t = c0*a0; t = fma(ci, ai, t) in table order -- the rounding sequence of each output is exactly the scalar one, and
 *     the golden-fixture tests stay bit-exact;
 *   * target_clones: the .so is built in one container and runs on the GPU box's host (2 x EPYC 9575F), so the ISA is picked
 *     at load time (avx512f / avx2+fma / baseline), never by -march=native;
 *   * OpenMP over (k, j) rows, static schedule -- drso_first_touch_* uses the same schedule so that on a two-socket box
 *     every thread's rows live on its own NUMA node. */
#define DRSO_BLK 128
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define DRSO_CLONES __attribute__((target_clones("avx512f", "arch=haswell", "default")))
#else
#define DRSO_CLONES
#endif

/* AVX-512 fast path of the contracted sweep (contract == 1: what the HIP kernels compute and what bench.py's cpu_baseline times):
 * 128 (fp32) / 64 (fp64) points per block live in EIGHT zmm accumulators for the whole tap loop -- the generic path above keeps them
 * in a stack array and pays two loads and a store per vector FMA.  Same arithmetic, lane for lane: t = c0*a0, then t = fma(ci, ai, t)
 * in table order (an IEEE fused multiply-add per lane), so the results are bit-identical to the generic path (test_oracle_golden.py
 * compares the two on every fixture).  Picked at run time when the CPU has avx512f. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#include <immintrin.h>
#define DRSO_HAVE_AVX512_PATH 1
__attribute__((target("avx512f")))
static void sweep_rows_avx512_f32(const drso_spec *s, const float *src, float *dst, const long *doff, const float *cf)
{
    const int H = s->halo, L = s->L, M = s->M, N = s->N, np = s->npts;
    const int klo = (s->ndim == 3) ? H : 0, khi = (s->ndim == 3) ? L - H : 1;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int k = klo; k < khi; k++)
        for (int j = H; j < M - H; j++) {
            const size_t row = ((size_t)k * M + j) * N;
            int i0 = H;
            for (; i0 + 128 <= N - H; i0 += 128) {
                const float *c = src + row + i0;
                __m512 w = _mm512_set1_ps(cf[0]);
                const float *a = c + doff[0];
                __m512 t0 = _mm512_mul_ps(w, _mm512_loadu_ps(a)), t1 = _mm512_mul_ps(w, _mm512_loadu_ps(a + 16)), t2 = _mm512_mul_ps(w, _mm512_loadu_ps(a + 32)),
                       t3 = _mm512_mul_ps(w, _mm512_loadu_ps(a + 48)), t4 = _mm512_mul_ps(w, _mm512_loadu_ps(a + 64)), t5 = _mm512_mul_ps(w, _mm512_loadu_ps(a + 80)),
                       t6 = _mm512_mul_ps(w, _mm512_loadu_ps(a + 96)), t7 = _mm512_mul_ps(w, _mm512_loadu_ps(a + 112));
                for (int p = 1; p < np; p++) {
                    w = _mm512_set1_ps(cf[p]);
                    a = c + doff[p];
                    t0 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a), t0); t1 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a + 16), t1);
                    t2 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a + 32), t2); t3 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a + 48), t3);
                    t4 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a + 64), t4); t5 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a + 80), t5);
                    t6 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a + 96), t6); t7 = _mm512_fmadd_ps(w, _mm512_loadu_ps(a + 112), t7);
                }
                float *d = dst + row + i0;
                _mm512_storeu_ps(d, t0); _mm512_storeu_ps(d + 16, t1); _mm512_storeu_ps(d + 32, t2); _mm512_storeu_ps(d + 48, t3);
                _mm512_storeu_ps(d + 64, t4); _mm512_storeu_ps(d + 80, t5); _mm512_storeu_ps(d + 96, t6); _mm512_storeu_ps(d + 112, t7);
            }
            for (; i0 < N - H; i0 += 16) {                       /* the row's tail, 16 points at a time under a mask */
                const int n = N - H - i0 < 16 ? N - H - i0 : 16;
                const __mmask16 m = (__mmask16)((1u << n) - 1u);
                const float *c = src + row + i0;
                __m512 t = _mm512_mul_ps(_mm512_set1_ps(cf[0]), _mm512_maskz_loadu_ps(m, c + doff[0]));
                for (int p = 1; p < np; p++) t = _mm512_fmadd_ps(_mm512_set1_ps(cf[p]), _mm512_maskz_loadu_ps(m, c + doff[p]), t);
                _mm512_mask_storeu_ps(dst + row + i0, m, t);
            }
        }
}
__attribute__((target("avx512f")))
static void sweep_rows_avx512_f64(const drso_spec *s, const double *src, double *dst, const long *doff, const double *cf)
{
    const int H = s->halo, L = s->L, M = s->M, N = s->N, np = s->npts;
    const int klo = (s->ndim == 3) ? H : 0, khi = (s->ndim == 3) ? L - H : 1;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int k = klo; k < khi; k++)
        for (int j = H; j < M - H; j++) {
            const size_t row = ((size_t)k * M + j) * N;
            int i0 = H;
            for (; i0 + 64 <= N - H; i0 += 64) {
                const double *c = src + row + i0;
                __m512d w = _mm512_set1_pd(cf[0]);
                const double *a = c + doff[0];
                __m512d t0 = _mm512_mul_pd(w, _mm512_loadu_pd(a)), t1 = _mm512_mul_pd(w, _mm512_loadu_pd(a + 8)), t2 = _mm512_mul_pd(w, _mm512_loadu_pd(a + 16)),
                        t3 = _mm512_mul_pd(w, _mm512_loadu_pd(a + 24)), t4 = _mm512_mul_pd(w, _mm512_loadu_pd(a + 32)), t5 = _mm512_mul_pd(w, _mm512_loadu_pd(a + 40)),
                        t6 = _mm512_mul_pd(w, _mm512_loadu_pd(a + 48)), t7 = _mm512_mul_pd(w, _mm512_loadu_pd(a + 56));
                for (int p = 1; p < np; p++) {
                    w = _mm512_set1_pd(cf[p]);
                    a = c + doff[p];
                    t0 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a), t0); t1 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a + 8), t1);
                    t2 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a + 16), t2); t3 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a + 24), t3);
                    t4 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a + 32), t4); t5 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a + 40), t5);
                    t6 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a + 48), t6); t7 = _mm512_fmadd_pd(w, _mm512_loadu_pd(a + 56), t7);
                }
                double *d = dst + row + i0;
                _mm512_storeu_pd(d, t0); _mm512_storeu_pd(d + 8, t1); _mm512_storeu_pd(d + 16, t2); _mm512_storeu_pd(d + 24, t3);
                _mm512_storeu_pd(d + 32, t4); _mm512_storeu_pd(d + 40, t5); _mm512_storeu_pd(d + 48, t6); _mm512_storeu_pd(d + 56, t7);
            }
            for (; i0 < N - H; i0 += 8) {
                const int n = N - H - i0 < 8 ? N - H - i0 : 8;
                const __mmask8 m = (__mmask8)((1u << n) - 1u);
                const double *c = src + row + i0;
                __m512d t = _mm512_mul_pd(_mm512_set1_pd(cf[0]), _mm512_maskz_loadu_pd(m, c + doff[0]));
                for (int p = 1; p < np; p++) t = _mm512_fmadd_pd(_mm512_set1_pd(cf[p]), _mm512_maskz_loadu_pd(m, c + doff[p]), t);
                _mm512_mask_storeu_pd(dst + row + i0, m, t);
            }
        }
}
static int g_avx512_path = -1;     /* -1 unknown, 0 off (DRSO_NO_AVX512_PATH=1 or no avx512f), 1 on */
static int use_avx512_path(void)
{
    if (g_avx512_path < 0) {
        __builtin_cpu_init();
        const char *off = getenv("DRSO_NO_AVX512_PATH");
        g_avx512_path = (__builtin_cpu_supports("avx512f") && !(off && off[0] == '1')) ? 1 : 0;
    }
    return g_avx512_path;
}
#endif
void drso_set_avx512_path(int on)
{
#ifdef DRSO_HAVE_AVX512_PATH
    __builtin_cpu_init();
    g_avx512_path = (on && __builtin_cpu_supports("avx512f")) ? 1 : 0;
#else
    (void)on;
#endif
}

#define DEF_SWEEP(NAME, T, FMA, FAST)                                                        \
DRSO_CLONES                                                                              \
void NAME(const drso_spec *s, const T *src, T *dst, int contract)                       \
{                                                                                        \
    const int H = s->halo, L = s->L, M = s->M, N = s->N, np = s->npts;                   \
    const int klo = (s->ndim == 3) ? H : 0, khi = (s->ndim == 3) ? L - H : 1;            \
    long *doff = (long *)malloc(sizeof(long) * (size_t)np);                              \
    T *cf = (T *)malloc(sizeof(T) * (size_t)np);                                         \
    for (int p = 0; p < np; p++) {                                                       \
        doff[p] = ((long)s->off[p][0] * M + s->off[p][1]) * N + s->off[p][2];            \
        cf[p] = (T)s->coef[p];                                                           \
    }                                                                                    \
    if (contract && FAST(s, src, dst, doff, cf)) { free(doff); free(cf); return; }       \
    _Pragma("omp parallel for collapse(2) schedule(static)")                             \
    for (int k = klo; k < khi; k++)                                                      \
        for (int j = H; j < M - H; j++) {                                                \
            const size_t row = ((size_t)k * M + j) * N;                                  \
            for (int i0 = H; i0 < N - H; i0 += DRSO_BLK) {                               \
                const int n = (N - H - i0 < DRSO_BLK) ? N - H - i0 : DRSO_BLK;           \
                const T *c = src + row + i0;                                             \
                T t[DRSO_BLK] __attribute__((aligned(64)));                                \
                {                                                                        \
                    const T *a = c + doff[0]; const T w = cf[0];                         \
                    for (int x = 0; x < n; x++) t[x] = w * a[x];                         \
                }                                                                        \
                if (contract) {                                                          \
                    for (int p = 1; p < np; p++) {                                       \
                        const T *a = c + doff[p]; const T w = cf[p];                     \
                        for (int x = 0; x < n; x++) t[x] = FMA(w, a[x], t[x]);           \
                    }                                                                    \
                } else {                                                                 \
                    for (int p = 1; p < np; p++) {                                       \
                        const T *a = c + doff[p]; const T w = cf[p];                     \
                        for (int x = 0; x < n; x++) t[x] = t[x] + w * a[x];              \
                    }                                                                    \
                }                                                                        \
                memcpy(dst + row + i0, t, sizeof(T) * (size_t)n);                        \
            }                                                                            \
        }                                                                                \
    free(doff); free(cf);                                                                \
}

#ifdef DRSO_HAVE_AVX512_PATH
static int fast_f64(const drso_spec *s, const double *src, double *dst, const long *doff, const double *cf) { if (!use_avx512_path()) return 0; sweep_rows_avx512_f64(s, src, dst, doff, cf); return 1; }
static int fast_f32(const drso_spec *s, const float *src, float *dst, const long *doff, const float *cf) { if (!use_avx512_path()) return 0; sweep_rows_avx512_f32(s, src, dst, doff, cf); return 1; }
#else
#define fast_f64(s, a, b, c, d) 0
#define fast_f32(s, a, b, c, d) 0
#endif
DEF_SWEEP(drso_sweep_f64, double, __builtin_fma, fast_f64)
DEF_SWEEP(drso_sweep_f32, float, __builtin_fmaf, fast_f32)

/* The pages of a freshly allocated (untouched) array are placed on the NUMA node of the thread that writes them first:
 * zero-fill with the sweep's own (k, j) schedule, whole rows including the ring. */
#define DEF_TOUCH(NAME, T)                                                               \
void NAME(const drso_spec *s, T *a)                                                      \
{                                                                                        \
    const int L = (s->ndim == 3) ? s->L : 1, M = s->M, N = s->N;                         \
    _Pragma("omp parallel for collapse(2) schedule(static)")                             \
    for (int k = 0; k < L; k++)                                                          \
        for (int j = 0; j < M; j++)                                                      \
            memset(a + ((size_t)k * M + j) * N, 0, sizeof(T) * (size_t)N);               \
}
DEF_TOUCH(drso_first_touch_f64, double)
DEF_TOUCH(drso_first_touch_f32, float)

/* parallel copy with the same schedule (fills a first-touched array from a host buffer) */
#define DEF_COPY(NAME, T)                                                                \
void NAME(const drso_spec *s, T *dst, const T *src)                                      \
{                                                                                        \
    const int L = (s->ndim == 3) ? s->L : 1, M = s->M, N = s->N;                         \
    _Pragma("omp parallel for collapse(2) schedule(static)")                             \
    for (int k = 0; k < L; k++)                                                          \
        for (int j = 0; j < M; j++)                                                      \
            memcpy(dst + ((size_t)k * M + j) * N, src + ((size_t)k * M + j) * N, sizeof(T) * (size_t)N); \
}
DEF_COPY(drso_copy_f64, double)
DEF_COPY(drso_copy_f32, float)

/* ---- the whole run: codegen.hpp:581-584 / codegen_2d.hpp host loop.
 * for (t = 0; t < Iterations; t += 2*step) { launch(A,B); launch(B,A); }
 * Result is in A; B's ring stays as initialised. Returns the launch count. */
int drso_run_f64(const drso_spec *s, double *A, double *B, int contract)
{
    int n = 0;
    for (int t = 0; t < s->iterations; t += 2 * s->step) {
        drso_sweep_f64(s, A, B, contract);
        drso_sweep_f64(s, B, A, contract);
        n += 2;
    }
    return n;
}

int drso_run_f32(const drso_spec *s, float *A, float *B, int contract)
{
    int n = 0;
    for (int t = 0; t < s->iterations; t += 2 * s->step) {
        drso_sweep_f32(s, A, B, contract);
        drso_sweep_f32(s, B, A, contract);
        n += 2;
    }
    return n;
}

/* ---- error metric: common.hpp:47-102 (+ call sites codegen.hpp:620,
 * codegen_2d.hpp:649).  Over the interior box only; max-abs starts at the
 * reference's 1e-13 floor; returns RMS.  max_rel is ours (the 1e-6 gate):
 * max |out-ref| / max(|ref|, tiny). */
#define DEF_CHECK(NAME, T)                                                               \
double NAME(const drso_spec *s, const T *out, const T *ref,                             \
            double *max_abs, long *max_idx, double *max_rel)                            \
{                                                                                        \
    const int H = s->halo, L = s->L, M = s->M, N = s->N;                                 \
    const int klo = (s->ndim == 3) ? H : 0, khi = (s->ndim == 3) ? L - H : 1;            \
    double err = 0.0, mx = 1e-13, mrel = 0.0;                                            \
    long at = 0;                                                                         \
    for (int k = klo; k < khi; k++)                                                      \
        for (int j = H; j < M - H; j++)                                                  \
            for (int i = H; i < N - H; i++) {                                            \
                size_t x = ((size_t)k * M + j) * N + i;                                  \
                double d = (double)out[x] - (double)ref[x];                              \
                if (d < 0.0) d = -d;                                                     \
                err += d * d;                                                            \
                if (d > mx) { mx = d; at = (long)x; }                                    \
                double r = fabs((double)ref[x]);                                         \
                double rel = d / (r > 1e-30 ? r : 1e-30);                                \
                if (rel > mrel) mrel = rel;                                              \
            }                                                                            \
    double cnt = (double)(khi - klo) * (double)(M - 2 * H) * (double)(N - 2 * H);        \
    if (max_abs) *max_abs = mx;                                                          \
    if (max_idx) *max_idx = at;                                                          \
    if (max_rel) *max_rel = mrel;                                                        \
    return sqrt(err / cnt);                                                              \
}

DEF_CHECK(drso_check_f64, double)
DEF_CHECK(drso_check_f32, float)

/* accessors for ctypes */
size_t drso_spec_size(void) { return sizeof(drso_spec); }
int drso_npts(const drso_spec *s) { return s->npts; }
int drso_halo(const drso_spec *s) { return s->halo; }
int drso_iterations(const drso_spec *s) { return s->iterations; }
void drso_set_iterations(drso_spec *s, int it) { s->iterations = it; }
void drso_set_dims(drso_spec *s, int L, int M, int N) { s->L = (s->ndim == 3) ? L : 1; s->M = M; s->N = N; }
void drso_dims(const drso_spec *s, int *L, int *M, int *N) { *L = s->L; *M = s->M; *N = s->N; }
void drso_point(const drso_spec *s, int p, int *k, int *j, int *i, double *c)
{
    *k = s->off[p][0]; *j = s->off[p][1]; *i = s->off[p][2]; *c = s->coef[p];
}
/* which clone of the sweep the loader picked on this host (reported by bench.py's cpu_baseline) */
const char *drso_isa(void)
{
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) return use_avx512_path() ? "avx512f (register-blocked intrinsics)" : "avx512f";
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return "avx2+fma";
#endif
    return "baseline";
}
/* bench.py's cpu_baseline: use as many threads as this process may really run on (a container's CPU quota is not visible to
 * omp_get_max_threads(): 128 threads on a 16-CPU quota spend their time throttled) */
void drso_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int drso_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
