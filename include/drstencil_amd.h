/*
 * drstencil_amd.h -- C ABI of libdrstencil_amd.so (MI355X-native DRStencil generator + runtime).
 *
 * The reference (simple86/DRStencil) has NO library/FFI surface: its boundary is the
 * `drstencil` command line, the .stc file format and the emitted program.
 * Each entry point below names the reference interface it stands for; INTEGRATION.md shows
 * how a maintainer binds it.  Plain C types only; device buffers are raw device pointers
 * (e.g. torch.Tensor.data_ptr()), streams are hipStream_t passed as void*.
 */
#ifndef DRSTENCIL_AMD_H
#define DRSTENCIL_AMD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct drs_spec drs_spec;     /* parsed + fused stencil, reuse analysis done */
typedef struct drs_kernel drs_kernel; /* generated, compiled (hipcc, gfx950) and loaded kernel */

/* ---- generator: the `drstencil [options] file.stc` command as a function.
 * Replaces main() of main.cpp:10-280 (option scan 118-230, pipeline 237-278).
 * argv excludes the program name.  Returns the process exit code the command would have
 * (0, 1 "No data to reuse", 255 "Illegal input."/"Invalid configuration!"/"Error opening
 * stencil file.").  *source (emitted HIP text, NULL if nothing was emitted) and *messages
 * (the command's stdout text, followed by what it prints on stderr: remarks such as "the tuner's configuration ... is used") are malloc'ed; release with drs_free.  No file is written. */
int drs_generate(int argc, const char *const *argv, char **source, char **messages);
void drs_free(void *p);

/* ---- stencil IR: DRStencil / DRStencil_2d (drstencil.hpp:15-49, drstencil_2d.hpp).
 * drs_spec_open = get_stencil (52-78) + fusing (262-282) + dataReuse (306-311).
 * *status: 0 ok, 1 cannot open the file, 2 "No data to reuse" (spec still returned). */
drs_spec *drs_spec_open(const char *stc_path, int ndim, int step, int dist, int merge_forward, int *status);
void drs_spec_close(drs_spec *s);
int drs_spec_halo(const drs_spec *s);        /* get_order()    drstencil.hpp:34 */
int drs_spec_dist(const drs_spec *s);        /* get_distance() drstencil.hpp:35 */
int drs_spec_range(const drs_spec *s);       /* high_k - low_k + 1, codegen.hpp:89 */
int drs_spec_npoints(const drs_spec *s);     /* fused stencil size */
int drs_spec_iterations(const drs_spec *s);  /* get_problem_size, drstencil.hpp:80-86 */
int drs_spec_launches(const drs_spec *s);    /* kernel launches of the timed loop, codegen.hpp:581-584 */
void drs_spec_dims(const drs_spec *s, int *L, int *M, int *N);
/* point idx in gold (lexicographic) order; coef_text = the 6-significant-digit literal
 * the emitted source carries (drstencil.hpp:192), at least 32 bytes */
int drs_spec_point(const drs_spec *s, int idx, int *k, int *j, int *i, double *coef, char *coef_text);
/* sizes of forward_k, forward_j, forward_i, backward (partition, drstencil.hpp:198-259) */
void drs_spec_partition(const drs_spec *s, int sizes[4]);

/* ---- runtime: what the emitted program's main() does (host_code_gen, codegen.hpp:547-635),
 * callable on caller-owned device buffers.
 * drs_kernel_build: generate (same argv as drs_generate), compile with hipcc for gfx950 into
 * cache_dir (NULL: <package>/_kcache) unless already cached, and load.  NULL on failure with
 * *log (malloc'ed, may be NULL) holding the generator messages / compiler output. */
drs_kernel *drs_kernel_build(int argc, const char *const *argv, const char *cache_dir, char **log);
void drs_kernel_close(drs_kernel *k);                /* frees the handle; the plugin stays mapped (safe at process exit) */
/* drs_kernel_close + the plugin is unloaded (hipDeviceSynchronize, then dlclose: its code object leaves the HIP runtime).
 * For long sweeps that load thousands of kernels (the tuner: benchmarks/3d7pt_star/tuning.py:102-142 starts one process
 * per configuration instead).  Not to be called while one of its launches may still be running on another device. */
int drs_kernel_unload(drs_kernel *k);
/* JSON: dims, dtype, halo, step, grid, lds ... and which arithmetic the kernel computes:
 *   "arithmetic": "gold-order"    the reference's fused sum (drstencil.hpp:182-196, 262-282) as one FMA chain -- bit-identical
 *                                  to gold_<name> for any iteration count ("tolerance_horizon_iterations": -1);
 *                 "reassociated"  on-chip time steps (--temporal 1 / force): equal to it up to rounding; the generator emits
 *                                  such a kernel only where its drift estimate ("drift_estimate", relative, for the spec's
 *                                  iterations) stays within 1e-6 (fp32) / 1e-12 (fp64), "tolerance_horizon_iterations" is the
 *                                  largest iteration count for which it does, and "temporal_forced": 1 marks --temporal force.
 *   "out_skew_bytes", "placement_period_bytes": where the output array should sit relative to the input array, modulo the period
 *                                  (64 MiB): see drs_kernel_pair_layout below. */
const char *drs_kernel_info(const drs_kernel *k);
const char *drs_kernel_path(const drs_kernel *k);   /* the loaded shared object */
/* JSON: vgprs, agprs, sgprs, scratch_bytes_per_lane, sgpr_spill, vgpr_spill, occupancy_waves_per_simd, lds_bytes of
 * dr_<name> (with --pair-launch: the maximum over dr_<name> and dr2_<name>) as reported by hipcc -- what the reference reads
 * from `ncu --set full`, getGpuMetrics.py:9 -- plus "verified": 1 when every field was found in the compiler's report.
 * drs_kernel_build refuses (NULL + log): a kernel that spills to scratch or spills scalar registers (sgpr_spill > 0;
 * DRS_ALLOW_SCRATCH=1 overrides either), a kernel whose
 * report could not be read (DRS_ALLOW_UNVERIFIED=1), a cache miss once this process has launched a kernel or when
 * DRS_NO_COMPILE=1 (hipcc is a child process: build first, launch afterwards), and --debug-drop-barrier kernels
 * (wrong results by design) without DRS_EXPERIMENTS=1. */
const char *drs_kernel_resources(const drs_kernel *k);
/* Where to put the two arrays.  The reference's host code makes two cudaMalloc calls (codegen.hpp:556-566); on MI355X the
 * launch time of a z-streaming kernel depends on (out - in) mod 64 MiB -- up to 14 % for the 1024^3 step-2 kernel
 * (profiles/r03_probe_skew4.log) -- so the emitted program and bench.py carve both arrays out of ONE allocation:
 * in = arena, out = arena + *out_offset (a multiple of the 64 MiB period that clears the array + the kernel's "out_skew_bytes"),
 * arena of *arena_bytes.  Advice only: every entry point accepts any two device pointers, results never depend on them. */
int drs_kernel_pair_layout(const drs_kernel *k, size_t *arena_bytes, size_t *out_offset);
/* one launch of dr_<name><<<grid, block, 0, stream>>>(in, out): codegen.hpp:577,582-583 */
int drs_kernel_launch(drs_kernel *k, const void *d_in, void *d_out, void *stream);
/* one launch of dr2_<name>: the same sweep over TWO (in, out) pairs (kernels generated with --pair-launch 1; -2 otherwise).
 * No reference counterpart: the two boundary views of a slab-decomposed run (drstencil_amd/multigpu.py) in one launch. */
int drs_kernel_launch_pair(drs_kernel *k, const void *d_in0, void *d_out0, const void *d_in1, void *d_out1, void *stream);
/* one launch of gold_<name> (the reference's verification kernel, codegen.hpp:611-612) */
int drs_kernel_launch_gold(drs_kernel *k, const void *d_in, void *d_out, void *stream);
/* the timed ping-pong loop: for (t = 0; t < iterations; t += 2*step) { k(A,B); k(B,A); }
 * (codegen.hpp:581-584).  gold != 0 runs gold_<name> instead.  Returns the number of
 * launches, -1 on a HIP error, or -3 when `iterations` exceeds the tolerance horizon of a reassociated (temporal) kernel
 * that was not built with --temporal force: build the fused kernel for such a run.  Asynchronous on `stream`. */
int drs_kernel_run(drs_kernel *k, void *d_a, void *d_b, int iterations, int gold, void *stream);
/* `warmup` untimed launches (A,B) (codegen.hpp:575-578), then the loop above bracketed by
 * HIP events recorded on `stream`; blocks until done; *ms = elapsed milliseconds. */
int drs_kernel_run_timed(drs_kernel *k, void *d_a, void *d_b, int iterations, int warmup, void *stream, float *ms);

/* ---- N > 1: one rank of a slab-decomposed run (z slabs in 3D, y slabs in 2D), one process per GPU -----------------------------
 * No reference counterpart: the reference is single-GPU (no cudaSetDevice / streams / NCCL anywhere; SURVEY.md section 5 sketches
 * this layer as `ncclGroupStart; ncclSend/ncclRecv x <= 4; ncclGroupEnd` on a comm stream).  What these entry points must
 * reproduce is the SINGLE-DOMAIN run of drs_kernel_run on the whole grid, bit for bit; drstencil_amd/multigpu.py (SlabPlan /
 * SlabRun, the torch.distributed path) is the reference implementation they are tested against.
 *   rank 0:      drs_slab_unique_id(id)                  -> hand the 128 bytes to every rank (MPI_Bcast, a file, a socket ...)
 *   every rank:  s = drs_slab_open(argc, argv, ..., world, rank, every, 0, cache, &log)   plan + kernels, BEFORE any GPU call
 *                                                          (it may start hipcc); argv = generator options + the whole grid's .stc
 *                hipSetDevice(local_rank); drs_slab_connect(s, id, stream)    ncclCommInitRank, side stream, events
 *                drs_slab_plan(s, p): this rank holds global planes [p[0], p[1]) = Lloc planes INCLUDING G ghost planes per
 *                                     interior face, owns [p[2], p[3]); A and B are caller-owned device buffers of that size
 *                n = drs_slab_run(s, d_a, d_b, iterations)                    the reference's loop (codegen.hpp:581-584) on the slab
 *                drs_slab_sync(s); drs_slab_close(s)
 * every = 1: each launch exchanges its output's H = step * order boundary planes with the <= 2 neighbours; every = 2: ghost
 * planes 2H wide, one exchange per ping-pong pair.  A pair is captured once into a HIP graph per (A, B) and replayed
 * (DRS_SLAB_GRAPH=0: eager); drs_slab_info tells ("graph": 1 captured, -1 refused -> eager).  rehearse_world > 0 plays rank
 * `rank` of `rehearse_world` on ONE GPU with itself as both neighbours (communicator of size 1): tests and one-GPU rehearsals.
 * drs_slab_run answers -1 on an error (drs_slab_error) and -3, like drs_kernel_run, when the slab was opened with `--temporal 1`
 * options (on-chip stages, reassociated arithmetic) and `iterations` exceeds the kernels' tolerance horizon. */
typedef struct drs_slab drs_slab;
#define DRS_SLAB_ID_BYTES 128
int drs_slab_unique_id(void *id128);
drs_slab *drs_slab_open(int argc, const char *const *argv, int alone_argc, const char *const *alone_argv, int world, int rank, int every,
                        int rehearse_world, const char *cache_dir, char **log);
void drs_slab_plan(const drs_slab *s, long out[8]);   /* lo, hi, z0, z1, Lloc, G, H, every */
int drs_slab_connect(drs_slab *s, const void *id128, void *main_stream /* hipStream_t or NULL: own stream */);
int drs_slab_run(drs_slab *s, void *d_a, void *d_b, int iterations /* < 0: the spec's */);
int drs_slab_sync(drs_slab *s);
void *drs_slab_stream(drs_slab *s);
const char *drs_slab_info(drs_slab *s);     /* JSON */
const char *drs_slab_error(const drs_slab *s);
void drs_slab_close(drs_slab *s);

/* ---- inputs and error metric: common.hpp:9-102 (host memory) */
void drs_fill_random_f64(double *a, size_t n, unsigned seed);   /* seed 1 == the reference's unseeded rand() */
void drs_fill_random_f32(float *a, size_t n, unsigned seed);
double drs_check_error_f64(int ndim, int L, int M, int N, int halo, const double *out, const double *ref,
                           double *max_abs, long *max_idx, double *max_rel);
double drs_check_error_f32(int ndim, int L, int M, int N, int halo, const float *out, const float *ref,
                           double *max_abs, long *max_idx, double *max_rel);

const char *drs_version(void);

#ifdef __cplusplus
}
#endif
#endif
