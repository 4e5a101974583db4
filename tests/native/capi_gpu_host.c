/* capi_gpu_host.c -- the INTEGRATION.md "C host" example as a program: build a kernel through the C ABI, run the
 * reference's warm-up + timed ping-pong loop (codegen.hpp:575-589) on device buffers, run the gold kernel on a second
 * pair, compare with checkError3D semantics (common.hpp:47-102).  usage: capi_gpu_host <3d .stc> <L> <M> <N> <halo> */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "drstencil_amd.h"

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const int L = atoi(argv[2]), M = atoi(argv[3]), N = atoi(argv[4]), H = atoi(argv[5]);
    const size_t n = (size_t)L * M * N, nbytes = n * sizeof(float);
    const char *args[] = {"--3d", "--dtype", "fp32", "--step", "2", argv[1]};
    char *log = NULL;
    drs_kernel *k = drs_kernel_build(6, args, NULL, &log);          /* runs hipcc on a cache miss: before any HIP call */
    if (!k) { printf("build failed: %s\n", log ? log : ""); return 1; }
    printf("resources %s", drs_kernel_resources(k));
    float *h = (float *)malloc(nbytes), *out = (float *)malloc(nbytes), *ref = (float *)malloc(nbytes);
    drs_fill_random_f32(h, n, 1);
    void *a, *b, *ga, *gb;
    /* the optimised kernel's pair in ONE allocation laid out as the kernel recommends (drs_kernel_pair_layout: launch time depends on
     * (out - in) mod 64 MiB on MI355X); the gold pair as two plain allocations like the reference's host code */
    size_t arena_bytes = 0, out_at = 0;
    char *arena;
    if (drs_kernel_pair_layout(k, &arena_bytes, &out_at) != 0 || out_at < nbytes || arena_bytes != out_at + nbytes || out_at % 4096) { printf("pair layout failed\n"); return 1; }
    if (hipMalloc((void **)&arena, arena_bytes) || hipMalloc(&ga, nbytes) || hipMalloc(&gb, nbytes)) { printf("hipMalloc failed\n"); return 1; }
    a = arena; b = arena + out_at;
    hipMemcpy(a, h, nbytes, hipMemcpyHostToDevice); hipMemset(b, 0, nbytes);
    hipMemcpy(ga, h, nbytes, hipMemcpyHostToDevice); hipMemset(gb, 0, nbytes);
    float ms = 0.f;
    int launches = drs_kernel_run_timed(k, a, b, 4, 0, NULL, &ms);      /* no warm-up: the result must equal one gold run */
    int glaunches = drs_kernel_run(k, ga, gb, 4, 1, NULL);
    hipDeviceSynchronize();
    hipMemcpy(out, a, nbytes, hipMemcpyDeviceToHost);
    hipMemcpy(ref, ga, nbytes, hipMemcpyDeviceToHost);
    double max_abs, max_rel; long at;
    double rms = drs_check_error_f32(3, L, M, N, H, out, ref, &max_abs, &at, &max_rel);
    printf("launches %d gold_launches %d ms_positive %d rms %.3e max_abs %.3e max_rel %.3e\n", launches, glaunches, ms > 0.f, rms, max_abs, max_rel);
    drs_kernel_close(k);
    return 0;
}
