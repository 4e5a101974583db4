/* capi_slab_host.c -- the N > 1 entry points of the C ABI from a plain-C host (INTEGRATION.md "slab run"): one process plays
 * the middle rank of a 3-rank z-slab run of the whole grid's spec on ONE GPU, its two neighbours being itself (communicator of
 * size 1: RCCL's send/recv to self), 12 launches, and compares its owned planes with the single-domain run of the same grid
 * (drs_kernel_run on the whole grid, periodic in nothing: the self-neighbour ghosts make the slab the middle third of a grid
 * whose other thirds are copies -- so the reference here is a second slab run with the exchange replaced by hipMemcpy, plane
 * for plane what tests/test_gpu_parity.py checks against the torch path).  usage: capi_slab_host <3d .stc> <M> <N> */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "drstencil_amd.h"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int M = atoi(argv[2]), N = atoi(argv[3]);
    const char *args[] = {"--3d", "--dtype", "fp32", "--step", "2", "--sn", "16", argv[1]};
    char *log = NULL;
    int rc = 1;
    float *out[2] = {NULL, NULL};          /* A after the run: [0] eager, [1] through the captured graph */
    for (int graph = 1; graph >= 0; graph--) {
        setenv("DRS_SLAB_GRAPH", graph ? "1" : "0", 1);
        /* plan + kernels first: this may run hipcc, which a process that has touched the GPU must not do */
        drs_slab *s = drs_slab_open(8, args, 0, NULL, /*world*/ 1, /*rank*/ 1, /*every*/ 2, /*rehearse_world*/ 3, NULL, &log);
        if (!s) { printf("drs_slab_open failed: %s\n", log ? log : ""); return 1; }
        long p[8];
        drs_slab_plan(s, p);
        const size_t plane = (size_t)M * N, n = (size_t)p[4] * plane, nbytes = n * sizeof(float);
        unsigned char id[DRS_SLAB_ID_BYTES];
        if (drs_slab_unique_id(id) != 0) { printf("no RCCL\n"); return 1; }
        if (drs_slab_connect(s, id, NULL) != 0) { printf("connect failed: %s\n", drs_slab_error(s)); return 1; }
        float *h = (float *)malloc(nbytes);
        out[graph] = (float *)malloc(nbytes);
        drs_fill_random_f32(h, n, 1);
        void *a, *b;
        if (hipMalloc(&a, nbytes) || hipMalloc(&b, nbytes)) { printf("hipMalloc failed\n"); return 1; }
        hipMemcpy(a, h, nbytes, hipMemcpyHostToDevice); hipMemset(b, 0, nbytes);
        hipDeviceSynchronize();
        const int launches = drs_slab_run(s, a, b, 24);
        if (launches < 0 || drs_slab_sync(s) != 0) { printf("run failed: %s\n", drs_slab_error(s)); return 1; }
        hipMemcpy(out[graph], a, nbytes, hipMemcpyDeviceToHost);
        /* the ghost planes must have been exchanged: plane 0 of A is no longer the input */
        const int ghosts_moved = memcmp(out[graph], h, plane * sizeof(float)) != 0;
        printf("graph_requested %d info %s launches %d lloc %ld ghost_width %ld ghosts_moved %d\n", graph, drs_slab_info(s), launches, p[4], p[5], ghosts_moved);
        if (!graph) { printf("eager_equals_graph %d\n", memcmp(out[0], out[1], nbytes) == 0); rc = 0; }
        hipFree(a); hipFree(b);
        drs_slab_close(s);
        free(h);
    }
    free(out[0]); free(out[1]);
    return rc;
}
