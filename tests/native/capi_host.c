/* capi_host.c -- a plain-C host of the C ABI (include/drstencil_amd.h), as a maintainer of the reference would write it:
 * the generator as a function (main.cpp:10-280) and the stencil IR getters (drstencil.hpp:24-48).  No GPU is touched.
 * usage: capi_host <file.stc> ; prints one line per fact, checked by tests/test_capi_and_tuner.py. */
#include <stdio.h>
#include <string.h>
#include "drstencil_amd.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const char *args[] = {"--3d", "--dtype", "fp32", "--step", "2", argv[1]};
    char *source = NULL, *messages = NULL;
    int rc = drs_generate(6, args, &source, &messages);
    printf("version %s\n", drs_version());
    printf("generate rc %d source_bytes %zu has_kernel %d has_gold %d\n", rc, source ? strlen(source) : 0,
           source && strstr(source, "__global__") != NULL, source && strstr(source, "gold_") != NULL);
    drs_free(source);
    drs_free(messages);
    const char *bad[] = {"--bx", argv[1]};     /* a value-taking flag in the second-to-last slot: main.cpp:119-131 */
    char *msg2 = NULL;
    rc = drs_generate(2, bad, NULL, &msg2);
    printf("illegal rc %d message %s", rc, msg2 ? msg2 : "(null)\n");
    drs_free(msg2);
    int status = -1;
    drs_spec *s = drs_spec_open(argv[1], 3, 2, 0, 5, &status);
    if (!s) { printf("spec open failed %d\n", status); return 1; }
    int L, M, N, sizes[4];
    drs_spec_dims(s, &L, &M, &N);
    drs_spec_partition(s, sizes);
    printf("spec status %d dims %d %d %d halo %d dist %d range %d points %d iterations %d launches %d partition %d %d %d %d\n", status, L, M, N,
           drs_spec_halo(s), drs_spec_dist(s), drs_spec_range(s), drs_spec_npoints(s), drs_spec_iterations(s), drs_spec_launches(s),
           sizes[0], sizes[1], sizes[2], sizes[3]);
    int k, j, i; double c; char text[32];
    drs_spec_point(s, 0, &k, &j, &i, &c, text);
    printf("point0 %d %d %d %s\n", k, j, i, text);
    drs_spec_close(s);
    float a[4];
    drs_fill_random_f32(a, 4, 1);
    printf("rand0 %.8f\n", a[0]);
    return 0;
}
