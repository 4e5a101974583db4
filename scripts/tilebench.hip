// tilebench.hip -- what does the ACCESS PATTERN of a z-streaming x-y tile cost on this MI355X, with no stencil in it?
// A copy kernel that reads and writes a 1024^3 fp32 grid exactly like the fused kernel's workgroups do (512 lanes = LX lanes
// along x times 512/LX rows of lanes, 16 bytes per lane and row, RY rows per lane, stream blocks of SN planes, DEPTH planes of
// loads in flight, non-temporal stores, XCD band map) but without halos, LDS or arithmetic -- for tile widths from 128 columns to
// a full row, and for the dense row pitch (4 KiB, what the reference's arrays have) against a pitch padded by 256 bytes.
// usage: tilebench   (prints one line per shape; build: hipcc -O3 --offload-arch=gfx950 -o tilebench tilebench.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int LX, int RY, int DEPTH, bool SYNC = false, int LDSKB = 0, bool STORE_FIRST = false>
__global__ __launch_bounds__(512) void tile_copy(const float* __restrict__ in, float* __restrict__ out, int N, int M, int L, long pitch, long plane, int sn, int band)
{
    constexpr int LY = 512 / LX;            // rows of lanes
    constexpr int TW = LX * 4, TH = LY * RY;
    const int tiles_x = N / TW, tiles_y = M / TH, tiles = tiles_x * tiles_y;
    // XCD band map: workgroup ids round-robin over 8 XCDs; give every XCD a contiguous band of tiles
    // band = 1: every XCD (workgroup id % 8) takes a contiguous eighth of the (stream block, tile) list
    // band = 2: the generator's --xcd-remap 2: every XCD owns a fixed band of the x-y tiles of EVERY stream block and all XCDs walk the
    //           stream blocks together (one compact front through memory)
    int wg = blockIdx.x, zb, t;
    if (band == 2) { const int xcd = wg % 8, slot = wg / 8, per = tiles / 8; t = xcd * per + slot % per; zb = slot / per; }
    else { if (band) { const int per = gridDim.x / 8; wg = (wg % 8) * per + wg / 8; } zb = wg / tiles; t = wg % tiles; }
    const int tx = t % tiles_x, ty = t / tiles_x;
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const long off = (long)(ty * TH + ly * RY) * pitch + (long)tx * TW + lx * 4;
    const int k0 = zb * sn, k1 = min(k0 + sn, L);
    const float* pin = in + (long)k0 * plane + off;
    float* pout = out + (long)k0 * plane + off;
    if (LDSKB > 0) {                        // occupancy limiter: LDSKB KiB of LDS per workgroup (160 KiB per CU)
        __shared__ float pad[LDSKB > 0 ? LDSKB * 256 : 1];
        if (pitch < 0) pad[threadIdx.x] = 1.0f;
        if (pitch < -1) pout[0] = pad[(threadIdx.x + 1) % 512];
    }
    v4 buf[DEPTH][RY];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (k0 + d < k1)
#pragma unroll
            for (int r = 0; r < RY; r++) buf[d][r] = *(const v4*)(pin + (long)d * plane + (long)r * pitch);
    for (int k = k0; k < k1; k += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (k + d >= k1) break;
            if (SYNC) __syncthreads();      // the stencil kernels meet at a barrier once per plane (LDS exchange): the waves of a workgroup stay on the same plane
            v4 cur[RY];
#pragma unroll
            for (int r = 0; r < RY; r++) cur[r] = buf[d][r];
            if (STORE_FIRST)
#pragma unroll
                for (int r = 0; r < RY; r++) __builtin_nontemporal_store(cur[r], (v4*)(pout + (long)(k - k0 + d) * plane + (long)r * pitch));
            if (k + d + DEPTH < k1)
#pragma unroll
                for (int r = 0; r < RY; r++) buf[d][r] = *(const v4*)(pin + (long)(k - k0 + d + DEPTH) * plane + (long)r * pitch);
            if (!STORE_FIRST)
#pragma unroll
                for (int r = 0; r < RY; r++) __builtin_nontemporal_store(cur[r], (v4*)(pout + (long)(k - k0 + d) * plane + (long)r * pitch));
        }
    }
}

template <int LX, int RY, int DEPTH, bool SYNC = false, int LDSKB = 0, bool STORE_FIRST = false>
static void run(const char* name, const float* a, float* b, int N, int M, int L, long pitch, int sn, int band)
{
    constexpr int TW = LX * 4, TH = (512 / LX) * RY;
    const long plane = pitch * M;
    const int grid = (N / TW) * (M / TH) * ((L + sn - 1) / sn);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) tile_copy<LX, RY, DEPTH, SYNC, LDSKB, STORE_FIRST><<<grid, 512>>>(a, b, N, M, L, pitch, plane, sn, band);
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 7; r++) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 5; i++) tile_copy<LX, RY, DEPTH, SYNC, LDSKB, STORE_FIRST><<<grid, 512>>>(a, b, N, M, L, pitch, plane, sn, band);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5; sum += ms; if (ms < best) best = ms;
    }
    const double bytes = 2.0 * 4.0 * N * (double)M * L;
    printf("%-44s pitch %5ld  sn %4d  tile %4d x %3d  %d workgroups: %.3f ms (best %.3f)  %.0f GB/s  (%.1f %% of 8 TB/s)\n", name, pitch, sn, TW, TH, grid, sum / 7, best,
           bytes / (sum / 7 * 1e-3) / 1e9, bytes / (sum / 7 * 1e-3) / 8e12 * 100);
    if (hipGetLastError() != hipSuccess) printf("  launch error\n");
}

int main()
{
    // TILEBENCH_DIM=512: the same copies on the 512^3 grid of BASELINE C3 (round 3: what is the ceiling at 195 us launches?)
    const int DIM = getenv("TILEBENCH_DIM") ? atoi(getenv("TILEBENCH_DIM")) : 1024;
    const int N = DIM, M = DIM, L = DIM;
    const long pad = 64;                                  // floats: 256 bytes
    const size_t elems = (size_t)(N + pad) * M * L + 4096;
    float *a, *b;
    if (hipMalloc(&a, elems * 4) != hipSuccess || hipMalloc(&b, elems * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(a, 0, elems * 4); (void)hipMemset(b, 0, elems * 4);
    for (int padded = 0; padded < (getenv("TILEBENCH_PADDED") ? 2 : 1); padded++) {
        const long pitch = N + (padded ? pad : 0);
        printf("---- %d^3 grid, row pitch %ld bytes%s\n", DIM, pitch * 4, padded ? " (padded by 256 bytes: NOT the reference's layout)" : " (dense: the reference's layout)");
        if (DIM != 1024) {      // tiles of up to 512 columns; stream blocks of 8-64 planes
            run<32, 2, 3, true>("128 x 32, barrier, band map, 32-plane blocks", a, b, N, M, L, pitch, 32, 2);
            run<32, 2, 3, true>("128 x 32, barrier, band map, 16-plane blocks", a, b, N, M, L, pitch, 16, 2);
            run<32, 2, 3, true>("128 x 32, barrier, band map, 8-plane blocks", a, b, N, M, L, pitch, 8, 2);
            run<32, 2, 3, true>("128 x 32, barrier, band map, 64-plane blocks", a, b, N, M, L, pitch, 64, 2);
            run<32, 2, 3, true, 60>("128 x 32, barrier, band map, 2 wg/CU", a, b, N, M, L, pitch, 32, 2);
            run<32, 2, 3, false>("128 x 32, no barrier, band map", a, b, N, M, L, pitch, 32, 2);
            run<32, 2, 3, true>("128 x 32, barrier, chunk map", a, b, N, M, L, pitch, 32, 1);
            run<32, 2, 3, true>("128 x 32, barrier, chunk map, 8-plane blocks", a, b, N, M, L, pitch, 8, 1);
            run<32, 2, 3, true>("128 x 32, barrier, dispatch order", a, b, N, M, L, pitch, 32, 0);
            run<32, 1, 3, true>("128 x 16, barrier, chunk map", a, b, N, M, L, pitch, 32, 1);
            run<64, 2, 3, true>("256 x 16, barrier, chunk map", a, b, N, M, L, pitch, 32, 1);
            run<64, 4, 3, true>("256 x 32, barrier, chunk map", a, b, N, M, L, pitch, 32, 1);
            run<128, 4, 3, true>("512 x 16 (full rows), barrier, chunk map", a, b, N, M, L, pitch, 32, 1);
            run<128, 4, 3, true>("512 x 16 (full rows), barrier, 4-plane blocks", a, b, N, M, L, pitch, 4, 1);
            run<128, 4, 1, true>("512 x 16 (full rows), barrier, 4-plane, depth 1", a, b, N, M, L, pitch, 4, 1);
            run<128, 2, 3, true>("512 x 8 (full rows), barrier, 4-plane blocks", a, b, N, M, L, pitch, 4, 1);
            continue;
        }
        run<32, 2, 3, true>("128 x 32, barrier, generator's band map", a, b, N, M, L, pitch, 32, 2);
        run<32, 2, 3, true>("128 x 32, barrier, band map, 8-plane blocks", a, b, N, M, L, pitch, 8, 2);
        run<32, 2, 3, true, 100>("128 x 32, barrier, band map, 1 wg/CU", a, b, N, M, L, pitch, 32, 2);
        run<32, 2, 3, true, 60>("128 x 32, barrier, band map, 2 wg/CU", a, b, N, M, L, pitch, 32, 2);
        run<32, 2, 3, false>("128 x 32, no barrier, band map", a, b, N, M, L, pitch, 32, 2);
        run<64, 4, 3, true>("256 x 32, barrier, band map", a, b, N, M, L, pitch, 32, 2);
        run<256, 4, 1, true, 100>("full rows x 8, barrier, band map, 1 wg/CU", a, b, N, M, L, pitch, 4, 2);
        run<256, 4, 1, true, 60>("full rows x 8, barrier, band map, 2 wg/CU", a, b, N, M, L, pitch, 4, 2);
        run<256, 4, 1, false, 60>("full rows x 8, no barrier, band map, 2 wg/CU", a, b, N, M, L, pitch, 4, 2);
        run<256, 4, 1, true>("full rows x 8, barrier, band map", a, b, N, M, L, pitch, 4, 2);
        run<32, 2, 3>("128-column tiles (the fused kernel's shape)", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 3, true>("128-column tiles, barrier per plane", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 3, true>("128-column tiles, barrier, 8-plane blocks", a, b, N, M, L, pitch, 8, 1);
        run<32, 2, 3, true, 100>("128-col, barrier, 1 workgroup per CU", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 3, true, 60>("128-col, barrier, 2 workgroups per CU", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 3, true, 100, true>("128-col, barrier, 1 wg/CU, stores first", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 3, true, 0, true>("128-col, barrier, stores first", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 1, true>("128-col, barrier, 1 plane in flight", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 5, true>("128-col, barrier, 5 planes in flight", a, b, N, M, L, pitch, 32, 1);
        run<32, 4, 2, true>("128 x 64 tiles, barrier", a, b, N, M, L, pitch, 32, 1);
        run<32, 1, 3, true>("128 x 16 tiles, barrier", a, b, N, M, L, pitch, 32, 1);
        run<64, 2, 3, true>("256 x 16 tiles, barrier", a, b, N, M, L, pitch, 32, 1);
        run<64, 4, 3, true>("256 x 32 tiles, barrier", a, b, N, M, L, pitch, 32, 1);
        run<128, 4, 3, true>("512 x 16 tiles, barrier", a, b, N, M, L, pitch, 32, 1);
        run<256, 4, 3, true>("full rows x 8, barrier, 32-plane blocks", a, b, N, M, L, pitch, 32, 1);
        run<256, 4, 3, true, 100>("full rows x 8, barrier, 1 wg/CU, sn 4", a, b, N, M, L, pitch, 4, 1);
        run<256, 4, 1, true, 100>("full rows x 8, barrier, 1 wg/CU, depth 1", a, b, N, M, L, pitch, 4, 1);
        run<256, 4, 1, true, 60>("full rows x 8, barrier, 2 wg/CU, depth 1", a, b, N, M, L, pitch, 4, 1);
        run<32, 2, 3>("128-column tiles, no band map", a, b, N, M, L, pitch, 32, 0);
        run<32, 2, 1>("128-column tiles, 1 plane in flight", a, b, N, M, L, pitch, 32, 1);
        run<32, 2, 3>("128-column tiles, 8-plane blocks", a, b, N, M, L, pitch, 8, 1);
        run<32, 2, 3>("128-column tiles, whole-column blocks", a, b, N, M, L, pitch, 1024, 1);
        run<64, 2, 3>("256-column tiles", a, b, N, M, L, pitch, 32, 1);
        run<64, 4, 3>("256-column tiles, 4 rows per lane", a, b, N, M, L, pitch, 32, 1);
        run<128, 2, 3>("512-column tiles", a, b, N, M, L, pitch, 32, 1);
        run<256, 2, 3>("full rows (1024 columns x 4 rows)", a, b, N, M, L, pitch, 4, 1);
        run<256, 4, 3>("full rows (1024 columns x 8 rows)", a, b, N, M, L, pitch, 4, 1);
        run<256, 4, 3, true>("full rows x 8 rows, barrier per plane", a, b, N, M, L, pitch, 4, 1);
        run<256, 4, 1>("full rows x 8 rows, 1 plane in flight", a, b, N, M, L, pitch, 4, 1);
        run<256, 4, 3>("full rows x 8 rows, 32-plane blocks", a, b, N, M, L, pitch, 32, 1);
    }
    return 0;
}
