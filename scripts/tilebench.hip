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


// Round 3: WAVE-SPECIALISED copy of the same tile shape.  NLW "loader" wavefronts per workgroup do nothing but
// request planes by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at consecutive LDS addresses) into a ring of NSLOT plane
// slots, DEPTH planes ahead, and count their own vmcnt; the 8 "consumer" wavefronts (the 512 lanes of tile_copy) never issue a load:
// barrier, read their points of the landed plane from LDS, store them.  Reads and writes sit in different wavefronts' queues, no
// consumer drains loads, no prefetch registers.  One barrier per plane as in the stencil kernels.
template <int LX, int RY, int DEPTH, int NLW>
__global__ __launch_bounds__(512 + 64 * NLW) void tile_copy_ws(const float* __restrict__ in, float* __restrict__ out, int N, int M, int L, long pitch, long plane, int sn, int band)
{
    constexpr int LY = 512 / LX, TW = LX * 4, TH = LY * RY, NSLOT = DEPTH + 1;
    constexpr int ROWB = TW * 4;                         // bytes of a tile row
    constexpr int PLANE_V4 = TW * TH / 4;                // 16-byte pieces of a tile plane
    constexpr int PIECES = PLANE_V4 / 64;                // wavefront instructions per plane
    __shared__ __attribute__((aligned(16))) float ring[NSLOT * TW * TH];
    const int tiles_x = N / TW, tiles_y = M / TH, tiles = tiles_x * tiles_y;
    int wg = blockIdx.x, zb, t;
    if (band == 2) { const int xcd = wg % 8, slot = wg / 8, per = tiles / 8; t = xcd * per + slot % per; zb = slot / per; }
    else { if (band) { const int per = gridDim.x / 8; wg = (wg % 8) * per + wg / 8; } zb = wg / tiles; t = wg % tiles; }
    const int tx = t % tiles_x, ty = t / tiles_x;
    const int k0 = zb * sn, k1 = min(k0 + sn, L);
    const long tile_org = (long)(ty * TH) * pitch + (long)tx * TW;
    const int tid = threadIdx.x;
    if (tid >= 512) {
        // ---- loader wavefront lw of NLW: pieces lw, lw + NLW, ... of every plane
        const int lw = (tid - 512) / 64, lane = tid & 63;
        auto request = [&](int k) {
            const float* src = in + (long)k * plane + tile_org;
            float* dst = ring + (size_t)((k - k0) % NSLOT) * TW * TH;
#pragma unroll
            for (int pc = lw; pc < PIECES; pc += NLW) {
                const int v = pc * 64 + lane;            // 16-byte piece of the tile plane, row-major
                const int row = v / (TW / 4), col4 = v % (TW / 4);
                __builtin_amdgcn_global_load_lds(src + (long)row * pitch + col4 * 4, dst + pc * 256, 16, 0, 0);
            }
        };
        for (int d = 0; d < DEPTH; d++) if (k0 + d < k1) request(k0 + d);
        constexpr int PER = (PIECES + NLW - 1) / NLW;    // my instructions per plane (PIECES divisible by NLW in the shapes used)
        for (int k = k0; k < k1; k++) {
            // plane k has landed when at most (planes still in flight behind it) x PER of my requests are outstanding
            const int behind = min(DEPTH - 1, k1 - 1 - k);
            if (behind >= 3) __builtin_amdgcn_s_waitcnt(0x0f70 | ((3 * PER) & 0xf) | ((((3 * PER) >> 4) & 3) << 14));
            else if (behind == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | ((2 * PER) & 0xf) | ((((2 * PER) >> 4) & 3) << 14));
            else if (behind == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | ((1 * PER) & 0xf) | ((((1 * PER) >> 4) & 3) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();                             // plane k is in LDS for everyone; the consumers are done with plane k - 1
            if (k + DEPTH < k1) request(k + DEPTH);      // into the slot plane k - 1 occupied
        }
        return;
    }
    // ---- consumer wavefronts: the 512 lanes of tile_copy
    const int lx = tid % LX, ly = tid / LX;
    float* pout = out + (long)k0 * plane + tile_org + (long)(ly * RY) * pitch + lx * 4;
    for (int k = k0; k < k1; k++) {
        __syncthreads();
        const float* slot = ring + (size_t)((k - k0) % NSLOT) * TW * TH;
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const v4 x = *(const v4*)(slot + (ly * RY + r) * TW + lx * 4);
            __builtin_nontemporal_store(x, (v4*)(pout + (long)(k - k0) * plane + (long)r * pitch));
        }
    }
}

template <int LX, int RY, int DEPTH, int NLW>
static void run_ws(const char* name, const float* a, float* b, int N, int M, int L, long pitch, int sn, int band)
{
    constexpr int TW = LX * 4, TH = (512 / LX) * RY;
    const long plane = pitch * M;
    const int grid = (N / TW) * (M / TH) * ((L + sn - 1) / sn);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) tile_copy_ws<LX, RY, DEPTH, NLW><<<grid, 512 + 64 * NLW>>>(a, b, N, M, L, pitch, plane, sn, band);
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 7; r++) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 5; i++) tile_copy_ws<LX, RY, DEPTH, NLW><<<grid, 512 + 64 * NLW>>>(a, b, N, M, L, pitch, plane, sn, band);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5; sum += ms; if (ms < best) best = ms;
    }
    const double bytes = 2.0 * 4.0 * N * (double)M * L;
    printf("%-52s sn %4d  tile %4d x %3d  %d loader waves, %d planes ahead: %.3f ms (best %.3f)  %.0f GB/s  (%.1f %% of 8 TB/s)\n", name, sn, TW, TH, NLW, DEPTH, sum / 7, best,
           bytes / (sum / 7 * 1e-3) / 1e9, bytes / (sum / 7 * 1e-3) / 8e12 * 100);
    if (hipGetLastError() != hipSuccess) printf("  launch error\n");
    // the copy must be a copy: compare a few words of the last plane
    float ha[4], hb[4];
    const size_t probe = (size_t)(L - 1) * plane + (size_t)(M - 1) * pitch + N - 4;
    (void)hipMemcpy(ha, a + probe, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, b + probe, 16, hipMemcpyDeviceToHost);
    if (ha[0] != hb[0] || ha[3] != hb[3]) printf("  COPY MISMATCH\n");
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
    // TILEBENCH_SKEW_MIB=<n> (round 3, second half): both arrays in ONE allocation, the output n MiB (mod 64 MiB) behind the input -- launch time of a
    // z-streaming kernel depends on (out - in) mod 64 MiB; unset: two separate allocations as in rounds 2-3 (an arbitrary phase)
    if (getenv("TILEBENCH_SKEW_MIB")) {
        const size_t period = 64UL << 20, skew = ((size_t)atol(getenv("TILEBENCH_SKEW_MIB")) << 20) % period;
        const size_t off = (elems * 4 + period - 1) / period * period + skew;
        char* arena;
        if (hipMalloc(&arena, off + elems * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
        a = (float*)arena; b = (float*)(arena + off);
        printf("one arena, output %zu MiB (mod 64 MiB) behind the input\n", skew >> 20);
    } else if (hipMalloc(&a, elems * 4) != hipSuccess || hipMalloc(&b, elems * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(a, 0x3c, elems * 4); (void)hipMemset(b, 0, elems * 4);
    for (int padded = 0; padded < (getenv("TILEBENCH_PADDED") ? 2 : 1); padded++) {
        const long pitch = N + (padded ? pad : 0);
        printf("---- %d^3 grid, row pitch %ld bytes%s\n", DIM, pitch * 4, padded ? " (padded by 256 bytes: NOT the reference's layout)" : " (dense: the reference's layout)");
        if (getenv("TILEBENCH_WS")) {     // round 3: the wave-specialised copy against the plain one of the same shape, interleaved
            for (int rep = 0; rep < 2; rep++) {
                run<32, 2, 3, true>("plain: 128 x 32, barrier, band map", a, b, N, M, L, pitch, 32, 2);
                run<32, 2, 3, true>("plain: 128 x 32, barrier, band map, 8-plane blocks", a, b, N, M, L, pitch, 8, 2);
                run_ws<32, 2, 3, 2>("wave-specialised: 128 x 32, band map", a, b, N, M, L, pitch, 32, 2);
                run_ws<32, 2, 3, 1>("wave-specialised: 128 x 32, band map", a, b, N, M, L, pitch, 32, 2);
                run_ws<32, 2, 3, 4>("wave-specialised: 128 x 32, band map", a, b, N, M, L, pitch, 32, 2);
                run_ws<32, 2, 2, 2>("wave-specialised: 128 x 32, band map", a, b, N, M, L, pitch, 32, 2);
                run_ws<32, 2, 3, 2>("wave-specialised: 128 x 32, band map, 8-plane blocks", a, b, N, M, L, pitch, 8, 2);
                run_ws<32, 2, 3, 2>("wave-specialised: 128 x 32, chunk map", a, b, N, M, L, pitch, 32, 1);
                run<64, 2, 3, true>("plain: 256 x 16, barrier, band map", a, b, N, M, L, pitch, 32, 2);
                run_ws<64, 2, 3, 2>("wave-specialised: 256 x 16, band map", a, b, N, M, L, pitch, 32, 2);
                run_ws<64, 4, 3, 2>("wave-specialised: 256 x 32, band map", a, b, N, M, L, pitch, 32, 2);
            }
            continue;
        }
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
