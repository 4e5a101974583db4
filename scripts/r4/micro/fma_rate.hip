// Issue rate of v_fma_f64 / v_fma_f32 / v_mov_b32_dpp per SIMD on gfx950: one workgroup of 256 lanes per CU-slot, WAVES waves per SIMD,
// 16 independent chains per lane.  Prints cycles per wave-instruction per SIMD (wall clock x effective clock is not known here: the
// figure is instructions / (time x 2.1e9) -- compare the rows with each other).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <typename T, int DPP>
__global__ void __launch_bounds__(1024) chains(T* out, int n, T a, T b) {
    T c[16];
    for (int i = 0; i < 16; i++) c[i] = (T)(threadIdx.x + i);
    for (int k = 0; k < n; k++) {
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = __builtin_fma(c[i], a, b);
        if (DPP) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                int v = __builtin_amdgcn_update_dpp(0, __float_as_int((float)c[i]), 0x138, 0xf, 0xf, false);   // wave_shr:1
                c[i] += (T)__int_as_float(v);
            }
        }
    }
    T s = 0;
    for (int i = 0; i < 16; i++) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename T, int DPP>
static void run(const char* name, int lanes, int wgs_per_cu) {
    int n = 4096, grid = 256 * wgs_per_cu;
    T* out; hipMalloc(&out, sizeof(T) * grid * lanes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chains<T, DPP><<<grid, lanes>>>(out, 64, (T)1.0000001, (T)1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<T, DPP><<<grid, lanes>>>(out, n, (T)1.0000001, (T)1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves_per_simd = (double)lanes / 64 * wgs_per_cu / 4;
    double inst = 16.0 * n * waves_per_simd;           // fma wave-instructions per SIMD
    printf("%-28s %4d lanes x %d WG/CU (%.1f waves/SIMD): %.3f ms, %.2f ns per fma wave-instruction per SIMD (x clock GHz = cycles), %.1f TFLOP/s\n",
           name, lanes, wgs_per_cu, waves_per_simd, ms, ms * 1e6 / inst, 2.0 * 16 * n * (double)grid * lanes / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<double, 0>("fma f64", 256, 1);  run<double, 0>("fma f64", 512, 1);  run<double, 0>("fma f64", 768, 1);  run<double, 0>("fma f64", 1024, 1);
    run<float, 0>("fma f32", 256, 1);   run<float, 0>("fma f32", 768, 1);   run<float, 0>("fma f32", 1024, 1);
    run<double, 1>("fma f64 + dpp + add", 768, 1);
    return 0;
}
