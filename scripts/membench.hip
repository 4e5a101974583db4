// membench.hip -- HBM copy/read/write ceilings on this MI355X with several kernel shapes,
// to calibrate what fraction of the 8 TB/s spec a streaming kernel can reach in practice.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int UNR, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_gs(const v4* __restrict__ a, v4* __restrict__ b, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
    v4 t[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) t[u] = NTL ? __builtin_nontemporal_load(&a[i + u * stride]) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; u++) { if (NTS) __builtin_nontemporal_store(t[u], &b[i + u * stride]); else b[i + u * stride] = t[u]; }
  }
  for (; i < n; i += stride) b[i] = a[i];
}
// each workgroup copies a contiguous chunk (like a tile), UNR vectors in flight per lane
template <int UNR, bool NTS>
__global__ __launch_bounds__(256) void copy_chunk(const v4* __restrict__ a, v4* __restrict__ b, size_t n, size_t per_wg) {
  size_t base = (size_t)blockIdx.x * per_wg;
  size_t end = base + per_wg < n ? base + per_wg : n;
  for (size_t i = base + threadIdx.x; i < end; i += 256 * UNR) {
    v4 t[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) if (i + u * 256 < end) t[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNR; u++) if (i + u * 256 < end) { if (NTS) __builtin_nontemporal_store(t[u], &b[i + u * 256]); else b[i + u * 256] = t[u]; }
  }
}
__global__ __launch_bounds__(256) void copy_simple(const v4* __restrict__ a, v4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}
// one shot: each workgroup handles PER consecutive vectors per lane (contiguous PER*4 KiB chunk), then exits
template <int PER, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_shot(const v4* __restrict__ a, v4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
  v4 t[PER];
#pragma unroll
  for (int u = 0; u < PER; u++) if (i + u * 256 < n) t[u] = NTL ? __builtin_nontemporal_load(&a[i + u * 256]) : a[i + u * 256];
#pragma unroll
  for (int u = 0; u < PER; u++) if (i + u * 256 < n) { if (NTS) __builtin_nontemporal_store(t[u], &b[i + u * 256]); else b[i + u * 256] = t[u]; }
}
template <int PER, bool NTS>
__global__ __launch_bounds__(256) void write_shot(v4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
  v4 s = {1, 2, 3, 4};
#pragma unroll
  for (int u = 0; u < PER; u++) if (i + u * 256 < n) { if (NTS) __builtin_nontemporal_store(s, &b[i + u * 256]); else b[i + u * 256] = s; }
}
template <int PER>
__global__ __launch_bounds__(256) void read_shot(const v4* __restrict__ a, float* out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
  v4 s = 0;
#pragma unroll
  for (int u = 0; u < PER; u++) if (i + u * 256 < n) s += a[i + u * 256];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1;
}
__global__ __launch_bounds__(256) void read_gs(const v4* __restrict__ a, float* out, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  v4 s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += a[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1;
}
__global__ __launch_bounds__(256) void write_gs(v4* __restrict__ b, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  v4 s = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(s, &b[i]);
}
#define TIME(name, bytes, launch)                                                     \
  do {                                                                                \
    for (int w = 0; w < 3; w++) { launch; }                                           \
    (void)hipEventRecord(e0, 0);                                                      \
    for (int r = 0; r < 10; r++) { launch; }                                          \
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);                       \
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;                       \
    printf("%-34s %8.3f ms %8.0f GB/s\n", name, ms, (double)(bytes) / ms / 1e6);      \
  } while (0)
int main(int argc, char** argv) {
  size_t nbytes = (argc > 1 ? atol(argv[1]) : 4096) * (1ul << 20);
  size_t n = nbytes / 16;
  v4 *a, *b; float* o;
  (void)hipMalloc(&a, nbytes); (void)hipMalloc(&b, nbytes); (void)hipMalloc(&o, 4);
  (void)hipMemset(a, 1, nbytes); (void)hipMemset(b, 0, nbytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("buffer %zu MiB each\n", nbytes >> 20);
  TIME("hipMemcpyDtoD", 2 * nbytes, (void)hipMemcpyAsync(b, a, nbytes, hipMemcpyDeviceToDevice, 0));
  TIME("copy_simple 1 v4/thread", 2 * nbytes, (copy_simple<<<dim3((n + 255) / 256), 256>>>(a, b, n)));
  int grids[] = {1024, 2048, 4096, 8192, 16384};
  for (int g : grids) {
    char nm[64];
    snprintf(nm, 64, "copy_gs<1> grid %d", g); TIME(nm, 2 * nbytes, (copy_gs<1, false, false><<<g, 256>>>(a, b, n)));
    snprintf(nm, 64, "copy_gs<4> grid %d", g); TIME(nm, 2 * nbytes, (copy_gs<4, false, false><<<g, 256>>>(a, b, n)));
    snprintf(nm, 64, "copy_gs<4,nts> grid %d", g); TIME(nm, 2 * nbytes, (copy_gs<4, false, true><<<g, 256>>>(a, b, n)));
    snprintf(nm, 64, "copy_gs<4,ntl,nts> grid %d", g); TIME(nm, 2 * nbytes, (copy_gs<4, true, true><<<g, 256>>>(a, b, n)));
    snprintf(nm, 64, "copy_gs<8,nts> grid %d", g); TIME(nm, 2 * nbytes, (copy_gs<8, false, true><<<g, 256>>>(a, b, n)));
  }
  for (int g : {2048, 8192, 32768}) {
    char nm[64];
    size_t per = (n + g - 1) / g;
    snprintf(nm, 64, "copy_chunk<4> grid %d", g); TIME(nm, 2 * nbytes, (copy_chunk<4, false><<<g, 256>>>(a, b, n, per)));
    snprintf(nm, 64, "copy_chunk<4,nts> grid %d", g); TIME(nm, 2 * nbytes, (copy_chunk<4, true><<<g, 256>>>(a, b, n, per)));
  }
  TIME("copy_shot<1>", 2 * nbytes, (copy_shot<1, false, false><<<dim3((n + 255) / 256), 256>>>(a, b, n)));
  TIME("copy_shot<1,nts>", 2 * nbytes, (copy_shot<1, false, true><<<dim3((n + 255) / 256), 256>>>(a, b, n)));
  TIME("copy_shot<1,ntl,nts>", 2 * nbytes, (copy_shot<1, true, true><<<dim3((n + 255) / 256), 256>>>(a, b, n)));
  TIME("copy_shot<2>", 2 * nbytes, (copy_shot<2, false, false><<<dim3((n + 511) / 512), 256>>>(a, b, n)));
  TIME("copy_shot<4>", 2 * nbytes, (copy_shot<4, false, false><<<dim3((n + 1023) / 1024), 256>>>(a, b, n)));
  TIME("copy_shot<4,nts>", 2 * nbytes, (copy_shot<4, false, true><<<dim3((n + 1023) / 1024), 256>>>(a, b, n)));
  TIME("copy_shot<8>", 2 * nbytes, (copy_shot<8, false, false><<<dim3((n + 2047) / 2048), 256>>>(a, b, n)));
  TIME("copy_shot<16>", 2 * nbytes, (copy_shot<16, false, false><<<dim3((n + 4095) / 4096), 256>>>(a, b, n)));
  TIME("write_shot<1>", nbytes, (write_shot<1, false><<<dim3((n + 255) / 256), 256>>>(b, n)));
  TIME("write_shot<1,nts>", nbytes, (write_shot<1, true><<<dim3((n + 255) / 256), 256>>>(b, n)));
  TIME("write_shot<4>", nbytes, (write_shot<4, false><<<dim3((n + 1023) / 1024), 256>>>(b, n)));
  TIME("write_shot<4,nts>", nbytes, (write_shot<4, true><<<dim3((n + 1023) / 1024), 256>>>(b, n)));
  TIME("read_shot<1>", nbytes, (read_shot<1><<<dim3((n + 255) / 256), 256>>>(a, o, n)));
  TIME("read_shot<4>", nbytes, (read_shot<4><<<dim3((n + 1023) / 1024), 256>>>(a, o, n)));
  TIME("read_gs grid 4096", nbytes, (read_gs<<<4096, 256>>>(a, o, n)));
  TIME("read_gs grid 16384", nbytes, (read_gs<<<16384, 256>>>(a, o, n)));
  TIME("write_gs(nt) grid 4096", nbytes, (write_gs<<<4096, 256>>>(b, n)));
  TIME("hipMemsetAsync", nbytes, (void)hipMemsetAsync(b, 0, nbytes, 0));
  return 0;
}
