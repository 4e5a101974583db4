// common.hpp -- host-side support for programs emitted by drstencil (MI355X back end).
// Same role and stdout protocol as the reference's common.hpp (get_random 9-11,
// getRandom*/getZero* 14-45, checkError2D/3D 47-102) but typed on the element type
// (the reference is double-only) and free of VLA pointer casts, which hipcc rejects.
#ifndef DRSTENCIL_AMD_COMMON_HPP
#define DRSTENCIL_AMD_COMMON_HPP

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// rand()/(RAND_MAX-1) from the unseeded glibc stream, row-major (first value
// 0.8401877175459523); fp32 runs use the same stream rounded to float.
template <typename T>
static T* getRandomArray(size_t n) {
  T* a = new T[n];
  for (size_t x = 0; x < n; x++) a[x] = (T)((double)rand() / (double)(RAND_MAX - 1));
  return a;
}

template <typename T>
static T* getZeroArray(size_t n) {
  T* a = new T[n];
  memset((void*)a, 0, sizeof(T) * n);
  return a;
}

// Running max-abs error over the interior box (print threshold 1e-13, one line per
// new maximum), returns the RMS error -- the reference's metric and output format.
template <typename T>
static double checkError2D(int width_x, const T* output, const T* reference, int y_lb, int y_ub, int x_lb, int x_ub) {
  double error = 0.0, max_error = 1e-13;
  int max_k = 0, max_j = 0;
  for (int j = y_lb; j < y_ub; j++)
    for (int k = x_lb; k < x_ub; k++) {
      size_t at = (size_t)j * width_x + k;
      double d = std::fabs((double)output[at] - (double)reference[at]);
      error += d * d;
      if (d > max_error) {
        printf("Values at index (%d,%d) differ : %.6f and %.6f\n", j, k, (double)reference[at], (double)output[at]);
        max_error = d; max_k = k; max_j = j;
      }
    }
  printf("[Test] Max Error : %e @ (,%d,%d)\n", max_error, max_j, max_k);
  return std::sqrt(error / ((double)(y_ub - y_lb) * (double)(x_ub - x_lb)));
}

template <typename T>
static double checkError3D(int width_y, int width_x, const T* output, const T* reference, int z_lb, int z_ub, int y_lb,
                           int y_ub, int x_lb, int x_ub) {
  double error = 0.0, max_error = 1e-13;
  int max_k = 0, max_j = 0, max_i = 0;
  for (int i = z_lb; i < z_ub; i++)
    for (int j = y_lb; j < y_ub; j++)
      for (int k = x_lb; k < x_ub; k++) {
        size_t at = ((size_t)i * width_y + j) * width_x + k;
        double d = std::fabs((double)output[at] - (double)reference[at]);
        error += d * d;
        if (d > max_error) {
          printf("Values at index (%d,%d,%d) differ : %.6f and %.6f\n", i, j, k, (double)reference[at], (double)output[at]);
          max_error = d; max_k = k; max_j = j; max_i = i;
        }
      }
  printf("[Test] Max Error : %e @ (%d,%d,%d)\n", max_error, max_i, max_j, max_k);
  return std::sqrt(error / ((double)(z_ub - z_lb) * (double)(y_ub - y_lb) * (double)(x_ub - x_lb)));
}

#endif
