// Gauge-step kernels of libmpsk: regularize!, QRpos / LQpos, truncated SVD.
#include <hip/hip_runtime.h>
#include <string>
#include "mpsk.h"
#include "mpsk_internal.h"

namespace mpsk {

// v[w] -= <lvec^T, v[w]> rvec      (transfermatrix.jl:70-76); one workgroup per slab
__global__ __launch_bounds__(256) void regularize_kernel(double* __restrict__ v, const double* __restrict__ lvec,
                                                         const double* __restrict__ rvec, int D1, int D2) {
  __shared__ double sh[4];
  __shared__ double coef;
  double* vw = v + (int64_t)blockIdx.x * D1 * D2;
  double acc = 0.0;
  const int64_t n = (int64_t)D1 * D2;
  for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
    int y = (int)(e % D1), x = (int)(e / D1);   // v[y, x]
    acc += lvec[x + (int64_t)D2 * y] * vw[e];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) coef = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  const double cf = coef;
  for (int64_t e = threadIdx.x; e < n; e += blockDim.x) vw[e] -= cf * rvec[e];
}

hipError_t regularize(int W, int D1, int D2, double* v, const double* lvec, const double* rvec, hipStream_t s) {
  hipLaunchKernelGGL(regularize_kernel, dim3(W), dim3(256), 0, s, v, lvec, rvec, D1, D2);
  return hipGetLastError();
}

}  // namespace mpsk
