// diagnostic: cycles and effective clock of a lone-workgroup latency-bound loop (64x64 Cholesky step pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void probe(double* out, unsigned long long* t) {
  __shared__ double rowbuf[2][64];
  const int tid = threadIdx.x, l = tid & 63, i0 = tid >> 6;
  double a[16];
  for (int k = 0; k < 16; ++k) a[k] = 1.0 + 0.001 * (tid + k);
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int j = 0; j < 64; ++j) {
    if (i0 == (j & 3)) { double v = 0; for (int k = 0; k < 16; ++k) if (k == (j >> 2)) v = a[k]; rowbuf[j & 1][l] = v; }
    __syncthreads();
    const double* rb = rowbuf[j & 1];
    double piv = rb[j]; double dinv2 = 1.0 / piv; double sjl = rb[l] * dinv2;
    double rv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) rv[k] = rb[i0 + 4 * k];
#pragma unroll
    for (int k = 0; k < 16; ++k) { int i = i0 + 4 * k; double u = rv[k] * sjl; a[k] -= (i > j && i <= l) ? u : 0.0; }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int k = 0; k < 16; ++k) s += a[k];
  out[tid] = s;
  if (tid == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
int main() {
  double* out; unsigned long long* t; hipMalloc(&out, 8 * 256); hipMalloc(&t, 16);
  unsigned long long h[2];
  for (int rep = 0; rep < 5; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, out, t); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("rep %d: kernel %.1f us, loop %llu shader cycles, %llu realtime ticks (100MHz) -> %.1f us, clock %.0f MHz, %.0f cycles/step\n", rep, ms * 1e3, h[0], h[1], h[1] / 100.0, (double)h[0] / h[1] * 100.0, h[0] / 64.0);
  }
  return 0;
}
