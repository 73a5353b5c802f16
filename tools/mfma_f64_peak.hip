// Measures the dense fp64 MFMA issue rate of the device (v_mfma_f64_16x16x4_f64), the number the
// roofline "peak" in bench.py is derived from (MI355X_MICROARCH.md has no fp64 row).
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, const double* in, int iters, unsigned long long* cyc) {
  d4 acc[NACC];
  double a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// v_fma_f64 VALU rate: NACC independent accumulator chains per lane
template <int NACC>
__global__ __launch_bounds__(256) void kfma(double* out, const double* in, int iters, unsigned long long* cyc) {
  double acc[NACC];
  double a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int i = 0; i < NACC; ++i) acc[i] = 1e-3 * i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// 4x4x4 (4 blocks) fp64 MFMA: 512 flop per instruction
template <int NACC>
__global__ __launch_bounds__(256) void k4(double* out, const double* in, int iters, unsigned long long* cyc) {
  double acc[NACC];
  double a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int KIND> void run2(int blocks, int threads, int iters, const char* tag) {
  double *out, *in; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&in, sizeof(double) * 512);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  std::vector<double> h(512);
  for (int i = 0; i < 512; ++i) h[i] = 1e-3 * ((i * 7919) % 1013 - 500);
  hipMemcpy(in, h.data(), sizeof(double) * 512, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = KIND == 0 ? kfma<NACC> : k4<NACC>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, in, iters / 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, in, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(blocks);
  hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double waves = (double)blocks * threads / 64;
  double per = KIND == 0 ? 128.0 : 512.0;
  double flops = waves * iters * NACC * per;
  printf("%s: blocks=%d nacc=%d  %.3f ms  %.2f TFLOP/s  ticks/instr(wave)=%.2f\n", tag, blocks, NACC, ms,
         flops / ms * 1e-9, (double)hc[0] / ((double)iters * NACC));
  hipFree(out); hipFree(in); hipFree(cyc);
}

template <int NACC> void run(int blocks, int threads, int iters, const char* tag) {
  double *out, *in; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&in, sizeof(double) * 512);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  std::vector<double> h(512);
  for (int i = 0; i < 512; ++i) h[i] = 1e-3 * ((i * 7919) % 1013 - 500);
  hipMemcpy(in, h.data(), sizeof(double) * 512, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, in, iters / 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(blocks);
  hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double waves = (double)blocks * threads / 64;
  double flops = waves * iters * NACC * 2048.0;
  double mfma_per_wave = (double)iters * NACC;
  printf("%s: blocks=%d threads=%d nacc=%d  %.3f ms  %.2f TFLOP/s  memtime-ticks/MFMA(wave)=%.2f\n", tag, blocks,
         threads, NACC, ms, flops / ms * 1e-9, (double)hc[0] / mfma_per_wave);
  hipFree(out); hipFree(in); hipFree(cyc);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  int cus = p.multiProcessorCount;
  run<4>(cus, 256, 20000, "1 wave/SIMD");
  run<4>(cus * 2, 256, 20000, "2 waves/SIMD");
  run<1>(cus, 256, 40000, "1 wave/SIMD dep-chain");
  run<8>(cus, 256, 10000, "1 wave/SIMD 8acc");
  run<4>(cus * 4, 256, 10000, "4 waves/SIMD");
  run2<8, 0>(cus, 256, 100000, "v_fma_f64 1 wave/SIMD");
  run2<8, 0>(cus * 2, 256, 100000, "v_fma_f64 2 waves/SIMD");
  run2<8, 0>(cus * 4, 256, 50000, "v_fma_f64 4 waves/SIMD");
  run2<8, 1>(cus, 256, 50000, "mfma_f64_4x4x4 1 wave/SIMD");
  run2<8, 1>(cus * 2, 256, 50000, "mfma_f64_4x4x4 2 waves/SIMD");
  return 0;
}
