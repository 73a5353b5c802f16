// LDS-fed fp64 MFMA inner loops, no global traffic: what an fp64 GEMM k-loop can sustain on this device when the
// operand fragments come out of LDS the way they do in mpsk_gemm.hip.
//   (a) v_mfma_f64_16x16x4, 2x2 register blocking (the production loop: 2 A + 2 B fragment reads per 4 MFMAs)
//   (b) v_mfma_f64_4x4x4 (4 blocks), 4x4 register blocking (4 A + 4 B fragment reads per 16 MFMAs)
// Same flops per k-step (8192 per wave), (b) reads twice the LDS bytes.  Fragment reads are conflict-free ds_read_b64
// (consecutive lanes, consecutive doubles); the lane -> matrix-element mapping does not matter for the rate.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_lds mfma_f64_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int LDS_DOUBLES = 4096;   // 32 KB

__global__ __launch_bounds__(256) void k16(double* out, const double* in, int ksteps) {
  extern __shared__ double sm[];
  for (int i = threadIdx.x; i < LDS_DOUBLES; i += 256) sm[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  d4 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
  int off = wave * 256;
  for (int s = 0; s < ksteps; ++s) {
    const double* p = sm + ((off + s * 256) & (LDS_DOUBLES - 1));
    double a0 = p[lane], a1 = p[64 + lane], b0 = p[128 + lane], b1 = p[192 + lane];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
  double r = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k4(double* out, const double* in, int ksteps) {
  extern __shared__ double sm[];
  for (int i = threadIdx.x; i < LDS_DOUBLES; i += 256) sm[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  int off = wave * 512;
  for (int s = 0; s < ksteps; ++s) {
    const double* p = sm + ((off + s * 512) & (LDS_DOUBLES - 1));
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = p[64 * i + lane]; b[i] = p[256 + 64 * i + lane]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename K>
static void run(const char* name, K kern, int wgs_per_cu, int ksteps, double* out, const double* in) {
  const int blocks = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS_DOUBLES * 8, 0, out, in, ksteps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDS_DOUBLES * 8, 0, out, in, ksteps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * ksteps * 8192.0;
  printf("%-34s %d WG/CU: %8.3f ms  %6.2f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms * 1e-9);
}

int main() {
  double *out, *in;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  hipMalloc(&in, LDS_DOUBLES * sizeof(double));
  std::vector<double> h(LDS_DOUBLES, 1e-3);
  hipMemcpy(in, h.data(), LDS_DOUBLES * sizeof(double), hipMemcpyHostToDevice);
  const int ksteps = 40000;
  for (int w : {1, 2, 4}) run("mfma_f64_16x16x4 2x2 blocking, LDS", k16, w, ksteps, out, in);
  for (int w : {1, 2, 4}) run("mfma_f64_4x4x4_4b 4x4 blocking, LDS", k4, w, ksteps, out, in);
  return 0;
}
