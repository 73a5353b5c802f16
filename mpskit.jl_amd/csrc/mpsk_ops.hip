// HBM-bound helper kernels of libmpsk: MPO "slab mixing" between the two MFMA GEMM stages and the
// Krylov vector kernels (dot / multi-dot / axpby / multi-axpy) with 64-lane wavefront reductions.
#include <hip/hip_runtime.h>
#include "mpsk_internal.h"

namespace mpsk {

typedef double d2 __attribute__((ext_vector_type(2)));

// -------------------------------------------------------------------------------------------
// slab mixing
// -------------------------------------------------------------------------------------------
__device__ inline int64_t slab_off(const SlabIndex& ix, int j) {
  int j0 = j % ix.n0;
  int t = j / ix.n0;
  int j1 = t % ix.n1;
  int j2 = t / ix.n1;
  return j0 * ix.s0 + j1 * ix.s1 + j2 * ix.s2;
}

template <bool VEC2>
__global__ __launch_bounds__(256) void mix_kernel(const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ src,
                                                  const double* __restrict__ coef,
                                                  const double* __restrict__ coef_im,
                                                  const double* __restrict__ in, SlabIndex iin,
                                                  double* __restrict__ out, SlabIndex iout, int R, int C) {
  const int o = blockIdx.y;
  const int t0 = rowptr[o], t1 = rowptr[o + 1];
  double* op = out + slab_off(iout, o);
  if (VEC2) {
    const int R2 = R >> 1;
    const int64_t total = (int64_t)R2 * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
      int r = (int)(e % R2) * 2, c = (int)(e / R2);
      d2 acc = {0.0, 0.0};
      if (coef_im) {          // complex plan: every 16-B vector is one complex number (re, im)
        for (int t = t0; t < t1; ++t) {
          const double* ip = in + slab_off(iin, src[t]) + r + (int64_t)c * iin.ld;
          d2 v = *reinterpret_cast<const d2*>(ip);
          const double cr = coef[t], ci = coef_im[t];
          acc.x += cr * v.x - ci * v.y;
          acc.y += cr * v.y + ci * v.x;
        }
      } else {
        for (int t = t0; t < t1; ++t) {
          const double* ip = in + slab_off(iin, src[t]) + r + (int64_t)c * iin.ld;
          d2 v = *reinterpret_cast<const d2*>(ip);
          double cf = coef[t];
          acc.x += cf * v.x;
          acc.y += cf * v.y;
        }
      }
      *reinterpret_cast<d2*>(op + r + (int64_t)c * iout.ld) = acc;
    }
  } else {
    const int64_t total = (int64_t)R * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
      int r = (int)(e % R), c = (int)(e / R);
      double acc = 0.0;
      for (int t = t0; t < t1; ++t)
        acc += coef[t] * in[slab_off(iin, src[t]) + r + (int64_t)c * iin.ld];
      op[r + (int64_t)c * iout.ld] = acc;
    }
  }
}

hipError_t mix_plan_create(const std::vector<MixTerm>& terms, int n_out, int n_in, MixPlan* plan) {
  std::vector<int32_t> rowptr(n_out + 1, 0), src(terms.size());
  std::vector<double> coef(terms.size()), coef_im(terms.size());
  bool cplx = false;
  for (auto& t : terms) { rowptr[t.out + 1]++; if (t.coef_im != 0.0) cplx = true; }
  int mx = 0;
  for (int i = 0; i < n_out; ++i) { if (rowptr[i + 1] > mx) mx = rowptr[i + 1]; rowptr[i + 1] += rowptr[i]; }
  std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
  for (auto& t : terms) { int p = fill[t.out]++; src[p] = t.in; coef[p] = t.coef; coef_im[p] = t.coef_im; }
  plan->n_out = n_out; plan->n_in = n_in; plan->nnz = (int)terms.size(); plan->max_terms = mx;
  hipError_t e;
  if ((e = hipMalloc(&plan->d_rowptr, sizeof(int32_t) * (n_out + 1))) != hipSuccess) return e;
  size_t nz = terms.size() ? terms.size() : 1;
  if ((e = hipMalloc(&plan->d_src, sizeof(int32_t) * nz)) != hipSuccess) return e;
  if ((e = hipMalloc(&plan->d_coef, sizeof(double) * nz)) != hipSuccess) return e;
  if ((e = hipMemcpy(plan->d_rowptr, rowptr.data(), sizeof(int32_t) * (n_out + 1), hipMemcpyHostToDevice)) != hipSuccess) return e;
  if (terms.size()) {
    if ((e = hipMemcpy(plan->d_src, src.data(), sizeof(int32_t) * terms.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(plan->d_coef, coef.data(), sizeof(double) * terms.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if (cplx) {
      if ((e = hipMalloc(&plan->d_coef_im, sizeof(double) * nz)) != hipSuccess) return e;
      if ((e = hipMemcpy(plan->d_coef_im, coef_im.data(), sizeof(double) * terms.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

void mix_plan_destroy(MixPlan* plan) {
  if (plan->d_rowptr) (void)hipFree(plan->d_rowptr);
  if (plan->d_src) (void)hipFree(plan->d_src);
  if (plan->d_coef) (void)hipFree(plan->d_coef);
  if (plan->d_coef_im) (void)hipFree(plan->d_coef_im);
  *plan = MixPlan();
}

hipError_t mix_apply(const MixPlan& plan, const double* in, SlabIndex iin, double* out, SlabIndex iout,
                     int R, int C, hipStream_t s) {
  if (R <= 0 || C <= 0 || plan.n_out <= 0) return hipSuccess;
  bool vec2 = (R % 2 == 0) && (iin.ld % 2 == 0) && (iout.ld % 2 == 0) && (iin.s0 % 2 == 0) &&
              (iin.s1 % 2 == 0) && (iin.s2 % 2 == 0) && (iout.s0 % 2 == 0) && (iout.s1 % 2 == 0) &&
              (iout.s2 % 2 == 0) && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (plan.d_coef_im && !vec2) return hipErrorInvalidValue;   // complex slabs are (re, im) row pairs: always even / aligned
  int64_t work = vec2 ? (int64_t)(R / 2) * C : (int64_t)R * C;
  int bx = (int)((work + 255) / 256);
  int cap = (2048 + plan.n_out - 1) / plan.n_out;
  if (cap < 1) cap = 1;
  if (bx > cap * 4) bx = cap * 4;
  if (bx < 1) bx = 1;
  dim3 grid(bx, plan.n_out, 1);
  if (vec2)
    hipLaunchKernelGGL(mix_kernel<true>, grid, dim3(256), 0, s, plan.d_rowptr, plan.d_src, plan.d_coef, plan.d_coef_im, in,
                       iin, out, iout, R, C);
  else
    hipLaunchKernelGGL(mix_kernel<false>, grid, dim3(256), 0, s, plan.d_rowptr, plan.d_src, plan.d_coef, plan.d_coef_im, in,
                       iin, out, iout, R, C);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_strided_kernel(const double* __restrict__ src, int64_t srs, int64_t scs,
                                                           double* __restrict__ dst, int64_t drs, int64_t dcs, int64_t R,
                                                           int64_t C) {
  const int64_t total = R * C;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % R, c = e / R;
    dst[r * drs + c * dcs] = src[r * srs + c * scs];
  }
}
// interleaved complex (re, im, re, im, ...) -> two planes, one pass
__global__ __launch_bounds__(256) void deinterleave_kernel(const d2* __restrict__ z, double* __restrict__ re,
                                                           double* __restrict__ im, int64_t n) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const d2 v = z[e];
    re[e] = v.x;
    im[e] = v.y;
  }
}
hipError_t deinterleave(const double* z, int64_t n, double* re, double* im, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)nb), dim3(256), 0, s, reinterpret_cast<const d2*>(z), re, im, n);
  return hipGetLastError();
}

hipError_t copy_strided(const double* src, int64_t srs, int64_t scs, double* dst, int64_t drs, int64_t dcs, int64_t R,
                        int64_t C, hipStream_t s) {
  if (R <= 0 || C <= 0) return hipSuccess;
  int64_t nb = (R * C + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)nb), dim3(256), 0, s, src, srs, scs, dst, drs, dcs, R, C);
  return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// vector kernels
// -------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-level sum of NV values per thread; result valid in thread 0
template <int NV> __device__ inline void block_sum(double (&v)[NV], double* sh /* [4*NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double w = wave_sum(v[i]);
    if (lane == 0) sh[wave * NV + i] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = sh[i] + sh[NV + i] + sh[2 * NV + i] + sh[3 * NV + i];
  }
  __syncthreads();
}

constexpr int DOT_BLOCKS = 1024;
constexpr int MD = 8;  // vectors per multidot pass

struct PtrPack { const double* p[MD]; };

// partial[j*DOT_BLOCKS + b] = sum over block b's elements of xs[j] . y
template <int NVEC>
__global__ __launch_bounds__(256) void multidot_kernel(PtrPack xs, const double* __restrict__ y, int64_t n,
                                                       double* __restrict__ partial) {
  __shared__ double sh[4 * NVEC];
  double acc[NVEC];
#pragma unroll
  for (int j = 0; j < NVEC; ++j) acc[j] = 0.0;
  const int64_t n2 = n >> 1;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) {
    d2 yv = *reinterpret_cast<const d2*>(y + 2 * e);
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
      d2 xv = *reinterpret_cast<const d2*>(xs.p[j] + 2 * e);
      acc[j] += xv.x * yv.x + xv.y * yv.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < NVEC; ++j) acc[j] += xs.p[j][n - 1] * y[n - 1];
  }
  block_sum<NVEC>(acc, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < NVEC; ++j) partial[(int64_t)j * DOT_BLOCKS + blockIdx.x] = acc[j];
  }
}

// out[j] = sum_b partial[j*DOT_BLOCKS + b]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void dot_final_kernel(const double* __restrict__ partial, int nblocks,
                                                        double* __restrict__ out) {
  __shared__ double sh[4];
  double acc[1] = {0.0};
  const double* p = partial + (int64_t)blockIdx.x * DOT_BLOCKS;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) acc[0] += p[i];
  block_sum<1>(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}

static int dot_grid(int64_t n) {
  int64_t b = (n / 2 + 255) / 256;
  if (b < 1) b = 1;
  if (b > DOT_BLOCKS) b = DOT_BLOCKS;
  return (int)b;
}

// xs: HOST array of k device pointers.  d_out: device [k].  d_partial: device [MD*DOT_BLOCKS].
// Requires all vectors 16-B aligned (library allocations are).
hipError_t vec_multidot(const double* const* xs, int k, const double* y, int64_t n, double* d_out,
                        double* d_partial, hipStream_t s) {
  const int nb = dot_grid(n);
  for (int j0 = 0; j0 < k; j0 += MD) {
    int nv = k - j0 < MD ? k - j0 : MD;
    PtrPack pk;
    for (int j = 0; j < MD; ++j) pk.p[j] = xs[j0 + (j < nv ? j : 0)];
    switch (nv) {
      case 1: hipLaunchKernelGGL(multidot_kernel<1>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
      case 2: hipLaunchKernelGGL(multidot_kernel<2>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
      case 3: hipLaunchKernelGGL(multidot_kernel<3>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
      case 4: hipLaunchKernelGGL(multidot_kernel<4>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
      case 5: hipLaunchKernelGGL(multidot_kernel<5>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
      case 6: hipLaunchKernelGGL(multidot_kernel<6>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
      case 7: hipLaunchKernelGGL(multidot_kernel<7>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
      default: hipLaunchKernelGGL(multidot_kernel<8>, dim3(nb), dim3(256), 0, s, pk, y, n, d_partial); break;
    }
    hipLaunchKernelGGL(dot_final_kernel, dim3(nv), dim3(256), 0, s, d_partial, nb, d_out + j0);
  }
  return hipGetLastError();
}

// ---- fused passes of the twice-iterated classical Gram-Schmidt step -------------------------------------------------
// y <- y + sign * sum_j coefs[j] xs[j]  AND, in the same pass over memory, either the NVEC dots <xs[j], y_new>
// (NORM == false: the second-round coefficients) or <y_new, y_new> (NORM == true: the squared norm of the remainder).
// The unfused sequence multiaxpy -> multidot re-reads y and all of xs for the dots: 4k + 9 vector passes per
// orthogonalisation step against 3k + 8 with these kernels (the step is HBM-bound: ~6 TB/s in both forms).
// Same grid and per-thread element order as multidot_kernel, so the sums are bit-identical to the unfused ones.
template <int NVEC, bool NORM>
__global__ __launch_bounds__(256) void multiaxpy_dot_kernel(PtrPack xs, const double* __restrict__ coefs, double sign,
                                                            double* __restrict__ y, int64_t n, double* __restrict__ partial) {
  constexpr int NA = NORM ? 1 : NVEC;
  __shared__ double sh[4 * NA];
  double c[NVEC], acc[NA];
#pragma unroll
  for (int j = 0; j < NVEC; ++j) c[j] = sign * coefs[j];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = 0.0;
  const int64_t n2 = n >> 1;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) {
    d2 yv = *reinterpret_cast<const d2*>(y + 2 * e);
    d2 xv[NVEC];
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
      xv[j] = *reinterpret_cast<const d2*>(xs.p[j] + 2 * e);
      yv.x += c[j] * xv[j].x;
      yv.y += c[j] * xv[j].y;
    }
    *reinterpret_cast<d2*>(y + 2 * e) = yv;
    if (NORM) acc[0] += yv.x * yv.x + yv.y * yv.y;
    else {
#pragma unroll
      for (int j = 0; j < NVEC; ++j) acc[NORM ? 0 : j] += xv[j].x * yv.x + xv[j].y * yv.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double v = y[n - 1];
#pragma unroll
    for (int j = 0; j < NVEC; ++j) v += c[j] * xs.p[j][n - 1];
    y[n - 1] = v;
    if (NORM) acc[0] += v * v;
    else {
#pragma unroll
      for (int j = 0; j < NVEC; ++j) acc[NORM ? 0 : j] += xs.p[j][n - 1] * v;
    }
  }
  block_sum<NA>(acc, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) partial[(int64_t)j * DOT_BLOCKS + blockIdx.x] = acc[j];
  }
}

template <bool NORM>
static void launch_axpy_dot(int nv, int nb, const PtrPack& pk, const double* cf, double sign, double* y, int64_t n,
                            double* d_partial, hipStream_t s) {
  switch (nv) {
    case 1: hipLaunchKernelGGL((multiaxpy_dot_kernel<1, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
    case 2: hipLaunchKernelGGL((multiaxpy_dot_kernel<2, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
    case 3: hipLaunchKernelGGL((multiaxpy_dot_kernel<3, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
    case 4: hipLaunchKernelGGL((multiaxpy_dot_kernel<4, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
    case 5: hipLaunchKernelGGL((multiaxpy_dot_kernel<5, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
    case 6: hipLaunchKernelGGL((multiaxpy_dot_kernel<6, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
    case 7: hipLaunchKernelGGL((multiaxpy_dot_kernel<7, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
    default: hipLaunchKernelGGL((multiaxpy_dot_kernel<8, NORM>), dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n, d_partial); break;
  }
}

// ---- long bases (MD < k <= ML: the restarted solvers at krylovdim 30) ------------------------------------------------
// The same two fused passes.  The first has to keep an element's k operand values in registers until the element's final
// y is known (the dots of pass two use both), so it works on single doubles (k + 2k accumulators / coefficients <= 192
// VGPRs at 32 vectors) in three sizes, padded with zero coefficients on xs[0] (re-reads of a line already in L1); the
// second needs no retention and loops over k at run time.  Vector passes per CGS2 step: 3k + 5 + k/8 instead of the
// 4k + 6 k/8 + 1 of the chunked fallback (k = 24: 80 against 115; the step is HBM-bound).
constexpr int ML = 32;
struct PtrPackL { const double* p[ML]; };

template <int NVEC>
__global__ __launch_bounds__(256) void multiaxpy_dot_long_kernel(PtrPackL xs, const double* __restrict__ coefs, double sign,
                                                                 double* __restrict__ y, int64_t n, int k,
                                                                 double* __restrict__ partial) {
  __shared__ double sh[4 * NVEC];
  double c[NVEC], acc[NVEC];
#pragma unroll
  for (int j = 0; j < NVEC; ++j) { c[j] = j < k ? sign * coefs[j] : 0.0; acc[j] = 0.0; }
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double yv = y[e];
    double xv[NVEC];
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
      xv[j] = xs.p[j][e];
      yv += c[j] * xv[j];
    }
    y[e] = yv;
#pragma unroll
    for (int j = 0; j < NVEC; ++j) acc[j] += xv[j] * yv;
  }
  block_sum<NVEC>(acc, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < NVEC; ++j)
      if (j < k) partial[(int64_t)j * DOT_BLOCKS + blockIdx.x] = acc[j];
  }
}

__global__ __launch_bounds__(256) void multiaxpy_norm_long_kernel(PtrPackL xs, const double* __restrict__ coefs, double sign,
                                                                  double* __restrict__ y, int64_t n, int k,
                                                                  double* __restrict__ partial) {
  __shared__ double sh[4];
  __shared__ double cs[ML];
  if (threadIdx.x < ML) cs[threadIdx.x] = threadIdx.x < k ? sign * coefs[threadIdx.x] : 0.0;
  __syncthreads();
  double acc[1] = {0.0};
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double yv = y[e];
    for (int j = 0; j < k; ++j) yv += cs[j] * xs.p[j][e];
    y[e] = yv;
    acc[0] += yv * yv;
  }
  block_sum<1>(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}

static int dot_grid_long(int64_t n) {          // one double per thread and trip
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > DOT_BLOCKS) b = DOT_BLOCKS;
  return (int)b;
}

// One CGS2 step of y against xs[0..k) with the remainder's squared norm, d_out = {h1[k], h2[k], |y_final|^2}:
//   h1 = X^T y ; y -= X h1 (+ h2 = X^T y in the same pass) ; y -= X h2 (+ |y|^2 in the same pass).
// k <= MD vectors go through the fused kernels, MD < k <= ML through their long twins; beyond that the separate passes (a
// chunked axpy cannot carry dots of the FINAL y).  y is left un-normalised.
hipError_t vec_cgs2(const double* const* xs, int k, double* y, int64_t n, double* d_out, double* d_partial, hipStream_t s) {
  if (n <= 0 || k <= 0) return hipErrorInvalidValue;
  hipError_t e = vec_multidot(xs, k, y, n, d_out, d_partial, s);
  if (e != hipSuccess) return e;
  if (k > MD && k <= ML) {
    const int nbl = dot_grid_long(n);
    PtrPackL pk;
    for (int j = 0; j < ML; ++j) pk.p[j] = xs[j < k ? j : 0];
    if (k <= 16) hipLaunchKernelGGL(multiaxpy_dot_long_kernel<16>, dim3(nbl), dim3(256), 0, s, pk, d_out, -1.0, y, n, k, d_partial);
    else if (k <= 24) hipLaunchKernelGGL(multiaxpy_dot_long_kernel<24>, dim3(nbl), dim3(256), 0, s, pk, d_out, -1.0, y, n, k, d_partial);
    else hipLaunchKernelGGL(multiaxpy_dot_long_kernel<32>, dim3(nbl), dim3(256), 0, s, pk, d_out, -1.0, y, n, k, d_partial);
    hipLaunchKernelGGL(dot_final_kernel, dim3(k), dim3(256), 0, s, d_partial, nbl, d_out + k);
    hipLaunchKernelGGL(multiaxpy_norm_long_kernel, dim3(nbl), dim3(256), 0, s, pk, d_out + k, -1.0, y, n, k, d_partial);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, d_partial, nbl, d_out + 2 * k);
    return hipGetLastError();
  }
  if (k > MD) {
    if ((e = vec_multiaxpy(xs, d_out, k, -1.0, y, n, s)) != hipSuccess) return e;
    if ((e = vec_multidot(xs, k, y, n, d_out + k, d_partial, s)) != hipSuccess) return e;
    if ((e = vec_multiaxpy(xs, d_out + k, k, -1.0, y, n, s)) != hipSuccess) return e;
    const double* ys[1] = {y};
    return vec_multidot(ys, 1, y, n, d_out + 2 * k, d_partial, s);
  }
  const int nb = dot_grid(n);
  PtrPack pk;
  for (int j = 0; j < MD; ++j) pk.p[j] = xs[j < k ? j : 0];
  launch_axpy_dot<false>(k, nb, pk, d_out, -1.0, y, n, d_partial, s);
  hipLaunchKernelGGL(dot_final_kernel, dim3(k), dim3(256), 0, s, d_partial, nb, d_out + k);
  launch_axpy_dot<true>(k, nb, pk, d_out + k, -1.0, y, n, d_partial, s);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, d_partial, nb, d_out + 2 * k);
  return hipGetLastError();
}

// y = a*x + b*y   (b == 0: y is not read, so it may hold NaN / be uninitialised)
__global__ __launch_bounds__(256) void axpby_kernel(double a, const double* __restrict__ x, double b,
                                                    double* __restrict__ y, int64_t n) {
  const int64_t n2 = n >> 1;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) {
    d2 xv = *reinterpret_cast<const d2*>(x + 2 * e);
    d2 r;
    if (b == 0.0) { r.x = a * xv.x; r.y = a * xv.y; }
    else {
      d2 yv = *reinterpret_cast<const d2*>(y + 2 * e);
      r.x = a * xv.x + b * yv.x; r.y = a * xv.y + b * yv.y;
    }
    *reinterpret_cast<d2*>(y + 2 * e) = r;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    y[n - 1] = (b == 0.0) ? a * x[n - 1] : a * x[n - 1] + b * y[n - 1];
}

// y = (I (x) J) x on interleaved row pairs, J = [[0, -1], [1, 0]]:  y[2a] = -x[2a+1], y[2a+1] = x[2a]  (every column;
// the leading dimension is even, so the flat buffer is a sequence of (re-row, im-row) pairs).  This is "times i" for
// complex tensors carried as 2x2 real blocks on the bond indices (mpskit.jl_amd/cplx.py).
__global__ __launch_bounds__(256) void times_i_kernel(const double* __restrict__ x, double* __restrict__ y, int64_t npairs) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < npairs; e += (int64_t)gridDim.x * blockDim.x) {
    const d2 v = reinterpret_cast<const d2*>(x)[e];
    reinterpret_cast<d2*>(y)[e] = d2{-v.y, v.x};
  }
}
hipError_t vec_times_i(const double* x, double* y, int64_t n, hipStream_t s) {
  const int64_t npairs = n / 2;
  int64_t nb = (npairs + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(times_i_kernel, dim3((int)nb), dim3(256), 0, s, x, y, npairs);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void scal_kernel(double a, double* x, int64_t n) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    x[e] *= a;
}

// x *= 1/sqrt(*d_n2)  (scale factor stays on the device; 0 if the squared norm is not positive)
__global__ __launch_bounds__(256) void scal_rsqrt_dev_kernel(const double* __restrict__ d_n2, double* x, int64_t n) {
  const double n2 = *d_n2;
  const double a = (n2 > 0.0) ? 1.0 / sqrt(n2) : 0.0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    x[e] *= a;
}

// y = x / sqrt(*d_n2), out of place (one pass instead of a device copy followed by the in-place scaling)
__global__ __launch_bounds__(256) void scal_rsqrt_dev_oop_kernel(const double* __restrict__ d_n2, const double* __restrict__ x,
                                                                 double* __restrict__ y, int64_t n) {
  const double n2 = *d_n2;
  const double a = (n2 > 0.0) ? 1.0 / sqrt(n2) : 0.0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    y[e] = a * x[e];
}
hipError_t vec_scal_rsqrt_dev_oop(const double* d_n2, const double* x, double* y, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(scal_rsqrt_dev_oop_kernel, dim3((int)nb), dim3(256), 0, s, d_n2, x, y, n);
  return hipGetLastError();
}

hipError_t vec_scal_rsqrt_dev(const double* d_n2, double* x, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(scal_rsqrt_dev_kernel, dim3((int)nb), dim3(256), 0, s, d_n2, x, n);
  return hipGetLastError();
}

// ---- Ritz step of a fixed-budget Krylov solve, on the device --------------------------------------------------------
// slot holds, for step k = 0 .. m-1 at offset k * stride, the 2 (k+1) + 1 scalars mpsk_vorth_step_dev left there:
// h1[0..k], h2[0..k], |remainder|^2.  One wave builds the projected matrix H (h = h1 + h2, beta = sqrt of the last),
// cuts it at the first beta <= 1e-13 max|H| (invariant subspace: the later columns were built from a renormalised
// rounding residual), diagonalises the symmetrised H by cyclic Jacobi and writes the eigenvector of the SMALLEST
// eigenvalue (sign: positive component on the start vector) to coef[0..m) (zeros beyond the cut) and
// info = {lambda, |beta_last * s_last|, m_eff}.  The host variant (numpy eigh on the downloaded slot) costs one stream
// stall and ~150 us of idle GPU per site of a sweep; this one keeps the whole solve asynchronous.
constexpr int RITZ_MAX = 32;
__device__ __forceinline__ double ritz_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  return r * (2.0 - x * r);
}
__device__ __forceinline__ double ritz_rsq(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  return r * (1.5 - 0.5 * x * r * r);
}
__global__ __launch_bounds__(64) void ritz_small_kernel(const double* __restrict__ slot, int m, int stride,
                                                        double* __restrict__ coef, double* __restrict__ info) {
  __shared__ double A[RITZ_MAX][RITZ_MAX + 1], V[RITZ_MAX][RITZ_MAX + 1], beta[RITZ_MAX];
  __shared__ int meff_s, rp[RITZ_MAX / 2], rq[RITZ_MAX / 2];
  __shared__ double rc[RITZ_MAX / 2], rs[RITZ_MAX / 2];
  const int t = threadIdx.x;
  for (int e = t; e < RITZ_MAX * RITZ_MAX; e += 64) { A[e / RITZ_MAX][e % RITZ_MAX] = 0.0; V[e / RITZ_MAX][e % RITZ_MAX] = (e / RITZ_MAX == e % RITZ_MAX) ? 1.0 : 0.0; }
  __syncthreads();
  if (t < m) {                               // column t of H
    const int kk = t + 1;
    const double* blk = slot + (int64_t)t * stride;
    for (int j = 0; j < kk && j < m; ++j) A[j][t] = blk[j] + blk[kk + j];
    const double n2 = blk[2 * kk];
    beta[t] = n2 > 0.0 ? sqrt(n2) : 0.0;
  }
  __syncthreads();
  if (t == 0) {
    double scale = 1e-300;
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) scale = fmax(scale, fabs(A[i][j]));
    // (the sub-diagonal entries H[k+1][k] = beta[k] belong to the (m+1) x m matrix; only k + 1 < m ones lie in the square part)
    for (int k = 0; k + 1 < m; ++k) scale = fmax(scale, beta[k]);
    int me = m;
    for (int k = 0; k < m; ++k) if (beta[k] <= 1e-13 * scale) { me = k + 1; break; }
    meff_s = me;
  }
  __syncthreads();
  const int me = meff_s;
  // square part of the Arnoldi matrix: upper triangle from the dots, sub-diagonal from the betas; symmetrise
  if (t == 0) {
    for (int k = 0; k + 1 < me; ++k) A[k + 1][k] = beta[k];
    for (int i = 0; i < me; ++i)
      for (int j = i + 1; j < me; ++j) { const double v = 0.5 * (A[i][j] + A[j][i]); A[i][j] = v; A[j][i] = v; }
  }
  __syncthreads();
  // cyclic Jacobi in the round-robin (tournament) order: ne / 2 disjoint rotations per round, one lane per rotation for
  // the angles, lanes x pairs for the column / row updates (the serial (p, q) order spent ~0.5 ms in dependent sqrt / div
  // chains for m = 8 -- longer than the host round trip this kernel replaces)
  const int ne = me + (me & 1);                         // even player count; index me (if odd) is a bye
  const int half = ne / 2;
  for (int sweep = 0; sweep < 24; ++sweep) {
    // off-diagonal / diagonal weight: lane t sums row t, the wave adds the rows (the all-lanes-read-everything form cost
    // 2 m^2 dependent LDS reads per sweep -- more than the seven rotation rounds of an m = 8 sweep)
    double off = 0.0, dia = 0.0;
    if (t < me) {
      for (int j = 0; j < me; ++j) { const double v = A[t][j]; if (j != t) off += v * v; else dia += v * v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dia += __shfl_xor(dia, o, 64); }
    if (off <= 1e-30 * dia) break;                       // (uniform: the butterfly leaves the same sums in every lane)
    for (int round = 0; round < ne - 1; ++round) {
      __syncthreads();
      if (t < half) {                                    // pair t of this round: circle method with player ne - 1 fixed
        int a = (t == 0) ? ne - 1 : (round + t) % (ne - 1);
        int b = (round + ne - 1 - t) % (ne - 1);
        int pp = a < b ? a : b, qq = a < b ? b : a;
        double c = 1.0, sn = 0.0;
        if (qq < me) {
          const double apq = A[pp][qq], app = A[pp][pp], aqq = A[qq][qq];
          if (apq * apq > 1e-36 * fabs(app * aqq) && fabs(apq) > 1e-300) {
            // hardware reciprocal / reciprocal-square-root seeds + Newton steps: the IEEE divide / sqrt expansions are
            // ~40-instruction dependent chains each and five of them sit on the critical path of every round
            const double theta = (aqq - app) * 0.5 * ritz_rcp(apq);
            const double h2 = theta * theta + 1.0;
            const double hyp = h2 * ritz_rsq(h2);                                  // sqrt(theta^2 + 1)
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) * ritz_rcp(fabs(theta) + hyp);
            c = ritz_rsq(tt * tt + 1.0); sn = tt * c;
          }
        } else { qq = pp; }                              // bye: identity on (pp, pp), never applied
        rp[t] = pp; rq[t] = qq; rc[t] = c; rs[t] = sn;
      }
      __syncthreads();
      for (int e = t; e < half * me; e += 64) {          // columns p, q of A and V (e -> pair, row)
        const int r = e / me, i = e - r * me, pp = rp[r], qq = rq[r];
        if (pp == qq) continue;
        const double c = rc[r], sn = rs[r];
        const double aip = A[i][pp], aiq = A[i][qq];
        A[i][pp] = c * aip - sn * aiq; A[i][qq] = sn * aip + c * aiq;
        const double vip = V[i][pp], viq = V[i][qq];
        V[i][pp] = c * vip - sn * viq; V[i][qq] = sn * vip + c * viq;
      }
      __syncthreads();
      for (int e = t; e < half * me; e += 64) {          // rows p, q of A
        const int r = e / me, i = e - r * me, pp = rp[r], qq = rq[r];
        if (pp == qq) continue;
        const double c = rc[r], sn = rs[r];
        const double api = A[pp][i], aqi = A[qq][i];
        A[pp][i] = c * api - sn * aqi; A[qq][i] = sn * api + c * aqi;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (t == 0) {
    int best = 0;
    for (int i = 1; i < me; ++i) if (A[i][i] < A[best][best]) best = i;
    double nrm = 0.0;
    for (int i = 0; i < me; ++i) nrm += V[i][best] * V[i][best];
    double sc = 1.0 / sqrt(nrm);
    if (V[0][best] < 0.0) sc = -sc;
    for (int i = 0; i < m; ++i) coef[i] = i < me ? sc * V[i][best] : 0.0;
    if (info) { info[0] = A[best][best]; info[1] = fabs(beta[me - 1] * sc * V[me - 1][best]); info[2] = (double)me; }
  }
}

hipError_t vec_ritz_small(const double* d_slot, int m, int stride, double* d_coef, double* d_info, hipStream_t s) {
  if (m <= 0 || m > RITZ_MAX) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ritz_small_kernel, dim3(1), dim3(64), 0, s, d_slot, m, stride, d_coef, d_info);
  return hipGetLastError();
}

hipError_t vec_scal(double a, double* x, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(scal_kernel, dim3((int)nb), dim3(256), 0, s, a, x, n);
  return hipGetLastError();
}

hipError_t vec_axpby(double a, const double* x, double b, double* y, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  int64_t nb = (n / 2 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(axpby_kernel, dim3((int)nb), dim3(256), 0, s, a, x, b, y, n);
  return hipGetLastError();
}

// y += sign * sum_j coefs[j] * xs[j]    (coefs on device: produced by vec_multidot, no host sync)
template <int NVEC>
__global__ __launch_bounds__(256) void multiaxpy_kernel(PtrPack xs, const double* __restrict__ coefs, double sign,
                                                        double* __restrict__ y, int64_t n) {
  double c[NVEC];
#pragma unroll
  for (int j = 0; j < NVEC; ++j) c[j] = sign * coefs[j];
  const int64_t n2 = n >> 1;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n2; e += (int64_t)gridDim.x * blockDim.x) {
    d2 yv = *reinterpret_cast<const d2*>(y + 2 * e);
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
      d2 xv = *reinterpret_cast<const d2*>(xs.p[j] + 2 * e);
      yv.x += c[j] * xv.x;
      yv.y += c[j] * xv.y;
    }
    *reinterpret_cast<d2*>(y + 2 * e) = yv;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double v = y[n - 1];
#pragma unroll
    for (int j = 0; j < NVEC; ++j) v += c[j] * xs.p[j][n - 1];
    y[n - 1] = v;
  }
}

// ---- basis rotation of a thick restart: ys[j] = sum_i coefs[i + k j] xs[i], j < m, in ONE pass ----------------------
// (KrylovKit's shrink step rotates the Krylov basis by the Schur vectors of the projected matrix.)  As m separate linear
// combinations every output re-reads all k inputs: m (k + k/8 + 2) vector passes -- 700 at k = 30, m = 18, 2.8 ms at
// D = 1024.  Here a thread keeps the k values of its element in registers and writes the m outputs: k + m passes.  The
// coefficients are wave-uniform (scalar loads).  xs and ys must not overlap.
template <int NVEC>
__global__ __launch_bounds__(256) void multilincomb_kernel(PtrPackL xs, PtrPackL ys, const double* __restrict__ coefs, int k,
                                                           int m, int64_t n) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double xv[NVEC];
#pragma unroll
    for (int i = 0; i < NVEC; ++i) xv[i] = i < k ? xs.p[i][e] : 0.0;
    for (int j = 0; j < m; ++j) {
      const double* cj = coefs + (int64_t)j * k;
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < NVEC; ++i) acc += (i < k ? cj[i] : 0.0) * xv[i];
      const_cast<double*>(ys.p[j])[e] = acc;
    }
  }
}

hipError_t vec_multilincomb(const double* const* xs, int k, double* const* ys, int m, const double* d_coefs, int64_t n,
                            hipStream_t s) {
  if (n <= 0 || k <= 0 || m <= 0 || k > ML || m > ML) return hipErrorInvalidValue;
  PtrPackL px, py;
  for (int j = 0; j < ML; ++j) { px.p[j] = xs[j < k ? j : 0]; py.p[j] = ys[j < m ? j : 0]; }
  int64_t nb64 = (n + 255) / 256;
  if (nb64 > 4096) nb64 = 4096;
  const int nb = (int)nb64;
  if (k <= 8) hipLaunchKernelGGL(multilincomb_kernel<8>, dim3(nb), dim3(256), 0, s, px, py, d_coefs, k, m, n);
  else if (k <= 16) hipLaunchKernelGGL(multilincomb_kernel<16>, dim3(nb), dim3(256), 0, s, px, py, d_coefs, k, m, n);
  else if (k <= 24) hipLaunchKernelGGL(multilincomb_kernel<24>, dim3(nb), dim3(256), 0, s, px, py, d_coefs, k, m, n);
  else hipLaunchKernelGGL(multilincomb_kernel<32>, dim3(nb), dim3(256), 0, s, px, py, d_coefs, k, m, n);
  return hipGetLastError();
}

hipError_t vec_multiaxpy(const double* const* xs, const double* d_coefs, int k, double sign, double* y,
                         int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  int64_t nb64 = (n / 2 + 255) / 256;
  if (nb64 < 1) nb64 = 1;
  if (nb64 > 4096) nb64 = 4096;
  const int nb = (int)nb64;
  for (int j0 = 0; j0 < k; j0 += MD) {
    int nv = k - j0 < MD ? k - j0 : MD;
    PtrPack pk;
    for (int j = 0; j < MD; ++j) pk.p[j] = xs[j0 + (j < nv ? j : 0)];
    const double* cf = d_coefs + j0;
    switch (nv) {
      case 1: hipLaunchKernelGGL(multiaxpy_kernel<1>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
      case 2: hipLaunchKernelGGL(multiaxpy_kernel<2>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
      case 3: hipLaunchKernelGGL(multiaxpy_kernel<3>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
      case 4: hipLaunchKernelGGL(multiaxpy_kernel<4>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
      case 5: hipLaunchKernelGGL(multiaxpy_kernel<5>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
      case 6: hipLaunchKernelGGL(multiaxpy_kernel<6>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
      case 7: hipLaunchKernelGGL(multiaxpy_kernel<7>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
      default: hipLaunchKernelGGL(multiaxpy_kernel<8>, dim3(nb), dim3(256), 0, s, pk, cf, sign, y, n); break;
    }
  }
  return hipGetLastError();
}

}  // namespace mpsk
