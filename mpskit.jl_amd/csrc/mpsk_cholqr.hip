// Fast path of QRpos: shifted CholeskyQR3 on the MFMA GEMM core.
//
//   pass(X):  G = X^T X  (+ shift on the first pass)  ->  G = R^T R (blocked Cholesky, 64-wide)
//             -> R^{-1} (64x64 inverses in LDS + recursive doubling with batched GEMMs)
//             -> Q = X R^{-1}
//   A --pass(shifted)--> Q1,R1 --pass--> Q2,R2 --pass--> Q3,R3 ;  Q = Q3, R = R3 R2 R1.
//
// Every flop-heavy step is a GEMM (Gram, trailing Cholesky update, inverse doubling, Q = X R^-1),
// so a (2048 x 1024) QR costs ~1 ms instead of the ~17 ms of the LDS-panel Householder kernel
// (profiles/r01_*).  Cholesky yields diag(R) > 0 directly, which is the QRpos convention
// (TensorKit leftorth!(; alg = QRpos())); for a full-column-rank matrix the factorisation is unique,
// so this agrees with Householder QRpos to O(cond * eps).
// Robustness: the first pass is shifted (Fukaya et al., "Shifted Cholesky QR", SIAM J. Sci. Comput.
// 2020: s = 11 (mn + n(n+1)) u ||A||^2) which covers cond(A) up to ~1e15; a non-positive pivot
// or a last-pass Gram matrix far from the identity raises a device flag and the caller falls back
// to the Householder kernel (rank-deficient input needs the orthonormal completion only
// Householder provides).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <string>
#include "mpsk_internal.h"

namespace mpsk {

constexpr int CB = 64;   // Cholesky block

// G (npad x npad): keep the n x n Gram block, identity elsewhere
__global__ __launch_bounds__(256) void cq_pad_identity_kernel(double* __restrict__ G, int npad, int n) {
  const int64_t total = (int64_t)npad * npad;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % npad), c = (int)(e / npad);
    if (r >= n || c >= n) G[e] = (r == c) ? 1.0 : 0.0;
  }
}

// G += s I with s = 11 (m n + n (n+1)) u trace(G)   (trace(G) = ||A||_F^2 >= ||A||_2^2)
__global__ __launch_bounds__(256) void cq_shift_kernel(double* __restrict__ G, int npad, int n, double factor) {
  __shared__ double red[4];
  __shared__ double sh;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += G[i + (int64_t)i * npad];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sh = factor * (red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  const double s = sh;
  for (int i = threadIdx.x; i < n; i += 256) G[i + (int64_t)i * npad] += s;
}

typedef double cq_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double cq_readlane(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// Cholesky (upper, G_kk = R^T R) of one 64x64 diagonal block on the MATRIX CORES: the block lives
// in MFMA accumulators (4 waves x 2x2 tiles of 16x16); rows are eliminated four at a time:
//   (a) the owners copy rows j0..j0+3 to LDS,  (b) wave 0 factors that 4 x 64 panel with
//   cross-lane reads (lane = column),  (c) every wave applies the rank-4 update
//   acc -= P^T P with v_mfma_f64_16x16x4_f64 (K = 4 is exactly the MFMA depth).
// 16 chunks x 2 barriers instead of 64 latency-bound column steps (profiles/r01: 79 us -> ~10 us).
// Writes R_kk over G_kk (strict lower part zeroed).
__global__ __launch_bounds__(256) void cq_potrf64_mfma_kernel(double* __restrict__ G, int npad, int k,
                                                              int* __restrict__ flag) {
  __shared__ double P[2][4][CB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane >> 4, lc = lane & 15;
  double* Gk = G + (int64_t)k * CB * (npad + 1);
  cq_d4 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        acc[ti][tj][rg] = Gk[(32 * wr + 16 * ti + lr + 4 * rg) + (int64_t)(32 * wc + 16 * tj + lc) * npad];
  int bad = 0;
  for (int c = 0; c < CB / 4; ++c) {
    const int j0 = 4 * c;
    double (*Pb)[CB] = P[c & 1];
    if (wr == (j0 >> 5)) {                 // (a) rows j0 + lr live in tile row ti, register q
      const int ti = (j0 >> 4) & 1, q = (j0 >> 2) & 3;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const cq_d4 v = ti ? acc[1][tj] : acc[0][tj];
        const double e = (q == 0) ? v[0] : (q == 1) ? v[1] : (q == 2) ? v[2] : v[3];
        Pb[lr][32 * wc + 16 * tj + lc] = e;
      }
    }
    __syncthreads();
    if (wave == 0) {                       // (b) 4 x 64 panel, lane = column
      double p[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) p[t] = Pb[t][lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double piv = cq_readlane(p[t], j0 + t);
        const bool ok = (piv > 0.0) && (piv < 1.0e300);
        if (!ok) bad = 1;
        const double sq = ok ? sqrt(piv) : 1.0;
        const double rs = ok ? 1.0 / sq : 0.0;
        p[t] = (lane > j0 + t) ? p[t] * rs : (lane == j0 + t ? sq : 0.0);
#pragma unroll
        for (int u = t + 1; u < 4; ++u) {
          const double r = cq_readlane(p[t], j0 + u);
          p[u] -= r * p[t];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Pb[t][lane] = p[t];
        Gk[(j0 + t) + (int64_t)lane * npad] = p[t];
      }
    }
    __syncthreads();
    double af[2], bf[2];                   // (c) rank-4 update on the matrix cores
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) af[ti] = -Pb[lr][32 * wr + 16 * ti + lc];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) bf[tj] = Pb[lr][32 * wc + 16 * tj + lc];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
  }
  if (wave == 0 && lane == 0 && bad) atomicOr(flag, 1);
}

// panel (64 x rest, ld npad)  <-  R_kk^{-T} panel : forward substitution, one thread per column with the
// column in registers and R_kk broadcast from LDS (fully unrolled: 2016 FMAs per thread).
__global__ __launch_bounds__(256) void cq_trsm_panel_kernel(const double* __restrict__ Rkk, double* __restrict__ panel,
                                                            int npad, int rest) {
  __shared__ double Rc[CB][CB];            // Rc[i][l] = R[l][i]  (column i contiguous in l)
  __shared__ double dinv[CB];
  const int tid = threadIdx.x;
  for (int e = tid; e < CB * CB; e += 256) {
    const int l = e & 63, i = e >> 6;
    Rc[i][l] = Rkk[l + (int64_t)i * npad];
  }
  if (tid < CB) dinv[tid] = 1.0 / Rkk[tid + (int64_t)tid * npad];
  __syncthreads();
  const int col = blockIdx.x * 256 + tid;
  if (col >= rest) return;
  double* pc = panel + (int64_t)col * npad;
  double x[CB];
#pragma unroll
  for (int i = 0; i < CB; ++i) x[i] = pc[i];
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    double s = x[i];
#pragma unroll
    for (int l = 0; l < i; ++l) s -= Rc[i][l] * x[l];
    x[i] = s * dinv[i];
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) pc[i] = x[i];
}

// Rinv diagonal blocks (all at once, off the critical path): inverse of each upper-triangular R_bb.
// Thread (i0 = tid >> 6, c = tid & 63) keeps rows i0 + 4k of column c of the inverse in registers;
// step i reads row i only (x_i = Y[i][c] / R[i][i]) and updates the rows above it.
__global__ __launch_bounds__(256) void cq_diag_inverse_kernel(const double* __restrict__ R, double* __restrict__ Rinv,
                                                              int npad) {
  __shared__ double S[CB][CB + 1];
  __shared__ double rowbuf[2][CB];
  __shared__ double dinvd[CB];
  const int tid = threadIdx.x, c = tid & 63, i0 = tid >> 6;
  const double* Rb = R + (int64_t)blockIdx.x * CB * (npad + 1);
  double* Xb = Rinv + (int64_t)blockIdx.x * CB * (npad + 1);
  for (int e = tid; e < CB * CB; e += 256) S[e & 63][e >> 6] = Rb[(e & 63) + (int64_t)(e >> 6) * npad];
  __syncthreads();
  if (tid < CB) dinvd[tid] = 1.0 / S[tid][tid];
  double x[CB / 4];
#pragma unroll
  for (int kk = 0; kk < CB / 4; ++kk) x[kk] = ((i0 + 4 * kk) == c) ? 1.0 : 0.0;
  __syncthreads();
  for (int i = CB - 1; i >= 0; --i) {
    if (i0 == (i & 3)) {
      double v = 0.0;
#pragma unroll
      for (int kk = 0; kk < CB / 4; ++kk) if (kk == (i >> 2)) v = x[kk];
      rowbuf[i & 1][c] = v;
    }
    __syncthreads();
    const double xi = rowbuf[i & 1][c] * dinvd[i];
    double sv[CB / 4];
#pragma unroll
    for (int kk = 0; kk < CB / 4; ++kk) sv[kk] = S[i0 + 4 * kk][i];
#pragma unroll
    for (int kk = 0; kk < CB / 4; ++kk) {
      const int r = i0 + 4 * kk;
      const double upd = sv[kk] * xi;
      x[kk] -= (r < i && c >= i) ? upd : 0.0;
    }
  }
#pragma unroll
  for (int kk = 0; kk < CB / 4; ++kk) {
    const int r = i0 + 4 * kk;
    Xb[r + (int64_t)c * npad] = (r <= c) ? x[kk] * dinvd[r] : 0.0;
  }
}

// zero everything outside the block upper triangle of R (garbage of the trailing updates) and
// everything outside the diagonal blocks of Rinv
__global__ __launch_bounds__(256) void cq_cleanup_kernel(double* __restrict__ R, double* __restrict__ Rinv, int npad) {
  const int64_t total = (int64_t)npad * npad;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % npad), c = (int)(e / npad);
    if (r / CB > c / CB) R[e] = 0.0;
    if (r / CB != c / CB) Rinv[e] = 0.0;
  }
}

// flag |= 2 when max |G - I| over the n x n block exceeds `thresh`
__global__ __launch_bounds__(256) void cq_check_identity_kernel(const double* __restrict__ G, int npad, int n,
                                                                double thresh, int* __restrict__ flag) {
  const int64_t total = (int64_t)n * n;
  int bad = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % n), c = (int)(e / n);
    double v = G[r + (int64_t)c * npad] - (r == c ? 1.0 : 0.0);
    if (!(fabs(v) <= thresh)) bad = 1;
  }
  if (bad) atomicOr(flag, 2);
}

size_t cholqr_workspace_doubles(int m, int n) {
  int nb = (n + CB - 1) / CB, p2 = 1;
  while (p2 < nb) p2 <<= 1;
  size_t npad = (size_t)p2 * CB;
  return 5 * npad * npad + 2 * (size_t)m * n + 16;
}

static GemmArgs cq_mk(const double* A, const double* B, double* C, int M, int N, int K, int64_t lda, int64_t ldb,
                      int64_t ldc, int tA, double alpha, double beta) {
  GemmArgs g;
  std::memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.batch = 1; g.nseg = 1; g.alpha = alpha; g.beta = beta; g.transA = tA; g.transB = 0;
  return g;
}

// One CholeskyQR pass.  X: m x n (ldx).  Writes Q (m x n, ldq) and the npad x npad upper factor Rp.
static hipError_t cq_pass(int m, int n, int npad, const double* X, int ldx, double* Q, int ldq, double* Rp,
                          double* Rinv, double* T, bool shifted, bool check_identity, int* flag, hipStream_t s) {
  hipError_t e;
  GemmArgs g = cq_mk(X, X, Rp, n, n, m, ldx, ldx, npad, 1, 1.0, 0.0);      // G = X^T X
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  if (npad > n) hipLaunchKernelGGL(cq_pad_identity_kernel, dim3(512), dim3(256), 0, s, Rp, npad, n);
  if (check_identity) hipLaunchKernelGGL(cq_check_identity_kernel, dim3(256), dim3(256), 0, s, Rp, npad, n, 0.5, flag);
  if (shifted) {
    const double u = 1.1102230246251565e-16;
    const double factor = 11.0 * ((double)m * n + (double)n * (n + 1)) * u;
    hipLaunchKernelGGL(cq_shift_kernel, dim3(1), dim3(256), 0, s, Rp, npad, n, factor);
  }
  const int nb = npad / CB;
  for (int k = 0; k < nb; ++k) {
    hipLaunchKernelGGL(cq_potrf64_mfma_kernel, dim3(1), dim3(256), 0, s, Rp, npad, k, flag);
    const int rest = npad - (k + 1) * CB;
    if (rest <= 0) break;
    double* panel = Rp + (int64_t)k * CB + (int64_t)(k + 1) * CB * npad;     // rows k-block, cols > k-block
    const double* Rkk = Rp + (int64_t)k * CB * (npad + 1);
    // panel <- R_kk^{-T} panel
    hipLaunchKernelGGL(cq_trsm_panel_kernel, dim3((rest + 255) / 256), dim3(256), 0, s, Rkk, panel, npad, rest);
    // trailing -= panel^T panel
    double* trail = Rp + (int64_t)(k + 1) * CB * (npad + 1);
    g = cq_mk(panel, panel, trail, rest, rest, CB, npad, npad, npad, 1, -1.0, 1.0);
    if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  }
  hipLaunchKernelGGL(cq_diag_inverse_kernel, dim3(nb), dim3(256), 0, s, Rp, Rinv, npad);
  hipLaunchKernelGGL(cq_cleanup_kernel, dim3(1024), dim3(256), 0, s, Rp, Rinv, npad);
  // R^{-1} by recursive doubling: inv([R11 R12; 0 R22]) = [i11, -i11 R12 i22; 0, i22]
  for (int b = CB; b < npad; b <<= 1) {
    const int pairs = npad / (2 * b);
    const int64_t bs = (int64_t)2 * b * (npad + 1);
    const int64_t off12 = (int64_t)b * npad;                 // (0, b) block of a pair
    const int64_t off22 = (int64_t)b * (npad + 1);
    // T12 = R12 * i22
    g = cq_mk(Rp + off12, Rinv + off22, T + off12, b, b, b, npad, npad, npad, 0, 1.0, 0.0);
    g.batch = pairs; g.bsA = bs; g.bsB = bs; g.bsC = bs;
    if ((e = gemm_f64(g, s)) != hipSuccess) return e;
    // i12 = -i11 * T12
    g = cq_mk(Rinv, T + off12, Rinv + off12, b, b, b, npad, npad, npad, 0, -1.0, 0.0);
    g.batch = pairs; g.bsA = bs; g.bsB = bs; g.bsC = bs;
    if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  }
  // Q = X R^{-1}
  g = cq_mk(X, Rinv, Q, m, n, n, ldx, npad, ldq, 0, 1.0, 0.0);
  return gemm_f64(g, s);
}

__global__ __launch_bounds__(256) void cq_copy_upper_kernel(const double* __restrict__ Rp, int npad, int n,
                                                            double* __restrict__ R, int ldr) {
  const int64_t total = (int64_t)n * n;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % n), c = (int)(e / n);
    R[r + (int64_t)c * ldr] = (r <= c) ? Rp[r + (int64_t)c * npad] : 0.0;
  }
}

// Third pass when Q2 is already orthogonal to ~1e-7: chol(I + E) = I + U + O(E^2) with
// U = striu(E) + diag(E)/2, so  R3 = I + U,  R3^{-1} = I - U  and  Q3 = Q2 (I - U) is orthogonal
// to O(|E|^2) <= 1e-14 -- no Cholesky, no inverse.  Sets flag bit 4 when max|E| > 1e-7 (the
// caller then repeats the pass with the full Cholesky).
__global__ __launch_bounds__(256) void cq_firstorder_kernel(const double* __restrict__ G, int npad, int n,
                                                            double* __restrict__ R3, double* __restrict__ Minv,
                                                            int* __restrict__ flag) {
  const int64_t total = (int64_t)npad * npad;
  int big = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % npad), c = (int)(e / npad);
    double u = 0.0, id = (r == c) ? 1.0 : 0.0;
    if (r < n && c < n) {
      const double ev = G[e] - id;
      if (!(fabs(ev) <= 1.0e-7)) big = 1;
      u = (r < c) ? ev : (r == c ? 0.5 * ev : 0.0);
    }
    R3[e] = id + u;
    Minv[e] = id - u;
  }
  if (big) atomicOr(flag, 4);
}

static hipError_t cq_pass_firstorder(int m, int n, int npad, const double* X, int ldx, double* Q, int ldq, double* Rp,
                                     double* Rinv, double* T, int* flag, hipStream_t s) {
  hipError_t e;
  GemmArgs g = cq_mk(X, X, T, n, n, m, ldx, ldx, npad, 1, 1.0, 0.0);      // G = X^T X  (into T)
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(cq_firstorder_kernel, dim3(1024), dim3(256), 0, s, T, npad, n, Rp, Rinv, flag);
  g = cq_mk(X, Rinv, Q, m, n, n, ldx, npad, ldq, 0, 1.0, 0.0);
  return gemm_f64(g, s);
}

// Shifted CholeskyQR3, split in two halves so that two factorizations can be in flight on two streams:
//   cholqr3_enqueue  : launches everything (passes 1, 2, first-order pass 3, R product, flag copy) -- no sync
//   cholqr3_finalize : syncs the stream, repeats pass 3 with the full Cholesky if the device asked for it.
// *flag_out != 0 after finalize means "not trustworthy, fall back to Householder".
struct CqBufs { double *R1, *R2, *R3, *Rinv, *T, *Qa, *Qb; int npad; };
static CqBufs cq_bufs(int m, int n, double* ws) {
  int nb = (n + CB - 1) / CB, p2 = 1;
  while (p2 < nb) p2 <<= 1;
  CqBufs b;
  b.npad = p2 * CB;
  const size_t np2 = (size_t)b.npad * b.npad;
  b.R1 = ws; b.R2 = b.R1 + np2; b.R3 = b.R2 + np2; b.Rinv = b.R3 + np2; b.T = b.Rinv + np2;
  b.Qa = b.T + np2; b.Qb = b.Qa + (size_t)m * n;
  return b;
}
static hipError_t cq_finish(const CqBufs& b, int n, double* R, int ldr, int* d_flag, int* flag_out, hipStream_t s) {
  hipError_t e;
  GemmArgs g = cq_mk(b.R2, b.R1, b.T, b.npad, b.npad, b.npad, b.npad, b.npad, b.npad, 0, 1.0, 0.0);
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  g = cq_mk(b.R3, b.T, b.Rinv, b.npad, b.npad, b.npad, b.npad, b.npad, b.npad, 0, 1.0, 0.0);
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(cq_copy_upper_kernel, dim3(1024), dim3(256), 0, s, b.Rinv, b.npad, n, R, ldr);
  return hipMemcpyAsync(flag_out, d_flag, sizeof(int), hipMemcpyDeviceToHost, s);
}

hipError_t cholqr3_enqueue(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                           int* d_flag, int* flag_out, hipStream_t s) {
  const CqBufs b = cq_bufs(m, n, ws);
  hipError_t e;
  if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, b.npad, A, lda, b.Qa, m, b.R1, b.Rinv, b.T, true, false, d_flag, s)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, b.npad, b.Qa, m, b.Qb, m, b.R2, b.Rinv, b.T, false, false, d_flag, s)) != hipSuccess) return e;
  if ((e = cq_pass_firstorder(m, n, b.npad, b.Qb, m, Q, ldq, b.R3, b.Rinv, b.T, d_flag, s)) != hipSuccess) return e;
  return cq_finish(b, n, R, ldr, d_flag, flag_out, s);
}

hipError_t cholqr3_finalize(int m, int n, double* Q, int ldq, double* R, int ldr, double* ws, int* d_flag,
                            int* flag_out, hipStream_t s) {
  const CqBufs b = cq_bufs(m, n, ws);
  hipError_t e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  if (*flag_out == 4) {                 // Q2 not yet orthogonal to 1e-7: full third pass
    if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
    if ((e = cq_pass(m, n, b.npad, b.Qb, m, Q, ldq, b.R3, b.Rinv, b.T, false, true, d_flag, s)) != hipSuccess) return e;
    if ((e = cq_finish(b, n, R, ldr, d_flag, flag_out, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

hipError_t cholqr3(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                   int* d_flag, int* flag_out, hipStream_t s) {
  hipError_t e = cholqr3_enqueue(m, n, A, lda, Q, ldq, R, ldr, ws, d_flag, flag_out, s);
  if (e != hipSuccess) return e;
  return cholqr3_finalize(m, n, Q, ldq, R, ldr, ws, d_flag, flag_out, s);
}

}  // namespace mpsk
