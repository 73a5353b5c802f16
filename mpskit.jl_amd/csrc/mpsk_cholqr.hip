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

// In-LDS Cholesky (upper, G_kk = R^T R) and triangular inverse of one 64x64 diagonal block.
// Writes R_kk back into G (strict lower part zeroed) and R_kk^{-1} into Rinv's diagonal block.
__global__ __launch_bounds__(256) void cq_potrf_diag_kernel(double* __restrict__ G, double* __restrict__ Rinv,
                                                            int npad, int k, int* __restrict__ flag) {
  __shared__ double S[CB][CB + 1];
  __shared__ double X[CB][CB + 1];
  __shared__ int bad;
  const int tid = threadIdx.x;
  double* Gk = G + (int64_t)k * CB * (npad + 1);
  double* Rk = Rinv + (int64_t)k * CB * (npad + 1);
  if (tid == 0) bad = 0;
  for (int e = tid; e < CB * CB; e += 256) S[e % CB][e / CB] = Gk[(e % CB) + (int64_t)(e / CB) * npad];
  __syncthreads();
  for (int j = 0; j < CB; ++j) {
    const double piv = S[j][j];
    if (!(piv > 0.0) || !isfinite(piv)) { if (tid == 0) bad = 1; }
    const double dinv = (piv > 0.0) ? 1.0 / sqrt(piv) : 0.0;
    __syncthreads();
    // row j of R
    if (tid < CB) {
      if (tid > j) S[j][tid] *= dinv;
      else if (tid == j) S[j][j] = (piv > 0.0) ? sqrt(piv) : 1.0;
    }
    __syncthreads();
    // trailing update of the upper triangle: S[i][l] -= R[j][i] R[j][l], j < i <= l
    const int rem = CB - 1 - j;
    for (int e = tid; e < rem * rem; e += 256) {
      const int i = j + 1 + e % rem, l = j + 1 + e / rem;
      if (i <= l) S[i][l] -= S[j][i] * S[j][l];
    }
    __syncthreads();
  }
  // inverse of the upper-triangular R: column c by back substitution (one thread per column)
  if (tid < CB) {
    const int c = tid;
    for (int i = CB - 1; i > c; --i) X[i][c] = 0.0;
    for (int i = c; i >= 0; --i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int l = i + 1; l <= c; ++l) s -= S[i][l] * X[l][c];
      X[i][c] = s / S[i][i];
    }
  }
  __syncthreads();
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e % CB, c = e / CB;
    Gk[r + (int64_t)c * npad] = (r <= c) ? S[r][c] : 0.0;
    Rk[r + (int64_t)c * npad] = X[r][c];
  }
  if (tid == 0 && bad) atomicOr(flag, 1);
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
    hipLaunchKernelGGL(cq_potrf_diag_kernel, dim3(1), dim3(256), 0, s, Rp, Rinv, npad, k, flag);
    const int rest = npad - (k + 1) * CB;
    if (rest <= 0) break;
    double* panel = Rp + (int64_t)k * CB + (int64_t)(k + 1) * CB * npad;     // rows k-block, cols > k-block
    const double* Rik = Rinv + (int64_t)k * CB * (npad + 1);
    // panel <- R_kk^{-T} panel  (in place: one 64-row tile reads all of its K rows before it writes)
    g = cq_mk(Rik, panel, panel, CB, rest, CB, npad, npad, npad, 1, 1.0, 0.0);
    if ((e = gemm_f64(g, s)) != hipSuccess) return e;
    // trailing -= panel^T panel
    double* trail = Rp + (int64_t)(k + 1) * CB * (npad + 1);
    g = cq_mk(panel, panel, trail, rest, rest, CB, npad, npad, npad, 1, -1.0, 1.0);
    if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  }
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

// Shifted CholeskyQR3.  *flag_out != 0 (host, after a stream sync) means "not trustworthy, fall back".
hipError_t cholqr3(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                   int* d_flag, int* flag_out, hipStream_t s) {
  int nb = (n + CB - 1) / CB, p2 = 1;
  while (p2 < nb) p2 <<= 1;
  const int npad = p2 * CB;
  const size_t np2 = (size_t)npad * npad;
  double* R1 = ws; double* R2 = R1 + np2; double* R3 = R2 + np2;
  double* Rinv = R3 + np2; double* T = Rinv + np2;
  double* Qa = T + np2; double* Qb = Qa + (size_t)m * n;
  hipError_t e;
  if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, npad, A, lda, Qa, m, R1, Rinv, T, true, false, d_flag, s)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, npad, Qa, m, Qb, m, R2, Rinv, T, false, false, d_flag, s)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, npad, Qb, m, Q, ldq, R3, Rinv, T, false, true, d_flag, s)) != hipSuccess) return e;
  // R = R3 R2 R1
  GemmArgs g = cq_mk(R2, R1, T, npad, npad, npad, npad, npad, npad, 0, 1.0, 0.0);
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  g = cq_mk(R3, T, Rinv, npad, npad, npad, npad, npad, npad, 0, 1.0, 0.0);
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(cq_copy_upper_kernel, dim3(1024), dim3(256), 0, s, Rinv, npad, n, R, ldr);
  if ((e = hipMemcpyAsync(flag_out, d_flag, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  return hipGetLastError();
}

}  // namespace mpsk
