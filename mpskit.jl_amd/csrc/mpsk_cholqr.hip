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

// Cholesky (upper, G_kk = R^T R) and triangular inverse of one 64x64 diagonal block.
// Thread (i0 = tid >> 6, l = tid & 63) keeps its 16 elements (rows i0 + 4k of column l) in
// REGISTERS; per elimination step the owners publish the pivot row to a double-buffered LDS row
// (one barrier per step, 17 independent LDS reads) -- an LDS-resident matrix with read-modify-write
// per element is latency-bound (~120 us per block, profiles/r01), this form is ~10x faster.
// Writes R_kk back into G (strict lower part zeroed) and R_kk^{-1} into Rinv's diagonal block.
__global__ __launch_bounds__(256) void cq_potrf_diag_kernel(double* __restrict__ G, double* __restrict__ Rinv,
                                                            int npad, int k, int* __restrict__ flag) {
  __shared__ double S[CB][CB + 1];       // R (written once after the factorisation)
  __shared__ double rowbuf[2][CB];
  __shared__ double dsq[CB];
  __shared__ double dinvd[CB];           // 1 / R[i][i]
  __shared__ int bad;
  const int tid = threadIdx.x;
  const int l = tid & 63, i0 = tid >> 6;
  double* Gk = G + (int64_t)k * CB * (npad + 1);
  double* Rk = Rinv + (int64_t)k * CB * (npad + 1);
  if (tid == 0) bad = 0;
  double a[CB / 4];
#pragma unroll
  for (int kk = 0; kk < CB / 4; ++kk) a[kk] = Gk[(i0 + 4 * kk) + (int64_t)l * npad];
  for (int j = 0; j < CB; ++j) {
    if (i0 == (j & 3)) {                 // owners of row j publish it (unscaled)
      double v = 0.0;
#pragma unroll
      for (int kk = 0; kk < CB / 4; ++kk) if (kk == (j >> 2)) v = a[kk];
      rowbuf[j & 1][l] = v;
    }
    __syncthreads();
    const double* rb = rowbuf[j & 1];
    const double piv = rb[j];
    const bool ok = (piv > 0.0) && isfinite(piv);
    const double dinv2 = ok ? 1.0 / piv : 0.0;
    if (tid == 0) { const double sq = ok ? sqrt(piv) : 1.0; dsq[j] = sq; dinvd[j] = 1.0 / sq; if (!ok) bad = 1; }
    const double sjl = rb[l] * dinv2;
    double rv[CB / 4];                   // issue all LDS reads back-to-back (one wait), then branch-free math
#pragma unroll
    for (int kk = 0; kk < CB / 4; ++kk) rv[kk] = rb[i0 + 4 * kk];
#pragma unroll
    for (int kk = 0; kk < CB / 4; ++kk) {
      const int i = i0 + 4 * kk;
      const double upd = rv[kk] * sjl;
      a[kk] -= (i > j && i <= l) ? upd : 0.0;
    }
  }
  __syncthreads();
  // R[i][l] = a / dsq[i] (i < l), dsq[i] on the diagonal, 0 below
#pragma unroll
  for (int kk = 0; kk < CB / 4; ++kk) {
    const int i = i0 + 4 * kk;
    const double r = (i < l) ? a[kk] * dinvd[i] : (i == l ? dsq[i] : 0.0);
    S[i][l] = r;
    Gk[i + (int64_t)l * npad] = r;
  }
  __syncthreads();
  // inverse: x[kk] = Y[i0 + 4kk][c] (unscaled rows), step i: xi = Y[i][c] / R[i][i], rows above -= R[r][i] xi
  const int c = l;
  double x[CB / 4];
#pragma unroll
  for (int kk = 0; kk < CB / 4; ++kk) x[kk] = ((i0 + 4 * kk) == c) ? 1.0 : 0.0;
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
    Rk[r + (int64_t)c * npad] = (r <= c) ? x[kk] * dinvd[r] : 0.0;
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
  auto finish = [&]() -> hipError_t {   // R = R3 R2 R1 ; flag to host
    GemmArgs g = cq_mk(R2, R1, T, npad, npad, npad, npad, npad, npad, 0, 1.0, 0.0);
    if ((e = gemm_f64(g, s)) != hipSuccess) return e;
    g = cq_mk(R3, T, Rinv, npad, npad, npad, npad, npad, npad, 0, 1.0, 0.0);
    if ((e = gemm_f64(g, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(cq_copy_upper_kernel, dim3(1024), dim3(256), 0, s, Rinv, npad, n, R, ldr);
    if ((e = hipMemcpyAsync(flag_out, d_flag, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    return hipStreamSynchronize(s);
  };
  if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, npad, A, lda, Qa, m, R1, Rinv, T, true, false, d_flag, s)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, npad, Qa, m, Qb, m, R2, Rinv, T, false, false, d_flag, s)) != hipSuccess) return e;
  if ((e = cq_pass_firstorder(m, n, npad, Qb, m, Q, ldq, R3, Rinv, T, d_flag, s)) != hipSuccess) return e;
  if ((e = finish()) != hipSuccess) return e;
  if (*flag_out == 4) {                 // Q2 not yet orthogonal to 1e-7: full third pass
    if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
    if ((e = cq_pass(m, n, npad, Qb, m, Q, ldq, R3, Rinv, T, false, true, d_flag, s)) != hipSuccess) return e;
    if ((e = finish()) != hipSuccess) return e;
  }
  return hipGetLastError();
}

}  // namespace mpsk
