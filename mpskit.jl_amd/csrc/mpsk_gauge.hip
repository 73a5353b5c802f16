// Gauge-step kernels of libmpsk: regularize!, QRpos / LQpos, truncated SVD.
#include <hip/hip_runtime.h>
#include <string>
#include <cstring>
#include "mpsk.h"
#include "mpsk_internal.h"

namespace mpsk {

// v[w] -= <lvec^T, v[w]> rvec      (transfermatrix.jl:70-76): coef_w = sum_{x,y} lvec[x, y] v_w[y, x].
// Phase 1: one workgroup per 32 x 32 tile of a slab; the lvec tile is read coalesced along x and transposed through LDS
// so that the v tile is read coalesced along y (the single-workgroup, stride-D2 version this replaces took 3 ms per
// 1024 x 1024 slab -- the dominant cost of every GMRES step on the regularised transfer matrix).
// Phase 2: every workgroup sums the tile partials of its slab in the same order (deterministic) and updates its chunk.
constexpr int RG_T = 32;
__global__ __launch_bounds__(256) void regularize_dot_kernel(const double* __restrict__ v, const double* __restrict__ lvec,
                                                             double* __restrict__ partial, int D1, int D2, int tx_n) {
  __shared__ double tile[RG_T][RG_T + 1];
  __shared__ double sh[4];
  const int t = blockIdx.x, w = blockIdx.y;
  const int x0 = (t % tx_n) * RG_T, y0 = (t / tx_n) * RG_T;      // v[y, x], y in [0, D1), x in [0, D2); lvec[x, y] (D2 x D1)
  const double* vw = v + (int64_t)w * D1 * D2;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;         // 8 rows of 32 threads
  for (int r = r0; r < RG_T; r += 8) {                           // lvec tile: fast index x
    const int x = x0 + c, y = y0 + r;
    tile[r][c] = (x < D2 && y < D1) ? lvec[x + (int64_t)D2 * y] : 0.0;
  }
  __syncthreads();
  double acc = 0.0;
  for (int r = r0; r < RG_T; r += 8) {                           // v tile: fast index y
    const int y = y0 + c, x = x0 + r;
    if (x < D2 && y < D1) acc += tile[c][r] * vw[y + (int64_t)D1 * x];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)w * gridDim.x + t] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void regularize_apply_kernel(double* __restrict__ v, const double* __restrict__ rvec,
                                                               const double* __restrict__ partial, int ntiles, int64_t n) {
  __shared__ double sh[4];
  __shared__ double coef;
  const int w = blockIdx.y;
  double acc = 0.0;
  for (int i = threadIdx.x; i < ntiles; i += 256) acc += partial[(int64_t)w * ntiles + i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) coef = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  const double cf = coef;
  double* vw = v + (int64_t)w * n;
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) vw[e] -= cf * rvec[e];
}

size_t regularize_workspace_doubles(int W, int D1, int D2) {
  return (size_t)W * ((D1 + RG_T - 1) / RG_T) * ((D2 + RG_T - 1) / RG_T);
}

hipError_t regularize(int W, int D1, int D2, double* v, const double* lvec, const double* rvec, double* partial, hipStream_t s) {
  const int tx_n = (D2 + RG_T - 1) / RG_T, ty_n = (D1 + RG_T - 1) / RG_T, ntiles = tx_n * ty_n;
  hipLaunchKernelGGL(regularize_dot_kernel, dim3(ntiles, W), dim3(256), 0, s, v, lvec, partial, D1, D2, tx_n);
  const int64_t n = (int64_t)D1 * D2;
  int nb = (int)((n + 2047) / 2048);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(regularize_apply_kernel, dim3(nb, W), dim3(256), 0, s, v, rvec, partial, ntiles, n);
  return hipGetLastError();
}

}  // namespace mpsk

// =============================================================================================
// QRpos: blocked Householder QR with positive diagonal  (TensorKit leftorth!(; alg = QRpos()),
// call sites src/states/orthoview.jl:56, finitemps.jl:149, ortho.jl:128-136 of the reference).
//
//  * outer blocks of NB columns; inside a block ONE workgroup (1024 threads) factors LDS-resident
//    inner panels of <= 8 columns (the whole (m-j) x 8 panel lives in the 160 KB LDS, so a column
//    step is one fused reduction (norm + dots with the remaining panel columns) + one update)
//    and applies each inner panel to the rest of the outer block in compact-WY form;
//  * trailing matrix update and the formation of Q use the MFMA GEMM core:
//        W1 = V^T C ;  W2 = T^T W1 (or T W1) ;  C -= V W2
//    with T never formed: T^{-1} = striu(V^T V) + diag(1/tau), so W2 comes from a triangular
//    solve (wy_solve_kernel);
//  * finally rows of R / columns of Q are sign-flipped so that diag(R) > 0.
// =============================================================================================
namespace mpsk {

constexpr int QR_T = 512;
constexpr int QR_NW = QR_T / 64;
constexpr int QR_NBI = 8;
constexpr int QR_KC = 2;
constexpr int QR_NB = 32;
constexpr int QR_LDS_DOUBLES = 19968;   // 156 KiB of the 160 KiB LDS
constexpr int QR_SCRATCH = 32 * QR_NW;

template <int NV> __device__ inline void block_allreduce(double (&v)[NV], double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double w = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    if (lane == 0) scratch[i * QR_NW + wave] = w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < QR_NW; ++w) s += scratch[i * QR_NW + w];
    v[i] = s;
  }
  __syncthreads();
}

// Factor columns [j0, j0+nbo) of A (rows j0..m-1).  On exit A holds R on/above the diagonal and
// the Householder vectors below; Vall[:, j0:j0+nbo] holds the explicit unit-lower-trapezoidal V
// (zeros above the diagonal, rows >= j0) and tau_all[j0:j0+nbo] the scalars.
__global__ __launch_bounds__(QR_T) void qr_block_kernel(double* __restrict__ A, int lda, int m, int j0, int nbo,
                                                        int nbi, int PS, double* __restrict__ Vall, int ldv,
                                                        double* __restrict__ tau_all) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* P = lds;                   // [nbi][PS]
  double* scratch = lds + (size_t)nbi * PS;
  __shared__ double tau_in[QR_NBI];
  const int tid = threadIdx.x;

  for (int ip = 0; ip < nbo; ip += nbi) {
    const int jc = j0 + ip;
    const int rows = m - jc;
    const int w = (nbo - ip < nbi) ? nbo - ip : nbi;
    for (int c = 0; c < w; ++c)
      for (int r = tid; r < rows; r += QR_T) P[c * PS + r] = A[(jc + r) + (int64_t)(jc + c) * lda];
    __syncthreads();

    // ---- unblocked Householder on the LDS panel ----
    for (int c = 0; c < w; ++c) {
      double part[QR_NBI];
#pragma unroll
      for (int i = 0; i < QR_NBI; ++i) part[i] = 0.0;
      for (int r = tid; r < rows; r += QR_T) {
        if (r > c) {
          const double ac = P[c * PS + r];
          part[0] += ac * ac;
#pragma unroll
          for (int k = 1; k < QR_NBI; ++k)
            if (c + k < w) part[k] += ac * P[(c + k) * PS + r];
        }
      }
      block_allreduce<QR_NBI>(part, scratch);
      const double alpha = P[c * PS + c];
      const double sigma = part[0];
      double tau, beta, scale;
      if (sigma == 0.0) { tau = 0.0; beta = alpha; scale = 0.0; }
      else {
        beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      double wk[QR_NBI];
#pragma unroll
      for (int k = 1; k < QR_NBI; ++k) wk[k] = (c + k < w) ? tau * (P[(c + k) * PS + c] + scale * part[k]) : 0.0;
      __syncthreads();  // every thread has read row c before it is modified
      for (int r = tid; r < rows; r += QR_T) {
        if (r > c) {
          const double v = P[c * PS + r] * scale;
          P[c * PS + r] = v;
#pragma unroll
          for (int k = 1; k < QR_NBI; ++k)
            if (c + k < w) P[(c + k) * PS + r] -= wk[k] * v;
        }
      }
      if (tid == 0) {
        P[c * PS + c] = beta;
#pragma unroll
        for (int k = 1; k < QR_NBI; ++k)
          if (c + k < w) P[(c + k) * PS + c] -= wk[k];
        tau_in[c] = tau;
      }
      __syncthreads();
    }

    // ---- write back: A (R + v), explicit V, tau ----
    for (int c = 0; c < w; ++c) {
      for (int r = tid; r < rows; r += QR_T) {
        const double pv = P[c * PS + r];
        A[(jc + r) + (int64_t)(jc + c) * lda] = pv;
        Vall[(jc + r) + (int64_t)(jc + c) * ldv] = (r > c) ? pv : (r == c ? 1.0 : 0.0);
      }
      for (int r = tid; r < ip; r += QR_T) Vall[(j0 + r) + (int64_t)(jc + c) * ldv] = 0.0;
    }
    if (tid < w) tau_all[jc + tid] = tau_in[tid];

    const int krem = nbo - (ip + w);
    if (krem <= 0) { __syncthreads(); continue; }

    // ---- Gram of the inner panel's vectors: g[i][c] = v_i . v_c (i < c) ----
    double g[QR_NBI * (QR_NBI - 1) / 2];
#pragma unroll
    for (int i = 0; i < QR_NBI * (QR_NBI - 1) / 2; ++i) g[i] = 0.0;
    for (int r = tid; r < rows; r += QR_T) {
      int idx = 0;
#pragma unroll
      for (int c = 1; c < QR_NBI; ++c) {
        const double vc = (c < w && r >= c) ? (r == c ? 1.0 : P[c * PS + r]) : 0.0;
#pragma unroll
        for (int i = 0; i < c; ++i) {
          if (c < w && r >= c) g[idx] += P[i * PS + r] * vc;   // r >= c > i  -> P[i][r] is v_i[r]
          ++idx;
        }
      }
    }
    block_allreduce<QR_NBI*(QR_NBI - 1) / 2>(g, scratch);

    // ---- apply (I - V T V^T)^T to the remaining columns of the outer block, KC columns at a time ----
    for (int k0 = ip + w; k0 < nbo; k0 += QR_KC) {
      const int kc = (nbo - k0 < QR_KC) ? nbo - k0 : QR_KC;
      double acc[QR_NBI * QR_KC];
#pragma unroll
      for (int i = 0; i < QR_NBI * QR_KC; ++i) acc[i] = 0.0;
      for (int r = tid; r < rows; r += QR_T) {
        double a[QR_KC];
#pragma unroll
        for (int kk = 0; kk < QR_KC; ++kk) a[kk] = (kk < kc) ? A[(jc + r) + (int64_t)(j0 + k0 + kk) * lda] : 0.0;
#pragma unroll
        for (int c = 0; c < QR_NBI; ++c) {
          if (c < w) {
            const double v = (r > c) ? P[c * PS + r] : (r == c ? 1.0 : 0.0);
#pragma unroll
            for (int kk = 0; kk < QR_KC; ++kk) acc[c * QR_KC + kk] += v * a[kk];
          }
        }
      }
      block_allreduce<QR_NBI * QR_KC>(acc, scratch);
      // W2 = T^T W : forward substitution with T^{-T} = (striu(G) + diag(1/tau))^T
#pragma unroll
      for (int kk = 0; kk < QR_KC; ++kk) {
#pragma unroll
        for (int c = 0; c < QR_NBI; ++c) {
          if (c < w) {
            double s = acc[c * QR_KC + kk];
#pragma unroll
            for (int i = 0; i < c; ++i) s -= g[c * (c - 1) / 2 + i] * acc[i * QR_KC + kk];
            acc[c * QR_KC + kk] = tau_in[c] * s;
          }
        }
      }
      for (int r = tid; r < rows; r += QR_T) {
#pragma unroll
        for (int kk = 0; kk < QR_KC; ++kk) {
          if (kk < kc) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < QR_NBI; ++c) {
              if (c < w) {
                const double v = (r > c) ? P[c * PS + r] : (r == c ? 1.0 : 0.0);
                s += v * acc[c * QR_KC + kk];
              }
            }
            A[(jc + r) + (int64_t)(j0 + k0 + kk) * lda] -= s;
          }
        }
      }
    }
    __syncthreads();
  }
}

// In-place triangular solve with U = striu(S) + diag(1/tau):
//   lowerT = 1: solve U^T X = W (forward)   -> X = T^T W
//   lowerT = 0: solve U   X = W (backward)  -> X = T W
__global__ __launch_bounds__(256) void wy_solve_kernel(const double* __restrict__ S, int lds_, const double* __restrict__ tau,
                                                       int nb, double* __restrict__ W, int ldw, int ncols, int lowerT) {
  extern __shared__ double sh[];
  double* Ss = sh;            // nb*nb
  double* ts = sh + nb * nb;  // nb
  for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) Ss[e] = S[(e % nb) + (int64_t)(e / nb) * lds_];
  for (int e = threadIdx.x; e < nb; e += blockDim.x) ts[e] = tau[e];
  __syncthreads();
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  double* w = W + (int64_t)col * ldw;
  if (lowerT) {
    for (int c = 0; c < nb; ++c) {
      double s = w[c];
      for (int i = 0; i < c; ++i) s -= Ss[i + c * nb] * w[i];
      w[c] = ts[c] * s;
    }
  } else {
    for (int c = nb - 1; c >= 0; --c) {
      double s = w[c];
      for (int i = c + 1; i < nb; ++i) s -= Ss[c + i * nb] * w[i];
      w[c] = ts[c] * s;
    }
  }
}

__global__ __launch_bounds__(256) void set_identity_kernel(double* __restrict__ Q, int ldq, int m, int n) {
  const int64_t total = (int64_t)m * n;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % m), c = (int)(e / m);
    Q[r + (int64_t)c * ldq] = (r == c) ? 1.0 : 0.0;
  }
}

// R = triu(Aw) with rows scaled by sign(diag); Q columns scaled by the same signs
__global__ __launch_bounds__(256) void qr_finish_kernel(const double* __restrict__ Aw, int lda, int m, int n,
                                                        double* __restrict__ Q, int ldq, double* __restrict__ R, int ldr) {
  const int64_t totalQ = (int64_t)m * n, totalR = (int64_t)n * n;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < totalQ + totalR; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < totalQ) {
      int r = (int)(e % m), c = (int)(e / m);
      if (Aw[c + (int64_t)c * lda] < 0.0) Q[r + (int64_t)c * ldq] = -Q[r + (int64_t)c * ldq];
    } else {
      int64_t f = e - totalQ;
      int i = (int)(f % n), j = (int)(f / n);
      double v = 0.0;
      if (i <= j) { v = Aw[i + (int64_t)j * lda]; if (Aw[i + (int64_t)i * lda] < 0.0) v = -v; }
      R[i + (int64_t)j * ldr] = v;
    }
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int ldi, int rows, int cols,
                                                        double* __restrict__ out, int ldo) {
  // out[c, r] = in[r, c]; 32x32 LDS tiles
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    int r = bx + tx, c = by + k;
    tile[k][tx] = (r < rows && c < cols) ? in[r + (int64_t)c * ldi] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int c = by + tx, r = bx + k;
    if (r < rows && c < cols) out[c + (int64_t)r * ldo] = tile[tx][k];
  }
}

hipError_t transpose(const double* in, int ldi, int rows, int cols, double* out, int ldo, hipStream_t s) {
  dim3 grid((rows + 31) / 32, (cols + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, ldi, rows, cols, out, ldo);
  return hipGetLastError();
}

size_t qrpos_workspace_doubles(int m, int n) { return (size_t)2 * m * n + (size_t)QR_NB * n * 2 + n + 64; }

// A (m x n, m >= n) -> Q (m x n), R (n x n).  ws: qrpos_workspace_doubles(m, n) doubles.
hipError_t qrpos(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                 hipStream_t s, std::string* err) {
  if (m < n) { *err = "qrpos needs m >= n"; return hipErrorInvalidValue; }
  if (m - 0 > QR_LDS_DOUBLES - QR_SCRATCH) { *err = "qrpos: more rows than fit one LDS-resident panel column"; return hipErrorInvalidValue; }
  double* Aw = ws;                         // m x n
  double* Vall = Aw + (size_t)m * n;       // m x n
  double* Sall = Vall + (size_t)m * n;     // QR_NB x n   (block j0 at column j0)
  double* Wb = Sall + (size_t)QR_NB * n;   // QR_NB x n
  double* tau = Wb + (size_t)QR_NB * n;    // n
  hipError_t e;
  if ((e = hipMemcpy2DAsync(Aw, sizeof(double) * m, A, sizeof(double) * lda, sizeof(double) * m, n,
                            hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
  static std::atomic<uint64_t> attr{0};
  if ((e = ensure_dyn_smem(attr, reinterpret_cast<const void*>(qr_block_kernel), QR_LDS_DOUBLES * 8)) != hipSuccess) return e;
  auto gemm = [&](const double* a, const double* b, double* c, int M, int N, int K, int64_t la, int64_t lb, int64_t lc,
                  int tA, double alpha, double beta) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = a; g.B = b; g.C = c; g.M = M; g.N = N; g.K = K; g.lda = la; g.ldb = lb; g.ldc = lc; g.batch = 1; g.nseg = 1;
    g.alpha = alpha; g.beta = beta; g.transA = tA; g.transB = 0;
    return gemm_f64(g, s);
  };
  for (int j0 = 0; j0 < n; j0 += QR_NB) {
    const int nbo = (n - j0 < QR_NB) ? n - j0 : QR_NB;
    const int PS = m - j0;
    int nbi = (QR_LDS_DOUBLES - QR_SCRATCH) / PS;
    if (nbi > QR_NBI) nbi = QR_NBI;
    if (nbi < 1) nbi = 1;
    size_t shmem = ((size_t)nbi * PS + QR_SCRATCH) * sizeof(double);
    hipLaunchKernelGGL(qr_block_kernel, dim3(1), dim3(QR_T), shmem, s, Aw, m, m, j0, nbo, nbi, PS, Vall, m, tau);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const double* Vb = Vall + j0 + (size_t)j0 * m;
    double* Sb = Sall + (size_t)QR_NB * j0;
    // S = V^T V  (nbo x nbo)
    if ((e = gemm(Vb, Vb, Sb, nbo, nbo, m - j0, m, m, QR_NB, 1, 1.0, 0.0)) != hipSuccess) return e;
    const int ntrail = n - (j0 + nbo);
    if (ntrail > 0) {
      double* C = Aw + j0 + (size_t)(j0 + nbo) * m;
      if ((e = gemm(Vb, C, Wb, nbo, ntrail, m - j0, m, m, QR_NB, 1, 1.0, 0.0)) != hipSuccess) return e;
      hipLaunchKernelGGL(wy_solve_kernel, dim3((ntrail + 255) / 256), dim3(256), (nbo * nbo + nbo) * sizeof(double), s,
                         Sb, QR_NB, tau + j0, nbo, Wb, QR_NB, ntrail, 1);
      if ((e = gemm(Vb, Wb, C, m - j0, ntrail, nbo, m, QR_NB, m, 0, -1.0, 1.0)) != hipSuccess) return e;
    }
  }
  // ---- form Q = H_1 ... H_k [I; 0] by backward accumulation ----
  hipLaunchKernelGGL(set_identity_kernel, dim3(1024), dim3(256), 0, s, Q, ldq, m, n);
  const int nblk = (n + QR_NB - 1) / QR_NB;
  for (int b = nblk - 1; b >= 0; --b) {
    const int j0 = b * QR_NB;
    const int nbo = (n - j0 < QR_NB) ? n - j0 : QR_NB;
    const double* Vb = Vall + j0 + (size_t)j0 * m;
    double* Sb = Sall + (size_t)QR_NB * j0;
    double* C = Q + j0 + (size_t)j0 * ldq;
    const int nc = n - j0;
    if ((e = gemm(Vb, C, Wb, nbo, nc, m - j0, m, ldq, QR_NB, 1, 1.0, 0.0)) != hipSuccess) return e;
    hipLaunchKernelGGL(wy_solve_kernel, dim3((nc + 255) / 256), dim3(256), (nbo * nbo + nbo) * sizeof(double), s, Sb,
                       QR_NB, tau + j0, nbo, Wb, QR_NB, nc, 0);
    if ((e = gemm(Vb, Wb, C, m - j0, nc, nbo, m, QR_NB, ldq, 0, -1.0, 1.0)) != hipSuccess) return e;
  }
  hipLaunchKernelGGL(qr_finish_kernel, dim3(1024), dim3(256), 0, s, Aw, m, m, n, Q, ldq, R, ldr);
  return hipGetLastError();
}

}  // namespace mpsk
