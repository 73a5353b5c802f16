// Fast path of QRpos: shifted CholeskyQR3 on the MFMA GEMM core.
//
//   pass(X):  G = X^T X  (+ shift on the first pass)  ->  G = R^T R : blocked right-looking Cholesky, ONE fused
//             launch per 64-column block (cq_step_kernel: trailing update through the 64x64 inverse, Cholesky of
//             the next diagonal block and its inverse in MFMA accumulators, out-of-place panel solve)
//             -> R^{-1} (diagonal-block inverses come out of the step kernel; recursive doubling with batched GEMMs)
//             -> Q = X R^{-1}
//   A --pass(shifted)--> Q1,R1 --pass--> Q2,R2 --pass--> Q3,R3 ;  Q = Q3, R = R3 R2 R1.
//
// Every flop-heavy step is a GEMM (Gram, trailing Cholesky update, inverse doubling, Q = X R^-1),
// so a (2048 x 1024) QR costs ~2 ms instead of the ~25 ms of the LDS-panel Householder kernel
// (profiles/r01_*); the latency-bound Cholesky chain (32 launches x ~29 us) is about half of that.  Cholesky yields diag(R) > 0 directly, which is the QRpos convention
// (TensorKit leftorth!(; alg = QRpos())); for a full-column-rank matrix the factorisation is unique,
// so this agrees with Householder QRpos to O(cond * eps).
// Shift policy: the bound of Fukaya et al. (s = 11 (mn + n(n+1)) u ||A||^2) is a worst-case constant, ~1e7 u here; a shift
// that large leaves cond(Q1) ~ sqrt(s) / sigma_min, and on the benchmark sweep 28 % of the factorizations then needed a full
// third Cholesky pass.  The callers (mpsk_api.hip) therefore try shift_scale * s with shift_scale = 1e-8 first (a shift at
// the rounding level of the Gram matrix: 2 % repeats, no breakdown on the benchmark states) and repeat the factorization
// with the published shift when the device flags a breakdown; MPSK_CQ_SHIFT_SCALE overrides the first attempt's scale.
// Robustness: the first pass is shifted (Fukaya et al., "Shifted Cholesky QR", SIAM J. Sci. Comput.
// 2020: s = 11 (mn + n(n+1)) u ||A||^2) which covers cond(A) up to ~1e15; a non-positive pivot
// or a last-pass Gram matrix far from the identity raises a device flag and the caller falls back
// to the Householder kernel (rank-deficient input needs the orthonormal completion only
// Householder provides).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "mpsk_internal.h"

namespace mpsk {

constexpr int CB = 64;   // Cholesky block
// first-order third pass is accepted when every |(Q2^T Q2 - I)_ij| <= CQ_FIRSTORDER_MAX.  (Measured with MPSK_CQ_DEBUG=1 on the
// benchmark sweep: 28 % of the factorizations repeat the pass, and the rate is the same for any bound up to 1e-5 -- those
// tensors are far from the first-order regime, a second-order series would not catch them.)
constexpr double CQ_FIRSTORDER_MAX = 1.0e-7;

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

// 1/sqrt(x) and sqrt(x) for a pivot on the serial critical path of the 64-step elimination: v_rsq_f64 seed (~2^-26)
// + two Newton steps (full fp64) instead of the IEEE sqrt + division sequences (~50 dependent instructions a pivot).
__device__ __forceinline__ void piv_rsqrt(double x, double* rs_out, double* sq_out) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  double s = x * r;
  s = s + 0.5 * r * (x - s * s);            // one correction: s = sqrt(x) to the last bit or two
  *rs_out = r;
  *sq_out = s;
}

// ---- fused right-looking Cholesky step --------------------------------------------------------------
// One launch per 64-column block k (instead of potrf + trsm + syrk = three dependent launches):
//   * trailing tiles (i, j), k <= i <= j:  G_ij -= Y_i^T Y_j with Y_x = R_{k-1,k-1}^-T G_{k-1,x} formed in the tile
//     from the 64x64 triangular inverse -- the panel solve is folded into the update, the tiles only need the
//     UNSOLVED row block k-1 of G and there is no separate triangular solve on the critical path;
//   * the workgroup that owns tile (k, k) keeps the updated tile in its MFMA accumulators and factors it on
//     the spot: Cholesky of [G_kk | I] by 4-row chunks with rank-4 MFMA updates gives R_kk AND R_kk^-T
//     (Gaussian elimination of the augmented block) for the next launch;
//   * panel tiles j >= k:  R_{k-1,j} = R_{k-1,k-1}^-T G_{k-1,j}  (out of place, G's row block stays intact
//     for the trailing tiles of the same launch).
// R (out) and the Rinv diagonal blocks (out) are written; G is consumed.  LDS tiles hold element (r, c) at c * CQ_SL + r.
constexpr int CQ_SL = 66;
constexpr size_t CQ_STEP_LDS = (size_t)3 * CB * CQ_SL * sizeof(double);

typedef double cq_d2 __attribute__((ext_vector_type(2)));
// 64 x 64 tile (ld npad) -> registers (all 8 loads in flight) -> LDS
__device__ __forceinline__ void cq_fetch_tile(cq_d2 (&r)[8], const double* __restrict__ src, int npad, int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int v = tid + 256 * q, r2 = (v & 31) * 2, c = v >> 5;
    r[q] = *reinterpret_cast<const cq_d2*>(&src[r2 + (int64_t)c * npad]);
  }
}
__device__ __forceinline__ void cq_put_tile(double* __restrict__ dst, const cq_d2 (&r)[8], int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int v = tid + 256 * q, r2 = (v & 31) * 2, c = v >> 5;
    *reinterpret_cast<cq_d2*>(&dst[c * CQ_SL + r2]) = r[q];
  }
}

// acc[ti][tj] (+)= sign * (tA^T tB) sub-tile of this wave:  D(r, c) = sum_l tA(l, r) tB(l, c)
__device__ __forceinline__ void cq_prod(cq_d4 (&acc)[2][2], const double* __restrict__ tA, const double* __restrict__ tB,
                                        double sign, int wr, int wc, int fr, int fq) {
  for (int l0 = 0; l0 < CB; l0 += 4) {
    double af[2], bf[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) af[ti] = sign * tA[(32 * wr + 16 * ti + fr) * CQ_SL + l0 + fq];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) bf[tj] = tB[(32 * wc + 16 * tj + fr) * CQ_SL + l0 + fq];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
  }
}

constexpr int CQ_GS = 16;  // K-splits of the in-step Gram tiles (each workgroup: <= ceil(m / 64 / 16) products of 64^3; 8: QRpos 1.28 ms, 16: 1.22 ms)

// G (upper block triangle) = sum of the CQ_GS partial Gram matrices the step launches left in Gp, in fixed order
__global__ __launch_bounds__(256) void cq_gram_reduce_kernel(const double* __restrict__ Gp, int npad, double* __restrict__ G) {
  const int64_t np2 = (int64_t)npad * npad;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < np2; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % npad), c = (int)(e / npad);
    if (r / CB > c / CB) continue;
    double acc = 0.0;
#pragma unroll
    for (int sp = 0; sp < CQ_GS; ++sp) acc += Gp[(int64_t)sp * np2 + e];
    G[e] = acc;
  }
}

// ---- in-place triangular solve B <- B R^-1 riding on the step launches (trsm != 0) ------------------------------------
// Q = X R^-1 used to cost the recursive-doubling inverse (12 dependent small-GEMM launches) plus one GEMM AFTER the
// latency-bound factorization chain, during which the chip is mostly idle.  The right-looking block substitution
//   Q_c = (B_c - sum_{l<c} Q_l R_lc) Rinv_cc
// is executed instead by extra workgroups of the step launches themselves, two launches behind the factorization:
//   launch k:  Qt(c = k-2):  Q_c  = (B_c - Q_{c-1} R_{c-1,c}) Rinv_cc      (the last pending update folded in)
//              Ut(l = k-3):  B_j -= Q_l R_lj  for j >= l + 2
// (R row c is written by the panel tiles of launch c+1, Rinv_cc by the diagonal workgroup of launch c; every tile is one
// or two 64^3 products, shorter than the diagonal workgroup's path, so the chain is not lengthened), and the solve is
// finished two launches after the factorization.  B (m x n, ld ldb) holds a copy of X on entry and Q on exit.
// The same launches can also form the NEXT pass's Gram matrix Q^T Q (Gt tiles, column block j in launch j + 3, K split over
// CQ_GS workgroups per tile): the 88 us Gram GEMM between two passes then shrinks to one more launch and a reduction.
__device__ __forceinline__ void cq_fetch_tile_g(cq_d2 (&r)[8], const double* __restrict__ src, int ld, int row0, int col0,
                                                int m, int n, int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int v = tid + 256 * q, r2 = (v & 31) * 2, c = v >> 5;
    const int gr = row0 + r2, gc = col0 + c;
    cq_d2 t = {0.0, 0.0};
    if (gc < n) {
      const double* p = src + gr + (int64_t)gc * ld;
      if (gr < m) t.x = p[0];
      if (gr + 1 < m) t.y = p[1];
    }
    r[q] = t;
  }
}
__device__ __forceinline__ void cq_put_tile_T(double* __restrict__ dst, const cq_d2 (&r)[8], int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {                 // element (r2, c) -> position r2 * SL + c: cq_prod then contracts the COLUMN index
    const int v = tid + 256 * q, r2 = (v & 31) * 2, c = v >> 5;
    dst[r2 * CQ_SL + c] = r[q].x;
    dst[(r2 + 1) * CQ_SL + c] = r[q].y;
  }
}

__global__ __launch_bounds__(256) void cq_step_kernel(double* __restrict__ G, double* __restrict__ R,
                                                      double* __restrict__ Rinv, int npad, int k,
                                                      int* __restrict__ flag, double* __restrict__ B, int m, int n,
                                                      int ldb, int trsm, double* __restrict__ Gp) {
  extern __shared__ __attribute__((aligned(16))) double cq_sm[];
  double* sM = cq_sm;                      // R_{k-1,k-1}^-1
  double* sJ = cq_sm + CB * CQ_SL;         // G_{k-1, j}
  double* sI = cq_sm + 2 * CB * CQ_SL;     // G_{k-1, i}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, fq = lane >> 4, fr = lane & 15;
  const int nb = npad / CB, kk = k - 1, nt = (nb - k > 0) ? nb - k : 0;
  const int ntrail = (k == 0) ? 1 : nt * (nt + 1) / 2;
  const int b = blockIdx.x;
  cq_d4 acc[2][2];
  if (trsm && b >= ntrail + nt) {
    const int mt = (m + CB - 1) / CB;
    int t = b - ntrail - nt;
    const int cQ = k - 2;                                    // column block solved in this launch
    const bool hasQ = (cQ >= 0 && cQ < nb);
    if (hasQ && t < mt) {
      // ---- Qt: Q_c = (B_c - Q_{c-1} R_{c-1,c}) Rinv_cc on row tile t
      const int row0 = t * CB, col0 = cQ * CB;
      cq_d2 t0[8], t1[8], t2[8];
      cq_fetch_tile(t2, Rinv + (int64_t)cQ * CB * (npad + 1), npad, tid);
      if (cQ > 0) {
        cq_fetch_tile_g(t0, B, ldb, row0, col0 - CB, m, n, tid);                              // Q_{c-1}[rows]
        cq_fetch_tile(t1, R + (int64_t)(cQ - 1) * CB + (int64_t)cQ * CB * npad, npad, tid);   // R_{c-1,c}
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int gr = row0 + 32 * wr + 16 * ti + fq + 4 * rg, gc = col0 + 32 * wc + 16 * tj + fr;
            acc[ti][tj][rg] = (gr < m && gc < n) ? B[gr + (int64_t)gc * ldb] : 0.0;
          }
      if (cQ > 0) {
        cq_put_tile_T(sM, t0, tid);
        cq_put_tile(sJ, t1, tid);
        __syncthreads();
        cq_prod(acc, sM, sJ, -1.0, wr, wc, fr, fq);          // B_c - Q_{c-1} R_{c-1,c}
        __syncthreads();
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)                     // sI <- P^T
            sI[(32 * wr + 16 * ti + fq + 4 * rg) * CQ_SL + 32 * wc + 16 * tj + fr] = acc[ti][tj][rg];
      cq_put_tile(sJ, t2, tid);
      __syncthreads();
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = cq_d4{0.0, 0.0, 0.0, 0.0};
      cq_prod(acc, sI, sJ, 1.0, wr, wc, fr, fq);             // P Rinv_cc
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int gr = row0 + 32 * wr + 16 * ti + fq + 4 * rg, gc = col0 + 32 * wc + 16 * tj + fr;
            if (gr < m && gc < n) B[gr + (int64_t)gc * ldb] = acc[ti][tj][rg];
          }
      return;
    }
    if (hasQ) t -= mt;
    const int nU = (k - 3 >= 0 && nb - k + 1 > 0) ? (nb - k + 1) * mt : 0;
    if (t >= nU) {
      // ---- Gt: the NEXT pass's Gram matrix, column block j = k - 3 (Q_j is final since launch j + 2): tile (i, j), i <= j,
      // over the row tiles of K-split sp -> partial tile sp (summed in fixed order by cq_gram_reduce_kernel)
      t -= nU;
      const int j = k - 3, i = t / CQ_GS, sp = t % CQ_GS;
      if (Gp == nullptr || j < 0 || j >= nb || i > j) return;
      const int chunk = (mt + CQ_GS - 1) / CQ_GS;
      const int r0 = sp * chunk, r1 = (r0 + chunk < mt) ? r0 + chunk : mt;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = cq_d4{0.0, 0.0, 0.0, 0.0};
      cq_d2 t0[8], t1[8];
      if (r0 < r1) {
        cq_fetch_tile_g(t0, B, ldb, r0 * CB, i * CB, m, n, tid);
        if (i != j) cq_fetch_tile_g(t1, B, ldb, r0 * CB, j * CB, m, n, tid);
      }
      for (int rt = r0; rt < r1; ++rt) {
        __syncthreads();                                     // the previous product has finished reading sM / sJ
        cq_put_tile(sM, t0, tid);
        if (i != j) cq_put_tile(sJ, t1, tid); else cq_put_tile(sJ, t0, tid);
        __syncthreads();
        if (rt + 1 < r1) {                                   // next row tile's loads fly while the MFMAs run
          cq_fetch_tile_g(t0, B, ldb, (rt + 1) * CB, i * CB, m, n, tid);
          if (i != j) cq_fetch_tile_g(t1, B, ldb, (rt + 1) * CB, j * CB, m, n, tid);
        }
        cq_prod(acc, sM, sJ, 1.0, wr, wc, fr, fq);           // += Q_i[rt]^T Q_j[rt]
      }
      double* Gt = Gp + (int64_t)sp * npad * npad + (int64_t)i * CB + (int64_t)j * CB * npad;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            Gt[(32 * wr + 16 * ti + fq + 4 * rg) + (int64_t)(32 * wc + 16 * tj + fr) * npad] = acc[ti][tj][rg];
      return;
    }
    // ---- Ut: B_j -= Q_l R_lj for l = k - 3, j = l + 2 + t / mt, row tile t % mt
    const int l = k - 3;
    const int j = l + 2 + t / mt, rt = t % mt;
    if (l < 0 || j >= nb) return;
    const int row0 = rt * CB, col0 = j * CB;
    cq_d2 t0[8], t1[8];
    cq_fetch_tile_g(t0, B, ldb, row0, l * CB, m, n, tid);                                      // Q_l[rows]
    cq_fetch_tile(t1, R + (int64_t)l * CB + (int64_t)j * CB * npad, npad, tid);               // R_lj
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int gr = row0 + 32 * wr + 16 * ti + fq + 4 * rg, gc = col0 + 32 * wc + 16 * tj + fr;
          acc[ti][tj][rg] = (gr < m && gc < n) ? B[gr + (int64_t)gc * ldb] : 0.0;
        }
    cq_put_tile_T(sM, t0, tid);
    cq_put_tile(sJ, t1, tid);
    __syncthreads();
    cq_prod(acc, sM, sJ, -1.0, wr, wc, fr, fq);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int gr = row0 + 32 * wr + 16 * ti + fq + 4 * rg, gc = col0 + 32 * wc + 16 * tj + fr;
          if (gr < m && gc < n) B[gr + (int64_t)gc * ldb] = acc[ti][tj][rg];
        }
    return;
  }
  if (!trsm && k >= 2 && (k & 1) == 0 && b == ntrail + nt) {
    // ---- first level of the inverse, off the critical path: the pair of diagonal blocks (a, a+1) = (k-2, k-1) is
    // complete (R_{a,a+1} from the panel tiles and R_{a+1,a+1}^-1 from the diagonal workgroup of launch k-1), so
    //   Rinv_{a,a+1} = - Rinv_aa R_{a,a+1} Rinv_{a+1,a+1}
    // is formed here instead of by the two batched GEMM launches of the b = 64 doubling level after the factorisation.
    const int a = k - 2;
    cq_d2 t0[8], t1[8], t2[8];
    cq_fetch_tile(t0, Rinv + (int64_t)a * CB * (npad + 1), npad, tid);
    cq_fetch_tile(t1, R + (int64_t)a * CB + (int64_t)(a + 1) * CB * npad, npad, tid);
    cq_fetch_tile(t2, Rinv + (int64_t)(a + 1) * CB * (npad + 1), npad, tid);
#pragma unroll
    for (int q = 0; q < 8; ++q) {          // sM <- Rinv_aa^T  (cq_prod contracts the ROW index of both tiles)
      const int v = tid + 256 * q, r2 = (v & 31) * 2, c = v >> 5;
      sM[r2 * CQ_SL + c] = t0[q].x;
      sM[(r2 + 1) * CQ_SL + c] = t0[q].y;
    }
    cq_put_tile(sJ, t1, tid);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = cq_d4{0.0, 0.0, 0.0, 0.0};
    cq_prod(acc, sM, sJ, 1.0, wr, wc, fr, fq);               // P = Rinv_aa R_{a,a+1}
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)                       // sI <- P^T
          sI[(32 * wr + 16 * ti + fq + 4 * rg) * CQ_SL + 32 * wc + 16 * tj + fr] = acc[ti][tj][rg];
    cq_put_tile(sJ, t2, tid);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = cq_d4{0.0, 0.0, 0.0, 0.0};
    cq_prod(acc, sI, sJ, -1.0, wr, wc, fr, fq);              // - P Rinv_{a+1,a+1}
    double* Xb = Rinv + (int64_t)a * CB + (int64_t)(a + 1) * CB * npad;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          Xb[(32 * wr + 16 * ti + fq + 4 * rg) + (int64_t)(32 * wc + 16 * tj + fr) * npad] = acc[ti][tj][rg];
    return;
  }
  if (b >= ntrail) {                       // ---- panel tile: R_{kk, j} = Rinv_kk^T G_{kk, j}
    const int j = k + (b - ntrail);
    cq_d2 t0[8], t1[8];
    cq_fetch_tile(t0, Rinv + (int64_t)kk * CB * (npad + 1), npad, tid);
    cq_fetch_tile(t1, G + (int64_t)kk * CB + (int64_t)j * CB * npad, npad, tid);
    cq_put_tile(sM, t0, tid);
    cq_put_tile(sJ, t1, tid);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = cq_d4{0.0, 0.0, 0.0, 0.0};
    cq_prod(acc, sM, sJ, 1.0, wr, wc, fr, fq);
    double* Rb = R + (int64_t)kk * CB + (int64_t)j * CB * npad;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          Rb[(32 * wr + 16 * ti + fq + 4 * rg) + (int64_t)(32 * wc + 16 * tj + fr) * npad] = acc[ti][tj][rg];
    return;
  }
  int i = k, j = k;
  if (k > 0) {
    int ii = 0, rem = b;
    while (rem >= nt - ii) { rem -= nt - ii; ++ii; }
    i = k + ii; j = i + rem;
  }
  double* Gij = G + (int64_t)i * CB + (int64_t)j * CB * npad;
  cq_d2 t0[8], t1[8], t2[8];
  if (k > 0) {                              // all operand loads in flight before anything is consumed
    cq_fetch_tile(t0, Rinv + (int64_t)kk * CB * (npad + 1), npad, tid);                 // R_{kk,kk}^-1
    cq_fetch_tile(t1, G + (int64_t)kk * CB + (int64_t)j * CB * npad, npad, tid);
    cq_fetch_tile(t2, G + (int64_t)kk * CB + (int64_t)i * CB * npad, npad, tid);
  }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        acc[ti][tj][rg] = Gij[(32 * wr + 16 * ti + fq + 4 * rg) + (int64_t)(32 * wc + 16 * tj + fr) * npad];
  if (k > 0) {
    cq_put_tile(sM, t0, tid);
    cq_put_tile(sJ, t1, tid);
    cq_put_tile(sI, t2, tid);
    __syncthreads();
    // Y_j = R_kk^-T G_{kk,j} and Y_i = R_kk^-T G_{kk,i} (the solved panels, recomputed per tile: 2 x 64^3 MFMA
    // flops), then G_ij -= Y_i^T Y_j.  Going through the triangular inverse keeps the error at cond(R_kk) u;
    // the shorter form G_kk^-1 = R^-1 R^-T squares the condition number and broke shifted Cholesky on graded input.
    cq_d4 yj[2][2], yi[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) { yj[ti][tj] = cq_d4{0.0, 0.0, 0.0, 0.0}; yi[ti][tj] = cq_d4{0.0, 0.0, 0.0, 0.0}; }
    cq_prod(yj, sM, sJ, 1.0, wr, wc, fr, fq);
    if (i == j) {                            // diagonal tile: the two solved panels coincide (uniform branch)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) yi[ti][tj] = yj[ti][tj];
    } else {
      cq_prod(yi, sM, sI, 1.0, wr, wc, fr, fq);
    }
    __syncthreads();                                        // everyone is done reading the unsolved panels
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int o = (32 * wc + 16 * tj + fr) * CQ_SL + 32 * wr + 16 * ti + fq + 4 * rg;
          sJ[o] = yj[ti][tj][rg];
          sI[o] = yi[ti][tj][rg];
        }
    __syncthreads();
    cq_prod(acc, sI, sJ, -1.0, wr, wc, fr, fq);             // G_ij -= Y_i^T Y_j
  }
  if (b != 0) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          Gij[(32 * wr + 16 * ti + fq + 4 * rg) + (int64_t)(32 * wc + 16 * tj + fr) * npad] = acc[ti][tj][rg];
    return;
  }
  // ---- tile (k, k): Cholesky of [G_kk | I] in the accumulators -> R_kk, R_kk^-T, then M_k
  __syncthreads();                                          // LDS is reused below
  double (*P)[4][2 * CB] = reinterpret_cast<double (*)[4][2 * CB]>(cq_sm);       // P[2][4][128]
  cq_d4 accE[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        accE[ti][tj][rg] = ((32 * wr + 16 * ti + fq + 4 * rg) == (32 * wc + 16 * tj + fr)) ? 1.0 : 0.0;
  double* Rk = R + (int64_t)k * CB * (npad + 1);
  double* Xk = Rinv + (int64_t)k * CB * (npad + 1);
  int bad = 0;
  // chunk c = 8 rb + 4 ti + q with (ti, q) compile-time: the accumulator registers are indexed statically
  // (a runtime (ti, q) made the compiler spill both accumulator sets to scratch every chunk)
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 8 * rb + 4 * ti + q;
    const int j0 = 4 * c;
    double (*Pb)[2 * CB] = P[c & 1];
    if (wr == rb) {                        // rows j0 + fq live in tile row ti, register q of wave row rb
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        Pb[fq][32 * wc + 16 * tj + fr] = acc[ti][tj][q];
        Pb[fq][CB + 32 * wc + 16 * tj + fr] = accE[ti][tj][q];
      }
    }
    __syncthreads();
    if (wave == 0) {                       // 4 x (64 | 64) panel, lane = column of both halves
      double p[4], e[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { p[t] = Pb[t][lane]; e[t] = Pb[t][CB + lane]; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double piv = cq_readlane(p[t], j0 + t);
        const bool ok = (piv > 0.0) && (piv < 1.0e300);
        if (!ok) bad = 1;
        double rs_, sq_;
        piv_rsqrt(ok ? piv : 1.0, &rs_, &sq_);
        const double sq = ok ? sq_ : 1.0;
        const double rs = ok ? rs_ : 0.0;
        p[t] = (lane > j0 + t) ? p[t] * rs : (lane == j0 + t ? sq : 0.0);
        e[t] *= rs;
#pragma unroll
        for (int u = t + 1; u < 4; ++u) {
          const double r = cq_readlane(p[t], j0 + u);
          p[u] -= r * p[t];
          e[u] -= r * e[t];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Pb[t][lane] = p[t];
        Pb[t][CB + lane] = e[t];
        Rk[(j0 + t) + (int64_t)lane * npad] = p[t];          // R_kk(j0+t, lane)
        Xk[lane + (int64_t)(j0 + t) * npad] = e[t];          // R_kk^-1(lane, j0+t) = E(j0+t, lane)
      }
    }
    __syncthreads();
    double af[2], bf[2], be[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) af[t2] = -Pb[fq][32 * wr + 16 * t2 + fr];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) { bf[tj] = Pb[fq][32 * wc + 16 * tj + fr]; be[tj] = Pb[fq][CB + 32 * wc + 16 * tj + fr]; }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        acc[t2][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t2], bf[tj], acc[t2][tj], 0, 0, 0);
        accE[t2][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t2], be[tj], accE[t2][tj], 0, 0, 0);
      }
  }
  if (wave == 0 && lane == 0 && bad) atomicOr(flag, 1);
}

// zero everything outside the block upper triangle of R (garbage of the trailing updates) and everything outside the
// 128 x 128 diagonal pair blocks' upper block triangle of Rinv (diagonal blocks + the pair inverses of the step launches)
__global__ __launch_bounds__(256) void cq_cleanup_kernel(double* __restrict__ R, double* __restrict__ Rinv, int npad) {
  const int64_t total = (int64_t)npad * npad;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % npad), c = (int)(e / npad);
    if (r / CB > c / CB) R[e] = 0.0;
    if (r / CB != c / CB && !(r / (2 * CB) == c / (2 * CB) && r / CB < c / CB)) Rinv[e] = 0.0;   // keep the pair inverses
  }
}

// flag |= 2 when max |G - I| over the n x n block exceeds `thresh`
__global__ __launch_bounds__(256) void cq_check_identity_kernel(const double* __restrict__ G, int npad, int n,
                                                                double thresh, int* __restrict__ flag) {
  const int64_t total = (int64_t)n * n;
  int bad = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % n), c = (int)(e / n);
    if (r > c) continue;                      // the Gram GEMM fills the upper block triangle only
    double v = G[r + (int64_t)c * npad] - (r == c ? 1.0 : 0.0);
    if (!(fabs(v) <= thresh)) bad = 1;
  }
  if (bad) atomicOr(flag, 2);
}

// in-step triangular solve (and in-step Gram of the next pass): worth it while one launch's tiles fit about two waves of
// workgroups: 2048 x 1024 1.59 -> 1.32 ms, 1100 x 700 1.39 -> 1.13 ms; at 4096^2 there are 4096 solve tiles per launch, the
// chain is lengthened and the GEMM path is faster (23.8 vs 24.8 ms)
static bool cq_use_trsm(int m, int npad) {
  static const bool enabled = !(getenv("MPSK_CQ_TRSM") && atoi(getenv("MPSK_CQ_TRSM")) == 0);
  return enabled && (int64_t)(npad / CB) * ((m + CB - 1) / CB) <= 768;     // 2048 x 1024: 512; 4096 x 1024 (1024): GEMM path 1.84 vs 1.97 ms
}
static bool cq_use_gram(int m, int npad) {
  static const bool enabled = !(getenv("MPSK_CQ_GRAM") && atoi(getenv("MPSK_CQ_GRAM")) == 0);
  return enabled && cq_use_trsm(m, npad);
}

size_t cholqr_workspace_doubles(int m, int n) {
  int nb = (n + CB - 1) / CB, p2 = 1;
  while (p2 < nb) p2 <<= 1;
  size_t npad = (size_t)p2 * CB;
  return 5 * npad * npad + 2 * (size_t)m * n + 16 + (cq_use_gram(m, (int)npad) ? (size_t)CQ_GS * npad * npad : 0);
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
// gram_ready: T already holds the Gram matrix of X (left there by the previous pass's step launches);
// Gp_next (CQ_GS x npad x npad doubles, or null): this pass's step launches also form the partial Gram matrices of Q, and
// their sum is left in T for the next pass.
static hipError_t cq_pass(int m, int n, int npad, const double* X, int ldx, double* Q, int ldq, double* Rp,
                          double* Rinv, double* T, bool shifted, bool check_identity, int* flag, hipStream_t s,
                          double shift_scale = 1.0, bool gram_ready = false, double* Gp_next = nullptr) {
  hipError_t e;
  double* Gw = T;                                                            // Gram matrix, consumed by the factorization
  GemmArgs g = cq_mk(X, X, Gw, n, n, m, ldx, ldx, npad, 1, 1.0, 0.0);      // G = X^T X: only the upper block triangle
  g.upper_only = 1;                                                          // is read by the factorization below
  if (!gram_ready && (e = gemm_f64(g, s)) != hipSuccess) return e;
  if (npad > n) hipLaunchKernelGGL(cq_pad_identity_kernel, dim3(512), dim3(256), 0, s, Gw, npad, n);
  if (check_identity) hipLaunchKernelGGL(cq_check_identity_kernel, dim3(256), dim3(256), 0, s, Gw, npad, n, 0.5, flag);
  if (shifted) {
    const double u = 1.1102230246251565e-16;
    const double factor = shift_scale * 11.0 * ((double)m * n + (double)n * (n + 1)) * u;
    hipLaunchKernelGGL(cq_shift_kernel, dim3(1), dim3(256), 0, s, Gw, npad, n, factor);
  }
  const int nb = npad / CB;
  static std::atomic<uint64_t> step_attr{0};
  if ((e = ensure_dyn_smem(step_attr, reinterpret_cast<const void*>(cq_step_kernel), CQ_STEP_LDS)) != hipSuccess) return e;
  const bool use_trsm = cq_use_trsm(m, npad);
  if (use_trsm) {
    // Q = X R^-1 by the in-step block substitution (see cq_step_kernel): Q starts as a copy of X and is solved in place,
    // two launches behind the factorization; no explicit inverse, no separate GEMM
    if (X != Q) {
      if ((e = hipMemcpy2DAsync(Q, sizeof(double) * (size_t)ldq, X, sizeof(double) * (size_t)ldx, sizeof(double) * (size_t)m, n,
                                hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
    }
    const int mt = (m + CB - 1) / CB;
    for (int k = 0; k <= nb + (Gp_next ? 2 : 1); ++k) {
      const int nt = (nb - k > 0) ? nb - k : 0;
      int nwg = (k == 0) ? 1 : nt * (nt + 1) / 2 + nt;
      if (k - 2 >= 0 && k - 2 < nb) nwg += mt;                               // Qt(k - 2)
      if (k - 3 >= 0 && nb - k + 1 > 0) nwg += (nb - k + 1) * mt;            // Ut(k - 3): columns k - 1 .. nb - 1
      if (Gp_next && k - 3 >= 0 && k - 3 < nb) nwg += (k - 2) * CQ_GS;       // Gt(k - 3): tiles (0 .. k-3, k-3) x K-splits
      if (nwg > 0)
        hipLaunchKernelGGL(cq_step_kernel, dim3(nwg), dim3(256), CQ_STEP_LDS, s, Gw, Rp, Rinv, npad, k, flag, Q, m, n, ldq, 1,
                           Gp_next);
    }
    hipLaunchKernelGGL(cq_cleanup_kernel, dim3(1024), dim3(256), 0, s, Rp, Rinv, npad);
    if (Gp_next) hipLaunchKernelGGL(cq_gram_reduce_kernel, dim3(1024), dim3(256), 0, s, Gp_next, npad, T);   // T is free again
    return hipGetLastError();
  }
  for (int k = 0; k <= nb; ++k) {           // one fused launch per block column (see cq_step_kernel); even launches
    const int nt = nb - k;                  // k >= 2 carry one more workgroup (pair inverse), k == nb is that one alone
    const int pairwg = (k >= 2 && (k & 1) == 0) ? 1 : 0;
    const int nwg = (k == 0) ? 1 : nt * (nt + 1) / 2 + nt + pairwg;
    if (nwg > 0)
      hipLaunchKernelGGL(cq_step_kernel, dim3(nwg), dim3(256), CQ_STEP_LDS, s, Gw, Rp, Rinv, npad, k, flag, (double*)nullptr, 0, 0, 0, 0,
                         (double*)nullptr);
  }
  hipLaunchKernelGGL(cq_cleanup_kernel, dim3(1024), dim3(256), 0, s, Rp, Rinv, npad);
  // R^{-1} by recursive doubling: inv([R11 R12; 0 R22]) = [i11, -i11 R12 i22; 0, i22]
  // (computing the block columns of R^-1 as extra tiles of the step launches -- X_ic = -(sum_l X_il R_lc) X_cc -- was
  //  measured and rejected: the serial l-loop of a tile, ~2 us per term, is longer than a step from column 12 on and puts
  //  the inverse ON the critical path: 1.95 instead of 1.80 ms at 2048 x 1024, 59 instead of 29 ms at 4096^2)
  for (int b = 2 * CB; b < npad; b <<= 1) {      // (the b = 64 level comes out of the step launches)
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
  // Q = X R^{-1}   (R^-1 upper triangular: column tile n0 only needs k < n0 + 64)
  g = cq_mk(X, Rinv, Q, m, n, n, ldx, npad, ldq, 0, 1.0, 0.0);
  g.b_upper = 1;
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
                                                            int* __restrict__ flag, double thr) {
  const int64_t total = (int64_t)npad * npad;
  int big = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e % npad), c = (int)(e / npad);
    double u = 0.0, id = (r == c) ? 1.0 : 0.0;
    if (r < n && c < n && r <= c) {           // the Gram GEMM fills the upper block triangle only
      const double ev = G[e] - id;
      if (!(fabs(ev) <= 1.0e-7)) big = 1;
      u = (r < c) ? ev : 0.5 * ev;
    }
    R3[e] = id + u;
    Minv[e] = id - u;
  }
  if (big) atomicOr(flag, 4);
}

static hipError_t cq_pass_firstorder(int m, int n, int npad, const double* X, int ldx, double* Q, int ldq, double* Rp,
                                     double* Rinv, double* T, int* flag, hipStream_t s, bool gram_ready = false) {
  hipError_t e;
  GemmArgs g = cq_mk(X, X, T, n, n, m, ldx, ldx, npad, 1, 1.0, 0.0);      // G = X^T X  (into T), upper block triangle
  g.upper_only = 1;
  if (!gram_ready && (e = gemm_f64(g, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(cq_firstorder_kernel, dim3(1024), dim3(256), 0, s, T, npad, n, Rp, Rinv, flag, CQ_FIRSTORDER_MAX);
  g = cq_mk(X, Rinv, Q, m, n, n, ldx, npad, ldq, 0, 1.0, 0.0);
  g.b_upper = 1;                                                           // Minv = I - U is upper triangular
  return gemm_f64(g, s);
}

// Shifted CholeskyQR3, split in two halves so that two factorizations can be in flight on two streams:
//   cholqr3_enqueue  : launches everything (passes 1, 2, first-order pass 3, R product, flag copy) -- no sync
//   cholqr3_finalize : syncs the stream, repeats pass 3 with the full Cholesky if the device asked for it.
// *flag_out != 0 after finalize means "not trustworthy, fall back to Householder".
struct CqBufs { double *R1, *R2, *R3, *Rinv, *T, *Qa, *Qb, *Gp; int npad; };
static CqBufs cq_bufs(int m, int n, double* ws) {
  int nb = (n + CB - 1) / CB, p2 = 1;
  while (p2 < nb) p2 <<= 1;
  CqBufs b;
  b.npad = p2 * CB;
  const size_t np2 = (size_t)b.npad * b.npad;
  b.R1 = ws; b.R2 = b.R1 + np2; b.R3 = b.R2 + np2; b.Rinv = b.R3 + np2; b.T = b.Rinv + np2;
  b.Qa = b.T + np2; b.Qb = b.Qa + (size_t)m * n;
  b.Gp = cq_use_gram(m, b.npad) ? b.Qb + (((size_t)m * n + 1) & ~(size_t)1) : nullptr;   // CQ_GS partial Gram matrices
  return b;
}
static hipError_t cq_finish(const CqBufs& b, int n, double* R, int ldr, int* d_flag, int* flag_out, hipStream_t s) {
  hipError_t e;
  // products of upper triangular factors: only the upper tiles, only the k-tiles between the two diagonals
  GemmArgs g = cq_mk(b.R2, b.R1, b.T, b.npad, b.npad, b.npad, b.npad, b.npad, b.npad, 0, 1.0, 0.0);
  g.a_upper = g.b_upper = g.upper_only = 1;
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  g = cq_mk(b.R3, b.T, b.Rinv, b.npad, b.npad, b.npad, b.npad, b.npad, b.npad, 0, 1.0, 0.0);
  g.a_upper = g.b_upper = g.upper_only = 1;
  if ((e = gemm_f64(g, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(cq_copy_upper_kernel, dim3(1024), dim3(256), 0, s, b.Rinv, b.npad, n, R, ldr);
  return hipMemcpyAsync(flag_out, d_flag, sizeof(int), hipMemcpyDeviceToHost, s);
}

static void cq_dbg_count(bool rep) {
  static long n = 0, r = 0;
  if (!getenv("MPSK_CQ_DEBUG")) return;
  ++n; if (rep) ++r;
  if (n % 100 == 0) fprintf(stderr, "[cholqr3] %ld factorizations, %ld repeated the third pass (thr %.1e)\n", n, r, CQ_FIRSTORDER_MAX);
}
hipError_t cholqr3_enqueue(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                           int* d_flag, int* flag_out, hipStream_t s, double shift_scale) {
  const CqBufs b = cq_bufs(m, n, ws);
  hipError_t e;
  if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
  // (with the in-step Gram, every pass leaves the Gram matrix of its Q in T for the next one)
  const bool gr = (b.Gp != nullptr);
  if ((e = cq_pass(m, n, b.npad, A, lda, b.Qa, m, b.R1, b.Rinv, b.T, true, false, d_flag, s, shift_scale, false, b.Gp)) != hipSuccess) return e;
  if ((e = cq_pass(m, n, b.npad, b.Qa, m, b.Qb, m, b.R2, b.Rinv, b.T, false, false, d_flag, s, 1.0, gr, b.Gp)) != hipSuccess) return e;
  if ((e = cq_pass_firstorder(m, n, b.npad, b.Qb, m, Q, ldq, b.R3, b.Rinv, b.T, d_flag, s, gr)) != hipSuccess) return e;
  return cq_finish(b, n, R, ldr, d_flag, flag_out, s);
}

hipError_t cholqr3_finalize(int m, int n, double* Q, int ldq, double* R, int ldr, double* ws, int* d_flag,
                            int* flag_out, hipStream_t s) {
  const CqBufs b = cq_bufs(m, n, ws);
  hipError_t e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  cq_dbg_count(*flag_out == 4);
  if (*flag_out == 4) {                 // Q2 not yet orthogonal to 1e-7: full third pass
    if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
    if ((e = cq_pass(m, n, b.npad, b.Qb, m, Q, ldq, b.R3, b.Rinv, b.T, false, true, d_flag, s)) != hipSuccess) return e;
    if ((e = cq_finish(b, n, R, ldr, d_flag, flag_out, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

// ---- robust variant for ill-conditioned / rank-deficient input (what used to go to Householder) ------------
// X = A + delta N with N = +-1 pseudo-random and delta = u ||A||_F / sqrt(m): a backward error at rounding level
// that gives exactly dependent columns a direction of their own (sigma_min ~ u ||A||), so cond(X) <~ 1e16.
// Each SHIFTED pass divides the condition number by ~1e4..1e5 (sigma -> sigma / sqrt(sigma^2 + s)), so three (four
// for n = 4096) shifted passes + one plain pass + the first-order pass reach orthogonality 1e-15:  R = R5 R4 R3 R2 R1.
__global__ __launch_bounds__(256) void cq_sumsq_kernel(const double* __restrict__ A, int lda, int m, int n,
                                                       double* __restrict__ out) {
  __shared__ double red[4];
  const int64_t total = (int64_t)m * n;
  double acc = 0.0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const double v = A[(e % m) + (e / m) * (int64_t)lda];
    acc += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[1 + blockIdx.x] = red[0] + red[1] + red[2] + red[3];   // per-block partial
}
// out[0] = sum of the per-block partials in block order (deterministic: no floating-point atomics)
__global__ __launch_bounds__(64) void cq_sumsq_final_kernel(double* __restrict__ out, int nblocks) {
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int i = 0; i < nblocks; ++i) acc += out[1 + i];
    out[0] = acc;
  }
}
__global__ __launch_bounds__(256) void cq_perturb_kernel(const double* __restrict__ A, int lda, int m, int n,
                                                         double* __restrict__ X, int ldx, const double* __restrict__ sumsq) {
  const double delta = 1.1102230246251565e-16 * sqrt(sumsq[0] / (double)m);
  const int64_t total = (int64_t)m * n;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const unsigned r = (unsigned)(e % m), c = (unsigned)(e / m);
    unsigned h = (r * 0x9E3779B1u) ^ (c * 0x85EBCA77u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    X[r + (int64_t)c * ldx] = A[r + (int64_t)c * lda] + ((h & 1u) ? delta : -delta);
  }
}

hipError_t cholqr_robust(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                         int* d_flag, int* flag_out, hipStream_t s) {
  const CqBufs b = cq_bufs(m, n, ws);
  const int npad = b.npad;
  hipError_t e;
  double* sumsq = b.T;                       // T is free until the first pass
  if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
  hipLaunchKernelGGL(cq_sumsq_kernel, dim3(512), dim3(256), 0, s, A, lda, m, n, sumsq);
  hipLaunchKernelGGL(cq_sumsq_final_kernel, dim3(1), dim3(64), 0, s, sumsq, 512);
  hipLaunchKernelGGL(cq_perturb_kernel, dim3(1024), dim3(256), 0, s, A, lda, m, n, b.Qa, m, sumsq);
  double *acc = b.R1, *cur = b.R2, *tmp = b.R3;
  double *X = b.Qa, *Y = b.Qb;
  auto accumulate = [&]() -> hipError_t {    // acc <- cur * acc
    GemmArgs g = cq_mk(cur, acc, tmp, npad, npad, npad, npad, npad, npad, 0, 1.0, 0.0);
    g.a_upper = g.b_upper = g.upper_only = 1;
    hipError_t ee = gemm_f64(g, s);
    double* t = acc; acc = tmp; tmp = t;
    return ee;
  };
  // each shifted pass multiplies sigma_min by ~1/sqrt(shift_rel); from cond 1e16 down to <~1e4 needs
  // ceil(12 / -log10(sqrt(shift_rel))) of them: 3 at 2048 x 1024, 4 at 8192 x 4096
  const double shift_rel = 11.0 * ((double)m * n + (double)n * (n + 1)) * 1.1102230246251565e-16;
  int nshift = (int)std::ceil(12.0 / (-0.5 * std::log10(shift_rel)));
  if (nshift < 3) nshift = 3;
  if (nshift > 6) nshift = 6;
  for (int p = 0; p < nshift + 1; ++p) {
    double* Rp = (p == 0) ? acc : cur;
    if ((e = cq_pass(m, n, npad, X, m, Y, m, Rp, b.Rinv, b.T, /*shifted=*/p < nshift, false, d_flag, s)) != hipSuccess) return e;
    if (p > 0 && (e = accumulate()) != hipSuccess) return e;
    double* t = X; X = Y; Y = t;
  }
  if ((e = cq_pass_firstorder(m, n, npad, X, m, Q, ldq, cur, b.Rinv, b.T, d_flag, s)) != hipSuccess) return e;
  if ((e = accumulate()) != hipSuccess) return e;
  hipLaunchKernelGGL(cq_copy_upper_kernel, dim3(1024), dim3(256), 0, s, acc, npad, n, R, ldr);
  if ((e = hipMemcpyAsync(flag_out, d_flag, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  if (*flag_out == 4) {                      // the plain pass left more than 1e-7: one more full pass (undo the last product)
    { double* t = acc; acc = tmp; tmp = t; }
    if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
    if ((e = cq_pass(m, n, npad, X, m, Y, m, cur, b.Rinv, b.T, false, false, d_flag, s)) != hipSuccess) return e;
    if ((e = accumulate()) != hipSuccess) return e;
    if ((e = cq_pass_firstorder(m, n, npad, Y, m, Q, ldq, cur, b.Rinv, b.T, d_flag, s)) != hipSuccess) return e;
    if ((e = accumulate()) != hipSuccess) return e;
    hipLaunchKernelGGL(cq_copy_upper_kernel, dim3(1024), dim3(256), 0, s, acc, npad, n, R, ldr);
    if ((e = hipMemcpyAsync(flag_out, d_flag, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

// One SHIFTED CholeskyQR pass, Q = X R^-1 with X^T X + s I = R^T R (the published shift: positive definite for every X).
// Not a QR factorization to working accuracy (orthogonality ~ u cond(X)^2, R is not returned): the cheap
// re-conditioning step between two multiplications of the subspace iteration of mpsk_tsplit (svd mode 3), where only
// span(Q) = span(X) matters and one Cholesky chain replaces three.  X and Q must not overlap.  Enqueues only.
// shift_scale: fraction of the published shift.  The published shift (1.0) never breaks down but DAMPS every direction
// with sigma_i < ~1e-4 sigma_max (Q's singular value there is sigma_i / sqrt(sigma_i^2 + s)): repeated over the iterations
// the small wanted directions of a two-site tensor (sigma_k / sigma_1 ~ 1e-6) fade away.  The caller therefore passes the
// rounding-level scale first and repeats the pass with 1.0 when *d_flag comes back non-zero (pivot breakdown).
hipError_t cholqr1_orth(int m, int n, const double* X, int ldx, double* Q, int ldq, double* ws, int* d_flag, hipStream_t s,
                        double shift_scale) {
  const CqBufs b = cq_bufs(m, n, ws);
  hipError_t e;
  if ((e = hipMemsetAsync(d_flag, 0, sizeof(int), s)) != hipSuccess) return e;
  return cq_pass(m, n, b.npad, X, ldx, Q, ldq, b.R1, b.Rinv, b.T, /*shifted=*/true, false, d_flag, s, shift_scale);
}

hipError_t cholqr3(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                   int* d_flag, int* flag_out, hipStream_t s, double shift_scale) {
  hipError_t e = cholqr3_enqueue(m, n, A, lda, Q, ldq, R, ldr, ws, d_flag, flag_out, s, shift_scale);
  if (e != hipSuccess) return e;
  return cholqr3_finalize(m, n, Q, ldq, R, ldr, ws, d_flag, flag_out, s);
}

}  // namespace mpsk
