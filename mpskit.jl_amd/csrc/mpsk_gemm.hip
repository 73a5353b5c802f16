// fp64 MFMA GEMM core for gfx950 (MI355X) -- the contraction engine behind every hot-path op
// (dAC / dC / dAC2 / transfer_left / transfer_right / gauge products).
//
//   C_z[m,n] = alpha * sum_seg sum_k opA(A_z,seg)[m,k] * opB(B_z,seg)[k,n] + beta * C_z[m,n]
//
// * all matrices column-major (TensorKit / Fortran order), z = batch index (blockIdx.y),
//   "seg" = K-segments with independent base offsets: this is how the sum over MPO levels
//   (derivatives.jl:85-89, transfer.jl:188-207 in the reference) becomes ONE launch.
// * v_mfma_f64_16x16x4_f64, 4 waves (2x2) per workgroup, LDS double-buffered with register
//   prefetch of the next K-tile; the MFMA operands are swapped (n-side fragment as MFMA-A) so
//   that a lane's accumulator column index (lane&15) is the memory-contiguous m index and the
//   epilogue writes full 128-B lines.
// * LDS images are padded so that ds_read_b64 fragment reads are bank-conflict free:
//     MN-major image [BK][BMN+16]   (k-row stride == 128 B mod 256 B)
//     K-major  image [BMN][BK]      (128-B rows, column index XOR-swizzled by the row)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "mpsk_internal.h"

namespace mpsk {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int NTHREADS = 256;

template <int BMN, bool KMAJOR> struct LdsImg {
  static_assert(BK == 16, "the K-major swizzle assumes 128-byte rows");
  static constexpr int STRIDE = KMAJOR ? BK : (BMN + 16);
  static constexpr int SIZE = KMAJOR ? BMN * BK : BK * (BMN + 16);  // doubles
  // K-major rows are 128 B, unpadded, with the 8-byte column index XOR-swizzled by the row: the 16 rows
  // a fragment read touches at one k land on 16 distinct bank pairs under BOTH banking rules the
  // compiler may pick (ds_read_b64: 32 lanes / 64 banks; ds_read2_b64: 16 lanes / 32 banks).
  __device__ static inline int idx(int mn, int k) { return KMAJOR ? mn * BK + (k ^ (mn & 15)) : k * STRIDE + mn; }
};

// Loader for one operand tile (BMN x BK).  KCONTIG: element (mn,k) at base[k + mn*ld], else
// base[mn + k*ld].  The LDS image is K-major iff KCONTIG, so global reads and LDS writes are both
// 16-B vectors along the contiguous direction.
template <int BMN, bool KCONTIG, bool ALIGNED> struct TileLoader {
  static constexpr int NV = BMN * BK / 2 / NTHREADS;  // 16-B vectors per thread
  using Img = LdsImg<BMN, KCONTIG>;
  d2 r[NV];

  __device__ inline void load(const double* __restrict__ base, int64_t ld, int mn0, int k0,
                              int MN, int K, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = tid + i * NTHREADS;
      int mn, k;
      if (KCONTIG) { k = (v % (BK / 2)) * 2; mn = v / (BK / 2); }
      else         { mn = (v % (BMN / 2)) * 2; k = v / (BMN / 2); }
      int gmn = mn0 + mn, gk = k0 + k;
      if (ALIGNED) {
        // wave-uniform 64-bit base (SGPRs) + loop-invariant 32-bit per-thread byte offset: the loads take the
        // saddr + voffset form and the k loop carries no 64-bit vector address arithmetic (gemm_f64 only selects the
        // aligned kernels when 128 * ld * 8 fits 32 bits)
        const char* ub = reinterpret_cast<const char*>(KCONTIG ? base + k0 + (int64_t)mn0 * ld : base + mn0 + (int64_t)k0 * ld);
        const uint32_t tob = (KCONTIG ? (uint32_t)k + (uint32_t)mn * (uint32_t)ld : (uint32_t)mn + (uint32_t)k * (uint32_t)ld) * 8u;
        r[i] = *reinterpret_cast<const d2*>(ub + tob);
      } else {
        d2 t = {0.0, 0.0};
        if (KCONTIG) {
          if (gmn < MN) {
            const double* p = base + gk + (int64_t)gmn * ld;
            if (gk < K) t.x = p[0];
            if (gk + 1 < K) t.y = p[1];
          }
        } else {
          if (gk < K) {
            const double* p = base + gmn + (int64_t)gk * ld;
            if (gmn < MN) t.x = p[0];
            if (gmn + 1 < MN) t.y = p[1];
          }
        }
        r[i] = t;
      }
    }
  }
  // jf (complex kernels only): every 16-B vector is one complex number (re, im) -- along M for an MN-contiguous
  // operand, along K for a K-contiguous one -- and is multiplied by i (jf = 1) or -i (jf = 2) on its way into LDS
  __device__ inline void store(double* __restrict__ lds, int tid, int jf = 0) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int v = tid + i * NTHREADS;
      int mn, k;
      if (KCONTIG) { k = (v % (BK / 2)) * 2; mn = v / (BK / 2); }
      else         { mn = (v % (BMN / 2)) * 2; k = v / (BMN / 2); }
      d2 w = r[i];
      if (jf == 1) w = d2{-w.y, w.x};
      else if (jf == 2) w = d2{w.y, -w.x};
      if (KCONTIG) {   // (k, k+1), k even, share one 16-B slot of the swizzled row; odd rows swap the halves
        if (mn & 1) w = d2{w.y, w.x};
        *reinterpret_cast<d2*>(&lds[Img::idx(mn, k) & ~1]) = w;
      } else {
        *reinterpret_cast<d2*>(&lds[Img::idx(mn, k)]) = w;
      }
    }
  }
};

// Accumulate the flattened k-tile range [ktb, kte) (k-tile t = segment t / tps, offset (t % tps) * BK)
// of the output tile at (m0, n0) into the MFMA accumulators.  Ends with a barrier: LDS is reusable.
// Compile-time experiment (-DMPSK_MFMA_4X4X4=1): the k loop on v_mfma_f64_4x4x4 instead of v_mfma_f64_16x16x4.
// MEASURED AND REJECTED on MI355X: bit-correct (all 142 GPU tests pass) but 42 instead of 56 TFLOP/s on the dAC stage
// kernels at D = 1024 (bench.py 0.446 instead of 0.545 sweeps/s), although the LDS-fed microbenchmark
// (tools/mfma_f64_lds.hip) sustains 68 vs 46 TFLOP/s for the two shapes: four times the MFMA issue slots plus the DPP
// rotations cost more than the shape's higher issue rate returns -- with the scheduling barriers that keep the ds_reads
// ahead of the MFMAs and without them (-DMPSK_MFMA_4X4X4_FREE_SCHED=1: 41-42 TFLOP/s as well).
#ifndef MPSK_MFMA_4X4X4
#define MPSK_MFMA_4X4X4 0
#endif
#ifndef MPSK_ROT1            // DPP row_ror amounts that bring block q + 1 / q + 3 of a 16-lane row into block q
#define MPSK_ROT1 0x12C
#define MPSK_ROT3 0x124
#endif
// rotate a double inside every 16-lane row (DPP row_ror:n = control 0x120 | n on the two 32-bit halves)
template <int CTRL>
__device__ __forceinline__ double row_rot(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// Wave-uniform 64-bit load through the scalar cache.  The per-batch K-segment tables (GemmArgs::zsegA / zsegB) are read
// inside the share loop of the stream-K body, after global stores of the previous share: the compiler can then no longer
// prove the table unclobbered and falls back to VECTOR loads followed by s_waitcnt vmcnt(0) -- which also drains the
// prefetched operand tiles (measured: 25 % of the wave cycles of the split-K stage kernel spent in s_waitcnt).
__device__ __forceinline__ int64_t uniform_load_i64(const int64_t* p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const int64_t* sp = reinterpret_cast<const int64_t*>(((uint64_t)hi << 32) | lo);
  int64_t v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(sp));
  return v;
}

template <int BM, int BN, bool TA, bool TB, bool ALIGNED, bool ZS = false, bool CJ = false>
__device__ __forceinline__ void gemm_accumulate(const GemmArgs& g, const double* __restrict__ Ab,
                                                const double* __restrict__ Bb, int z, int m0, int n0, int ktb, int kte,
                                                d4 (&acc)[BM / 32][BN / 32], double* smem) {
  constexpr int WTM = BM / 2, WTN = BN / 2;   // wave tile (2x2 waves)
  constexpr int TM = WTM / 16, TN = WTN / 16; // MFMA tiles per wave
  // A is "k-contiguous" when transposed, B is "k-contiguous" when NOT transposed
  using LA = TileLoader<BM, TA, ALIGNED>;
  using LB = TileLoader<BN, !TB, ALIGNED>;
  using IA = typename LA::Img;
  using IB = typename LB::Img;
  double* const sA = smem;                  // two A images, then two B images
  double* const sB = smem + 2 * IA::SIZE;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int tps = (g.K + BK - 1) / BK;  // k-tiles per segment
  LA la; LB lb;
  int seg = ktb / tps, kt = ktb % tps;
  // K-segment offsets: from the argument block, or (ZS, compile time: a run-time choice cost the plain kernels 30 %)
  // per batch from device tables -- uniform scalar loads
  const int64_t* const zsa = ZS ? g.zsegA + (int64_t)z * g.nseg : nullptr;
  const int64_t* const zsb = ZS ? g.zsegB + (int64_t)z * g.nseg : nullptr;
  auto offA = [&](int sg) -> int64_t { if constexpr (ZS) return uniform_load_i64(zsa + sg); else return g.segA[sg]; };
  auto offB = [&](int sg) -> int64_t { if constexpr (ZS) return uniform_load_i64(zsb + sg); else return g.segB[sg]; };
  la.load(Ab + offA(seg), g.lda, m0, kt * BK, g.M, g.K, tid);
  lb.load(Bb + offB(seg), g.ldb, n0, kt * BK, g.N, g.K, tid);
  la.store(sA, tid, CJ ? (int)g.segJ[seg] : 0);
  lb.store(sB, tid);
  __syncthreads();
  for (int t = ktb; t < kte; ++t) {
    const int cur = (t - ktb) & 1;
    int kt2 = kt + 1, seg2 = seg;
    if (kt2 == tps) { kt2 = 0; seg2 = seg + 1; }
    if (t + 1 < kte) {
      la.load(Ab + offA(seg2), g.lda, m0, kt2 * BK, g.M, g.K, tid);
      lb.load(Bb + offB(seg2), g.ldb, n0, kt2 * BK, g.N, g.K, tid);
    }
    const double* a_s = sA + cur * IA::SIZE;
    const double* b_s = sB + cur * IB::SIZE;
    // fragment reads are software-pipelined one k-step ahead of the MFMAs that consume them
    double af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = a_s[IA::idx(wm * WTM + i * 16 + fr, fq)];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = b_s[IB::idx(wn * WTN + j * 16 + fr, fq)];
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      const int cb = (ks >> 2) & 1, nb = cb ^ 1;
      if (ks + 4 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nb][i] = a_s[IA::idx(wm * WTM + i * 16 + fr, ks + 4 + fq)];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nb][j] = b_s[IB::idx(wn * WTN + j * 16 + fr, ks + 4 + fq)];
      }
      // keep the next k-step's ds_reads ABOVE this k-step's MFMAs (the machine scheduler otherwise sinks
      // them to their use and every MFMA group starts with s_waitcnt lgkmcnt(0) on a just-issued read)
#if !(MPSK_MFMA_4X4X4 && MPSK_MFMA_4X4X4_FREE_SCHED)
      __builtin_amdgcn_sched_barrier(0);
#endif
#if MPSK_MFMA_4X4X4
      // v_mfma_f64_4x4x4 (4 independent 4x4x4 blocks q: operand lanes 16k + 4q + {i | j}, result lane 16i + 4q + j):
      // with the B fragment as the first operand and the A fragment as the second, block q yields
      // C[m = 4q + j][n = 4q + i]; rotating the B fragment by r blocks inside each 16-lane row (DPP row_ror) gives the
      // column block (q + r) & 3, so 4 instructions cover the same 16 x 16 x 4 product as one v_mfma_f64_16x16x4 --
      // out of the same two LDS fragments, at the higher issue rate this shape sustains (tools/mfma_f64_lds.hip:
      // 68 vs 46 TFLOP/s out of LDS).  acc[i][j][r] is un-rotated into the 16x16x4 register layout after the k loop.
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const double b0 = bf[cb][j];
        const double b1 = row_rot<MPSK_ROT1>(b0), b2 = row_rot<0x128>(b0), b3 = row_rot<MPSK_ROT3>(b0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          acc[i][j][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(b0, af[cb][i], acc[i][j][0], 0, 0, 0);
          acc[i][j][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(b1, af[cb][i], acc[i][j][1], 0, 0, 0);
          acc[i][j][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(b2, af[cb][i], acc[i][j][2], 0, 0, 0);
          acc[i][j][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(b3, af[cb][i], acc[i][j][3], 0, 0, 0);
        }
      }
#else
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[cb][j], af[cb][i], acc[i][j], 0, 0, 0);
#endif
#if !(MPSK_MFMA_4X4X4 && MPSK_MFMA_4X4X4_FREE_SCHED)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (t + 1 < kte) {
      la.store(sA + (cur ^ 1) * IA::SIZE, tid, CJ ? (int)g.segJ[seg2] : 0);
      lb.store(sB + (cur ^ 1) * IB::SIZE, tid);
    }
    __syncthreads();
    kt = kt2; seg = seg2;
  }
#if MPSK_MFMA_4X4X4
  {  // register r of a lane in block q holds the column block (q + r) & 3: bring block rg into register rg
    const int q = (threadIdx.x >> 2) & 3;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const d4 v = acc[i][j];
        d4 o;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = (rg - q) & 3;
          o[rg] = r == 0 ? v[0] : (r == 1 ? v[1] : (r == 2 ? v[2] : v[3]));
        }
        acc[i][j] = o;
      }
  }
#endif
}

// Same contract as gemm_accumulate, with the global loads running TWO k-tiles ahead of the MFMAs (two register sets
// next to the LDS double buffer).  The 64-wide tiles spend 16 MFMAs (~1000 MFMA-pipe cycles per wave) on a k-tile, so a
// one-tile prefetch distance only covers the load latency when 4 waves share the SIMD and nothing else is in the way;
// the PMC pass showed the MFMA pipe busy 72 % of the time at D = 1024 against 81 % with 128x128 tiles (whose k-tile is
// 4x longer).  The steady-state loop is unrolled by two and free of conditionals: a branch around the loads makes the
// compiler's s_waitcnt insertion merge the two paths into vmcnt(0), which serialises exactly what this is meant to overlap.
#ifndef MPSK_MID_KS
#define MPSK_MID_KS 8      // k-step (0, 4, 8, 12) behind whose MFMA group the next tile's LDS writes are issued
#endif
template <int BM, int BN, bool TA, bool TB, bool ALIGNED, bool ZS = false, bool CJ = false>
__device__ __forceinline__ void gemm_accumulate_pf2(const GemmArgs& g, const double* __restrict__ Ab,
                                                    const double* __restrict__ Bb, int z, int m0, int n0, int ktb, int kte,
                                                    d4 (&acc)[BM / 32][BN / 32], double* smem) {
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  using LA = TileLoader<BM, TA, ALIGNED>;
  using LB = TileLoader<BN, !TB, ALIGNED>;
  using IA = typename LA::Img;
  using IB = typename LB::Img;
  double* const sA = smem;
  double* const sB = smem + 2 * IA::SIZE;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int tps = (g.K + BK - 1) / BK;
  const int64_t* const zsa = ZS ? g.zsegA + (int64_t)z * g.nseg : nullptr;
  const int64_t* const zsb = ZS ? g.zsegB + (int64_t)z * g.nseg : nullptr;
  auto offA = [&](int sg) -> int64_t { if constexpr (ZS) return uniform_load_i64(zsa + sg); else return g.segA[sg]; };
  auto offB = [&](int sg) -> int64_t { if constexpr (ZS) return uniform_load_i64(zsb + sg); else return g.segB[sg]; };
  LA la0, la1; LB lb0, lb1;
  int jf0 = 0, jf1 = 0;
  int segL = ktb / tps, ktL = ktb % tps;          // the next k-tile to be requested from memory
  const double* pa = Ab + offA(segL);             // segment bases: looked up when the segment changes, not per k-tile
  const double* pb = Bb + offB(segL);
  int jfL = CJ ? (int)g.segJ[segL] : 0;
  auto issue = [&](LA& a, LB& b, int& jf) {
    a.load(pa, g.lda, m0, ktL * BK, g.M, g.K, tid);
    b.load(pb, g.ldb, n0, ktL * BK, g.N, g.K, tid);
    jf = jfL;
    if (++ktL == tps) {
      ktL = 0; ++segL;
      if (segL * tps < kte) {                     // (uniform; scalar loads only: the vmcnt bookkeeping of the two paths agrees)
        pa = Ab + offA(segL); pb = Bb + offB(segL);
        if (CJ) jfL = (int)g.segJ[segL];
      }
    }
  };
  auto compute = [&](const double* __restrict__ a_s, const double* __restrict__ b_s, auto&& mid) {
    double af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = a_s[IA::idx(wm * WTM + i * 16 + fr, fq)];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = b_s[IB::idx(wn * WTN + j * 16 + fr, fq)];
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      const int cb = (ks >> 2) & 1, nb = cb ^ 1;
      if (ks + 4 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nb][i] = a_s[IA::idx(wm * WTM + i * 16 + fr, ks + 4 + fq)];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nb][j] = b_s[IB::idx(wn * WTN + j * 16 + fr, ks + 4 + fq)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[cb][j], af[cb][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the LDS image of the NEXT k-tile is written behind the first MFMA group (its registers arrived a phase ago, the
      // buffer was released by the barrier that ended the previous phase): the writes and their swizzle moves issue
      // while the matrix pipe works instead of in a serial tail before the barrier
      if (ks == MPSK_MID_KS) { mid(); __builtin_amdgcn_sched_barrier(0); }
    }
  };
  auto none = [] {};
  double* const sA0 = sA; double* const sA1 = sA + IA::SIZE;
  double* const sB0 = sB; double* const sB1 = sB + IB::SIZE;
  issue(la0, lb0, jf0);
  if (ktb + 1 < kte) issue(la1, lb1, jf1);
  la0.store(sA0, tid, jf0);
  lb0.store(sB0, tid);
  __syncthreads();
  int t = ktb;
  // invariant at the top: LDS image 0 holds k-tile t, register set 1 holds (or is receiving) k-tile t + 1
  while (t + 3 < kte) {
    issue(la0, lb0, jf0);                          // k-tile t + 2
    compute(sA0, sB0, [&] { la1.store(sA1, tid, jf1); lb1.store(sB1, tid); });
    __syncthreads();
    issue(la1, lb1, jf1);                          // k-tile t + 3
    compute(sA1, sB1, [&] { la0.store(sA0, tid, jf0); lb0.store(sB0, tid); });
    __syncthreads();
    t += 2;
  }
  const int rem = kte - t;                         // 1, 2 or 3 k-tiles left; nothing beyond t + 1 requested yet
  if (rem == 3) issue(la0, lb0, jf0);
  compute(sA0, sB0, none);
  if (rem >= 2) {
    la1.store(sA1, tid, jf1);
    lb1.store(sB1, tid);
    __syncthreads();
    compute(sA1, sB1, none);
    if (rem == 3) {
      la0.store(sA0, tid, jf0);
      lb0.store(sB0, tid);
      __syncthreads();
      compute(sA0, sB0, none);
    }
  }
  __syncthreads();
}

// the deep prefetch for every tile but 128x128 (which sits at the 256-VGPR cap and hides the latency behind its 64 MFMAs
// per k-tile); -DMPSK_PF2=0 builds the one-tile-ahead loop everywhere
#ifndef MPSK_PF2
#define MPSK_PF2 1
#endif
template <int BM, int BN, bool TA, bool TB, bool ALIGNED, bool ZS, bool CJ>
__device__ __forceinline__ void gemm_acc_dispatch(const GemmArgs& g, const double* __restrict__ Ab,
                                                  const double* __restrict__ Bb, int z, int m0, int n0, int ktb, int kte,
                                                  d4 (&acc)[BM / 32][BN / 32], double* smem) {
  if constexpr (MPSK_PF2 && !MPSK_MFMA_4X4X4 && BM * BN < 128 * 128)
    gemm_accumulate_pf2<BM, BN, TA, TB, ALIGNED, ZS, CJ>(g, Ab, Bb, z, m0, n0, ktb, kte, acc, smem);
  else
    gemm_accumulate<BM, BN, TA, TB, ALIGNED, ZS, CJ>(g, Ab, Bb, z, m0, n0, ktb, kte, acc, smem);
}

// XCD-ordered linear tile index t -> tile coordinates.  Tiles [x*q, (x+1)*q) run on XCD x (q = ntiles / 8).
// With an XCD grid gx x gy each XCD owns a (tilesM/gx) x (tilesN/gy) RECTANGLE of the tile grid, so its private L2
// only has to stream M/gx rows of A and N/gy columns of B instead of all of A (a column strip): the PMC passes
// showed 5-10x the compulsory L2->fabric traffic with strip chunks (profiles/r01_pmc_traffic.json).
__device__ __forceinline__ void tile_coords(const GemmArgs& g, int t, int tilesM, int tilesN, int* bm, int* bn) {
  if (g.xcd_gx > 0) {
    const int q = (tilesM * tilesN) >> 3;
    const int xcd = t / q, pos = t - xcd * q;
    const int rm = tilesM / g.xcd_gx, rn = tilesN / g.xcd_gy;
    const int xi = xcd % g.xcd_gx, yi = xcd / g.xcd_gx;
    const int ln = pos / rm, lm = pos - ln * rm;
    *bm = xi * rm + lm;
    *bn = yi * rn + ln;
    (void)rn;
  } else {
    *bn = t / tilesM;
    *bm = t - *bn * tilesM;
  }
}

// epilogue: lane holds C[m = .. + fr][n = .. + fq + 4*reg]
template <int BM, int BN, bool ALIGNED, bool CJ = false>
__device__ __forceinline__ void gemm_store_c(const GemmArgs& g, double* __restrict__ Cb, int z, int m0, int n0,
                                             const d4 (&acc)[BM / 32][BN / 32]) {
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1, fr = lane & 15, fq = lane >> 4;
  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + wn * WTN + j * 16 + fq + 4 * rg;
        if (ALIGNED || (m < g.M && n < g.N)) {
          const int64_t mo = CJ ? (int64_t)m * g.c_rs : (int64_t)m;
          double* p = (g.tabC2 != nullptr && n >= g.splitN)
                          ? g.C + g.tabC2[z] + mo + (int64_t)(n - g.splitN) * g.ldc
                          : Cb + mo + (int64_t)n * g.ldc;
          double v = alpha * acc[i][j][rg];
          if (beta != 0.0) v += beta * (*p);
          *p = v;
        }
      }
    }
  }
}

// partial tile -> dense BM x BN workspace slot (stream-K parts that do not start at k = 0)
template <int BM, int BN>
__device__ __forceinline__ void gemm_store_ws(double alpha, double* __restrict__ slot, const d4 (&acc)[BM / 32][BN / 32]) {
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1, fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        slot[(wm * WTM + i * 16 + fr) + (wn * WTN + j * 16 + fq + 4 * rg) * BM] = alpha * acc[i][j][rg];
}

template <int BM, int BN, bool TA, bool TB, bool ALIGNED, bool ZS = false, bool CJ = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // XCD-aware tile mapping: blocks b, b+8, ... share an XCD (L2); give each XCD a contiguous
  // chunk of the tile list so neighbouring tiles (sharing an A row panel) hit the same L2.
  const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
  const int ntiles = tilesM * tilesN;
  int bid = blockIdx.x;
  if (!g.xcd_interleave) {
    int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, pos = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  int bm, bn;
  tile_coords(g, bid, tilesM, tilesN, &bm, &bn);
  if (g.upper_only && bm > bn) return;      // symmetric result, lower tiles never read (uniform per workgroup)
  const int z = blockIdx.y;
  const int m0 = bm * BM, n0 = bn * BN;
  const double* Ab = g.A + (g.tabA ? g.tabA[z] : (int64_t)z * g.bsA);
  const double* Bb = g.B + (g.tabB ? g.tabB[z] : (int64_t)z * g.bsB);
  double* Cb = g.C + (g.tabC ? g.tabC[z] : (int64_t)z * g.bsC);
  d4 acc[BM / 32][BN / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i)
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  const int nt = ((g.K + BK - 1) / BK) * g.nseg;
  int ktlo = 0, kthi = nt;
  if (g.b_upper) { const int lim = (n0 + BN + BK - 1) / BK; if (lim < kthi) kthi = lim; }
  if (g.a_upper) ktlo = m0 / BK;
  if (ktlo < kthi) gemm_acc_dispatch<BM, BN, TA, TB, ALIGNED, ZS, CJ>(g, Ab, Bb, z, m0, n0, ktlo, kthi, acc, smem);
  gemm_store_c<BM, BN, ALIGNED, CJ>(g, Cb, z, m0, n0, acc);
}

// ---- stream-K: equal shares of the flattened (tile, k-tile) work list ---------------------------
// Workgroup i owns units [i*g.sk_units, (i+1)*g.sk_units) of the list  unit = (z*ntiles + tile)*KT + kt.
// A share that starts at kt == 0 of a tile writes C (with alpha / beta); a share that starts in the
// middle of a tile writes its partial tile to workspace slot i, and gemm_sk_fixup_kernel adds the
// slots to C in a fixed order (deterministic, no atomics).  This removes the wave-quantisation loss
// of big tiles (640 tiles of 128x128 on 512 resident workgroups at the north-star point).
template <int BM, int BN, bool TA, bool TB, bool ALIGNED, bool ZS = false>
__device__ __forceinline__ void gemm_sk_body(const GemmArgs& g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 16, TN = WTN / 16;
  const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
  const int ntiles = tilesM * tilesN;
  const int KT = ((g.K + BK - 1) / BK) * g.nseg;
  const int U = ntiles * g.batch * KT;
  // XCD-aware share order: workgroups b, b+8, ... share an L2 -> give each XCD a contiguous run of shares
  int sid = blockIdx.x;
  if (!g.xcd_interleave) {
    const int nwg = gridDim.x;
    int q = nwg / 8, r = nwg % 8, xcd = sid % 8, pos = sid / 8;
    sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  int u = sid * g.sk_units;
  int slot = sid;                            // workspace slot == share index in the (tile, k) order the fixup walks
  if (g.sk_split_major) {
    // split-K launches (sk_units divides KT): shares ordered split-major, so that the 128 workgroups resident on one XCD
    // are 128 DIFFERENT tiles at the SAME k position (a 16 x 8 tile rectangle whose A / B panels its L2 streams once)
    // instead of 64 tiles at two k positions that share nothing
    const int TZ = ntiles * g.batch;
    const int split = sid / TZ, tzs = sid - split * TZ;
    u = tzs * KT + split * g.sk_units;
    slot = tzs * (KT / g.sk_units - 1) + split - 1;      // compact: the first share of a tile writes C, not a slot
  }
  const int uend = (u + g.sk_units < U) ? u + g.sk_units : U;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1, fr = lane & 15, fq = lane >> 4;
  while (u < uend) {
    const int tz = u / KT;
    const int kt0 = u - tz * KT;
    const int kt1 = (uend - u < KT - kt0) ? kt0 + (uend - u) : KT;
    const int z = tz / ntiles, tile = tz - z * ntiles;
    int bm, bn;
    tile_coords(g, tile, tilesM, tilesN, &bm, &bn);
    if (g.upper_only && bm > bn) { u += kt1 - kt0; continue; }     // uniform: the whole workgroup skips the share
    const int m0 = bm * BM, n0 = bn * BN;
    const double* Ab = g.A + (g.tabA ? g.tabA[z] : (int64_t)z * g.bsA);
    const double* Bb = g.B + (g.tabB ? g.tabB[z] : (int64_t)z * g.bsB);
    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    {
      int ka = kt0, kb = kt1;                 // triangular operands: clamp the share to the structurally non-zero k-tiles
      if (g.b_upper) { const int lim = (n0 + BN + BK - 1) / BK; if (lim < kb) kb = lim; }
      if (g.a_upper) { const int lo = m0 / BK; if (lo > ka) ka = lo; }
      if (ka < kb) gemm_acc_dispatch<BM, BN, TA, TB, ALIGNED, ZS, false>(g, Ab, Bb, z, m0, n0, ka, kb, acc, smem);
    }
    // one epilogue for both destinations (uniform parameters): C tile (kt0 == 0) or workspace slot
    const bool toC = (kt0 == 0);
    double* base = toC ? g.C + (g.tabC ? g.tabC[z] : (int64_t)z * g.bsC) + m0 + (int64_t)n0 * g.ldc
                       : g.sk_ws + (int64_t)slot * BM * BN;
    const int64_t ld = toC ? g.ldc : BM;
    const double beta = toC ? g.beta : 0.0;
    const double alpha = g.alpha;
    const int mlim = toC ? g.M - m0 : BM, nlim = toC ? g.N - n0 : BN;
    const bool split = toC && g.tabC2 != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = wm * WTM + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int n = wn * WTN + j * 16 + fq + 4 * rg;
          if (m < mlim && n < nlim) {
            double* p = base + m + (int64_t)n * ld;
            if (split && n0 + n >= g.splitN) p = g.C + g.tabC2[z] + (m0 + m) + (int64_t)(n0 + n - g.splitN) * g.ldc;
            double v = alpha * acc[i][j][rg];
            if (beta != 0.0) v += beta * (*p);
            *p = v;
          }
        }
      }
    }
    u += kt1 - kt0;
  }
}

// waves per SIMD the register allocation has to leave room for: 4 for the 64x64 tile (<= 128 VGPRs; its LDS images allow
// 4 workgroups per CU and the deep prefetch relies on them), 2 otherwise
constexpr int min_waves(int bm, int bn) { return (bm == 64 && bn == 64) ? 4 : 2; }

template <int BM, int BN, bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, min_waves(BM, BN)) void gemm_sk_f64_kernel(GemmArgs g) {
  gemm_sk_body<BM, BN, TA, TB, ALIGNED>(g);
}
template <int BM, int BN, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, min_waves(BM, BN)) void dac_gemm_sk_f64_kernel(GemmArgs g) {
  gemm_sk_body<BM, BN, false, false, ALIGNED>(g);
}
// per-batch K-segment tables (stage 3 of the prepared operator, mpsk_hac_apply)
template <int BM, int BN, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, min_waves(BM, BN)) void dac_gemm_sk_zs_f64_kernel(GemmArgs g) {
  gemm_sk_body<BM, BN, false, false, ALIGNED, true>(g);
}

// C tile += sum of the workspace slots of the shares that start strictly inside this tile's k-range
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_sk_fixup_kernel(GemmArgs g) {
  const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
  const int ntiles = tilesM * tilesN;
  const int KT = ((g.K + BK - 1) / BK) * g.nseg;
  const int tz = blockIdx.x;
  const int z = tz / ntiles, tile = tz % ntiles;
  const int u0 = tz * KT, u1 = u0 + KT;
  const int i = u0 / g.sk_units + 1;           // first (logical) share that can start inside (u0, u1)
  if (i * g.sk_units >= u1) return;            // tile not split
  int bm, bn;
  tile_coords(g, tile, tilesM, tilesN, &bm, &bn);
  if (g.upper_only && bm > bn) return;
  const int m0 = bm * BM, n0 = bn * BN;
  double* Cb = g.C + (g.tabC ? g.tabC[z] : (int64_t)z * g.bsC);
  // 8 elements per thread and pass, every load of a pass issued before the first use: the earlier one-element loop
  // waited out a full memory round trip per element (23 us per launch for a few MB of traffic)
  // gridDim.y workgroups share a tile (a 64x64 tile is two passes of 8 elements per thread for one workgroup: four
  // workgroups of half a pass each cut the launch from 9-15 us to the latency of one round of loads)
  constexpr int UNR = 8;
  const int per = (BM * BN + (int)gridDim.y - 1) / (int)gridDim.y;
  const int ebeg = (int)blockIdx.y * per, eend = (ebeg + per < BM * BN) ? ebeg + per : BM * BN;
  for (int e0 = ebeg + threadIdx.x; e0 < eend; e0 += 256 * UNR) {
    double sacc[UNR], cv[UNR];
    double* pp[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int e = e0 + 256 * q;
      const int m = e % BM, n = e / BM;
      const bool ok = (e < eend) && (m0 + m < g.M) && (n0 + n < g.N);
      const int gn = n0 + n;
      pp[q] = !ok ? nullptr
                  : ((g.tabC2 != nullptr && gn >= g.splitN) ? g.C + g.tabC2[z] + (m0 + m) + (int64_t)(gn - g.splitN) * g.ldc
                                                            : Cb + (m0 + m) + (int64_t)gn * g.ldc);
      cv[q] = ok ? *pp[q] : 0.0;
      sacc[q] = 0.0;
    }
    for (int w = i; w * g.sk_units < u1; ++w) {
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const int e = e0 + 256 * q;
        if (e < eend) sacc[q] += g.sk_ws[(int64_t)(g.sk_split_major ? w - tz - 1 : w) * BM * BN + e];
      }
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q)
      if (pp[q]) *pp[q] = cv[q] + sacc[q];
  }
}

template <int BM, int BN, bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, min_waves(BM, BN)) void gemm_f64_kernel(GemmArgs g) {
  gemm_body<BM, BN, TA, TB, ALIGNED>(g);
}

// Same body under its own symbol for the two big GEMM stages of the effective-Hamiltonian matvecs
// (dAC / dC / dAC2): rocprofv3 --stats and the in-library event profile then report the hot kernel
// separately from the small gauge-step GEMMs.
template <int BM, int BN, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, min_waves(BM, BN)) void dac_gemm_f64_kernel(GemmArgs g) {
  gemm_body<BM, BN, false, false, ALIGNED>(g);
}
template <int BM, int BN, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, min_waves(BM, BN)) void dac_gemm_zs_f64_kernel(GemmArgs g) {
  gemm_body<BM, BN, false, false, ALIGNED, true>(g);
}

// complex128 family (A / C rows interleaved re-im, B planar; GemmArgs::segJ / c_rs): same body with the J-aware loader
template <int BM, int BN, bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, min_waves(BM, BN)) void cgemm_f64_kernel(GemmArgs g) {
  gemm_body<BM, BN, TA, TB, ALIGNED, false, true>(g);
}

// ---- event profile of the tagged (matvec) launches ---------------------------------------------
struct ProfRec { hipEvent_t e0, e1; double flops; int bm, bn, aligned, sk, zs; };
// The profile is process-wide (every ctx's tagged launches land in one list); the list is mutex-protected because
// distinct ctxs may launch from distinct host threads.
static std::atomic<bool> g_prof_on{false};
static std::atomic<uint64_t> g_prof_seq{0};
static const int g_prof_stride = getenv("MPSK_PROF_STRIDE") && atoi(getenv("MPSK_PROF_STRIDE")) > 0 ? atoi(getenv("MPSK_PROF_STRIDE")) : 1;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;

void gemm_prof_enable(bool on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (on && !g_prof_on.load()) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
  }
  if (on) g_prof_seq.store(0);
  g_prof_on.store(on);
}
static void prof_push(const ProfRec& r) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
}

// JSON list of {kernel, launches, total_ms, avg_ms, flops}; synchronises the device.
std::string gemm_prof_summary() {
  (void)hipDeviceSynchronize();
  struct Agg { long n = 0; double ms = 0, flops = 0; };
  std::map<std::string, Agg> agg;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    char key[96];
    snprintf(key, sizeof(key), "%s<%d,%d,%s>",
             r.sk ? (r.zs ? "dac_gemm_sk_zs_f64_kernel" : "dac_gemm_sk_f64_kernel")
                  : (r.zs ? "dac_gemm_zs_f64_kernel" : "dac_gemm_f64_kernel"),
             r.bm, r.bn, r.aligned ? "true" : "false");
    Agg& a = agg[key];
    a.n++; a.ms += ms; a.flops += r.flops;
  }
  std::string out = "[";
  bool first = true;
  for (auto& kv : agg) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s{\"kernel\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"avg_ms\": %.6f, \"flops\": %.6e}",
             first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.ms / kv.second.n, kv.second.flops);
    out += buf;
    first = false;
  }
  out += "]";
  return out;
}

// one launch (+ its event pair when the profile is on and the launch is tagged)
template <class K>
static hipError_t launch_one(K kern, std::atomic<uint64_t>& attr, dim3 grid, size_t smem, const GemmArgs& g, hipStream_t s,
                             bool prof, ProfRec* r) {
  if (hipError_t e = ensure_dyn_smem(attr, reinterpret_cast<const void*>(kern), smem); e != hipSuccess) return e;
  if (prof) {
    (void)hipEventCreateWithFlags(&r->e0, hipEventDisableSystemFence); (void)hipEventCreateWithFlags(&r->e1, hipEventDisableSystemFence);   // device-scope release: no system-wide cache flush per record
    (void)hipEventRecord(r->e0, s);
  }
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), smem, s, g);
  return hipGetLastError();
}

template <int BM, int BN, bool TA, bool TB, bool ALIGNED>
static hipError_t launch_cfg(const GemmArgs& g, hipStream_t s) {
  using LA = TileLoader<BM, TA, ALIGNED>;
  using LB = TileLoader<BN, !TB, ALIGNED>;
  constexpr size_t smem = (2 * LA::Img::SIZE + 2 * LB::Img::SIZE) * sizeof(double);
  const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
  dim3 grid(tilesM * tilesN, g.batch, 1);
  const bool zs = g.zsegA != nullptr;
  if (zs && !(g.tag == 1 && !TA && !TB && g.zsegB != nullptr)) return hipErrorInvalidValue;   // tables: tagged NN launches only
  const bool tagged = (g.tag == 1 && !TA && !TB);
  // an event pair created with the default flags costs a ~10 us gap on the stream per launch (system-scope release on
  // every record; kernel trace of bench.py: 10.3 us between stage 1 and stage 3 with the profile on, < 3 us without).
  // With hipEventDisableSystemFence the cost is not measurable (same-box A/B, 0.643 vs 0.644 sweeps/s), so every tagged
  // launch is recorded; MPSK_PROF_STRIDE=n samples both stage launches of every n-th matvec instead
  bool prof = tagged && g_prof_on.load(std::memory_order_relaxed);
  if (prof) {
    const uint64_t n = g_prof_seq.fetch_add(1, std::memory_order_relaxed);
    prof = ((n >> 1) % (uint64_t)g_prof_stride) == 0;
  }
  ProfRec r;
  r.flops = 2.0 * g.M * g.N * (double)g.K * g.nseg * g.batch; r.bm = BM; r.bn = BN; r.aligned = ALIGNED; r.sk = 0; r.zs = zs;
  hipError_t e = hipSuccess;
  if (g.cplx) {             // complex128 family: J-aware A loader, optional row-strided C; data-parallel tiles only
    if (zs || g.sk_units > 0) return hipErrorInvalidValue;
    static std::atomic<uint64_t> c0{0};
    return launch_one(cgemm_f64_kernel<BM, BN, TA, TB, ALIGNED>, c0, grid, smem, g, s, false, &r);
  }
  if constexpr ((BM == 128 && BN == 128) || (BM == 64 && BN == 64)) {
    if (g.sk_units > 0) {   // stream-K launch: equal k-tile shares + fixed-order fixup of split tiles
      const int KT = ((g.K + BK - 1) / BK) * g.nseg;
      const int64_t U = (int64_t)tilesM * tilesN * g.batch * KT;
      const dim3 sgrid((unsigned)((U + g.sk_units - 1) / g.sk_units));
      r.sk = 1;
      static std::atomic<uint64_t> a0{0}, a1{0}, a2{0};
      if constexpr (!TA && !TB) {
        if (tagged && zs) e = launch_one(dac_gemm_sk_zs_f64_kernel<BM, BN, ALIGNED>, a2, sgrid, smem, g, s, prof, &r);
        else if (tagged) e = launch_one(dac_gemm_sk_f64_kernel<BM, BN, ALIGNED>, a1, sgrid, smem, g, s, prof, &r);
        else e = launch_one(gemm_sk_f64_kernel<BM, BN, TA, TB, ALIGNED>, a0, sgrid, smem, g, s, false, &r);
      } else {
        e = launch_one(gemm_sk_f64_kernel<BM, BN, TA, TB, ALIGNED>, a0, sgrid, smem, g, s, false, &r);
      }
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((gemm_sk_fixup_kernel<BM, BN>), dim3(tilesM * tilesN * g.batch, (BM * BN) / 1024), dim3(256), 0, s, g);
      // the event window covers the split GEMM AND its fixup launch: the flops credited are those of the whole stage
      if (prof) { (void)hipEventRecord(r.e1, s); prof_push(r); }
      return hipGetLastError();
    }
  }
  static std::atomic<uint64_t> b0{0}, b1{0}, b2{0};
  if constexpr (!TA && !TB) {
    if (tagged && zs) e = launch_one(dac_gemm_zs_f64_kernel<BM, BN, ALIGNED>, b2, grid, smem, g, s, prof, &r);
    else if (tagged) e = launch_one(dac_gemm_f64_kernel<BM, BN, ALIGNED>, b1, grid, smem, g, s, prof, &r);
    else e = launch_one(gemm_f64_kernel<BM, BN, TA, TB, ALIGNED>, b0, grid, smem, g, s, false, &r);
  } else {
    e = launch_one(gemm_f64_kernel<BM, BN, TA, TB, ALIGNED>, b0, grid, smem, g, s, false, &r);
  }
  if (e != hipSuccess) return e;
  if (prof) { (void)hipEventRecord(r.e1, s); prof_push(r); }
  return hipGetLastError();
}

template <bool TA, bool TB, bool ALIGNED>
static hipError_t launch_tile(const GemmArgs& g, int bm, int bn, hipStream_t s) {
  if (bm == 128 && bn == 128) return launch_cfg<128, 128, TA, TB, ALIGNED>(g, s);
  if (bm == 64 && bn == 128) return launch_cfg<64, 128, TA, TB, ALIGNED>(g, s);
  if (bm == 128 && bn == 64) return launch_cfg<128, 64, TA, TB, ALIGNED>(g, s);
  return launch_cfg<64, 64, TA, TB, ALIGNED>(g, s);
}

// Tile choice (measured on MI355X, tools/ab_dac.py, whole prepared-operator matvec in TFLOP/s after the deep-prefetch /
// interleaved-store rework of the 64-wide loop):
//     D      64x64   128x64   128x128          (stage 1: M = D, N = 2D, K = D, batch 5 -> 64x64-tile count T64)
//   1024     65.2     58.9     --              T64 =  2 560
//   2048     66.2     65.4     64.2            T64 = 10 240
//   3072     63.7     65.7     60.2            T64 = 23 040
//   4096     60.9     65.5     64.8            T64 = 40 960
// The small tile wins while its extra operand traffic (twice the panel re-reads of 128x128) stays inside the L2 / MALL;
// beyond that the wider tile's reuse pays.  128x128 (227 VGPRs, no room for the second prefetch set) is no longer chosen
// by default; MPSK_STREAMK=1 and mpsk_ctx_force_tile still reach it.
static void choose_tile(int M, int N, int K, int batch, int* bm, int* bn) {
  // short-K products (the tall-skinny Jacobi updates X_p W_p: K = 64 = 4 k-tiles) never reach the steady state of the
  // pipeline: the smallest tile gives the most workgroups to hide the prologue (tools/svd_tiles.py: -2.5 % on a 4096^2 split)
  *bm = 64; *bn = 64;
  if (K <= 4 * BK) return;
  const int64_t T64 = (int64_t)((M + 63) / 64) * ((N + 63) / 64) * batch;
  if (T64 <= 16384 && (M < 3072 || N < 3072)) return;     // (stage 3 at D >= 3072: few tiles, but 10 D deep)
  if (M > 64) *bm = 128;                                   // never pad a 64-wide side to 128
  else if (N > 64) *bn = 128;
}

static std::atomic<int> g_force_bm{0}, g_force_bn{0};   // benchmarking knob (process-wide)
void gemm_force_tile(int bm, int bn) { g_force_bm = bm; g_force_bn = bn; }
// Stream-K is implemented and parity-tested but OFF by default: on MI355X it measured no gain over
// the data-parallel 64x64 tiling (D = 1024: 47.5 vs 47.8-50.4 TF/s, tools/bench_dac.py A/B) -- the
// 128x128 stream-K body sits at the 256-VGPR cap and the fixup pass eats the balance it buys.
// MPSK_STREAMK=1 enables it (single-stream use only: the partial-tile workspace is per device).
static std::atomic<bool> g_sk_enabled{(getenv("MPSK_STREAMK") != nullptr) && (getenv("MPSK_STREAMK")[0] == '1')};
void gemm_enable_streamk(bool on) { g_sk_enabled = on; }
static bool g_xcd_grid_enabled = (getenv("MPSK_XCDGRID") == nullptr) || (getenv("MPSK_XCDGRID")[0] != '0');
static bool g_splitk_enabled = (getenv("MPSK_SPLITK") == nullptr) || (getenv("MPSK_SPLITK")[0] != '0');
static int g_splitk_force = getenv("MPSK_SPLITK_F") ? atoi(getenv("MPSK_SPLITK_F")) : 0;
constexpr size_t SK_WS_DOUBLES = (size_t)1024 * 128 * 128 / 2;   // 512 slots of 128x128 == 2048 slots of 64x64
// partial-tile workspace: one per (device, stream) so that concurrent streams never share slots (launches on ONE
// stream are ordered, so ctxs that share a stream may share its slots); allocated on first use (64 MiB each), reference
// counted by the ctxs bound to the stream (gemm_retain_stream / gemm_release_stream) and freed with the last of them.  The table is mutex-protected: distinct ctxs launch from
// distinct host threads.
static std::mutex g_skws_mu;
struct SkWs { double* p = nullptr; int refs = 0; };     // refs: ctxs currently bound to the stream (gemm_retain_stream)
static std::map<std::pair<int, hipStream_t>, SkWs> g_skws;
static double* sk_workspace(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_skws_mu);
  SkWs& e = g_skws[std::make_pair(dev, s)];
  if (e.p) return e.p;
  if (hipMalloc(&e.p, SK_WS_DOUBLES * sizeof(double)) != hipSuccess) { e.p = nullptr; return nullptr; }
  return e.p;
}
// Several ctxs may be bound to one stream (every Backend binds torch's current stream by default): the workspace of a
// (device, stream) is freed when the LAST ctx bound to it lets go, never under a ctx that still launches on it.
void gemm_retain_stream(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(g_skws_mu);
  g_skws[std::make_pair(dev, s)].refs++;
}
void gemm_release_stream(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  double* p = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_skws_mu);
    auto it = g_skws.find(std::make_pair(dev, s));
    if (it == g_skws.end()) return;
    if (--it->second.refs > 0) return;
    p = it->second.p;
    g_skws.erase(it);
  }
  if (p) { (void)hipStreamSynchronize(s); (void)hipFree(p); }
}

hipError_t gemm_f64(const GemmArgs& g_in, hipStream_t s) {
  GemmArgs g = g_in;
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return hipSuccess;
  if (g.K <= 0 || g.nseg <= 0) {  // pure scaling C = beta*C is not needed by any caller
    return hipErrorInvalidValue;
  }
  int bm = 128, bn = 128;
  choose_tile(g.M, g.N, g.K * g.nseg, g.batch, &bm, &bn);
  g.sk_units = 0;
  g.sk_ws = nullptr;
  g.sk_split_major = 0;
  if (g.cplx && g.c_rs == 0) g.c_rs = 1;
  if (g.upper_only && g.M != g.N) g.upper_only = 0;
  if (g.nseg != 1 || g.transA || g.transB) g.a_upper = g.b_upper = 0;
  if (g_force_bm.load()) { bm = g_force_bm.load(); bn = g_force_bn.load(); }
  else if (g.cplx) { /* no split-K / stream-K: the complex epilogue lives in gemm_store_c only */ }
  else {
    const int KT = ((g.K + BK - 1) / BK) * g.nseg;
    const int64_t Tb = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch;
    const int64_t Ts = (int64_t)((g.M + 63) / 64) * ((g.N + 63) / 64) * g.batch;
    if (g_sk_enabled) {
      // full stream-K (experimental, MPSK_STREAMK=1): big tiles with equal k-tile shares
      const int64_t Ub = Tb * KT;
      if (Tb >= 128 && Tb < 4 * 512 && Tb % 512 != 0 && Ub / 512 >= 16) {
        double* ws = sk_workspace(s);
        if (ws) { bm = bn = 128; g.sk_units = (int)((Ub + 511) / 512); g.sk_ws = ws; }
      }
    }
    // (the fixed-order fixup is a second, dependent launch of ~18 us: below ~64 k-tiles the split loses.  An
    //  in-kernel fixup by the last-arriving share was measured and rejected: its agent-scope release/acquire
    //  fences write back / invalidate the XCD's L2 and cost 45 % on the big dAC stage.)
    if (g.sk_units == 0 && g_splitk_enabled && bm == 64 && bn == 64 && KT >= 64 && Ts <= 4096) {
      // split-K for long-K GEMMs whose 64x64 tile count does not balance the 256 CUs: f shares per tile through the
      // stream-K body (first share writes C, the others a workspace slot, fixed-order fixup).  Model of the launch, in
      // k-tile times: the busiest CU runs n = ceil(Ts f / 256) shares of KT / f + 6 (prologue / epilogue) each, four at a
      // time, plus 3 per extra share for its partial tile and 12 for the fixup pass.  Measured (tools/ab_dac.py with
      // MPSK_SPLITK_F, stage 3 in us for f = 1 / 2 / 4): D = 1024: 352 / 340 / 347, D = 1536: 1239 / 1119 / 1118,
      // D = 768: 262 / 198 / 176, D = 512: 109 / 67 / 63.
      const int64_t slots = (int64_t)(SK_WS_DOUBLES / (64 * 64));
      int best_f = 1;
      double best = 0.0;
      for (int f = 1; f <= 8; ++f) {
        if (f > 1 && (KT / f < 16 || Ts * (f - 1) > slots)) break;
        const int64_t n = (Ts * f + 255) / 256;          // shares on the busiest CU; 4 run at a time
        const int r = (int)(n % 4);                      // the last, partial group shares the matrix pipes r ways:
        const double phi[4] = {1.0, 0.55, 0.92, 0.98};   // measured pipe utilisation with 1 / 2 / 3 resident workgroups
        const double len = (double)KT / f + 6.0;
        const double cost = (double)(n - r) * len + (r ? r * len / phi[r] : 0.0) + 3.0 * (f - 1) * (double)n / f + (f > 1 ? 12.0 : 0.0);
        if (f == 1 || cost < 0.97 * best) { best = cost; best_f = f; }
      }
      if (g_splitk_force > 0 && (g_splitk_force == 1 || (KT / g_splitk_force >= 16 && Ts * (g_splitk_force - 1) <= slots)))
        best_f = g_splitk_force;                // MPSK_SPLITK_F: tuning knob
      if (best_f >= 2) {
        double* ws = sk_workspace(s);
        if (ws) {
          g.sk_units = (KT + best_f - 1) / best_f;
          g.sk_ws = ws;
          g.sk_split_major = (KT % g.sk_units == 0 && KT / g.sk_units == best_f) ? 1 : 0;
          if (!g.sk_split_major) {          // uneven shares: logical share index == slot index, needs one slot per share
            const int64_t U = Ts * KT;
            if ((U + g.sk_units - 1) / g.sk_units > slots) { g.sk_units = 0; g.sk_ws = nullptr; }
          }
        }
      }
    }
  }
  // triangular work (symmetric result / triangular operands): the k-range of a tile grows with its column (row) index, so
  // contiguous per-XCD chunks -- rectangles or strips -- leave some XCDs with several times the work of others (Q = X R^-1
  // with the upper-triangular operand: 5.6x between the first and the last column group); consecutive tiles then go to
  // consecutive XCDs (the hardware's round-robin) instead
  g.xcd_interleave = (g.upper_only || g.a_upper || g.b_upper) ? 1 : 0;
  {  // XCD grid: minimise the operand rows + columns an XCD's L2 has to stream, M/gx + N/gy
    const int tM = (g.M + bm - 1) / bm, tN = (g.N + bn - 1) / bn;
    g.xcd_gx = g.xcd_gy = 0;
    const bool chunks_ok = ((int64_t)tM * tN) % 8 == 0 &&
                           (g.sk_units == 0 || (((g.K + BK - 1) / BK) * g.nseg) % g.sk_units == 0);
    if (chunks_ok && g_xcd_grid_enabled && !g.xcd_interleave) {
      double best = 0.0;
      const int cand[4][2] = {{1, 8}, {2, 4}, {4, 2}, {8, 1}};
      for (int c = 0; c < 4; ++c) {
        if (tM % cand[c][0] || tN % cand[c][1]) continue;
        const double cost = (double)g.M / cand[c][0] + (double)g.N / cand[c][1];
        if (g.xcd_gx == 0 || cost < best) { best = cost; g.xcd_gx = cand[c][0]; g.xcd_gy = cand[c][1]; }
      }
    }
  }
  bool aligned = (g.M % bm == 0) && (g.N % bn == 0) && (g.K % BK == 0) && (g.lda % 2 == 0) &&
                 (g.ldb % 2 == 0) && (g.bsA % 2 == 0) && (g.bsB % 2 == 0) &&
                 ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0) &&
                 g.lda < ((int64_t)1 << 21) && g.ldb < ((int64_t)1 << 21);   // 32-bit per-thread byte offsets
  if (g.tabA || g.tabB || g.zsegA || g.zsegB) aligned = aligned && g.tabs_even;
  if (!g.zsegA) for (int i = 0; i < g.nseg && i < MAXSEG; ++i) aligned = aligned && (g.segA[i] % 2 == 0);
  if (!g.zsegB) for (int i = 0; i < g.nseg && i < MAXSEG; ++i) aligned = aligned && (g.segB[i] % 2 == 0);
  if ((!g.zsegA || !g.zsegB) && g.nseg > MAXSEG) return hipErrorInvalidValue;
  if (g.upper_only && bm != bn) g.upper_only = 0;     // the tile test bm > bn needs square tiles
#define MPSK_DISPATCH(TA_, TB_)                                                        \
  return aligned ? launch_tile<TA_, TB_, true>(g, bm, bn, s) : launch_tile<TA_, TB_, false>(g, bm, bn, s)
  if (!g.transA && !g.transB) { MPSK_DISPATCH(false, false); }
  if (g.transA && !g.transB) { MPSK_DISPATCH(true, false); }
  if (!g.transA && g.transB) { MPSK_DISPATCH(false, true); }
  MPSK_DISPATCH(true, true);
#undef MPSK_DISPATCH
}

}  // namespace mpsk
