// Internal declarations shared by the HIP translation units of libmpsk (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include <vector>
#include <string>

namespace mpsk {

// Once-per-device hipFuncSetAttribute(MaxDynamicSharedMemorySize).  `done` is a static bit mask owned by the call site
// (one bit per device ordinal); distinct ctxs may be driven from distinct host threads, so the flag is atomic -- a race
// only repeats the idempotent call.
inline hipError_t ensure_dyn_smem(std::atomic<uint64_t>& done, const void* kernel, size_t bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  done.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

constexpr int MAXSEG = 32;

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int M, N, K;               // K = length of ONE segment
  int64_t lda, ldb, ldc;
  int64_t bsA, bsB, bsC;     // batch strides (elements)
  int batch;
  int nseg;
  int64_t segA[MAXSEG], segB[MAXSEG];  // element offsets of each K-segment
  double alpha, beta;
  int transA, transB;        // 0: col-major as given; 1: operand is the transpose of a col-major matrix
  // optional per-batch element offsets (device arrays of length batch); override bsA/bsB/bsC when set
  const int64_t* tabA;
  const int64_t* tabB;
  const int64_t* tabC;
  // optional column split of C: columns n >= splitN of batch z go to C + tabC2[z] + (n - splitN)*ldc
  const int64_t* tabC2;
  int splitN;
  // stream-K (set by gemm_f64 itself): units of k-tiles per workgroup and the partial-tile workspace
  int sk_units;
  int sk_split_major;   // split-K launches: share index = split * (tiles * batch) + tile (see gemm_sk_body)
  double* sk_ws;
  // XCD grid (set by gemm_f64): the 8 XCDs' tile chunks are xcd_gx x xcd_gy rectangles of the tile grid (0: linear)
  int xcd_gx, xcd_gy;
  int xcd_interleave;   // triangular work (upper_only / a_upper / b_upper): tiles round-robin over the XCDs instead of chunks
  // optional per-batch K-segment offsets (device arrays [batch][nseg], element offsets): batch z reads segment i at
  // A + tabA/bsA offset + zsegA[z * nseg + i] (override segA / segB when set; nseg is uniform over the batches)
  const int64_t* zsegA;
  const int64_t* zsegB;
  // complex128 operands carried as REAL matrices with (re, im) interleaved along the first (row) index of the A operand
  // and of C ("half-embedded" real view of TensorKit's interleaved complex storage), B planar (separate re / im planes):
  //   C_half = A_half Br + (J A_half) Bi,   J (re, im) = (-im, re)  -- the multiplication by i on a row pair.
  // segJ[i]: what the A loader applies to K-segment i: 0 nothing, 1 J, 2 -J.  cplx selects the kernel family with that
  // loader (the real kernels do not carry the test).  c_rs: row stride of C (1, or 2 to write one plane of an interleaved
  // complex result: transfer_left's Ab^H T2).
  signed char segJ[MAXSEG];
  int b_upper, a_upper;      // operand is upper triangular (single K segment, K index aligned with N resp. M): k-tiles that
                             // only meet structural zeros are skipped -- B upper: k < n0 + BN ; A upper: k >= m0
  int upper_only;            // C is symmetric (Gram matrix) and only its upper triangle is consumed: tiles strictly below the
                             // block diagonal are skipped (M == N, square tiles)
  int cplx;
  int c_rs;
  int tag;                   // 1: matvec-stage launch (own kernel symbol + event profile)
  int tabs_even;             // caller guarantees every tabA/tabB entry is even (16-B aligned operands)
};

hipError_t gemm_f64(const GemmArgs& g, hipStream_t s);
void gemm_force_tile(int bm, int bn);
// reference count of the ctxs bound to (current device, stream); the split-K partial-tile workspace attached to the pair is
// freed when the last of them releases it (ctx destroyed or bound to another stream)
void gemm_retain_stream(hipStream_t s);
void gemm_release_stream(hipStream_t s);
void gemm_enable_streamk(bool on);
void gemm_prof_enable(bool on);
std::string gemm_prof_summary();

// ---- slab mixing:  out_slab[o][r,c] = sum_t coef[t] * in_slab[src[t]][r,c]  --------------------
// A "slab" is an R x C column-major matrix view (ld, base offset) inside a larger tensor.  This
// is the (w,s) -> (v,t) application of the small MPO tensor between the two big GEMMs.
struct MixTerm { int32_t out, in; double coef; double coef_im = 0.0; };

struct MixPlan {          // device-resident CSR: terms grouped by output slab
  int n_out = 0, n_in = 0;
  int32_t* d_rowptr = nullptr;   // [n_out+1]
  int32_t* d_src = nullptr;      // [nnz]
  double* d_coef = nullptr;      // [nnz]
  double* d_coef_im = nullptr;   // [nnz] imaginary parts (complex plans: slabs hold (re, im) row pairs), else null
  int nnz = 0;
  int max_terms = 0;
};

hipError_t mix_plan_create(const std::vector<MixTerm>& terms, int n_out, int n_in, MixPlan* plan);
void mix_plan_destroy(MixPlan* plan);

// slab j = j0 + n0*(j1 + n1*j2) lives at element offset j0*s0 + j1*s1 + j2*s2; rows contiguous,
// column stride ld.
struct SlabIndex { int n0, n1; int64_t s0, s1, s2; int64_t ld; };
hipError_t mix_apply(const MixPlan& plan, const double* in, SlabIndex iin, double* out, SlabIndex iout,
                     int R, int C, hipStream_t s);
// dst[r * drs + c * dcs] = src[r * srs + c * scs]   (R x C elements, arbitrary strides): interleaved <-> planar complex,
// extraction of the real / imaginary planes of an embedded tensor
hipError_t deinterleave(const double* z, int64_t n, double* re, double* im, hipStream_t s);   // z 16-byte aligned
hipError_t copy_strided(const double* src, int64_t srs, int64_t scs, double* dst, int64_t drs, int64_t dcs, int64_t R,
                        int64_t C, hipStream_t s);

// ---- vector kernels ---------------------------------------------------------------------------
// xs: HOST array of k device pointers; d_out / d_coefs: device [k]; d_partial: device scratch
// [MPSK_DOT_SCRATCH doubles].  Vectors must be 16-B aligned.
constexpr int MPSK_DOT_SCRATCH = 32 * 1024;     // up to 32 dots x DOT_BLOCKS partial sums (the long fused Gram-Schmidt passes)
hipError_t vec_multidot(const double* const* xs, int k, const double* y, int64_t n, double* d_out,
                        double* d_partial, hipStream_t s);
hipError_t vec_axpby(double a, const double* x, double b, double* y, int64_t n, hipStream_t s);
hipError_t vec_times_i(const double* x, double* y, int64_t n, hipStream_t s);
hipError_t vec_scal(double a, double* x, int64_t n, hipStream_t s);
hipError_t vec_scal_rsqrt_dev(const double* d_n2, double* x, int64_t n, hipStream_t s);
hipError_t vec_scal_rsqrt_dev_oop(const double* d_n2, const double* x, double* y, int64_t n, hipStream_t s);
// CGS2 of y against xs[0..k) + squared norm of the remainder, fused passes (mpsk_ops.hip); d_out = {h1[k], h2[k], |y|^2}
hipError_t vec_cgs2(const double* const* xs, int k, double* y, int64_t n, double* d_out, double* d_partial, hipStream_t s);
// Ritz step of a fixed-budget Krylov solve on the device (mpsk_ops.hip: ritz_small_kernel); m <= 32
hipError_t vec_ritz_small(const double* d_slot, int m, int stride, double* d_coef, double* d_info, hipStream_t s);
hipError_t vec_multilincomb(const double* const* xs, int k, double* const* ys, int m, const double* d_coefs, int64_t n,
                            hipStream_t s);   // ys[j] = sum_i d_coefs[i + k j] xs[i]; k, m <= 32
hipError_t vec_multiaxpy(const double* const* xs, const double* d_coefs, int k, double sign, double* y,
                         int64_t n, hipStream_t s);

// ---- gauge kernels ----------------------------------------------------------------------------
size_t regularize_workspace_doubles(int W, int D1, int D2);
hipError_t regularize(int W, int D1, int D2, double* v, const double* lvec, const double* rvec, double* partial, hipStream_t s);
hipError_t transpose(const double* in, int ldi, int rows, int cols, double* out, int ldo, hipStream_t s);
size_t qrpos_workspace_doubles(int m, int n);
hipError_t qrpos(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                 hipStream_t s, std::string* err);

size_t cholqr_workspace_doubles(int m, int n);
hipError_t cholqr3(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                   int* d_flag, int* flag_out, hipStream_t s, double shift_scale = 1.0);
hipError_t cholqr_robust(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                         int* d_flag, int* flag_out, hipStream_t s);
hipError_t cholqr1_orth(int m, int n, const double* X, int ldx, double* Q, int ldq, double* ws, int* d_flag, hipStream_t s,
                        double shift_scale);
hipError_t cholqr3_enqueue(int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr, double* ws,
                           int* d_flag, int* flag_out, hipStream_t s, double shift_scale = 1.0);
hipError_t cholqr3_finalize(int m, int n, double* Q, int ldq, double* R, int ldr, double* ws, int* d_flag,
                            int* flag_out, hipStream_t s);
size_t tsvd_workspace_bytes(int m, int n);
hipError_t tsvd(int m, int n, const double* theta, int ldt, double* U, int ldu, double* S, double* Vh, int ldv,
                int max_keep, double trunc_err, int* kept, double* disc_norm, void* ws, hipStream_t s,
                std::string* err, int* sweeps_out, const double* Qpre = nullptr, int ldq = 0, int q_rows = 0,
                int outer_transposed = 0, const hipStream_t* xs = nullptr, int nxs = 0, int vfree = 0);

}  // namespace mpsk
