/* libmpsk -- C ABI of the MI355X-native (gfx950) DMRG / VUMPS hot path.
 *
 * This is the drop-in boundary for the per-site inner loop of stecrotti/MPSKit.jl v0.10.2: every
 * entry point replaces one Julia method of the reference (cited as file:line relative to the
 * reference root) and is what a `ccall((:mpsk_xxx, libmpsk), Cint, ...)` shim binds.  The
 * reference itself has no FFI: the seams are the multiple-dispatch methods listed below
 * (SURVEY.md section 8b); INTEGRATION.md shows the Julia-side overloads.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; mpsk_last_error() gives the
 *    thread-local message.  No exceptions or longjmp cross the boundary.
 *  - all tensor arguments are DEVICE pointers to fp64, column-major ("Fortran", TensorKit's
 *    trivial-sector storage) in TensorKit index order:
 *        MPS tensor      x [Dl, d, Dr]             (V_l (x) P <- V_r)        src/states/abstractmps.jl:30-33
 *        two-site tensor x2[Dl, d, Dr, d]          (V_l (x) P <- V_r (x) P)  src/algorithms/groundstate/dmrg.jl:92
 *        bond matrix     c [Dl, Dr]
 *        environment     G [W][Dbra, Dket]  = W column-major slabs, slab (off_i + k) holding
 *                        GL[i][:, k, :] of the reference's Vector of [D, chi_i, D] tensors
 *                        (src/environments/FinEnv.jl:49-59, mpohaminfenv.jl:25-30); W = sum_i chi_i.
 *    The host owns all device memory (any allocator: hipMalloc, torch, AMDGPU.jl ROCArray);
 *    mpsk_malloc & co. are provided for hosts that have none.
 *  - one mpsk_ctx = one device + one HIP stream + a private workspace.  Calls on a ctx are
 *    asynchronous on its stream; functions that return a host scalar synchronise.  Distinct
 *    ctxs may be driven concurrently from distinct host threads (the reference applies its
 *    operators from several Julia tasks: vumps.jl:39-49,78-86); give each its own stream
 *    (mpsk_ctx_set_stream) if the work is to overlap.  ONE ctx must not be used from two threads
 *    at once.  The only process-wide state of the library (per-device kernel attributes, the
 *    split-K workspace table keyed by (device, stream), the event profile) is atomic /
 *    mutex-protected; tests/test_gpu_threads.py drives two ctxs from two threads and compares
 *    bit-for-bit with the serial run.  mpsk_ctx_force_tile and mpsk_prof_* are process-wide
 *    diagnostics.
 *  - dtype: MPSK_F64 (real fp64) everywhere; MPSK_C128 (complex128, the reference's default scalar type,
 *    defaults.jl:18) for the matvec / transfer family: mpsk_dAC, mpsk_dC, mpsk_dAC2, mpsk_hac_*, mpsk_transfer_left,
 *    mpsk_transfer_right.  A slice created with MPSK_C128 makes every tensor argument of a call that takes it complex;
 *    the slice-less calls (mpsk_dC, pass-through transfers) follow mpsk_ctx_set_dtype.  Complex tensors are
 *    INTERLEAVED complex128 in the same column-major index order (TensorKit / Julia Array{ComplexF64} storage); a
 *    complex MPO slice takes interleaved scalars[2 odim^2] and dense blocks.  Internally a complex product runs on
 *    the real fp64 MFMA core as two K-segments (re / im of the second operand, the first one through a loader that
 *    multiplies by i): 4x the real flops, the complex optimum.  The single gauge steps mpsk_qrpos / mpsk_lqpos and the
 *    two-site split mpsk_tsplit follow mpsk_ctx_set_dtype too: with MPSK_C128 their operands are interleaved complex matrices (leading dimensions count
 *    COMPLEX elements), Q / R / L come back complex with a real positive diagonal on the triangular factor
 *    (TensorKit leftorth! / rightorth! with QRpos() / LQpos() on ComplexF64 tensors).  They run on the real 2m x 2n
 *    embedding INSIDE the library -- 2x the GEMM flops of a native complex kernel and a Cholesky chain of 2n columns --
 *    and keep the complex structure also for ill-conditioned / rank-deficient input (structured part + one more
 *    factorization, R = triu(Q^H A)); the caller's tensors stay interleaved (2x the real memory, not 4x).  Complex
 *    mpsk_tsplit (max_keep only: trunc_err must be 0) returns complex isometries AL / AR, C lower triangular with a real
 *    positive diagonal and the kept COMPLEX singular values; the real split of the embedding (every value twice, arbitrary
 *    basis inside each pair) only supplies the kept subspace, which is made an embedding again by projecting structured
 *    random vectors on it, with a J-invariant choice inside a cluster that straddles the cut.  mpsk_gemm under MPSK_C128
 *    multiplies interleaved complex matrices (trans = conjugate transpose, alpha / beta real, leading dimensions in complex
 *    elements) as ONE real GEMM on the embedded A and the interleaved B: 8 M N K flops, the complex optimum.  The remaining
 *    entry points (mpsk_qrpos2, mpsk_qrlq_pair, mpsk_tsvd, Krylov vector helpers, mpsk_regularize) are fp64 only and ignore
 *    the ctx dtype: a complex host runs its vector arithmetic on the 2n doubles of an interleaved vector (real inner
 *    products suffice for the Hermitian Lanczos solvers).
 */
#ifndef MPSK_H
#define MPSK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpsk_ctx mpsk_ctx;
typedef struct mpsk_mposlice mpsk_mposlice;

enum { MPSK_F64 = 0, MPSK_C128 = 1 };
enum { MPSK_OK = 0, MPSK_ERR_INVALID = 1, MPSK_ERR_HIP = 2, MPSK_ERR_UNSUPPORTED = 3, MPSK_ERR_NOMEM = 4 };
/* MPO block kinds, src/operators/sparsempo/sparseslice.jl:74-106 (contains / isscal) */
enum { MPSK_BLOCK_ZERO = 0, MPSK_BLOCK_SCALAR = 1, MPSK_BLOCK_DENSE = 2 };

int mpsk_version(void);
const char* mpsk_last_error(void);

/* ---- context ---------------------------------------------------------------------------- */
int mpsk_ctx_create(int device, mpsk_ctx** out);
int mpsk_ctx_destroy(mpsk_ctx* ctx);
int mpsk_ctx_set_stream(mpsk_ctx* ctx, void* hip_stream);      /* hipStream_t; NULL = default */
int mpsk_ctx_set_dtype(mpsk_ctx* ctx, int dtype);              /* scalar type of the slice-less calls (mpsk_dC, pass-through
                                                                 * transfers, mpsk_qrpos, mpsk_lqpos, mpsk_tsplit, mpsk_gemm) */
int mpsk_ctx_get_stream(mpsk_ctx* ctx, void** hip_stream);
int mpsk_ctx_get_device(mpsk_ctx* ctx, int* device);
int mpsk_ctx_synchronize(mpsk_ctx* ctx);
int mpsk_ctx_workspace_reserve(mpsk_ctx* ctx, size_t bytes);   /* pre-size the private workspace */
/* QRpos algorithm: 0 auto (shifted CholeskyQR3 on the GEMM core; when the device flags an ill-conditioned /
 * rank-deficient input: perturbed, repeatedly shifted CholeskyQR, then blocked Householder as the last resort),
 * 1 Householder only, 2 CholeskyQR3 only (error on flag).
 * Counters: factorizations finished by CholeskyQR3 / by Householder / CholeskyQR3 attempts that were flagged /
 * flagged inputs finished by the robust CholeskyQR variant (any pointer may be NULL). */
int mpsk_ctx_set_qr_mode(mpsk_ctx* ctx, int mode);
int mpsk_ctx_qr_stats(mpsk_ctx* ctx, long* n_chol, long* n_house, long* n_fallback, long* n_robust);
/* CholeskyQR3 factorizations whose first attempt (shift at the rounding level of the Gram matrix) broke down and that were
 * repeated with the shift of Fukaya et al. (s = 11 (mn + n(n+1)) u ||A||_F^2); counted in n_chol when the repeat succeeds */
int mpsk_ctx_qr_retries(mpsk_ctx* ctx, long* n_retry);
/* tsvd algorithm switch: 0 = Jacobi on theta directly; 1 = the tall orientation of theta is factored with QRpos first and
 * the block-Jacobi iteration runs on R^T (Drmac-Veselic preconditioning; far fewer sweeps on graded spectra); 2
 * = additionally R^T = Q1 R1 and the iteration runs on R1^T (V-free in mpsk_tsplit, with accumulated rotations in
 * mpsk_tsvd: 15 -> 10 sweeps on graded 4096^2 tensors for one more n x n QRpos).  mpsk_ctx_svd_stats returns the number of
 * Jacobi sweeps of the last mpsk_tsvd / mpsk_tsplit.
 * 3 (default) = mode 2, and mpsk_tsplit becomes TRUNCATION-AWARE when max_keep > 0 and r = max_keep + max(64, max_keep / 2) (rounded
 * up to 64; max_keep / 4 if that is too wide) is at most 5/8 of min(m, n): a randomized subspace iteration on the GEMM core finds the dominant r-dimensional
 * subspace, the Jacobi split runs on r columns instead of min(m, n), and the result is accepted only after a CHECK -- the
 * part of the kept triplets outside the iterated subspace, || U_k^T theta (I - W W^T) ||_F <= 1e-12 ||theta||_F
 * (MPSK_SPLIT_TOL) -- otherwise the iteration continues or the call falls through to mode 2 (flat spectra).  Same AL / C / AR
 * contract; S then holds the r leading values and NaN behind them.  mpsk_ctx_split_stats: path of the last mpsk_tsplit
 * (0 full iteration, 1 subspace stage accepted, 2 stage gave up -> full iteration), subspace iterations, check value.
 * Mode 0 has no relative accuracy on matrices whose COLUMNS are nearly parallel (condition number of the column-scaled
 * matrix >= 1/u, e.g. U diag(logspace(0,-14)) V^T behind random orthogonal factors): every rotation against an O(1) column
 * injects rounding noise u |a_big| into columns whose norm is of that order, so their mutual cosines stay O(1) and the
 * scale-invariant convergence test cannot be met; the call then returns MPSK_ERR_HIP "did not converge" instead of
 * non-isometric factors.  (Demmel-Veselic: one-sided Jacobi is accurate for A = B D with B well conditioned -- which is
 * what the QR preconditioning of modes 1 / 2 produces.) */
int mpsk_ctx_set_svd_mode(mpsk_ctx* ctx, int precondition);
int mpsk_ctx_svd_stats(mpsk_ctx* ctx, int* last_sweeps);
int mpsk_ctx_split_stats(mpsk_ctx* ctx, int* path, int* iterations, double* residual);   /* any pointer may be NULL */
/* tile override for benchmarking the GEMM core (0,0 restores the heuristic) */
int mpsk_ctx_force_tile(mpsk_ctx* ctx, int bm, int bn);

/* HIP-event profile of the matvec-stage GEMM launches (the dominant kernel, dac_gemm_f64_kernel):
 * enable, run, then read a JSON summary [{kernel, launches, total_ms, avg_ms, flops}] (synchronises). */
int mpsk_prof_enable(mpsk_ctx* ctx, int on);
int mpsk_prof_summary(mpsk_ctx* ctx, char* buf, size_t buflen);

/* ---- memory helpers (optional) ------------------------------------------------------------ */
int mpsk_malloc(mpsk_ctx* ctx, size_t bytes, void** dptr);
int mpsk_free(mpsk_ctx* ctx, void* dptr);
int mpsk_memcpy_h2d(mpsk_ctx* ctx, void* dst, const void* src, size_t bytes);
int mpsk_memcpy_d2h(mpsk_ctx* ctx, void* dst, const void* src, size_t bytes);
int mpsk_memcpy_d2d(mpsk_ctx* ctx, void* dst, const void* src, size_t bytes);

/* ---- MPO slice: SparseMPOSlice, src/operators/sparsempo/sparseslice.jl:13-27 ----------------
 * odim x odim blocks (column-major tables: entry (i,j) at i + odim*j).  chi_l[i] / chi_r[j] are
 * the MPO bond dimensions of level i (left) / j (right).  kind: MPSK_BLOCK_*.  scalars[i,j]: value of a
 * scalar block (c * identity, needs chi_l[i] == chi_r[j]).  blocks[i,j]: HOST pointer to a dense
 * block O[chi_l[i], d, d, chi_r[j]] column-major, index order [w, t(out), s(in), v]
 * (docs/src/man/intro.md:78-83, derivatives.jl:97), or NULL. */
int mpsk_mposlice_create(mpsk_ctx* ctx, int dtype, int odim, const int32_t* chi_l, const int32_t* chi_r,
                         int d, const int32_t* kind, const double* scalars, const void* const* blocks,
                         mpsk_mposlice** out);
int mpsk_mposlice_destroy(mpsk_mposlice* s);
int mpsk_mposlice_dims(const mpsk_mposlice* s, int* Wl, int* Wr, int* d);

/* ---- effective-Hamiltonian matvecs: src/algorithms/derivatives.jl ---------------------------
 * mpsk_dAC  == dAC(x, H::SparseMPOSlice, leftenv, rightenv)   derivatives.jl:77-104
 *   y[p,t,q] = sum GL[w][p,a] x[a,s,b] O[w,t,s,v] GR[v][b,q]
 *   GL: Wl slabs [Dlo, Dl]   (Dlo = number of output rows held by this rank; == Dl unsharded)
 *   GR: Wr slabs [Dr, Dr];  x: [Dl, d, Dr];  y: [Dlo, d, Dr]. */
int mpsk_dAC(mpsk_ctx* ctx, const mpsk_mposlice* H, int Dlo, int Dl, int Dr, const void* GL,
             const void* GR, const void* x, void* y);
/* The same matvec with x in the bond-sharded ("blocked") vector layout of the multi-GPU path: x is given
 * as nblk row blocks, block q = x[q Dl/nblk : (q+1) Dl/nblk, :, :] stored as a contiguous [Dl/nblk, d, Dr] tensor at
 * offset q (Dl/nblk) d Dr -- exactly what an all-gather of the ranks' output row blocks produces, so the Krylov vectors
 * of a sharded eigensolve never have to be re-interleaved.  y: [Dlo, d, Dr] (this rank's row block; the caller points
 * it INTO the blocked destination vector and all-gathers in place).  nblk == 1 is mpsk_dAC. */
int mpsk_dAC_blocked(mpsk_ctx* ctx, const mpsk_mposlice* H, int nblk, int Dlo, int Dl, int Dr, const void* GL,
                     const void* GR, const void* xblk, void* y);
/* Prepared effective Hamiltonian == the reference's MPO_ddAC object (derivatives.jl:11-15, 44-46): `ddAC(pos, psi, H, envs)`
 * builds it ONCE per site visit from (H[pos], leftenv, rightenv) and the Krylov solver applies it many times.
 * mpsk_hac_create folds the MPO tensor into the right environment (GRc[(w,s,t)] = sum_v O[w,t,s,v] GR[v], one
 * elementwise pass) whenever that does not cost extra GEMM work (Heisenberg / TFI-type MPOs): the application is then
 * TWO GEMM launches with no slab mix and one intermediate instead of two; otherwise it applies exactly what mpsk_dAC
 * does.  GL / GR must stay valid and unchanged while the object lives.  x may be in the blocked layout (nblk row
 * blocks, see mpsk_dAC_blocked; nblk = 1: plain).  mpsk_hac_info: mode 1 = combined environment (nslabs of them). */
typedef struct mpsk_hac mpsk_hac;
int mpsk_hac_create(mpsk_ctx* ctx, const mpsk_mposlice* H, int Dlo, int Dl, int Dr, const void* GL, const void* GR,
                    mpsk_hac** out);
int mpsk_hac_apply(mpsk_hac* h, const void* x, int nblk, void* y);
/* Fixed-budget smallest-real eigensolve with the prepared operator in ONE call: V[0] = x0 / |x0|, m Krylov steps (apply,
 * CGS2 + normalise), Ritz step of the projected matrix on the device, y = normalised Ritz vector -- fixedpoint(H_AC, AC, :SR,
 * Arnoldi(; krylovdim = m, maxiter = 1)) of dmrg.jl:36 without a convergence test and without a host synchronisation.
 * V: HOST array of m + 2 device vectors of the operator's size; scal: device scratch, >= m (2m + 1) + 40 doubles;
 * first_image (optional): H (x0 / |x0|) as the first step produced it (what calc_galerkin of the old tensor needs,
 * toolbox.jl:18).  1 <= m <= 32, MPSK_F64 operators, unblocked vectors. */
int mpsk_hac_eigsolve_fixed(mpsk_hac* hac, const void* x0, int m, void* const* V, void* scal, void* y, void* first_image);
int mpsk_hac_destroy(mpsk_hac* h);
int mpsk_hac_info(const mpsk_hac* h, int* mode, int* nslabs);
/* mpsk_dC == dC(x, leftenv::Vector, rightenv::Vector)   derivatives.jl:171-193
 *   y[p,q] = sum_w GL[w][p,a] c[a,b] GR[w][b,q] */
int mpsk_dC(mpsk_ctx* ctx, int W, int Dlo, int Dl, int Dr, const void* GL, const void* GR,
            const void* c, void* y);
/* mpsk_dAC2 == dAC2(x, h1, h2, leftenv, rightenv)   derivatives.jl:119-158
 *   x2, y2: [Dl, d1, Dr, d2] with (d1, d2) the physical dims of H1 / H2 */
int mpsk_dAC2(mpsk_ctx* ctx, const mpsk_mposlice* H1, const mpsk_mposlice* H2, int Dlo, int Dl, int Dr,
              const void* GL, const void* GR, const void* x2, void* y2);

/* ---- transfer matrices / environment updates: src/transfermatrix/transfer.jl ---------------
 * mpsk_transfer_left == transfer_left(vec, ham::SparseMPOSlice, A, Ab)   transfer.jl:166-211
 *   GLout[v][q,b] = sum GLin[w][p,a] A[a,s,b] O[w,t,s,v] conj(Ab[p,t,q])
 *   GLin: Wl slabs [Dlb, Dl]; A: [Dl,d,Dr]; Ab: [Dlb,d,Drb]; GLout: Wr slabs [Drb, Dr]
 * mpsk_transfer_right == transfer_right(vec, ham, A, Ab)                  transfer.jl:212-259
 *   GRout[w][a,p] = sum A[a,s,b] O[w,t,s,v] conj(Ab[p,t,q]) GRin[v][b,q]
 *   GRin: Wr slabs [Dr, Drb]; GRout: Wl slabs [Dl, Dlb]
 * H == NULL: pass-through legs, W independent slabs (transfer.jl:18-25,38-45,66-75). */
int mpsk_transfer_left(mpsk_ctx* ctx, const mpsk_mposlice* H, int W, int d, int Dl, int Dr, int Dlb,
                       int Drb, const void* GLin, const void* A, const void* Ab, void* GLout);
int mpsk_transfer_right(mpsk_ctx* ctx, const mpsk_mposlice* H, int W, int d, int Dl, int Dr, int Dlb,
                        int Drb, const void* A, const void* Ab, const void* GRin, void* GRout);
/* regularize!(v, lvec, rvec)   src/transfermatrix/transfermatrix.jl:70-76
 *   v[w] -= <lvec^T, v[w]> * rvec  for each of the W slabs [D1, D2]; lvec: [D2, D1]; rvec: [D1, D2] */
int mpsk_regularize(mpsk_ctx* ctx, int W, int D1, int D2, void* v, const void* lvec, const void* rvec);

/* ---- gauge steps: TensorKit leftorth!(QRpos) / rightorth!(LQpos) / tsvd! ----------------------
 * call sites: src/states/orthoview.jl:52,56 ; finitemps.jl:149 ; ortho.jl:128-136 ; dmrg.jl:96 */
/* A (m x n, m >= n, lda) = Q (m x n, ldq) * R (n x n upper, ldr), diag(R) > 0.  A is not modified. */
int mpsk_qrpos(mpsk_ctx* ctx, int m, int n, const void* A, int lda, void* Q, int ldq, void* R, int ldr);
/* two independent QRpos of equal shape issued together (two streams inside the ctx): the sweep needs
 * leftorth of the old AC (toolbox.jl:17-22) and of the new AC (orthoview.jl:56) at the same moment */
int mpsk_qrpos2(mpsk_ctx* ctx, int m, int n, const void* A1, int lda1, void* Q1, int ldq1, void* R1, int ldr1,
                const void* A2, int lda2, void* Q2, int ldq2, void* R2, int ldr2);
/* QRpos of A1 (m x n) and LQpos of A2 (n x m: A2 = L2 Q2, L2 n x n lower, Q2 n x m) issued together: the
 * left-moving site update needs leftorth of the old AC (toolbox.jl:17-22) and rightorth of the new AC
 * (orthoview.jl:52) at the same moment */
int mpsk_qrlq_pair(mpsk_ctx* ctx, int m, int n, const void* A1, int lda1, void* Q1, int ldq1, void* R1, int ldr1,
                   const void* A2, int lda2, void* L2, int ldl2, void* Q2, int ldq2);
/* A (m x n, m <= n) = L (m x m lower) * Q (m x n), diag(L) > 0 */
int mpsk_lqpos(mpsk_ctx* ctx, int m, int n, const void* A, int lda, void* L, int ldl, void* Q, int ldq);
/* Deferred completion of a gauge step.  After mpsk_ctx_qr_defer the NEXT mpsk_qrpos2 / mpsk_lqpos call on the ctx that
 * takes the CholeskyQR3 path returns as soon as its launches are enqueued: Q / R of mpsk_qrpos2 are speculative, L / Q of
 * mpsk_lqpos not yet written, until mpsk_qr_commit has read the device's success flag (and run the repeated third pass /
 * the fallbacks where it asks for them).  *redone: bit 0 / bit 1 = the first / second factorization of the pair was
 * corrected after its enqueue -- whatever was computed from the speculative factor has to be recomputed.  Between the
 * two calls only entry points that do not use the ctx workspace are accepted (mpsk_gemm, mpsk_v*): the DMRG sweep
 * enqueues the galerkin evaluation of the site there (toolbox.jl:17-22), so the stream has work while the host waits. */
int mpsk_ctx_qr_defer(mpsk_ctx* ctx);
/* Side stream: mpsk_ctx_side_mark records "now" on the ctx stream; after mpsk_ctx_side_begin every mpsk_* call of the ctx
 * runs on its second stream, ordered after the mark only; mpsk_ctx_side_end switches back and makes the main stream wait
 * for the side work.  Workspace users and the calls that use the second stream themselves (mpsk_qrpos2, mpsk_qrlq_pair,
 * mpsk_tsplit) are refused in between.  The sweep runs the galerkin evaluation of a left-moving visit there, under the
 * latency-bound CholeskyQR chain of the LQ step it has just enqueued on the main stream. */
int mpsk_ctx_side_mark(mpsk_ctx* ctx);
int mpsk_ctx_side_begin(mpsk_ctx* ctx);
int mpsk_ctx_side_end(mpsk_ctx* ctx);
int mpsk_qr_commit(mpsk_ctx* ctx, int* redone);
/* thin SVD of theta (m x n): theta = U diag(S) Vh, S descending.  U: m x kmax, S: kmax, Vh: kmax x n
 * buffers with kmax = min(m, n).  Truncation (TensorKit truncdim & truncerr, dmrg.jl:75,96):
 * keep at most max_keep (<= 0: no limit) values and drop the tail while ||S_dropped||_2 <= trunc_err -- an ABSOLUTE
 * bound, as TensorKit 0.12's truncerr(eps) (p = 2) applies it [restated from the published TensorKit source, which the
 * reference does not vendor: parity unpinned; equal to the relative rule for the normalised theta of DMRG2 / IDMRG2 /
 * real-time TDVP2].  *kept / *disc_norm are written on the host (sync).
 * Returns MPSK_ERR_HIP with "did not converge" if the Jacobi iteration is not orthogonal to 1e-9 after 40 sweeps. */
int mpsk_tsvd(mpsk_ctx* ctx, int m, int n, const void* theta, int ldt, void* U, int ldu, void* S, void* Vh,
              int ldv, int max_keep, double trunc_err, int* kept, double* disc_norm);
/* Truncated two-site split theta (m x n, min(m, n) > 64) ~ AL (m x k) . C (k x k) . AR (k x n): what dmrg.jl:96-104 /
 * tdvp.jl:124-126 build from tsvd! (al, c, ar), with the same truncation arguments as mpsk_tsvd, but computed without
 * accumulating the Jacobi rotations (a third less memory traffic per round): AL, AR are isometries to rounding,
 * AL C AR = theta projected on the kept singular subspace, C is TRIANGULAR instead of diag(S) -- lower or upper depending on
 * the orientation and on the svd mode (mode 2 delivers the left vectors of the tall orientation, the other factor comes from
 * an LQpos); S (min(m, n) doubles) receives all singular values (svd mode 3: see mpsk_ctx_set_svd_mode), *kept = k.
 * Buffers: AL m x min(m,n), C min(m,n)^2, AR min(m,n) x n (only the leading k columns / rows are written).
 * Any scale: a theta whose largest entry is outside [1e-100, 1e100] is split as theta / max|theta_ij| (C, S, disc_norm scaled
 * back); theta = 0 returns C = 0 with identity isometries. */
int mpsk_tsplit(mpsk_ctx* ctx, int m, int n, const void* theta, int ldt, int max_keep, double trunc_err,
                void* AL, int ldal, void* C, int ldc, void* AR, int ldar, void* S, int* kept, double* disc_norm);
/* Interleaved complex matrix (m x n complex: 2m x n doubles, ldh in doubles) <-> its real 2m x 2n embedding
 * E[:, 2b] = h_b, E[:, 2b+1] = J h_b, one launch each.  mpsk_cx_half takes the STRUCTURED PART of a nearly embedded matrix
 * (re = (E00 + E11) / 2, im = (E10 - E01) / 2 per 2 x 2 block).  For hosts that keep complex tensors bond-embedded
 * (mpskit.jl_amd/cplx.py) and iterate their Krylov solvers on interleaved vectors. */
int mpsk_cx_embed(mpsk_ctx* ctx, int m, int n, const void* H, int64_t ldh, void* E, int64_t lde);
int mpsk_cx_half(mpsk_ctx* ctx, int m, int n, const void* E, int64_t lde, void* H, int64_t ldh);
/* general column-major product C = alpha op(A) op(B) + beta C for the small gauge products
 * AC = AL*C, AC = C*AR, theta = AC*AR, AL = Q_AC*Q_C'  (orthoview.jl:99,103; dmrg.jl:92; ortho.jl:130) */
int mpsk_gemm(mpsk_ctx* ctx, int transA, int transB, int M, int N, int K, double alpha, const void* A,
              int64_t lda, const void* B, int64_t ldb, double beta, void* C, int64_t ldc);

/* strided column-major matrix copy dst[r, c] = src[r, c] (index permutations of small tensors, e.g.
 * AR[k, s, b] <- Vh[k, b, s] after tsvd!: dmrg.jl:104 `_transpose_front(ar)`) */
int mpsk_copy2d(mpsk_ctx* ctx, int rows, int cols, const void* src, int64_t lds, void* dst, int64_t ldd);

/* ---- Krylov vector protocol (VectorInterface: inner / add!! / scale!! / zerovector) -----------
 * KrylovKit needs these of the iterate type (quasiparticle_state.jl:357-411 is the in-repo example). */
int mpsk_vdot(mpsk_ctx* ctx, int64_t n, const void* x, const void* y, double* host_out);
int mpsk_vnrm2(mpsk_ctx* ctx, int64_t n, const void* x, double* host_out);
int mpsk_vaxpby(mpsk_ctx* ctx, int64_t n, double alpha, const void* x, double beta, void* y);
int mpsk_vscal(mpsk_ctx* ctx, int64_t n, double alpha, void* x);
/* y = (I (x) J) x, J = [[0,-1],[1,0]], on interleaved row pairs of a tensor whose first dimension is even: the
 * multiplication by i of a complex tensor carried as 2x2 real blocks on its bond indices (integrators.jl:21
 * `-1im * dt` on the embedded representation, mpskit.jl_amd/cplx.py) */
int mpsk_vtimes_i(mpsk_ctx* ctx, int64_t n, const void* x, void* y);
int mpsk_vcopy(mpsk_ctx* ctx, int64_t n, const void* x, void* y);
int mpsk_vzero(mpsk_ctx* ctx, int64_t n, void* x);
/* fused Gram-Schmidt helpers: xs = HOST array of k device pointers.
 *   mpsk_vmultidot : host_out[j] = <xs[j], y>                       (one sync for k dots)
 *   mpsk_vgs_step  : h = [<xs[j], y>]_j ; y -= sum_j h[j] xs[j] ; host_out[j] = h[j]
 *                    (coefficients stay on the device between the two kernels) */
int mpsk_vmultidot(mpsk_ctx* ctx, int64_t n, int k, const void* const* xs, const void* y, double* host_out);
int mpsk_vgs_step(mpsk_ctx* ctx, int64_t n, int k, const void* const* xs, void* y, double* host_out);
/*   mpsk_vorth_step: one full Krylov orthogonalisation step with a single host sync: CGS2 of y against
 *                    xs[0..k), then y <- y / ||y||; host_h[j] = coefficient on xs[j], *host_beta = ||y|| */
int mpsk_vorth_step(mpsk_ctx* ctx, int64_t n, int k, const void* const* xs, void* y, double* host_h,
                    double* host_beta);
/* the same step without a host synchronisation: the 2k+1 scalars (h1[k], h2[k], |remainder|^2; h = h1 + h2,
 * beta = sqrt of the last) stay in dev_out (device memory, >= 2k+1 doubles) */
int mpsk_vorth_step_dev(mpsk_ctx* ctx, int64_t n, int k, const void* const* xs, void* y, void* dev_out);
/* y = sum_j coefs[j] xs[j]   (Ritz vector assembly); asynchronous (host_coefs is copied before the call returns) */
int mpsk_vlincomb(mpsk_ctx* ctx, int64_t n, int k, const void* const* xs, const double* host_coefs, void* y);
/* ys[j] = sum_i host_coefs[i + k j] xs[i] for j < m, all in ONE pass over the vectors (k + m instead of ~m (k + 6) vector
 * passes): the basis rotation of KrylovKit's thick restart (shrink step of eigsolve / schursolve at krylovdim 30, which
 * fixedpoint.jl:19-30 runs with maxiter = 100).  k, m <= 32; outputs must not alias inputs; asynchronous like mpsk_vlincomb. */
int mpsk_vmultilincomb(mpsk_ctx* ctx, int64_t n, int k, const void* const* xs, int m, void* const* ys, const double* host_coefs);
/* y = x / |x| (y may be x) and dev_out[0] = |x|^2, both without a host synchronisation: the start / Ritz-vector
 * normalisations of a fixed-budget Krylov solve and the per-site galerkin norms of a sweep are read back once per
 * solve / sweep instead of stalling the stream at every use (normalize! in toolbox.jl:18, fixedpoint.jl:19-30).
 * dev_n2 (optional, device memory, one double) receives |x|^2. */
int mpsk_vnormalize_dev(mpsk_ctx* ctx, int64_t n, const void* x, void* y, void* dev_n2);
/* Ritz step of a fixed-budget solve without leaving the device: dev_slot holds, for step k at offset k * stride, the
 * 2 (k + 1) + 1 scalars of mpsk_vorth_step_dev; dev_coef[0..m) receives the eigenvector of the smallest eigenvalue of
 * the projected (m <= 32) matrix (positive component on the start vector; zeros beyond an invariant-subspace cut) and
 * dev_info (optional, 3 doubles) {eigenvalue, residual estimate, effective m}.  mpsk_vlincomb_dev assembles
 * y = sum_j dev_coefs[j] xs[j] from device-resident coefficients.  (KrylovKit's eigsolve does this step on the host;
 * fixedpoint.jl:19-30 only uses the vector.) */
int mpsk_vritz_dev(mpsk_ctx* ctx, int m, int stride, const void* dev_slot, void* dev_coef, void* dev_info);
int mpsk_vlincomb_dev(mpsk_ctx* ctx, int64_t n, int k, const void* const* xs, const void* dev_coefs, void* y);
int mpsk_vnrm2_dev(mpsk_ctx* ctx, int64_t n, const void* x, void* dev_out);

#ifdef __cplusplus
}
#endif
#endif /* MPSK_H */
