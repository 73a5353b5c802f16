// C ABI of libmpsk (see include/mpsk.h): argument checking, workspace management and the
// decomposition of every hot-path operator into  MFMA-GEMM -> slab-mix -> MFMA-GEMM  launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <string>
#include <vector>
#include "mpsk.h"
#include "mpsk_internal.h"

using namespace mpsk;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(MPSK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
  } while (0)
#define REQUIRE(cond, msg)                                                  \
  do {                                                                      \
    if (!(cond)) return fail(MPSK_ERR_INVALID, std::string(__func__) + ": " + (msg)); \
  } while (0)

struct mpsk_mposlice {
  mpsk_ctx* ctx;
  int Wl, Wr, d;
  int dtype = MPSK_F64;       // MPSK_C128: complex entries (Ofull_im), every operand of a call with this slice is complex128
  std::vector<double> Ofull;  // [Wl, d, d, Wr] column-major, index (w,t,s,v)
  std::vector<double> Ofull_im;
  MixPlan fwd;                // (w,s) -> (v,t)   in = s + d*w, out = t + d*v
  MixPlan bwd;                // (v,t) -> (w,s)   in = t + d*v, out = s + d*w
  MixPlan rgt;                // (v,s) -> (w,t)   in = s + d*v, out = t + d*w   (complex transfer_right: A GR first, then Ab^H)
  double Oi(int w, int t, int s, int v) const {
    return Ofull_im.empty() ? 0.0 : Ofull_im[w + (size_t)Wl * (t + d * (s + (size_t)d * v))];
  }
  std::vector<char> row_used, col_used;
  // "right-combined" form of the matvec (mpsk_hac): the MPO tensor is folded into the right environment once per
  // site visit,  GRc[c] = sum_v O[w_c, t_c, s_c, v] GR[v]  for every (w, s, t) with a non-zero entry, so that
  //   y[:, t, :] = sum_{c : t_c = t} T1[w_c][:, s_c, :] GRc[c]   needs no slab mix between the two GEMM stages.
  MixPlan rc;                      // out slab c <- in slab v
  std::vector<int> rc_w, rc_s, rc_t;
  int rc_nseg = 0;                 // max over t of the number of c with t_c = t (lists are padded to this length)
  double O(int w, int t, int s, int v) const { return Ofull[w + (size_t)Wl * (t + d * (s + (size_t)d * v))]; }
};

struct PoolBuf { void* p; size_t bytes; bool used; hipStream_t last = nullptr; hipEvent_t ev = nullptr; bool ev_set = false; };

struct mpsk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  double* d_scal = nullptr;     // [MAXK] device scalars
  double* d_partial = nullptr;  // dot scratch
  double* h_scal = nullptr;     // pinned host mirror
  double* d_coef = nullptr;     // [MAXK] coefficients of mpsk_vlincomb + their pinned source + the event that frees it
  double* h_coef = nullptr;
  hipEvent_t ev_coef = nullptr;
  bool coef_pending = false;
  int dtype = MPSK_F64;         // scalar type of the slice-less entry points (mpsk_ctx_set_dtype)
  int last_svd_sweeps = 0;
  int split_skip = 0, split_backoff = 0;           // calls that skip the stage after it gave up
  // subspace iterations the next truncation-aware mpsk_tsplit of the same (min(m, n), r) starts with: what the last one needed
  // (the bonds near the ends of a chain have other shapes and other spectra than the bulk; 8 for a shape not seen yet)
  std::map<std::pair<int, int>, int> split_q_hint;
  int last_split_iters = 0, last_split_path = 0;   // path: 0 full iteration (mode <= 2 / not applicable), 1 subspace stage, 2 stage gave up -> full
  double last_split_resid = 0.0;
  int svd_precondition = 3;     // mpsk_ctx_set_svd_mode: 0 plain, 1 QR-preconditioned, 2 QR + QR of R^T (mpsk_tsplit V-free, mpsk_tsvd with accumulated rotations), 3 = 2 + truncation-aware mpsk_tsplit
  int qr_mode = 0;              // 0 auto (CholeskyQR3 + Householder fallback), 1 Householder, 2 CholeskyQR3 only
  int* d_flag = nullptr;
  long n_qr_chol = 0, n_qr_house = 0, n_qr_fallback = 0, n_qr_robust = 0, n_qr_retry = 0;
  // first attempt of CholeskyQR3 with shift = qr_shift_fast * (bound of Fukaya et al.); a flagged breakdown repeats it with the bound
  double qr_shift_fast = getenv("MPSK_CQ_SHIFT_SCALE") ? atof(getenv("MPSK_CQ_SHIFT_SCALE")) : 1.0e-8;
  // second stream + workspace for two concurrent factorizations (mpsk_qrpos2)
  hipStream_t stream2 = nullptr;
  hipStream_t xstreams[3] = {nullptr, nullptr, nullptr};   // {stream2, two more}: chains of the block-Jacobi schedule (mpsk_svd.hip)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void* ws2 = nullptr;
  size_t ws2_bytes = 0;
  void* ws3 = nullptr;          // transposed operands of mpsk_qrlq_pair
  size_t ws3_bytes = 0;
  void* cxws[4] = {nullptr, nullptr, nullptr, nullptr};   // scratch: complex gauge steps (embedded operands; conjugate transposes; split), rescaled theta
  size_t cxws_bytes[4] = {0, 0, 0, 0};
  int* h_flags = nullptr;       // pinned [2]
  // deferred completion of a CholeskyQR gauge step (mpsk_ctx_qr_defer / mpsk_qr_commit): the launches are enqueued, the
  // success flag is read (and a fallback run) only at commit -- the caller fills the gap with work that does not need c->ws
  struct PendingQR {
    int active = 0;               // 0 none, 1 qrpos2, 2 lqpos
    int m = 0, n = 0;
    const void *A1 = nullptr, *A2 = nullptr; void *Q1 = nullptr, *R1 = nullptr, *Q2 = nullptr, *R2 = nullptr;
    int lda1 = 0, ldq1 = 0, ldr1 = 0, lda2 = 0, ldq2 = 0, ldr2 = 0;
    double *At = nullptr, *Qt = nullptr, *Rt = nullptr, *ws = nullptr;   // lqpos: transposed problem in c->ws
    void *L = nullptr, *Qo = nullptr; int ldl = 0, ldqo = 0;
  } pend;
  bool defer_next = false;
  bool on_side = false;         // mpsk_ctx_side_begin .. mpsk_ctx_side_end: `stream` and `stream2` are swapped
  std::map<std::pair<const mpsk_mposlice*, const mpsk_mposlice*>, MixPlan> pair_plans;
  std::vector<PoolBuf> pool;    // device buffers of prepared operators (mpsk_hac), reused across site visits
};

// prepared effective Hamiltonian of one site (MPO_ddAC of derivatives.jl:11-15: built once, applied by every Krylov step)
struct mpsk_hac {
  mpsk_ctx* ctx;
  const mpsk_mposlice* H;
  int Dlo, Dl, Dr;
  const double* GL;
  const double* GR;
  int mode;                     // 0: GEMM -> slab mix -> GEMM (mpsk_dAC);  1: right-combined environment, no mix
  double* GRc = nullptr;        // [nc + 1][Dr, Dr]   (last slab: zeros, pads the shorter segment lists)
  int64_t* zseg = nullptr;      // device [2][d][nseg]: A offsets (into T1), B offsets (into GRc)
  std::vector<int64_t> zseg_host;   // source of the asynchronous upload of zseg: lives as long as the handle (no sync)
  hipEvent_t ev_up = nullptr;       // completion of that upload (waited for before the handle is freed)
  int pool_idx = -1;
};
constexpr int MAXK = 256;
constexpr int MAXCOEF = 1024;   // coefficient staging: MAXK for mpsk_vlincomb, 32 x 32 for mpsk_vmultilincomb

extern "C" {

int mpsk_version(void) { return 100; }
const char* mpsk_last_error(void) { return g_err.c_str(); }

int mpsk_ctx_create(int device, mpsk_ctx** out) {
  REQUIRE(out != nullptr, "out is NULL");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  REQUIRE(device >= 0 && device < ndev, "no such HIP device (is a GPU visible?)");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MPSK_ERR_UNSUPPORTED, std::string("libmpsk is built for gfx950 only, device is ") + prop.gcnArchName);
  mpsk_ctx* c = new mpsk_ctx();
  c->device = device;
  HIPCHK(hipMalloc(&c->d_scal, sizeof(double) * MAXK));
  HIPCHK(hipMalloc(&c->d_partial, sizeof(double) * MPSK_DOT_SCRATCH));
  HIPCHK(hipHostMalloc(&c->h_scal, sizeof(double) * MAXK, hipHostMallocDefault));
  HIPCHK(hipMalloc(&c->d_coef, sizeof(double) * MAXCOEF));
  HIPCHK(hipHostMalloc(&c->h_coef, sizeof(double) * MAXCOEF, hipHostMallocDefault));
  HIPCHK(hipEventCreateWithFlags(&c->ev_coef, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipMalloc(&c->d_flag, 64));
  HIPCHK(hipHostMalloc(&c->h_flags, 64, hipHostMallocDefault));
  HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  c->xstreams[0] = c->stream2;
  HIPCHK(hipStreamCreateWithFlags(&c->xstreams[1], hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&c->xstreams[2], hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming | hipEventDisableSystemFence));
  HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming | hipEventDisableSystemFence));
  gemm_retain_stream(c->stream);
  gemm_retain_stream(c->stream2);
  gemm_retain_stream(c->xstreams[1]);
  gemm_retain_stream(c->xstreams[2]);
  *out = c;
  return MPSK_OK;
}

int mpsk_ctx_destroy(mpsk_ctx* c) {
  if (!c) return MPSK_OK;
  (void)hipSetDevice(c->device);
  // a ctx destroyed inside a side-stream section or with a deferred factorization pending: route back to the main stream
  // (otherwise the caller's stream would be destroyed below and the private one leaked) and complete the factorization
  if (c->on_side) (void)mpsk_ctx_side_end(c);
  if (c->pend.active) (void)mpsk_qr_commit(c, nullptr);
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  gemm_release_stream(c->stream);     // split-K partial-tile workspaces attached to this ctx's streams
  gemm_release_stream(c->stream2);
  for (auto& kv : c->pair_plans) mix_plan_destroy(&kv.second);
  for (auto& b : c->pool) { if (b.p) (void)hipFree(b.p); if (b.ev) (void)hipEventDestroy(b.ev); }
  if (c->ws) (void)hipFree(c->ws);
  if (c->d_scal) (void)hipFree(c->d_scal);
  if (c->d_partial) (void)hipFree(c->d_partial);
  if (c->h_scal) (void)hipHostFree(c->h_scal);
  if (c->d_coef) (void)hipFree(c->d_coef);
  if (c->h_coef) (void)hipHostFree(c->h_coef);
  if (c->ev_coef) (void)hipEventDestroy(c->ev_coef);
  if (c->d_flag) (void)hipFree(c->d_flag);
  if (c->h_flags) (void)hipHostFree(c->h_flags);
  if (c->ws2) (void)hipFree(c->ws2);
  if (c->ws3) (void)hipFree(c->ws3);
  for (int i = 0; i < 4; ++i) if (c->cxws[i]) (void)hipFree(c->cxws[i]);
  for (int i = 1; i < 3; ++i)
    if (c->xstreams[i]) { (void)hipStreamSynchronize(c->xstreams[i]); gemm_release_stream(c->xstreams[i]); (void)hipStreamDestroy(c->xstreams[i]); }
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  delete c;
  return MPSK_OK;
}

int mpsk_ctx_set_stream(mpsk_ctx* c, void* s) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(!c->on_side, "not inside a side-stream section (mpsk_ctx_side_end first)");
  REQUIRE(!c->pend.active, "a deferred factorization is pending (mpsk_qr_commit first)");
  if (c->stream != (hipStream_t)s) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    gemm_release_stream(c->stream);
    gemm_retain_stream((hipStream_t)s);
  }
  c->stream = (hipStream_t)s;
  return MPSK_OK;
}

int mpsk_ctx_set_dtype(mpsk_ctx* c, int dtype) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(dtype == MPSK_F64 || dtype == MPSK_C128, "dtype must be MPSK_F64 or MPSK_C128");
  c->dtype = dtype;
  return MPSK_OK;
}
int mpsk_ctx_get_stream(mpsk_ctx* c, void** s) {
  REQUIRE(c && s, "NULL argument");
  *s = (void*)c->stream;
  return MPSK_OK;
}
int mpsk_ctx_get_device(mpsk_ctx* c, int* device) {
  REQUIRE(c && device, "NULL argument");
  *device = c->device;
  return MPSK_OK;
}

int mpsk_ctx_synchronize(mpsk_ctx* c) {
  REQUIRE(c, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(c->stream));
  return MPSK_OK;
}

int mpsk_ctx_workspace_reserve(mpsk_ctx* c, size_t bytes) {
  REQUIRE(c, "ctx is NULL");
  if (bytes <= c->ws_bytes) return MPSK_OK;
  REQUIRE(!c->on_side && !c->pend.active, "the workspace is in use (side-stream section / deferred factorization)");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->ws) HIPCHK(hipFree(c->ws));
  c->ws = nullptr; c->ws_bytes = 0;
  size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
  hipError_t e = hipMalloc(&c->ws, want);
  if (e != hipSuccess) return fail(MPSK_ERR_NOMEM, "workspace hipMalloc failed");
  c->ws_bytes = want;
  return MPSK_OK;
}

int mpsk_ctx_force_tile(mpsk_ctx* c, int bm, int bn) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE((bm == 0 && bn == 0) || ((bm == 64 || bm == 128) && (bn == 64 || bn == 128)), "tile must be 64/128");
  gemm_force_tile(bm, bn);
  return MPSK_OK;
}

// Event profile of the matvec-stage GEMM launches (kernel dac_gemm_f64_kernel): enable, run, then
// read a JSON summary [{kernel, launches, total_ms, avg_ms, flops}] (device-synchronising).
int mpsk_prof_enable(mpsk_ctx* c, int on) {
  REQUIRE(c, "ctx is NULL");
  gemm_prof_enable(on != 0);
  return MPSK_OK;
}
int mpsk_prof_summary(mpsk_ctx* c, char* buf, size_t buflen) {
  REQUIRE(c && buf && buflen > 0, "NULL argument");
  std::string s = gemm_prof_summary();
  if (s.size() + 1 > buflen) return fail(MPSK_ERR_INVALID, "mpsk_prof_summary: buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return MPSK_OK;
}

int mpsk_malloc(mpsk_ctx* c, size_t bytes, void** dptr) {
  REQUIRE(c && dptr, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
  if (e != hipSuccess) return fail(MPSK_ERR_NOMEM, "hipMalloc failed");
  return MPSK_OK;
}
int mpsk_free(mpsk_ctx* c, void* dptr) {
  REQUIRE(c, "ctx is NULL");
  if (dptr) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(dptr)); }
  return MPSK_OK;
}
int mpsk_memcpy_h2d(mpsk_ctx* c, void* dst, const void* src, size_t bytes) {
  REQUIRE(c, "ctx is NULL");
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MPSK_OK;
}
int mpsk_memcpy_d2h(mpsk_ctx* c, void* dst, const void* src, size_t bytes) {
  REQUIRE(c, "ctx is NULL");
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MPSK_OK;
}
int mpsk_memcpy_d2d(mpsk_ctx* c, void* dst, const void* src, size_t bytes) {
  REQUIRE(c, "ctx is NULL");
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return MPSK_OK;
}

// --------------------------------------------------------------------------------------------
// MPO slices
// --------------------------------------------------------------------------------------------
int mpsk_mposlice_create(mpsk_ctx* c, int dtype, int odim, const int32_t* chi_l, const int32_t* chi_r,
                         int d, const int32_t* kind, const double* scalars, const void* const* blocks,
                         mpsk_mposlice** out) {
  REQUIRE(c && out && chi_l && chi_r && kind, "NULL argument");
  REQUIRE(dtype == MPSK_F64 || dtype == MPSK_C128, "dtype must be MPSK_F64 or MPSK_C128");
  const bool cx = dtype == MPSK_C128;     // scalars / dense blocks are then interleaved complex128 (re, im)
  REQUIRE(odim > 0 && d > 0, "odim and d must be positive");
  HIPCHK(hipSetDevice(c->device));
  std::vector<int> offl(odim + 1, 0), offr(odim + 1, 0);
  for (int i = 0; i < odim; ++i) {
    REQUIRE(chi_l[i] > 0 && chi_r[i] > 0, "chi must be positive");
    offl[i + 1] = offl[i] + chi_l[i];
    offr[i + 1] = offr[i] + chi_r[i];
  }
  auto* s = new mpsk_mposlice();
  s->ctx = c; s->Wl = offl[odim]; s->Wr = offr[odim]; s->d = d; s->dtype = dtype;
  s->Ofull.assign((size_t)s->Wl * d * d * s->Wr, 0.0);
  if (cx) s->Ofull_im.assign(s->Ofull.size(), 0.0);
  auto at = [&](int w, int t, int si, int v) -> double& {
    return s->Ofull[w + (size_t)s->Wl * (t + d * (si + (size_t)d * v))];
  };
  auto ati = [&](int w, int t, int si, int v) -> double& {
    return s->Ofull_im[w + (size_t)s->Wl * (t + d * (si + (size_t)d * v))];
  };
  for (int j = 0; j < odim; ++j)
    for (int i = 0; i < odim; ++i) {
      int k = kind[i + odim * j];
      if (k == MPSK_BLOCK_ZERO) continue;
      if (k == MPSK_BLOCK_SCALAR) {
        if (!scalars || chi_l[i] != chi_r[j]) { delete s; return fail(MPSK_ERR_INVALID, "scalar block needs scalars[] and chi_l[i] == chi_r[j]"); }
        const size_t e = (size_t)i + (size_t)odim * j;
        const double cv = cx ? scalars[2 * e] : scalars[e], ci = cx ? scalars[2 * e + 1] : 0.0;
        for (int a = 0; a < chi_l[i]; ++a)
          for (int t = 0; t < d; ++t) {
            at(offl[i] + a, t, t, offr[j] + a) = cv;
            if (cx) ati(offl[i] + a, t, t, offr[j] + a) = ci;
          }
      } else if (k == MPSK_BLOCK_DENSE) {
        if (!blocks || !blocks[i + odim * j]) { delete s; return fail(MPSK_ERR_INVALID, "dense block pointer is NULL"); }
        const double* b = (const double*)blocks[i + odim * j];
        const int cl = chi_l[i], cr = chi_r[j];
        for (int v = 0; v < cr; ++v)
          for (int si = 0; si < d; ++si)
            for (int t = 0; t < d; ++t)
              for (int w = 0; w < cl; ++w) {
                const size_t e = w + (size_t)cl * (t + d * (si + (size_t)d * v));
                at(offl[i] + w, t, si, offr[j] + v) = cx ? b[2 * e] : b[e];
                if (cx) ati(offl[i] + w, t, si, offr[j] + v) = b[2 * e + 1];
              }
      } else { delete s; return fail(MPSK_ERR_INVALID, "bad block kind"); }
    }
  std::vector<MixTerm> fwd, bwd, rgt;
  s->row_used.assign(s->Wl, 0); s->col_used.assign(s->Wr, 0);
  for (int v = 0; v < s->Wr; ++v)
    for (int si = 0; si < d; ++si)
      for (int t = 0; t < d; ++t)
        for (int w = 0; w < s->Wl; ++w) {
          const double cv = s->O(w, t, si, v), ci = s->Oi(w, t, si, v);
          if (cv == 0.0 && ci == 0.0) continue;
          fwd.push_back({t + d * v, si + d * w, cv, ci});
          bwd.push_back({si + d * w, t + d * v, cv, ci});
          rgt.push_back({t + d * w, si + d * v, cv, ci});
          s->row_used[w] = 1; s->col_used[v] = 1;
        }
  HIPCHK(mix_plan_create(fwd, d * s->Wr, d * s->Wl, &s->fwd));
  HIPCHK(mix_plan_create(bwd, d * s->Wl, d * s->Wr, &s->bwd));
  HIPCHK(mix_plan_create(rgt, d * s->Wl, d * s->Wr, &s->rgt));
  if (!cx) {  // right-combined plan (real slices; complex operators use the mix form)
    std::vector<MixTerm> rc;
    std::vector<int> per_t(d, 0);
    for (int t = 0; t < d; ++t)
      for (int w = 0; w < s->Wl; ++w)
        for (int si = 0; si < d; ++si) {
          bool any = false;
          for (int v = 0; v < s->Wr; ++v)
            if (s->O(w, t, si, v) != 0.0) { rc.push_back({(int32_t)s->rc_w.size(), v, s->O(w, t, si, v)}); any = true; }
          if (any) { s->rc_w.push_back(w); s->rc_s.push_back(si); s->rc_t.push_back(t); per_t[t]++; }
        }
    for (int t = 0; t < d; ++t) if (per_t[t] > s->rc_nseg) s->rc_nseg = per_t[t];
    HIPCHK(mix_plan_create(rc, (int)s->rc_w.size(), s->Wr, &s->rc));
  }
  *out = s;
  return MPSK_OK;
}

int mpsk_mposlice_destroy(mpsk_mposlice* s) {
  if (!s) return MPSK_OK;
  mpsk_ctx* c = s->ctx;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto it = c->pair_plans.begin(); it != c->pair_plans.end();) {
    if (it->first.first == s || it->first.second == s) { mix_plan_destroy(&it->second); it = c->pair_plans.erase(it); }
    else ++it;
  }
  mix_plan_destroy(&s->fwd);
  mix_plan_destroy(&s->bwd);
  mix_plan_destroy(&s->rgt);
  mix_plan_destroy(&s->rc);
  delete s;
  return MPSK_OK;
}

int mpsk_mposlice_dims(const mpsk_mposlice* s, int* Wl, int* Wr, int* d) {
  REQUIRE(s, "slice is NULL");
  if (Wl) *Wl = s->Wl;
  if (Wr) *Wr = s->Wr;
  if (d) *d = s->d;
  return MPSK_OK;
}

}  // extern "C"

// --------------------------------------------------------------------------------------------
// helpers
// --------------------------------------------------------------------------------------------
static int ensure_ws(mpsk_ctx* c, size_t bytes) {
  if (c->pend.active) return fail(MPSK_ERR_INVALID, "a deferred factorization still owns the workspace: call mpsk_qr_commit first");
  if (c->on_side) return fail(MPSK_ERR_INVALID, "workspace users are not accepted on the side stream: call mpsk_ctx_side_end first");
  if (bytes <= c->ws_bytes) return MPSK_OK;
  return mpsk_ctx_workspace_reserve(c, bytes + bytes / 4);
}

static GemmArgs mk(const double* A, const double* B, double* C, int M, int N, int K, int64_t lda, int64_t ldb,
                   int64_t ldc, int tA = 0, int tB = 0) {
  GemmArgs g;
  std::memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.batch = 1; g.nseg = 1; g.alpha = 1.0; g.beta = 0.0; g.transA = tA; g.transB = tB;
  return g;
}

// run a GEMM whose K dimension is a list of segments longer than MAXSEG by chunking (beta = 1)
static hipError_t gemm_segments(GemmArgs g, const std::vector<int64_t>& sa, const std::vector<int64_t>& sb,
                                hipStream_t s, const std::vector<int>* sj = nullptr) {
  size_t n = sa.size();
  if (n == 0) return hipErrorInvalidValue;
  double beta0 = g.beta;
  for (size_t i0 = 0; i0 < n; i0 += MAXSEG) {
    int ns = (int)std::min<size_t>(MAXSEG, n - i0);
    g.nseg = ns;
    for (int i = 0; i < ns; ++i) {
      g.segA[i] = sa[i0 + i]; g.segB[i] = sb[i0 + i];
      g.segJ[i] = sj ? (signed char)(*sj)[i0 + i] : 0;
    }
    g.beta = (i0 == 0) ? beta0 : 1.0;
    hipError_t e = gemm_f64(g, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

static hipError_t zero_async(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s); }

extern "C" {

// --------------------------------------------------------------------------------------------
// derivatives
// --------------------------------------------------------------------------------------------
// ---- complex128 operators -----------------------------------------------------------------------------------------
// Every complex tensor argument is interleaved complex128 in TensorKit / Julia layout = a REAL column-major tensor whose
// first dimension is doubled, rows (2 a, 2 a + 1) = (re, im) of row a.  With A_half such a view and B = Br + i Bi:
//   (A B)_half = A_half Br + (J A_half) Bi ,     J (re, im) = (-im, re)
// so a complex product is the real GEMM core with two K-segments (the second through the J-aware A loader, GemmArgs::segJ)
// on a B operand split into planes -- 4x the real flops, the complex optimum; intermediates stay interleaved.
static hipError_t cx_planes(const double* z, size_t n, double* re, double* im, hipStream_t s) {     // interleaved -> planar
  if ((uintptr_t)z % 16 == 0) return deinterleave(z, (int64_t)n, re, im, s);
  hipError_t e = copy_strided(z, 2, 0, re, 1, 0, (int64_t)n, 1, s);
  if (e != hipSuccess) return e;
  return copy_strided(z + 1, 2, 0, im, 1, 0, (int64_t)n, 1, s);
}
static size_t ev2(size_t v) { return (v + 1) & ~(size_t)1; }

// GRp: planar copy of GR ([re plane | im plane], each Wr Dr Dr doubles) if the caller has one (prepared operator), else null
static int dAC_c128(mpsk_ctx* c, const mpsk_mposlice* H, int Dlo, int Dl, int Dr, const double* GL, const double* GR,
                    const double* GRp, const double* x, double* y) {   // GRp planes are ev2(Wr Dr Dr) doubles apart
  HIPCHK(hipSetDevice(c->device));
  const int d = H->d, Wl = H->Wl, Wr = H->Wr;
  const size_t nx = (size_t)Dl * d * Dr, nG = (size_t)Wr * Dr * Dr, slab = (size_t)2 * Dlo * d * Dr;
  const size_t o_g = 2 * ev2(nx), o_t1 = o_g + (GRp ? 0 : 2 * ev2(nG)), o_t2 = o_t1 + slab * Wl;
  if (int rc = ensure_ws(c, sizeof(double) * (o_t2 + slab * Wr))) return rc;
  double* xp = (double*)c->ws;
  double* T1 = xp + o_t1;
  double* T2 = xp + o_t2;
  HIPCHK(cx_planes(x, nx, xp, xp + ev2(nx), c->stream));
  const double* gr = GRp;
  size_t gplane = ev2(nG);
  if (!gr) { double* g = xp + o_g; HIPCHK(cx_planes(GR, nG, g, g + ev2(nG), c->stream)); gr = g; gplane = ev2(nG); }
  // stage 1: T1[w] = GL[w] x
  GemmArgs g1 = mk(GL, xp, T1, 2 * Dlo, d * Dr, Dl, 2 * Dlo, Dl, 2 * Dlo);
  g1.batch = Wl; g1.bsA = (int64_t)2 * Dlo * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.nseg = 2; g1.segA[0] = g1.segA[1] = 0; g1.segB[0] = 0; g1.segB[1] = (int64_t)ev2(nx); g1.segJ[0] = 0; g1.segJ[1] = 1;
  g1.cplx = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  SlabIndex ix{d, 1 << 30, (int64_t)2 * Dlo, (int64_t)slab, 0, (int64_t)2 * Dlo * d};
  HIPCHK(mix_apply(H->fwd, T1, ix, T2, ix, 2 * Dlo, Dr, c->stream));
  std::vector<int64_t> sa, sb;
  std::vector<int> sj;
  for (int v = 0; v < Wr; ++v)
    if (H->col_used[v])
      for (int pl = 0; pl < 2; ++pl) { sa.push_back((int64_t)v * slab); sb.push_back((int64_t)(pl * gplane + (size_t)v * Dr * Dr)); sj.push_back(pl); }
  if (sa.empty()) { HIPCHK(zero_async(y, sizeof(double) * slab, c->stream)); return MPSK_OK; }
  GemmArgs g3 = mk(T2, gr, y, 2 * Dlo * d, Dr, Dr, (int64_t)2 * Dlo * d, Dr, (int64_t)2 * Dlo * d);
  g3.cplx = 1;
  HIPCHK(gemm_segments(g3, sa, sb, c->stream, &sj));
  return MPSK_OK;
}

static int dC_c128(mpsk_ctx* c, int W, int Dlo, int Dl, int Dr, const double* GL, const double* GR, const double* cm,
                   double* y) {
  HIPCHK(hipSetDevice(c->device));
  const size_t nx = (size_t)Dl * Dr, nG = (size_t)W * Dr * Dr, slab = (size_t)2 * Dlo * Dr;
  const size_t o_g = 2 * ev2(nx), o_t1 = o_g + 2 * ev2(nG);
  if (int rc = ensure_ws(c, sizeof(double) * (o_t1 + slab * W))) return rc;
  double* xp = (double*)c->ws;
  double* g = xp + o_g;
  double* T1 = xp + o_t1;
  HIPCHK(cx_planes(cm, nx, xp, xp + ev2(nx), c->stream));
  HIPCHK(cx_planes(GR, nG, g, g + ev2(nG), c->stream));
  GemmArgs g1 = mk(GL, xp, T1, 2 * Dlo, Dr, Dl, 2 * Dlo, Dl, 2 * Dlo);
  g1.batch = W; g1.bsA = (int64_t)2 * Dlo * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.nseg = 2; g1.segB[1] = (int64_t)ev2(nx); g1.segJ[1] = 1; g1.cplx = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  std::vector<int64_t> sa, sb;
  std::vector<int> sj;
  for (int w = 0; w < W; ++w)
    for (int pl = 0; pl < 2; ++pl) { sa.push_back((int64_t)w * slab); sb.push_back((int64_t)(pl * ev2(nG) + (size_t)w * Dr * Dr)); sj.push_back(pl); }
  GemmArgs g3 = mk(T1, g, y, 2 * Dlo, Dr, Dr, 2 * Dlo, Dr, 2 * Dlo);
  g3.cplx = 1;
  HIPCHK(gemm_segments(g3, sa, sb, c->stream, &sj));
  return MPSK_OK;
}

static int pair_plan(mpsk_ctx* c, const mpsk_mposlice* H1, const mpsk_mposlice* H2, const MixPlan** out);

static int dAC2_c128(mpsk_ctx* c, const mpsk_mposlice* H1, const mpsk_mposlice* H2, int Dlo, int Dl, int Dr,
                     const double* GL, const double* GR, const double* x2, double* y2) {
  HIPCHK(hipSetDevice(c->device));
  const int d1 = H1->d, d2 = H2->d, Wl = H1->Wl, Wr = H2->Wr;
  const size_t nx = (size_t)Dl * d1 * Dr * d2, nG = (size_t)Wr * Dr * Dr;
  const size_t plane = (size_t)2 * Dlo * d1 * Dr, slab = plane * d2;
  const size_t o_g = 2 * ev2(nx), o_t1 = o_g + 2 * ev2(nG), o_t2 = o_t1 + slab * Wl;
  if (int rc = ensure_ws(c, sizeof(double) * (o_t2 + slab * Wr))) return rc;
  double* xp = (double*)c->ws;
  double* g = xp + o_g;
  double* T1 = xp + o_t1;
  double* T2 = xp + o_t2;
  const MixPlan* plan = nullptr;
  if (int rc = pair_plan(c, H1, H2, &plan)) return rc;
  HIPCHK(cx_planes(x2, nx, xp, xp + ev2(nx), c->stream));
  HIPCHK(cx_planes(GR, nG, g, g + ev2(nG), c->stream));
  GemmArgs g1 = mk(GL, xp, T1, 2 * Dlo, d1 * Dr * d2, Dl, 2 * Dlo, Dl, 2 * Dlo);
  g1.batch = Wl; g1.bsA = (int64_t)2 * Dlo * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.nseg = 2; g1.segB[1] = (int64_t)ev2(nx); g1.segJ[1] = 1; g1.cplx = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  SlabIndex ix{d1, d2, (int64_t)2 * Dlo, (int64_t)plane, (int64_t)slab, (int64_t)2 * Dlo * d1};
  HIPCHK(mix_apply(*plan, T1, ix, T2, ix, 2 * Dlo, Dr, c->stream));
  std::vector<int64_t> sa, sb;
  std::vector<int> sj;
  for (int v = 0; v < Wr; ++v)
    if (H2->col_used[v])
      for (int pl = 0; pl < 2; ++pl) { sa.push_back((int64_t)v * slab); sb.push_back((int64_t)(pl * ev2(nG) + (size_t)v * Dr * Dr)); sj.push_back(pl); }
  if (sa.empty()) { HIPCHK(zero_async(y2, sizeof(double) * slab, c->stream)); return MPSK_OK; }
  GemmArgs g3 = mk(T2, g, y2, 2 * Dlo * d1, Dr, Dr, (int64_t)2 * Dlo * d1, Dr, (int64_t)2 * Dlo * d1);
  g3.batch = d2; g3.bsA = (int64_t)plane; g3.bsB = 0; g3.bsC = (int64_t)plane; g3.cplx = 1;
  HIPCHK(gemm_segments(g3, sa, sb, c->stream, &sj));
  return MPSK_OK;
}

// GLout[v][q,b] = sum GLin[w][p,a] A[a,s,b] O[w,t,s,v] conj(Ab[p,t,q])
static int transfer_left_c128(mpsk_ctx* c, const mpsk_mposlice* H, int W, int d, int Dl, int Dr, int Dlb, int Drb,
                              const double* GLin, const double* A, const double* Ab, double* GLout) {
  HIPCHK(hipSetDevice(c->device));
  int Wl = W, Wr = W;
  if (H) { Wl = H->Wl; Wr = H->Wr; d = H->d; }
  const size_t nA = (size_t)Dl * d * Dr, slab = (size_t)2 * Dlb * d * Dr;
  const size_t o_t1 = 2 * ev2(nA), o_t2 = o_t1 + slab * Wl;
  if (int rc = ensure_ws(c, sizeof(double) * (o_t2 + (H ? slab * Wr : 0)))) return rc;
  double* Ap = (double*)c->ws;
  double* T1 = Ap + o_t1;
  double* T2 = H ? Ap + o_t2 : T1;
  HIPCHK(cx_planes(A, nA, Ap, Ap + ev2(nA), c->stream));
  // T1[w] = GLin[w] A      (2 Dlb x d Dr, rows interleaved)
  GemmArgs g1 = mk(GLin, Ap, T1, 2 * Dlb, d * Dr, Dl, 2 * Dlb, Dl, 2 * Dlb);
  g1.batch = Wl; g1.bsA = (int64_t)2 * Dlb * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.nseg = 2; g1.segB[1] = (int64_t)ev2(nA); g1.segJ[1] = 1; g1.cplx = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  if (H) {
    SlabIndex ix{d, 1 << 30, (int64_t)2 * Dlb, (int64_t)slab, 0, (int64_t)2 * Dlb * d};
    HIPCHK(mix_apply(H->fwd, T1, ix, T2, ix, 2 * Dlb, Dr, c->stream));
  }
  // GLout[v] = Ab^H T2[v]:  K = (re / im, p, t) interleaved on both operands ->  real part = Ab_half^T T2_half,
  // imaginary part = (J Ab_half)^T T2_half ; plane alpha goes to the rows 2 q + alpha of the interleaved result
  for (int al = 0; al < 2; ++al) {
    GemmArgs g3 = mk(Ab, T2, GLout + al, Drb, Dr, 2 * Dlb * d, (int64_t)2 * Dlb * d, (int64_t)2 * Dlb * d, (int64_t)2 * Drb, 1, 0);
    g3.batch = Wr; g3.bsA = 0; g3.bsB = (int64_t)slab; g3.bsC = (int64_t)2 * Drb * Dr;
    g3.cplx = 1; g3.c_rs = 2; g3.segJ[0] = (signed char)al;
    HIPCHK(gemm_f64(g3, c->stream));
  }
  return MPSK_OK;
}

// GRout[w][a,p] = sum A[a,s,b] O[w,t,s,v] conj(Ab[p,t,q]) GRin[v][b,q]   evaluated as  ((A GRin) mixed) Ab^H
static int transfer_right_c128(mpsk_ctx* c, const mpsk_mposlice* H, int W, int d, int Dl, int Dr, int Dlb, int Drb,
                               const double* A, const double* Ab, const double* GRin, double* GRout) {
  HIPCHK(hipSetDevice(c->device));
  int Wl = W, Wr = W;
  if (H) { Wl = H->Wl; Wr = H->Wr; d = H->d; }
  const size_t nG = (size_t)Wr * Dr * Drb, nB = (size_t)Dlb * d * Drb, slab = (size_t)2 * Dl * d * Drb;
  const size_t o_b = 2 * ev2(nG), o_x = o_b + 2 * ev2(nB), o_y = o_x + slab * Wr;
  if (int rc = ensure_ws(c, sizeof(double) * (o_y + (H ? slab * Wl : 0)))) return rc;
  double* Gp = (double*)c->ws;
  double* Bp = Gp + o_b;
  double* X = Gp + o_x;
  double* Y = H ? Gp + o_y : X;
  HIPCHK(cx_planes(GRin, nG, Gp, Gp + ev2(nG), c->stream));
  HIPCHK(cx_planes(Ab, nB, Bp, Bp + ev2(nB), c->stream));
  // X[v][a,s,q] = sum_b A[a,s,b] GRin[v][b,q]      A as the (2 Dl d) x Dr interleaved matrix
  GemmArgs g1 = mk(A, Gp, X, 2 * Dl * d, Drb, Dr, (int64_t)2 * Dl * d, Dr, (int64_t)2 * Dl * d);
  g1.batch = Wr; g1.bsA = 0; g1.bsB = (int64_t)Dr * Drb; g1.bsC = (int64_t)slab;
  g1.nseg = 2; g1.segB[1] = (int64_t)ev2(nG); g1.segJ[1] = 1; g1.cplx = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  if (H) {   // Y[w][a,t,q] = sum_{v,s} O[w,t,s,v] X[v][a,s,q]
    SlabIndex ix{d, 1 << 30, (int64_t)2 * Dl, (int64_t)slab, 0, (int64_t)2 * Dl * d};
    HIPCHK(mix_apply(H->rgt, X, ix, Y, ix, 2 * Dl, Drb, c->stream));
  }
  // GRout[w][a,p] = sum_{t,q} Y[w][a,(t,q)] conj(Ab[p,(t,q)]) = Y_half Abr^T + (-J Y_half) Abi^T
  GemmArgs g3 = mk(Y, Bp, GRout, 2 * Dl, Dlb, d * Drb, (int64_t)2 * Dl, Dlb, (int64_t)2 * Dl, 0, 1);
  g3.batch = Wl; g3.bsA = (int64_t)slab; g3.bsB = 0; g3.bsC = (int64_t)2 * Dl * Dlb;
  g3.nseg = 2; g3.segB[1] = (int64_t)ev2(nB); g3.segJ[1] = 2; g3.cplx = 1;
  HIPCHK(gemm_f64(g3, c->stream));
  return MPSK_OK;
}

// x may be given in `nblk` row blocks (block q = rows [q Dl/nblk, (q+1) Dl/nblk) as a contiguous [Dl/nblk, d, Dr]
// tensor): the blocks become K-segments of the stage-1 GEMM, nothing is re-interleaved (mpsk_dAC_blocked).
static int dAC_impl(mpsk_ctx* c, const mpsk_mposlice* H, int Dlo, int Dl, int Dr, const void* GL, const void* GR,
                    const void* x, int nblk, void* y) {
  HIPCHK(hipSetDevice(c->device));
  const int d = H->d, Wl = H->Wl, Wr = H->Wr;
  const size_t slab = (size_t)Dlo * d * Dr;
  if (int rc = ensure_ws(c, sizeof(double) * slab * (Wl + Wr))) return rc;
  double* T1 = (double*)c->ws;
  double* T2 = T1 + slab * Wl;
  // stage 1: T1[w] = GL[w] * x     (batched over w)
  const int kb = Dl / nblk;
  GemmArgs g1 = mk((const double*)GL, (const double*)x, T1, Dlo, d * Dr, kb, Dlo, kb, Dlo);
  g1.batch = Wl; g1.bsA = (int64_t)Dlo * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.nseg = nblk;
  for (int q = 0; q < nblk; ++q) { g1.segA[q] = (int64_t)q * kb * Dlo; g1.segB[q] = (int64_t)q * kb * d * Dr; }
  g1.tag = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  // stage 2: T2[v][:,t,:] = sum_{w,s} O[w,t,s,v] T1[w][:,s,:]
  SlabIndex ix{d, 1 << 30, (int64_t)Dlo, (int64_t)slab, 0, (int64_t)Dlo * d};
  HIPCHK(mix_apply(H->fwd, T1, ix, T2, ix, Dlo, Dr, c->stream));
  // stage 3: y = sum_v T2[v] * GR[v]   (K-segments over v)
  std::vector<int64_t> sa, sb;
  for (int v = 0; v < Wr; ++v) if (H->col_used[v]) { sa.push_back((int64_t)v * slab); sb.push_back((int64_t)v * Dr * Dr); }
  if (sa.empty()) { HIPCHK(zero_async(y, sizeof(double) * slab, c->stream)); return MPSK_OK; }
  GemmArgs g3 = mk(T2, (const double*)GR, (double*)y, Dlo * d, Dr, Dr, (int64_t)Dlo * d, Dr, (int64_t)Dlo * d);
  g3.tag = 1;
  HIPCHK(gemm_segments(g3, sa, sb, c->stream));
  return MPSK_OK;
}

int mpsk_dAC(mpsk_ctx* c, const mpsk_mposlice* H, int Dlo, int Dl, int Dr, const void* GL, const void* GR,
             const void* x, void* y) {
  REQUIRE(c && H && GL && GR && x && y, "NULL argument");
  REQUIRE(Dlo > 0 && Dl > 0 && Dr > 0, "dimensions must be positive");
  if (H->dtype == MPSK_C128)
    return dAC_c128(c, H, Dlo, Dl, Dr, (const double*)GL, (const double*)GR, nullptr, (const double*)x, (double*)y);
  return dAC_impl(c, H, Dlo, Dl, Dr, GL, GR, x, 1, y);
}

int mpsk_dAC_blocked(mpsk_ctx* c, const mpsk_mposlice* H, int nblk, int Dlo, int Dl, int Dr, const void* GL,
                     const void* GR, const void* xblk, void* y) {
  REQUIRE(c && H && GL && GR && xblk && y, "NULL argument");
  REQUIRE(Dlo > 0 && Dl > 0 && Dr > 0, "dimensions must be positive");
  REQUIRE(nblk >= 1 && nblk <= MAXSEG && Dl % nblk == 0, "nblk must divide Dl (and be <= 32)");
  if (H->dtype == MPSK_C128) {
    if (nblk != 1) return fail(MPSK_ERR_UNSUPPORTED, "mpsk_dAC_blocked: the blocked layout is implemented for MPSK_F64 only");
    return dAC_c128(c, H, Dlo, Dl, Dr, (const double*)GL, (const double*)GR, nullptr, (const double*)xblk, (double*)y);
  }
  return dAC_impl(c, H, Dlo, Dl, Dr, GL, GR, xblk, nblk, y);
}

// ---- prepared operator ---------------------------------------------------------------------------
static int pool_take(mpsk_ctx* c, size_t bytes, void** p, int* idx) {
  int best = -1;
  for (size_t i = 0; i < c->pool.size(); ++i)
    if (!c->pool[i].used && c->pool[i].bytes >= bytes && (best < 0 || c->pool[i].bytes < c->pool[best].bytes)) best = (int)i;
  if (best < 0) {
    // drop idle buffers that are too small before growing (bond dimensions grow monotonically along a chain)
    for (auto& b : c->pool)
      if (!b.used && b.p) { (void)hipStreamSynchronize(c->stream); (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    void* q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) return fail(MPSK_ERR_NOMEM, "prepared-operator buffer hipMalloc failed");
    int slot = -1;
    for (size_t i = 0; i < c->pool.size(); ++i) if (!c->pool[i].p) { slot = (int)i; break; }
    if (slot < 0) { c->pool.push_back({nullptr, 0, false}); slot = (int)c->pool.size() - 1; }
    c->pool[slot] = {q, bytes, false};
    best = slot;
  }
  c->pool[best].used = true;
  // the buffer's previous owner may have been applied on another stream (side-stream sections, rebound ctx): its last
  // applications must have completed before the new owner overwrites GRc
  if (c->pool[best].ev_set && c->pool[best].last != c->stream) (void)hipStreamWaitEvent(c->stream, c->pool[best].ev, 0);
  c->pool[best].ev_set = false;
  *p = c->pool[best].p;
  *idx = best;
  return MPSK_OK;
}

int mpsk_hac_create(mpsk_ctx* c, const mpsk_mposlice* H, int Dlo, int Dl, int Dr, const void* GL, const void* GR,
                    mpsk_hac** out) {
  REQUIRE(c && H && GL && GR && out, "NULL argument");
  REQUIRE(Dlo > 0 && Dl > 0 && Dr > 0, "dimensions must be positive");
  HIPCHK(hipSetDevice(c->device));
  const int d = H->d, Wl = H->Wl, Wr = H->Wr;
  int wr_used = 0;
  for (int v = 0; v < Wr; ++v) wr_used += H->col_used[v] ? 1 : 0;
  // cost model (stage 1 is the same in both forms): stage-3 slab products at ~50 TFLOP/s, plus -- for the mix form --
  // the slab-mix pass over T1 / T2 (HBM / L3 traffic at ~3 TB/s) and its launch.  Measured on MI355X: the mix costs
  // 24 us at D = 1024 (Heisenberg), a dependent launch ~6 us including its gap.
  const double u3 = 2.0 * Dlo * (double)Dr * Dr;
  const double slab_bytes = 8.0 * Dlo * (double)d * Dr;
  const double t_mix = (double)wr_used * d * u3 / 50e12 + (Wl + wr_used) * slab_bytes / 3e12 + 6e-6;
  const double t_rc = (double)H->rc_nseg * d * u3 / 50e12;
  auto* h = new mpsk_hac();
  h->ctx = c; h->H = H; h->Dlo = Dlo; h->Dl = Dl; h->Dr = Dr; h->GL = (const double*)GL; h->GR = (const double*)GR;
  h->mode = (H->rc_nseg > 0 && t_rc <= t_mix) ? 1 : 0;
  if (const char* ev = getenv("MPSK_HAC_MODE")) h->mode = (ev[0] == '1' && H->rc_nseg > 0) ? 1 : 0;
  if (H->dtype == MPSK_C128) {
    // complex128: mix form; what is prepared once per site is the planar copy of the right environment (the B operand
    // of stage 3), so that an application converts only x
    h->mode = 2;
    const size_t nG = (size_t)Wr * Dr * Dr;
    void* buf = nullptr;
    if (int rc = pool_take(c, sizeof(double) * 2 * ev2(nG), &buf, &h->pool_idx)) { delete h; return rc; }
    h->GRc = (double*)buf;
    hipError_t e = cx_planes((const double*)GR, nG, h->GRc, h->GRc + ev2(nG), c->stream);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(c->stream);
      if (h->ev_up) (void)hipEventDestroy(h->ev_up);
      c->pool[h->pool_idx].used = false; delete h; return fail(MPSK_ERR_HIP, hipGetErrorString(e));
    }
    *out = h;
    return MPSK_OK;
  }
  if (h->mode == 1) {
    const int nc = (int)H->rc_w.size(), ns = H->rc_nseg;
    const size_t slabR = (size_t)Dr * Dr;
    const size_t bytes = sizeof(double) * slabR * (nc + 1) + sizeof(int64_t) * 2 * d * ns + 64;
    void* buf = nullptr;
    if (int rc = pool_take(c, bytes, &buf, &h->pool_idx)) { delete h; return rc; }
    h->GRc = (double*)buf;
    h->zseg = (int64_t*)((char*)buf + ((sizeof(double) * slabR * (nc + 1) + 15) & ~(size_t)15));
    // GRc[c] = sum_v O[w_c, t_c, s_c, v] GR[v] ; zero pad slab at index nc
    SlabIndex ix{1 << 30, 1, (int64_t)slabR, 0, 0, (int64_t)Dr};
    hipError_t e = mix_apply(H->rc, (const double*)GR, ix, h->GRc, ix, Dr, Dr, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->GRc + slabR * nc, 0, sizeof(double) * slabR, c->stream);
    // segment tables: stage-3 batch z = t;  A offset into T1 (slab w, plane s), B offset into GRc (slab c)
    const size_t slab1 = (size_t)Dlo * d * Dr;
    h->zseg_host.assign((size_t)2 * d * ns, 0);
    std::vector<int64_t>& tab = h->zseg_host;
    for (int t = 0; t < d; ++t) {
      int k = 0;
      for (int ci = 0; ci < nc; ++ci)
        if (H->rc_t[ci] == t) {
          tab[(size_t)t * ns + k] = (int64_t)H->rc_w[ci] * slab1 + (int64_t)H->rc_s[ci] * Dlo;
          tab[(size_t)(d + t) * ns + k] = (int64_t)ci * slabR;
          ++k;
        }
      for (; k < ns; ++k) { tab[(size_t)t * ns + k] = 0; tab[(size_t)(d + t) * ns + k] = (int64_t)nc * slabR; }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h->zseg, tab.data(), sizeof(int64_t) * tab.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming | hipEventDisableSystemFence);
    if (e == hipSuccess) e = hipEventRecord(h->ev_up, c->stream);
    if (e != hipSuccess) { c->pool[h->pool_idx].used = false; delete h; return fail(MPSK_ERR_HIP, hipGetErrorString(e)); }
  }
  *out = h;
  return MPSK_OK;
}

int mpsk_hac_destroy(mpsk_hac* h) {
  if (!h) return MPSK_OK;
  if (h->pool_idx >= 0 && h->pool_idx < (int)h->ctx->pool.size()) {
    PoolBuf& b = h->ctx->pool[h->pool_idx];
    b.used = false;
    // stream ordering of the hand-back: everything enqueued so far on the ctx's current stream still reads the buffer
    (void)hipSetDevice(h->ctx->device);
    if (!b.ev) (void)hipEventCreateWithFlags(&b.ev, hipEventDisableTiming | hipEventDisableSystemFence);
    if (b.ev && hipEventRecord(b.ev, h->ctx->stream) == hipSuccess) { b.last = h->ctx->stream; b.ev_set = true; }
  }
  if (h->ev_up) { (void)hipEventSynchronize(h->ev_up); (void)hipEventDestroy(h->ev_up); }   // zseg_host is still being read until then
  delete h;
  return MPSK_OK;
}

int mpsk_hac_info(const mpsk_hac* h, int* mode, int* nslabs) {
  REQUIRE(h, "hac is NULL");
  if (mode) *mode = h->mode;
  if (nslabs) *nslabs = h->mode == 1 ? (int)h->H->rc_w.size() : 0;
  return MPSK_OK;
}

int mpsk_hac_apply(mpsk_hac* h, const void* x, int nblk, void* y) {
  REQUIRE(h && x && y, "NULL argument");
  mpsk_ctx* c = h->ctx;
  const mpsk_mposlice* H = h->H;
  const int Dlo = h->Dlo, Dl = h->Dl, Dr = h->Dr;
  REQUIRE(nblk >= 1 && nblk <= MAXSEG && Dl % nblk == 0, "nblk must divide Dl (and be <= 32)");
  if (h->mode == 2) {
    if (nblk != 1) return fail(MPSK_ERR_UNSUPPORTED, "mpsk_hac_apply: the blocked layout is implemented for MPSK_F64 only");
    return dAC_c128(c, H, Dlo, Dl, Dr, h->GL, h->GR, h->GRc, (const double*)x, (double*)y);
  }
  if (h->mode == 0) return dAC_impl(c, H, Dlo, Dl, Dr, h->GL, h->GR, x, nblk, y);
  HIPCHK(hipSetDevice(c->device));
  const int d = H->d, Wl = H->Wl, ns = H->rc_nseg;
  const size_t slab = (size_t)Dlo * d * Dr;
  if (int rc = ensure_ws(c, sizeof(double) * slab * Wl)) return rc;
  double* T1 = (double*)c->ws;
  // stage 1: T1[w] = GL[w] * x     (batched over w; x possibly in row blocks = K segments)
  const int kb = Dl / nblk;
  GemmArgs g1 = mk(h->GL, (const double*)x, T1, Dlo, d * Dr, kb, Dlo, kb, Dlo);
  g1.batch = Wl; g1.bsA = (int64_t)Dlo * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.nseg = nblk;
  for (int q = 0; q < nblk; ++q) { g1.segA[q] = (int64_t)q * kb * Dlo; g1.segB[q] = (int64_t)q * kb * d * Dr; }
  g1.tag = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  // stage 3: y[:, t, :] = sum_{c: t_c = t} T1[w_c][:, s_c, :] GRc[c]      (batch over t, per-batch K-segment tables)
  GemmArgs g3 = mk(T1, h->GRc, (double*)y, Dlo, Dr, Dr, (int64_t)Dlo * d, Dr, (int64_t)Dlo * d);
  g3.batch = d; g3.bsA = 0; g3.bsB = 0; g3.bsC = (int64_t)Dlo;
  g3.nseg = ns; g3.zsegA = h->zseg; g3.zsegB = h->zseg + (size_t)d * ns;
  g3.tabs_even = (Dlo % 2 == 0) && (Dr % 2 == 0);
  g3.tag = 1;
  HIPCHK(gemm_f64(g3, c->stream));
  return MPSK_OK;
}

// Fixed-budget smallest-real solve with the prepared operator, one call per site (fixedpoint(H_AC, AC, :SR, alg) of
// dmrg.jl:36 with Arnoldi(; krylovdim = m, maxiter = 1) and no convergence test): V[0] = x0 / |x0|, m Krylov steps
// (apply, CGS2 + normalise), Ritz step of the m x m projected matrix on the device, y = normalised Ritz vector.  Nothing is
// read back; the ~30 entry-point calls the host made per site become one (at D = 256 the host could not keep the stream fed).
// V: HOST array of m + 2 device vectors of the operator's size (Krylov basis + assembly scratch); scal: device scratch of
// >= m (2m + 1) + 40 doubles; first_image (optional): receives H (x0 / |x0|) before it is orthogonalised (calc_galerkin of the
// old tensor needs exactly that vector, toolbox.jl:18).  MPSK_F64 operators, unblocked vector layout.
int mpsk_hac_eigsolve_fixed(mpsk_hac* h, const void* x0, int m, void* const* V, void* scal, void* y, void* first_image) {
  REQUIRE(h && x0 && V && scal && y, "NULL argument");
  REQUIRE(m >= 1 && m <= 32, "needs 1 <= m <= 32");
  REQUIRE(h->mode != 2, "mpsk_hac_eigsolve_fixed: MPSK_F64 operators only");
  mpsk_ctx* c = h->ctx;
  HIPCHK(hipSetDevice(c->device));
  const int64_t n = (int64_t)h->Dlo * h->H->d * h->Dr;
  REQUIRE(h->Dlo == h->Dl, "mpsk_hac_eigsolve_fixed: the operator must be square (unsharded)");
  double* slot = (double*)scal;
  const int stride = 2 * m + 1;
  double* rbuf = slot + (size_t)m * stride;           // Ritz coefficients [0:32], info [32:35]
  if (int rc = mpsk_vnormalize_dev(c, n, x0, V[0], nullptr)) return rc;
  for (int k = 0; k < m; ++k) {
    if (int rc = mpsk_hac_apply(h, V[k], 1, V[k + 1])) return rc;
    if (k == 0 && first_image) HIPCHK(hipMemcpyAsync(first_image, V[1], sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(vec_cgs2((const double* const*)V, k + 1, (double*)V[k + 1], n, slot + (size_t)k * stride, c->d_partial, c->stream));
    HIPCHK(vec_scal_rsqrt_dev(slot + (size_t)k * stride + 2 * (k + 1), (double*)V[k + 1], n, c->stream));
  }
  HIPCHK(vec_ritz_small(slot, m, stride, rbuf, rbuf + 32, c->stream));
  HIPCHK(hipMemsetAsync(V[m + 1], 0, sizeof(double) * n, c->stream));
  HIPCHK(vec_multiaxpy((const double* const*)V, rbuf, m, 1.0, (double*)V[m + 1], n, c->stream));
  return mpsk_vnormalize_dev(c, n, V[m + 1], y, nullptr);
}

int mpsk_dC(mpsk_ctx* c, int W, int Dlo, int Dl, int Dr, const void* GL, const void* GR, const void* cm, void* y) {
  REQUIRE(c && GL && GR && cm && y, "NULL argument");
  REQUIRE(W > 0 && Dlo > 0 && Dl > 0 && Dr > 0, "dimensions must be positive");
  if (c->dtype == MPSK_C128)
    return dC_c128(c, W, Dlo, Dl, Dr, (const double*)GL, (const double*)GR, (const double*)cm, (double*)y);
  HIPCHK(hipSetDevice(c->device));
  const size_t slab = (size_t)Dlo * Dr;
  if (int rc = ensure_ws(c, sizeof(double) * slab * W)) return rc;
  double* T1 = (double*)c->ws;
  GemmArgs g1 = mk((const double*)GL, (const double*)cm, T1, Dlo, Dr, Dl, Dlo, Dl, Dlo);
  g1.batch = W; g1.bsA = (int64_t)Dlo * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.tag = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  std::vector<int64_t> sa, sb;
  for (int w = 0; w < W; ++w) { sa.push_back((int64_t)w * slab); sb.push_back((int64_t)w * Dr * Dr); }
  GemmArgs g3 = mk(T1, (const double*)GR, (double*)y, Dlo, Dr, Dr, Dlo, Dr, Dlo);
  g3.tag = 1;
  HIPCHK(gemm_segments(g3, sa, sb, c->stream));
  return MPSK_OK;
}

static int pair_plan(mpsk_ctx* c, const mpsk_mposlice* H1, const mpsk_mposlice* H2, const MixPlan** out) {
  auto key = std::make_pair(H1, H2);
  auto it = c->pair_plans.find(key);
  if (it != c->pair_plans.end()) { *out = &it->second; return MPSK_OK; }
  const int d1 = H1->d, d2 = H2->d, Wl = H1->Wl, Wm = H1->Wr, Wr = H2->Wr;
  // O12[(t1,t2,v) <- (s1,s2,w)] = sum_u O1[w,t1,s1,u] O2[u,t2,s2,v]   (derivatives.jl:128-147)
  std::map<std::pair<int, int>, std::pair<double, double>> acc;
  for (int u = 0; u < Wm; ++u)
    for (int w = 0; w < Wl; ++w)
      for (int t1 = 0; t1 < d1; ++t1)
        for (int s1 = 0; s1 < d1; ++s1) {
          const double a = H1->O(w, t1, s1, u), ai = H1->Oi(w, t1, s1, u);
          if (a == 0.0 && ai == 0.0) continue;
          for (int v = 0; v < Wr; ++v)
            for (int t2 = 0; t2 < d2; ++t2)
              for (int s2 = 0; s2 < d2; ++s2) {
                const double b = H2->O(u, t2, s2, v), bi = H2->Oi(u, t2, s2, v);
                if (b == 0.0 && bi == 0.0) continue;
                int o = t1 + d1 * (t2 + d2 * v), in = s1 + d1 * (s2 + d2 * w);
                auto& e = acc[{o, in}];
                e.first += a * b - ai * bi;
                e.second += a * bi + ai * b;
              }
        }
  std::vector<MixTerm> terms;
  for (auto& kv : acc)
    if (kv.second.first != 0.0 || kv.second.second != 0.0)
      terms.push_back({kv.first.first, kv.first.second, kv.second.first, kv.second.second});
  MixPlan p;
  HIPCHK(mix_plan_create(terms, d1 * d2 * Wr, d1 * d2 * Wl, &p));
  auto res = c->pair_plans.emplace(key, p);
  *out = &res.first->second;
  return MPSK_OK;
}

int mpsk_dAC2(mpsk_ctx* c, const mpsk_mposlice* H1, const mpsk_mposlice* H2, int Dlo, int Dl, int Dr,
              const void* GL, const void* GR, const void* x2, void* y2) {
  REQUIRE(c && H1 && H2 && GL && GR && x2 && y2, "NULL argument");
  REQUIRE(H1->Wr == H2->Wl, "MPO bond dimensions of the two slices do not match");
  REQUIRE(Dlo > 0 && Dl > 0 && Dr > 0, "dimensions must be positive");
  REQUIRE(H1->dtype == H2->dtype, "the two slices have different scalar types");
  if (H1->dtype == MPSK_C128)
    return dAC2_c128(c, H1, H2, Dlo, Dl, Dr, (const double*)GL, (const double*)GR, (const double*)x2, (double*)y2);
  HIPCHK(hipSetDevice(c->device));
  const int d1 = H1->d, d2 = H2->d, Wl = H1->Wl, Wr = H2->Wr;
  const size_t plane = (size_t)Dlo * d1 * Dr;      // one s2-plane
  const size_t slab = plane * d2;                  // one w
  if (int rc = ensure_ws(c, sizeof(double) * slab * (Wl + Wr))) return rc;
  double* T1 = (double*)c->ws;
  double* T2 = T1 + slab * Wl;
  const MixPlan* plan = nullptr;
  if (int rc = pair_plan(c, H1, H2, &plan)) return rc;
  GemmArgs g1 = mk((const double*)GL, (const double*)x2, T1, Dlo, d1 * Dr * d2, Dl, Dlo, Dl, Dlo);
  g1.batch = Wl; g1.bsA = (int64_t)Dlo * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  g1.tag = 1;
  HIPCHK(gemm_f64(g1, c->stream));
  SlabIndex ix{d1, d2, (int64_t)Dlo, (int64_t)plane, (int64_t)slab, (int64_t)Dlo * d1};
  HIPCHK(mix_apply(*plan, T1, ix, T2, ix, Dlo, Dr, c->stream));
  std::vector<int64_t> sa, sb;
  for (int v = 0; v < Wr; ++v) if (H2->col_used[v]) { sa.push_back((int64_t)v * slab); sb.push_back((int64_t)v * Dr * Dr); }
  if (sa.empty()) { HIPCHK(zero_async(y2, sizeof(double) * slab, c->stream)); return MPSK_OK; }
  GemmArgs g3 = mk(T2, (const double*)GR, (double*)y2, Dlo * d1, Dr, Dr, (int64_t)Dlo * d1, Dr, (int64_t)Dlo * d1);
  g3.tag = 1;
  g3.batch = d2; g3.bsA = (int64_t)plane; g3.bsB = 0; g3.bsC = (int64_t)plane;
  HIPCHK(gemm_segments(g3, sa, sb, c->stream));
  return MPSK_OK;
}

// --------------------------------------------------------------------------------------------
// transfers
// --------------------------------------------------------------------------------------------
int mpsk_transfer_left(mpsk_ctx* c, const mpsk_mposlice* H, int W, int d, int Dl, int Dr, int Dlb, int Drb,
                       const void* GLin, const void* A, const void* Ab, void* GLout) {
  REQUIRE(c && GLin && A && Ab && GLout, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  int Wl = W, Wr = W;
  if (H) { Wl = H->Wl; Wr = H->Wr; d = H->d; }
  REQUIRE(Wl > 0 && d > 0 && Dl > 0 && Dr > 0 && Dlb > 0 && Drb > 0, "dimensions must be positive");
  if ((H ? H->dtype : c->dtype) == MPSK_C128)
    return transfer_left_c128(c, H, W, d, Dl, Dr, Dlb, Drb, (const double*)GLin, (const double*)A, (const double*)Ab, (double*)GLout);
  const size_t slab = (size_t)Dlb * d * Dr;
  if (int rc = ensure_ws(c, sizeof(double) * slab * (Wl + (H ? Wr : 0)))) return rc;
  double* T1 = (double*)c->ws;
  double* T2 = H ? T1 + slab * Wl : T1;
  // T1[w][p,s,b] = sum_a GLin[w][p,a] A[a,s,b]
  GemmArgs g1 = mk((const double*)GLin, (const double*)A, T1, Dlb, d * Dr, Dl, Dlb, Dl, Dlb);
  g1.batch = Wl; g1.bsA = (int64_t)Dlb * Dl; g1.bsB = 0; g1.bsC = (int64_t)slab;
  HIPCHK(gemm_f64(g1, c->stream));
  if (H) {
    SlabIndex ix{d, 1 << 30, (int64_t)Dlb, (int64_t)slab, 0, (int64_t)Dlb * d};
    HIPCHK(mix_apply(H->fwd, T1, ix, T2, ix, Dlb, Dr, c->stream));
  }
  // GLout[v][q,b] = sum_{(p,t)} Ab[(p,t),q] T2[v][(p,t),b]
  GemmArgs g3 = mk((const double*)Ab, T2, (double*)GLout, Drb, Dr, Dlb * d, (int64_t)Dlb * d, (int64_t)Dlb * d, Drb, 1, 0);
  g3.batch = Wr; g3.bsA = 0; g3.bsB = (int64_t)slab; g3.bsC = (int64_t)Drb * Dr;
  HIPCHK(gemm_f64(g3, c->stream));
  return MPSK_OK;
}

int mpsk_transfer_right(mpsk_ctx* c, const mpsk_mposlice* H, int W, int d, int Dl, int Dr, int Dlb, int Drb,
                        const void* A, const void* Ab, const void* GRin, void* GRout) {
  REQUIRE(c && GRin && A && Ab && GRout, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  int Wl = W, Wr = W;
  if (H) { Wl = H->Wl; Wr = H->Wr; d = H->d; }
  REQUIRE(Wl > 0 && d > 0 && Dl > 0 && Dr > 0 && Dlb > 0 && Drb > 0, "dimensions must be positive");
  if ((H ? H->dtype : c->dtype) == MPSK_C128)
    return transfer_right_c128(c, H, W, d, Dl, Dr, Dlb, Drb, (const double*)A, (const double*)Ab, (const double*)GRin, (double*)GRout);
  const size_t plane = (size_t)Dr * Dlb;   // one physical index
  const size_t slab = plane * d;
  if (int rc = ensure_ws(c, sizeof(double) * slab * (Wr + (H ? Wl : 0)))) return rc;
  double* U = (double*)c->ws;
  double* V = H ? U + slab * Wr : U;
  // U[v][b,p,t] = sum_q GRin[v][b,q] Ab[(p,t),q]
  GemmArgs g1 = mk((const double*)GRin, (const double*)Ab, U, Dr, Dlb * d, Drb, Dr, (int64_t)Dlb * d, Dr, 0, 1);
  g1.batch = Wr; g1.bsA = (int64_t)Dr * Drb; g1.bsB = 0; g1.bsC = (int64_t)slab;
  HIPCHK(gemm_f64(g1, c->stream));
  if (H) {
    SlabIndex ix{d, 1 << 30, (int64_t)plane, (int64_t)slab, 0, (int64_t)Dr};
    HIPCHK(mix_apply(H->bwd, U, ix, V, ix, Dr, Dlb, c->stream));
  }
  // GRout[w][a,p] = sum_s sum_b A[a,s,b] V[w][s][b,p]
  std::vector<int64_t> sa, sb;
  for (int s = 0; s < d; ++s) { sa.push_back((int64_t)s * Dl); sb.push_back((int64_t)s * plane); }
  GemmArgs g3 = mk((const double*)A, V, (double*)GRout, Dl, Dlb, Dr, (int64_t)Dl * d, Dr, Dl);
  g3.batch = Wl; g3.bsA = 0; g3.bsB = (int64_t)slab; g3.bsC = (int64_t)Dl * Dlb;
  HIPCHK(gemm_segments(g3, sa, sb, c->stream));
  return MPSK_OK;
}

int mpsk_regularize(mpsk_ctx* c, int W, int D1, int D2, void* v, const void* lvec, const void* rvec) {
  REQUIRE(c && v && lvec && rvec, "NULL argument");
  REQUIRE(W > 0 && D1 > 0 && D2 > 0, "dimensions must be positive");
  HIPCHK(hipSetDevice(c->device));
  if (int rc = ensure_ws(c, sizeof(double) * regularize_workspace_doubles(W, D1, D2))) return rc;
  HIPCHK(regularize(W, D1, D2, (double*)v, (const double*)lvec, (const double*)rvec, (double*)c->ws, c->stream));
  return MPSK_OK;
}

// QRpos dispatcher: CholeskyQR3 (GEMM-rich) for n > 64; when the device flag reports a non-positive pivot / a
// Gram matrix far from the identity (ill-conditioned or rank-deficient input) the perturbed, repeatedly shifted
// variant (cholqr_robust) runs, and only if that fails too the blocked Householder kernel.
static int qrpos_dispatch(mpsk_ctx* c, int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr,
                          double* ws) {
  std::string err;
  if (c->qr_mode != 1 && n > 64) {
    int flag = 0;
    hipError_t e = cholqr3(m, n, A, lda, Q, ldq, R, ldr, ws, c->d_flag, &flag, c->stream, c->qr_shift_fast);
    if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3: ") + hipGetErrorString(e));
    if (flag == 0) { c->n_qr_chol++; return MPSK_OK; }
    if (c->qr_shift_fast < 1.0) {          // breakdown with the rounding-level shift: the published shift next
      c->n_qr_retry++;
      e = cholqr3(m, n, A, lda, Q, ldq, R, ldr, ws, c->d_flag, &flag, c->stream, 1.0);
      if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3 (retry): ") + hipGetErrorString(e));
      if (flag == 0) { c->n_qr_chol++; return MPSK_OK; }
    }
    if (c->qr_mode == 2) return fail(MPSK_ERR_INVALID, "cholqr3: matrix too ill-conditioned / rank deficient");
    c->n_qr_fallback++;
    // second line of defence, still on the GEMM core: perturbed, repeatedly shifted CholeskyQR
    e = cholqr_robust(m, n, A, lda, Q, ldq, R, ldr, ws, c->d_flag, &flag, c->stream);
    if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr_robust: ") + hipGetErrorString(e));
    if (flag == 0) { c->n_qr_robust++; return MPSK_OK; }
  }
  c->n_qr_house++;
  hipError_t e = qrpos(m, n, A, lda, Q, ldq, R, ldr, ws, c->stream, &err);
  if (e != hipSuccess) return fail(err.empty() ? MPSK_ERR_HIP : MPSK_ERR_INVALID, "qrpos: " + (err.empty() ? std::string(hipGetErrorString(e)) : err));
  return MPSK_OK;
}

static size_t qr_ws_doubles(int m, int n) {
  size_t a = qrpos_workspace_doubles(m, n), b = cholqr_workspace_doubles(m, n);
  return a > b ? a : b;
}

int mpsk_ctx_set_qr_mode(mpsk_ctx* c, int mode) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (auto), 1 (Householder) or 2 (CholeskyQR3 only)");
  c->qr_mode = mode;
  return MPSK_OK;
}
int mpsk_ctx_qr_stats(mpsk_ctx* c, long* n_chol, long* n_house, long* n_fallback, long* n_robust) {
  REQUIRE(c, "ctx is NULL");
  if (n_chol) *n_chol = c->n_qr_chol;
  if (n_house) *n_house = c->n_qr_house;
  if (n_fallback) *n_fallback = c->n_qr_fallback;
  if (n_robust) *n_robust = c->n_qr_robust;
  return MPSK_OK;
}

int mpsk_ctx_qr_retries(mpsk_ctx* c, long* n_retry) {
  REQUIRE(c && n_retry, "NULL argument");
  *n_retry = c->n_qr_retry;
  return MPSK_OK;
}

static int qrpos_c128(mpsk_ctx* c, int m, int n, const void* A, int lda, void* Q, int ldq, void* R, int ldr);
static int lqpos_c128(mpsk_ctx* c, int m, int n, const void* A, int lda, void* L, int ldl, void* Q, int ldq);
static int lqpos_f64(mpsk_ctx* c, int m, int n, const void* A, int lda, void* L, int ldl, void* Q, int ldq);

static int qrpos_f64(mpsk_ctx* c, int m, int n, const void* A, int lda, void* Q, int ldq, void* R, int ldr) {
  REQUIRE(c && A && Q && R, "NULL argument");
  REQUIRE(m >= n && n > 0, "needs m >= n > 0");
  REQUIRE(lda >= m && ldq >= m && ldr >= n, "leading dimension too small");
  HIPCHK(hipSetDevice(c->device));
  c->defer_next = false;                       // (single factorizations complete at once)
  if (int rc = ensure_ws(c, sizeof(double) * qr_ws_doubles(m, n))) return rc;
  return qrpos_dispatch(c, m, n, (const double*)A, lda, (double*)Q, ldq, (double*)R, ldr, (double*)c->ws);
}
// the ABI entry: fp64, or complex128 when the ctx dtype says so (mpsk_ctx_set_dtype; interleaved complex operands)
int mpsk_qrpos(mpsk_ctx* c, int m, int n, const void* A, int lda, void* Q, int ldq, void* R, int ldr) {
  REQUIRE(c, "ctx is NULL");
  return c->dtype == MPSK_C128 ? qrpos_c128(c, m, n, A, lda, Q, ldq, R, ldr) : qrpos_f64(c, m, n, A, lda, Q, ldq, R, ldr);
}

// ---- complex128 gauge steps (ctx dtype MPSK_C128) ---------------------------------------------------------------------
// Operands are interleaved complex128 in column-major order (Julia Array{ComplexF64}): a complex m x n matrix is a real
// 2m x n matrix H whose row pairs hold (re, im); leading dimensions count COMPLEX elements.  The factorization runs on the
// real 2m x 2n embedding E = [h_0 | J h_0 | h_1 | J h_1 ...] (J (re, im) = (-im, re)): the real QRpos of an embedding is the
// embedding of the complex QRpos (R_E upper triangular with a positive diagonal, and the factorization is unique), so the
// whole CholeskyQR3 / fallback ladder is reused; the price is 2x the flops of a native complex kernel in the GEMM parts and a
// Cholesky chain of 2n instead of n columns.  Numerically the columns of Q_E along weakly determined directions are not
// embeddings (perturbed CholeskyQR, Householder completion of rank-deficient input); then Q := the structured part of Q_E,
// re-orthonormalised by a factorization that is now well conditioned, and R = triu(Q^H A) -- backward stable because those
// directions carry weight sigma_j (cplx.py: qrpos_structured, the same remedy on the host side).
__global__ __launch_bounds__(256) void cx_embed_kernel(const double* __restrict__ H, int64_t ldh, int m, int n,
                                                       double* __restrict__ E, int64_t lde) {
  const int64_t tot = (int64_t)m * n;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(t % m), b = (int)(t / m);
    const double re = H[2 * a + ldh * b], im = H[2 * a + 1 + ldh * b];
    double* e0 = E + 2 * a + lde * (2 * (int64_t)b);
    double* e1 = e0 + lde;
    e0[0] = re; e0[1] = im; e1[0] = -im; e1[1] = re;
  }
}
// H = structured part of the (nearly) embedded E: re = (E00 + E11) / 2, im = (E10 - E01) / 2; D (optional) = E - embed(H);
// upper != 0: complex entries below the diagonal are zeroed and the diagonal is made real (triangular factors)
__global__ __launch_bounds__(256) void cx_half_kernel(const double* __restrict__ E, int64_t lde, int m, int n,
                                                      double* __restrict__ H, int64_t ldh, double* __restrict__ D, int64_t ldd,
                                                      int upper) {
  const int64_t tot = (int64_t)m * n;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(t % m), b = (int)(t / m);
    const double* e0 = E + 2 * a + lde * (2 * (int64_t)b);
    const double* e1 = e0 + lde;
    double re = 0.5 * (e0[0] + e1[1]), im = 0.5 * (e0[1] - e1[0]);
    if (upper) { if (a > b) re = im = 0.0; else if (a == b) im = 0.0; }
    H[2 * a + ldh * b] = re; H[2 * a + 1 + ldh * b] = im;
    if (D) {
      double* d0 = D + 2 * a + ldd * (2 * (int64_t)b);
      double* d1 = d0 + ldd;
      d0[0] = e0[0] - re; d0[1] = e0[1] - im; d1[0] = e1[0] + im; d1[1] = e1[1] - re;
    }
  }
}
// out (n x m complex) = conjugate transpose of in (m x n complex)
__global__ __launch_bounds__(256) void cx_ctranspose_kernel(const double* __restrict__ in, int64_t ldi, int m, int n,
                                                            double* __restrict__ out, int64_t ldo) {
  const int64_t tot = (int64_t)m * n;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t % m), j = (int)(t / m);
    out[2 * j + ldo * i] = in[2 * i + ldi * j];
    out[2 * j + 1 + ldo * i] = -in[2 * i + 1 + ldi * j];
  }
}
static int cx_scratch(mpsk_ctx* c, int which, size_t bytes, double** out) {
  if (c->cxws_bytes[which] < bytes) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->cxws[which]) HIPCHK(hipFree(c->cxws[which]));
    c->cxws[which] = nullptr; c->cxws_bytes[which] = 0;
    if (hipMalloc(&c->cxws[which], bytes) != hipSuccess) return fail(MPSK_ERR_NOMEM, "complex gauge step: workspace hipMalloc failed");
    c->cxws_bytes[which] = bytes;
  }
  *out = (double*)c->cxws[which];
  return MPSK_OK;
}
static int qrpos_c128(mpsk_ctx* c, int m, int n, const void* A, int lda, void* Q, int ldq, void* R, int ldr) {
  REQUIRE(c && A && Q && R, "NULL argument");
  REQUIRE(m >= n && n > 0, "needs m >= n > 0");
  REQUIRE(lda >= m && ldq >= m && ldr >= n, "leading dimension too small");
  HIPCHK(hipSetDevice(c->device));
  const int m2 = 2 * m, n2 = 2 * n;
  const size_t en = (size_t)m2 * n2, rn = (size_t)n2 * n2;
  double* buf = nullptr;
  if (int rc = cx_scratch(c, 0, sizeof(double) * (3 * en + rn), &buf)) return rc;
  double *E = buf, *QE = E + en, *E2 = QE + en, *RE = E2 + en;
  const int grid = 1024;
  hipLaunchKernelGGL(cx_embed_kernel, dim3(grid), dim3(256), 0, c->stream, (const double*)A, (int64_t)2 * lda, m, n, E, (int64_t)m2);
  if (int rc = qrpos_f64(c, m2, n2, E, m2, QE, m2, RE, n2)) return rc;
  hipLaunchKernelGGL(cx_half_kernel, dim3(grid), dim3(256), 0, c->stream, QE, (int64_t)m2, m, n, (double*)Q, (int64_t)2 * ldq, E2,
                     (int64_t)m2, 0);
  double defect = 0.0;
  if (int rc = mpsk_vnrm2(c, (int64_t)en, E2, &defect)) return rc;
  if (defect <= 1.0e-13 * std::sqrt((double)n2)) {
    hipLaunchKernelGGL(cx_half_kernel, dim3(grid), dim3(256), 0, c->stream, RE, (int64_t)n2, n, n, (double*)R, (int64_t)2 * ldr,
                       (double*)nullptr, (int64_t)0, 1);
    return MPSK_OK;
  }
  // structured part -> re-orthonormalise (well conditioned: structure preserving to rounding) -> R = triu(Q^H A)
  hipLaunchKernelGGL(cx_embed_kernel, dim3(grid), dim3(256), 0, c->stream, (const double*)Q, (int64_t)2 * ldq, m, n, E2, (int64_t)m2);
  if (int rc = qrpos_f64(c, m2, n2, E2, m2, QE, m2, RE, n2)) return rc;
  hipLaunchKernelGGL(cx_half_kernel, dim3(grid), dim3(256), 0, c->stream, QE, (int64_t)m2, m, n, (double*)Q, (int64_t)2 * ldq,
                     (double*)nullptr, (int64_t)0, 0);
  hipLaunchKernelGGL(cx_embed_kernel, dim3(grid), dim3(256), 0, c->stream, (const double*)Q, (int64_t)2 * ldq, m, n, QE, (int64_t)m2);
  GemmArgs g = mk(QE, E, RE, n2, n2, m2, m2, m2, n2, 1, 0);
  HIPCHK(gemm_f64(g, c->stream));
  hipLaunchKernelGGL(cx_half_kernel, dim3(grid), dim3(256), 0, c->stream, RE, (int64_t)n2, n, n, (double*)R, (int64_t)2 * ldr,
                     (double*)nullptr, (int64_t)0, 1);
  return MPSK_OK;
}
// A (m x n complex, m <= n) = L Q: the QRpos of A^H, conjugate-transposed back
static int lqpos_c128(mpsk_ctx* c, int m, int n, const void* A, int lda, void* L, int ldl, void* Q, int ldq) {
  REQUIRE(c && A && Q && L, "NULL argument");
  REQUIRE(m <= n && m > 0, "needs 0 < m <= n");
  REQUIRE(lda >= m && ldq >= m && ldl >= m, "leading dimension too small");
  HIPCHK(hipSetDevice(c->device));
  const size_t an = (size_t)2 * n * m, ln = (size_t)2 * m * m;
  double* buf = nullptr;
  if (int rc = cx_scratch(c, 1, sizeof(double) * (2 * an + ln), &buf)) return rc;
  double *At = buf, *Qt = At + an, *Rt = Qt + an;
  hipLaunchKernelGGL(cx_ctranspose_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)A, (int64_t)2 * lda, m, n, At, (int64_t)2 * n);
  if (int rc = qrpos_c128(c, n, m, At, n, Qt, n, Rt, m)) return rc;
  hipLaunchKernelGGL(cx_ctranspose_kernel, dim3(1024), dim3(256), 0, c->stream, Qt, (int64_t)2 * n, n, m, (double*)Q, (int64_t)2 * ldq);
  hipLaunchKernelGGL(cx_ctranspose_kernel, dim3(1024), dim3(256), 0, c->stream, Rt, (int64_t)2 * m, m, m, (double*)L, (int64_t)2 * ldl);
  return MPSK_OK;
}

// completion of ONE CholeskyQR3 factorization whose launches are already on `s`: wait, read the device flag, repeat the
// third pass / run the fallbacks (robust CholeskyQR, Householder) if it asks for them.  *redone: the outputs changed
// after the enqueue (anything a caller computed from them speculatively has to be recomputed).
static int qr_complete_one(mpsk_ctx* c, int m, int n, const double* A, int lda, double* Q, int ldq, double* R, int ldr,
                           double* ws, int* d_flag, int* h_flag, hipStream_t s, bool* redone) {
  HIPCHK(hipStreamSynchronize(s));
  const int pre = *h_flag;
  hipError_t e = cholqr3_finalize(m, n, Q, ldq, R, ldr, ws, d_flag, h_flag, s);
  if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3 finalize: ") + hipGetErrorString(e));
  if (redone) *redone = (pre != 0);
  if (*h_flag == 0) { c->n_qr_chol++; return MPSK_OK; }
  int flag = 0;
  if (c->qr_shift_fast < 1.0) {            // breakdown with the rounding-level shift: the published shift next
    c->n_qr_retry++;
    e = cholqr3(m, n, A, lda, Q, ldq, R, ldr, ws, c->d_flag, &flag, c->stream, 1.0);
    if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3 (retry): ") + hipGetErrorString(e));
    if (flag == 0) { c->n_qr_chol++; return MPSK_OK; }
  }
  if (c->qr_mode == 2) return fail(MPSK_ERR_INVALID, "cholqr3: matrix too ill-conditioned / rank deficient");
  c->n_qr_fallback++;
  e = cholqr_robust(m, n, A, lda, Q, ldq, R, ldr, ws, c->d_flag, &flag, c->stream);
  if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr_robust: ") + hipGetErrorString(e));
  if (flag == 0) { c->n_qr_robust++; return MPSK_OK; }
  c->n_qr_house++;
  std::string err;
  e = qrpos(m, n, A, lda, Q, ldq, R, ldr, ws, c->stream, &err);
  if (e != hipSuccess) return fail(err.empty() ? MPSK_ERR_HIP : MPSK_ERR_INVALID, "qrpos: " + (err.empty() ? std::string(hipGetErrorString(e)) : err));
  return MPSK_OK;
}

static int qrpos2_complete(mpsk_ctx* c, int* redone) {
  mpsk_ctx::PendingQR P = c->pend;
  c->pend.active = 0;
  bool r1 = false, r2 = false;
  // (the first factorization's fallbacks use c->ws / the main stream: both free once its own stream is drained; the
  //  second one's are run after the join, on the main stream as well)
  if (int rc = qr_complete_one(c, P.m, P.n, (const double*)P.A1, P.lda1, (double*)P.Q1, P.ldq1, (double*)P.R1, P.ldr1,
                               (double*)c->ws, c->d_flag, &c->h_flags[0], c->stream, &r1)) return rc;
  HIPCHK(hipStreamSynchronize(c->stream2));
  const int pre2 = c->h_flags[1];
  hipError_t e = cholqr3_finalize(P.m, P.n, (double*)P.Q2, P.ldq2, (double*)P.R2, P.ldr2, (double*)c->ws2, c->d_flag + 8,
                                  &c->h_flags[1], c->stream2);
  if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3 finalize (2): ") + hipGetErrorString(e));
  // join: later work on the main stream is ordered after stream2 (already drained by finalize)
  HIPCHK(hipEventRecord(c->ev_join, c->stream2));
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  r2 = (pre2 != 0);
  bool ok2 = (c->h_flags[1] == 0);
  if (!ok2 && c->qr_shift_fast < 1.0) {    // (main stream, c->ws: the first factorization is complete)
    c->n_qr_retry++;
    int flag = 0;
    hipError_t e3 = cholqr3(P.m, P.n, (const double*)P.A2, P.lda2, (double*)P.Q2, P.ldq2, (double*)P.R2, P.ldr2,
                            (double*)c->ws, c->d_flag, &flag, c->stream, 1.0);
    if (e3 != hipSuccess) return fail(MPSK_ERR_HIP, "cholqr3 (pair retry) failed");
    ok2 = (flag == 0);
  }
  if (ok2) c->n_qr_chol++;
  else {
    if (c->qr_mode == 2) return fail(MPSK_ERR_INVALID, "cholqr3: matrix too ill-conditioned / rank deficient");
    c->n_qr_fallback++;
    int flag = 0;
    hipError_t e2 = cholqr_robust(P.m, P.n, (const double*)P.A2, P.lda2, (double*)P.Q2, P.ldq2, (double*)P.R2, P.ldr2,
                                  (double*)c->ws, c->d_flag, &flag, c->stream);
    if (e2 != hipSuccess) return fail(MPSK_ERR_HIP, "cholqr_robust (pair) failed");
    if (flag == 0) c->n_qr_robust++;
    else {
      c->n_qr_house++;
      std::string err;
      e2 = qrpos(P.m, P.n, (const double*)P.A2, P.lda2, (double*)P.Q2, P.ldq2, (double*)P.R2, P.ldr2, (double*)c->ws, c->stream, &err);
      if (e2 != hipSuccess) return fail(MPSK_ERR_HIP, "qrpos fallback (pair) failed");
    }
  }
  if (redone) *redone = (r1 ? 1 : 0) | (r2 ? 2 : 0);
  return MPSK_OK;
}

// Two independent QRpos factorizations of equal shape, in flight together on two streams (the
// CholeskyQR3 launch chain is latency-bound, so the two chains interleave on the GPU): the DMRG sweep
// needs leftorth of the OLD AC (galerkin projector, toolbox.jl:17-22) and of the NEW AC (next site's
// AL, orthoview.jl:56) at the same moment.
int mpsk_qrpos2(mpsk_ctx* c, int m, int n, const void* A1, int lda1, void* Q1, int ldq1, void* R1, int ldr1,
                const void* A2, int lda2, void* Q2, int ldq2, void* R2, int ldr2) {
  REQUIRE(c && A1 && Q1 && R1 && A2 && Q2 && R2, "NULL argument");
  REQUIRE(m >= n && n > 0, "needs m >= n > 0");
  REQUIRE(lda1 >= m && ldq1 >= m && ldr1 >= n && lda2 >= m && ldq2 >= m && ldr2 >= n, "leading dimension too small");
  HIPCHK(hipSetDevice(c->device));
  const bool defer = c->defer_next;
  c->defer_next = false;
  const size_t wsd = qr_ws_doubles(m, n);
  if (int rc = ensure_ws(c, sizeof(double) * wsd)) return rc;
  if (c->qr_mode == 1 || n <= 64) {
    if (int rc = qrpos_dispatch(c, m, n, (const double*)A1, lda1, (double*)Q1, ldq1, (double*)R1, ldr1, (double*)c->ws)) return rc;
    return qrpos_dispatch(c, m, n, (const double*)A2, lda2, (double*)Q2, ldq2, (double*)R2, ldr2, (double*)c->ws);
  }
  if (c->ws2_bytes < sizeof(double) * wsd) {
    HIPCHK(hipStreamSynchronize(c->stream2));
    if (c->ws2) HIPCHK(hipFree(c->ws2));
    c->ws2 = nullptr; c->ws2_bytes = 0;
    if (hipMalloc(&c->ws2, sizeof(double) * wsd) != hipSuccess) return fail(MPSK_ERR_NOMEM, "mpsk_qrpos2: workspace hipMalloc failed");
    c->ws2_bytes = sizeof(double) * wsd;
  }
  // fork: stream2 sees everything enqueued on the main stream so far (A2 / Q2 / R2 allocations and producers)
  HIPCHK(hipEventRecord(c->ev_fork, c->stream));
  HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  c->h_flags[0] = c->h_flags[1] = 0;
  hipError_t e = cholqr3_enqueue(m, n, (const double*)A1, lda1, (double*)Q1, ldq1, (double*)R1, ldr1, (double*)c->ws,
                                 c->d_flag, &c->h_flags[0], c->stream, c->qr_shift_fast);
  if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3 (1): ") + hipGetErrorString(e));
  e = cholqr3_enqueue(m, n, (const double*)A2, lda2, (double*)Q2, ldq2, (double*)R2, ldr2, (double*)c->ws2,
                      c->d_flag + 8, &c->h_flags[1], c->stream2, c->qr_shift_fast);
  if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3 (2): ") + hipGetErrorString(e));
  mpsk_ctx::PendingQR& P = c->pend;
  P.active = 1; P.m = m; P.n = n;
  P.A1 = A1; P.lda1 = lda1; P.Q1 = Q1; P.ldq1 = ldq1; P.R1 = R1; P.ldr1 = ldr1;
  P.A2 = A2; P.lda2 = lda2; P.Q2 = Q2; P.ldq2 = ldq2; P.R2 = R2; P.ldr2 = ldr2;
  if (defer) return MPSK_OK;                   // outputs are speculative until mpsk_qr_commit
  return qrpos2_complete(c, nullptr);
}

static int lqpos_complete(mpsk_ctx* c, int* redone) {
  mpsk_ctx::PendingQR P = c->pend;
  c->pend.active = 0;
  bool r1 = false;
  // the transposed problem: At (n x m) = Qt Rt; fallbacks write the same Qt / Rt
  if (int rc = qr_complete_one(c, P.n, P.m, P.At, P.n, P.Qt, P.n, P.Rt, P.m, P.ws, c->d_flag, &c->h_flags[0], c->stream, &r1)) return rc;
  HIPCHK(transpose(P.Qt, P.n, P.n, P.m, (double*)P.Qo, P.ldqo, c->stream));
  HIPCHK(transpose(P.Rt, P.m, P.m, P.m, (double*)P.L, P.ldl, c->stream));
  if (redone) *redone = 0;                     // L / Q are only written here: nothing speculative to redo
  (void)r1;
  return MPSK_OK;
}

// Side stream.  mpsk_ctx_side_mark records "now" on the ctx stream; mpsk_ctx_side_begin makes the ctx's second stream wait
// for that mark and routes every following mpsk_* call to it; mpsk_ctx_side_end routes calls back to the main stream, which
// waits for the side work.  The sweep marks, enqueues the (latency-bound, mostly idle) CholeskyQR chain of the next gauge
// step on the main stream, and runs the site's galerkin evaluation on the side stream underneath it.  Calls that use the
// second stream themselves (mpsk_qrpos2, mpsk_qrlq_pair, mpsk_tsplit) or the ctx workspace are refused in between.
int mpsk_ctx_side_mark(mpsk_ctx* c) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(!c->on_side, "already on the side stream");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventRecord(c->ev_fork, c->stream));
  return MPSK_OK;
}
int mpsk_ctx_side_begin(mpsk_ctx* c) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(!c->on_side, "already on the side stream");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  std::swap(c->stream, c->stream2);
  c->on_side = true;
  return MPSK_OK;
}
int mpsk_ctx_side_end(mpsk_ctx* c) {
  REQUIRE(c, "ctx is NULL");
  if (!c->on_side) return MPSK_OK;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventRecord(c->ev_join, c->stream));          // (the side stream)
  std::swap(c->stream, c->stream2);
  c->on_side = false;
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  return MPSK_OK;
}

// The NEXT mpsk_qrpos2 / mpsk_lqpos call on this ctx that takes the CholeskyQR3 path returns as soon as its launches
// are enqueued; its outputs are speculative (qrpos2) / not yet written (lqpos) until mpsk_qr_commit.  Between the two
// calls only entry points that do not use the ctx workspace are accepted (mpsk_gemm, the mpsk_v* family): the sweep
// enqueues the galerkin evaluation there, so the GPU is not idle while the host waits for the success flag.
int mpsk_ctx_qr_defer(mpsk_ctx* c) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(!c->pend.active, "a deferred factorization is already pending");
  c->defer_next = true;
  return MPSK_OK;
}

// *redone: bit 0 / bit 1 = the first / second factorization of a deferred mpsk_qrpos2 was corrected after its enqueue
// (third pass repeated or a fallback ran): results computed from the speculative Q / R must be recomputed.
int mpsk_qr_commit(mpsk_ctx* c, int* redone) {
  REQUIRE(c, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  c->defer_next = false;
  if (redone) *redone = 0;
  if (c->pend.active == 1) return qrpos2_complete(c, redone);
  if (c->pend.active == 2) return lqpos_complete(c, redone);
  return MPSK_OK;
}

// QRpos of A1 (m x n) and LQpos of A2 (n x m) in flight together: the left-moving DMRG site update needs
// leftorth of the OLD AC (galerkin projector) and rightorth of the NEW AC (next site's AR / C) at the same moment;
// for Dl == Dr the transposed tail of the new AC has the shape of the old AC's front matrix.
int mpsk_qrlq_pair(mpsk_ctx* c, int m, int n, const void* A1, int lda1, void* Q1, int ldq1, void* R1, int ldr1,
                   const void* A2, int lda2, void* L2, int ldl2, void* Q2, int ldq2) {
  REQUIRE(c && A1 && Q1 && R1 && A2 && L2 && Q2, "NULL argument");
  REQUIRE(m >= n && n > 0, "needs m >= n > 0");
  REQUIRE(lda1 >= m && ldq1 >= m && ldr1 >= n && lda2 >= n && ldl2 >= n && ldq2 >= n, "leading dimension too small");
  REQUIRE(!c->pend.active, "a deferred factorization is pending (mpsk_qr_commit first)");
  c->defer_next = false;                       // deferral is defined for mpsk_qrpos2 / mpsk_lqpos only: this call completes at once
  HIPCHK(hipSetDevice(c->device));
  const size_t need = sizeof(double) * ((size_t)2 * m * n + (size_t)n * n + 8);
  if (c->ws3_bytes < need) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->stream2));
    if (c->ws3) HIPCHK(hipFree(c->ws3));
    c->ws3 = nullptr; c->ws3_bytes = 0;
    if (hipMalloc(&c->ws3, need) != hipSuccess) return fail(MPSK_ERR_NOMEM, "mpsk_qrlq_pair: workspace hipMalloc failed");
    c->ws3_bytes = need;
  }
  double* At = (double*)c->ws3;                                 // m x n  (= A2^T)
  double* Qt = At + (((size_t)m * n + 1) & ~(size_t)1);         // m x n
  double* Rt = Qt + (((size_t)m * n + 1) & ~(size_t)1);         // n x n
  HIPCHK(transpose((const double*)A2, lda2, n, m, At, m, c->stream));
  if (int rc = mpsk_qrpos2(c, m, n, A1, lda1, Q1, ldq1, R1, ldr1, At, m, Qt, m, Rt, n)) return rc;
  HIPCHK(transpose(Qt, m, m, n, (double*)Q2, ldq2, c->stream));
  HIPCHK(transpose(Rt, n, n, n, (double*)L2, ldl2, c->stream));
  return MPSK_OK;
}

// LQ of A (m x n, m <= n) through the QR of A^T:  A^T = Qt Rt  ->  L = Rt^T, Q = Qt^T
int mpsk_lqpos(mpsk_ctx* c, int m, int n, const void* A, int lda, void* L, int ldl, void* Q, int ldq) {
  REQUIRE(c, "ctx is NULL");
  return c->dtype == MPSK_C128 ? lqpos_c128(c, m, n, A, lda, L, ldl, Q, ldq) : lqpos_f64(c, m, n, A, lda, L, ldl, Q, ldq);
}
static int lqpos_f64(mpsk_ctx* c, int m, int n, const void* A, int lda, void* L, int ldl, void* Q, int ldq) {
  REQUIRE(c && A && Q && L, "NULL argument");
  REQUIRE(m <= n && m > 0, "needs 0 < m <= n");
  REQUIRE(lda >= m && ldq >= m && ldl >= m, "leading dimension too small");
  HIPCHK(hipSetDevice(c->device));
  const bool defer = c->defer_next;
  c->defer_next = false;
  const size_t qws = qr_ws_doubles(n, m);
  const size_t extra = (size_t)2 * n * m + (size_t)m * m;
  if (int rc = ensure_ws(c, sizeof(double) * (qws + extra))) return rc;
  double* At = (double*)c->ws;            // n x m
  double* Qt = At + (size_t)n * m;        // n x m
  double* Rt = Qt + (size_t)n * m;        // m x m
  double* ws2 = Rt + (size_t)m * m;
  HIPCHK(transpose((const double*)A, lda, m, n, At, n, c->stream));
  if (defer && c->qr_mode != 1 && m > 64) {
    c->h_flags[0] = 0;
    hipError_t e = cholqr3_enqueue(n, m, At, n, Qt, n, Rt, m, ws2, c->d_flag, &c->h_flags[0], c->stream, c->qr_shift_fast);
    if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("cholqr3: ") + hipGetErrorString(e));
    mpsk_ctx::PendingQR& P = c->pend;
    P.active = 2; P.m = m; P.n = n; P.At = At; P.Qt = Qt; P.Rt = Rt; P.ws = ws2;
    P.L = L; P.ldl = ldl; P.Qo = Q; P.ldqo = ldq;
    return MPSK_OK;
  }
  if (int rc = qrpos_dispatch(c, n, m, At, n, Qt, n, Rt, m, ws2)) return rc;
  HIPCHK(transpose(Qt, n, n, m, (double*)Q, ldq, c->stream));
  HIPCHK(transpose(Rt, m, m, m, (double*)L, ldl, c->stream));
  return MPSK_OK;
}
int mpsk_tsvd(mpsk_ctx* c, int m, int n, const void* theta, int ldt, void* U, int ldu, void* S, void* Vh,
              int ldv, int max_keep, double trunc_err, int* kept, double* disc_norm) {
  REQUIRE(c && theta && U && S && Vh && kept && disc_norm, "NULL argument");
  REQUIRE(m > 0 && n > 0, "dimensions must be positive");
  const int kmax = m < n ? m : n;
  REQUIRE(ldt >= m && ldu >= m && ldv >= kmax, "leading dimension too small");
  REQUIRE(trunc_err >= 0.0, "trunc_err must be >= 0");
  REQUIRE(!c->pend.active, "a deferred factorization is pending (mpsk_qr_commit first)");
  c->defer_next = false;
  HIPCHK(hipSetDevice(c->device));
  std::string err;
  const int mm = m < n ? n : m, nn = m < n ? m : n, transposed = m < n ? 1 : 0;
  if (c->svd_precondition && nn > 64) {
    // QR-preconditioned one-sided Jacobi: A' = theta or theta^T (tall) = Qb Rb, Jacobi on Rb^T
    const auto ev = [](size_t v) { return (v + 1) & ~(size_t)1; };   // keep every sub-buffer 16-byte aligned
    const bool dbl = c->svd_precondition >= 2;
    const size_t a_d = transposed ? ev((size_t)mm * nn) : 0, q_d = ev((size_t)mm * nn), r_d = ev((size_t)nn * nn);
    const size_t x_d = dbl ? 3 * r_d : 0;                              // Q1 / R^T, U'', Vh'' of the double preconditioning
    // (the workspace of a factorization is NOT monotone in its shape: the in-step solve / Gram of the small regime adds
    //  CQ_GS npad^2 doubles, so the square second QR can need more than the tall first one)
    const size_t qws = sizeof(double) * std::max(qr_ws_doubles(mm, nn), qr_ws_doubles(nn, nn)), sws = tsvd_workspace_bytes(nn, nn);
    if (int rc = ensure_ws(c, sizeof(double) * (a_d + q_d + r_d + x_d) + (qws > sws ? qws : sws) + 256)) return rc;
    double* At = (double*)c->ws;
    double* Qb = At + a_d;
    double* Rb = Qb + q_d;
    double* X0 = Rb + r_d;
    double* rest = X0 + x_d;
    const double* Ap = (const double*)theta;
    int lda = ldt;
    if (transposed) {
      HIPCHK(transpose((const double*)theta, ldt, m, n, At, mm, c->stream));
      Ap = At; lda = mm;
    }
    if (int rc = qrpos_dispatch(c, mm, nn, Ap, lda, Qb, mm, Rb, nn, rest)) return rc;
    if (dbl) {
      // Double preconditioning (Drmac-Veselic "QR of R^T", as mpsk_tsplit): R^T = Q1 R1, Jacobi on the columns of R1^T with
      // the rotations accumulated:  R^T = U2 S V2^T  (U2 = Q1 W, V2 = G S^-1)  =>  A' = Qb R = (Qb V2) S U2^T.
      // 3-4 sweeps fewer on graded spectra (each with the V update: 36 ms at 4096^2) for one n x n QRpos and one GEMM.
      double* Q1 = X0;                       // first R^T, then Q1
      double* U2 = X0 + r_d;                 // nn x kmax
      double* V2h = U2 + r_d;                // kmax x nn
      HIPCHK(transpose(Rb, nn, nn, nn, U2, nn, c->stream));
      if (int rc = qrpos_dispatch(c, nn, nn, U2, nn, Q1, nn, Rb, nn, rest)) return rc;      // Rb <- R1
      hipError_t e = tsvd(nn, nn, Rb, nn, U2, nn, (double*)S, V2h, nn, max_keep, trunc_err, kept, disc_norm, rest, c->stream,
                          &err, &c->last_svd_sweeps, Q1, nn, nn, 0, c->xstreams, 3);
      if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("mpsk_tsvd: ") + (err.empty() ? hipGetErrorString(e) : err.c_str()));
      const int k = *kept;
      if (!transposed) {                     // theta = A':  U = Qb V2,  Vh = U2^T
        GemmArgs g = mk(Qb, V2h, (double*)U, mm, k, nn, mm, nn, ldu, 0, 1);
        HIPCHK(gemm_f64(g, c->stream));
        HIPCHK(transpose(U2, nn, nn, k, (double*)Vh, ldv, c->stream));
      } else {                               // theta^T = A':  theta = U2 S (Qb V2)^T :  U = U2,  Vh = V2^T Qb^T
        HIPCHK(hipMemcpy2DAsync(U, sizeof(double) * ldu, U2, sizeof(double) * nn, sizeof(double) * nn, k,
                                hipMemcpyDeviceToDevice, c->stream));
        GemmArgs g = mk(V2h, Qb, (double*)Vh, k, mm, nn, nn, mm, ldv, 0, 1);
        HIPCHK(gemm_f64(g, c->stream));
      }
      HIPCHK(hipStreamSynchronize(c->stream));
      return MPSK_OK;
    }
    hipError_t e = tsvd(nn, nn, Rb, nn, (double*)U, ldu, (double*)S, (double*)Vh, ldv, max_keep, trunc_err, kept,
                        disc_norm, rest, c->stream, &err, &c->last_svd_sweeps, Qb, mm, mm, transposed, c->xstreams, 3);
    if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("mpsk_tsvd: ") + (err.empty() ? hipGetErrorString(e) : err.c_str()));
    return MPSK_OK;
  }
  if (int rc = ensure_ws(c, tsvd_workspace_bytes(m, n))) return rc;
  hipError_t e = tsvd(m, n, (const double*)theta, ldt, (double*)U, ldu, (double*)S, (double*)Vh, ldv, max_keep,
                      trunc_err, kept, disc_norm, c->ws, c->stream, &err, &c->last_svd_sweeps, nullptr, 0, 0, 0, c->xstreams, 3);
  if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("mpsk_tsvd: ") + (err.empty() ? hipGetErrorString(e) : err.c_str()));
  return MPSK_OK;
}

// Truncated two-site split  theta ~ AL . C . AR  (tsvd!(theta; trunc) followed by al, c, ar of dmrg.jl:96-104 /
// tdvp.jl:124-126) WITHOUT accumulating the Jacobi rotations: QRpos of the tall orientation, V-free block Jacobi on
// R^T -> singular values and the singular vectors of ONE side (exact, orthonormal); the other factor is rebuilt from
// theta itself and re-orthonormalised by QRpos / LQpos, so AL, AR are isometries to rounding and
// AL C AR = theta projected on the kept singular subspace; C is triangular instead of diag(S) (S is returned too).
// ---- truncation-aware stage of mpsk_tsplit (svd mode 3) ---------------------------------------------------------------
// When only k = max_keep << n singular triplets are kept, the block-Jacobi iteration on all n columns (127 rounds a sweep at
// n = 4096, ~10 sweeps) is replaced by
//   1. a randomized subspace iteration for the dominant right subspace of the tall orientation A' (mm x nn):
//        Y <- orth(A'^T orth(A' Y)),   Y: nn x r,  r = k + max(64, k / 2) rounded up to 64
//      -- two GEMMs on the MFMA core and two ONE-pass shifted CholeskyQR re-conditionings per iteration (only the span
//      matters; the error of the i-th direction shrinks by (sigma_{r+1} / sigma_i)^2 per iteration);
//   2. W = QRpos(Y) (working accuracy), B' = A' W (mm x r), and the usual preconditioned V-free Jacobi split on B': r
//      columns instead of nn (r = 1536 of 4096: 47 rounds a sweep instead of 127, 24 pairs a round instead of 64);
//   3. a CHECK, not an estimate: with U_k the kept left vectors and M = U_k^T A' (formed anyway for the LQpos / QRpos that
//      delivers the other factor),  rho = || M (I - W W^T) ||_F / ||theta||_F  is the part of the kept triplets that lies
//      outside the iterated subspace.  rho <= MPSK_SPLIT_TOL (1e-12): done.  Otherwise the iteration continues from the r
//      left Ritz vectors for the number of iterations the measured Ritz ratio predicts, or -- if that would cost more than
//      the full iteration -- the call falls through to mode 2.  Nothing is returned that has not passed the check.
// M is computed with the full theta, so C . AR = U_k^T theta exactly as in mode 2; S holds the r leading values (Ritz values
// of a converged subspace: errors O(rho^2)), the remaining entries are NaN.
__global__ __launch_bounds__(256) void split_fill_kernel(double* __restrict__ y, int64_t n, unsigned seed) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)(e & 0xffffffffu) * 0x9E3779B1u ^ ((unsigned)(e >> 32) + seed) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    y[e] = ((double)h + 0.5) * (2.0 / 4294967296.0) - 1.0;       // uniform in (-1, 1)
  }
}
__global__ __launch_bounds__(256) void split_fill_nan_kernel(double* __restrict__ y, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) y[e] = __longlong_as_double(0x7ff8000000000000LL);
}

// X (rows x cols, ld) = alpha * A (lda)   /   leading rows x cols of the identity
__global__ __launch_bounds__(256) void split_scale_copy_kernel(const double* __restrict__ A, int64_t lda, int rows, int cols,
                                                               double alpha, double* __restrict__ X, int64_t ldx) {
  const int64_t tot = (int64_t)rows * cols;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(t % rows), c = (int)(t / rows);
    X[r + ldx * c] = A ? alpha * A[r + lda * c] : (r == c ? 1.0 : 0.0);
  }
}

// *out = max |A[r, c]| as the bit pattern of a non-negative double (order preserving under unsigned compare); NaN -> +inf bits
__global__ __launch_bounds__(256) void split_absmax_kernel(const double* __restrict__ A, int64_t lda, int rows, int cols,
                                                           unsigned long long* __restrict__ out) {
  const int64_t tot = (int64_t)rows * cols;
  double mx = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x) {
    const double v = fabs(A[(t % rows) + lda * (t / rows)]);
    mx = (v > mx || v != v) ? (v != v ? INFINITY : v) : mx;
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
}

struct SplitSub {             // state of the subspace stage across refinements
  int r = 0;
  double *Yb = nullptr, *Zb = nullptr, *Wb = nullptr, *Bp = nullptr;   // nn x r, mm x r, nn x r, mm x r
};

// one re-conditioning pass X -> Q (span preserved): rounding-level shift first, the published shift if a pivot broke down
static int split_orth1(mpsk_ctx* c, int m, int r, const double* X, double* Q) {
  HIPCHK(cholqr1_orth(m, r, X, m, Q, m, (double*)c->ws, c->d_flag, c->stream, c->qr_shift_fast));
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (flag) HIPCHK(cholqr1_orth(m, r, X, m, Q, m, (double*)c->ws, c->d_flag, c->stream, 1.0));
  return MPSK_OK;
}
// q iterations  Y <- orth1(A'^T orth1(A' Y))  starting from Yb (nn x r); result in Yb.  Zb / Wb / Bp are scratch.
static int split_iterate(mpsk_ctx* c, int mm, int nn, const double* Ap, int lda, const SplitSub& sb, int q) {
  const int r = sb.r;
  const bool trace = getenv("MPSK_SPLIT_TRACE") != nullptr;
  auto stage = [&](const char* name, int i) {
    if (!trace) return;
    hipError_t e_ = hipStreamSynchronize(c->stream);
    fprintf(stderr, "[mpsk_tsplit]   iteration %d %-12s %s\n", i, name, hipGetErrorString(e_)); fflush(stderr);
  };
  for (int i = 0; i < q; ++i) {
    GemmArgs g = mk(Ap, sb.Yb, sb.Zb, mm, r, nn, lda, nn, mm);
    HIPCHK(gemm_f64(g, c->stream));
    stage("Z = A' Y", i);
    if (int rc = split_orth1(c, mm, r, sb.Zb, sb.Bp)) return rc;
    stage("orth1(Z)", i);
    GemmArgs g2 = mk(Ap, sb.Bp, sb.Wb, nn, r, mm, lda, mm, nn, 1, 0);
    HIPCHK(gemm_f64(g2, c->stream));
    stage("Y = A'^T Z", i);
    if (int rc = split_orth1(c, nn, r, sb.Wb, sb.Yb)) return rc;
    stage("orth1(Y)", i);
  }
  return MPSK_OK;
}

static int tsplit_core(mpsk_ctx* c, int m, int n, const void* theta, int ldt, int max_keep, double trunc_err,
                       void* AL, int ldal, void* Cm, int ldc, void* AR, int ldar, void* S, int* kept, double* disc_norm);

// Front end of the fp64 split: every factorization inside squares the scale of theta (Gram matrices), so tensors whose norm
// is outside [1e-100, 1e100] are split as theta / |theta|_F and the values scaled back (the split is homogeneous of degree
// one: AL, AR unchanged, C, S, the discarded norm and trunc_err scale); theta = 0 gets the trivial answer (C = 0, any
// isometries) instead of a factorization of nothing.
static int tsplit_f64(mpsk_ctx* c, int m, int n, const void* theta, int ldt, int max_keep, double trunc_err,
                      void* AL, int ldal, void* Cm, int ldc, void* AR, int ldar, void* S, int* kept, double* disc_norm) {
  REQUIRE(c && theta && AL && Cm && AR && S && kept && disc_norm, "NULL argument");
  REQUIRE(m > 0 && n > 0 && ldt >= m, "dimensions must be positive");
  HIPCHK(hipSetDevice(c->device));
  // scale = max |theta_ij| (a norm would square the entries: 1e-200 underflows there already)
  double nrm = 1.0;
  {
    unsigned long long* d_mx = (unsigned long long*)c->d_partial;
    unsigned long long h_mx = 0;
    HIPCHK(hipMemsetAsync(d_mx, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(split_absmax_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)theta, (int64_t)ldt, m, n, d_mx);
    HIPCHK(hipMemcpyAsync(&h_mx, d_mx, sizeof(h_mx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::memcpy(&nrm, &h_mx, sizeof(double));
  }
  const int kfull = m < n ? m : n;
  if (nrm == 0.0) {
    const int k = (max_keep > 0 && max_keep < kfull) ? max_keep : kfull;
    REQUIRE(ldal >= m && ldc >= k && ldar >= k, "leading dimension too small");
    hipLaunchKernelGGL(split_scale_copy_kernel, dim3(256), dim3(256), 0, c->stream, (const double*)nullptr, (int64_t)0, m, k, 0.0, (double*)AL, (int64_t)ldal);
    hipLaunchKernelGGL(split_scale_copy_kernel, dim3(256), dim3(256), 0, c->stream, (const double*)nullptr, (int64_t)0, k, n, 0.0, (double*)AR, (int64_t)ldar);
    HIPCHK(hipMemset2DAsync(Cm, sizeof(double) * ldc, 0, sizeof(double) * k, k, c->stream));
    HIPCHK(hipMemsetAsync(S, 0, sizeof(double) * kfull, c->stream));
    *kept = k; *disc_norm = 0.0;
    c->last_split_path = 0; c->last_split_iters = 0; c->last_split_resid = 0.0; c->last_svd_sweeps = 0;
    return MPSK_OK;
  }
  if (std::isfinite(nrm) && (nrm < 1.0e-100 || nrm > 1.0e100)) {
    double* Ts = nullptr;
    if (int rc = cx_scratch(c, 3, sizeof(double) * (size_t)m * n, &Ts)) return rc;
    hipLaunchKernelGGL(split_scale_copy_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)theta, (int64_t)ldt, m, n, 1.0 / nrm, Ts, (int64_t)m);
    if (int rc = tsplit_core(c, m, n, Ts, m, max_keep, trunc_err / nrm, AL, ldal, Cm, ldc, AR, ldar, S, kept, disc_norm)) return rc;
    const int k = *kept;
    hipLaunchKernelGGL(split_scale_copy_kernel, dim3(256), dim3(256), 0, c->stream, (const double*)Cm, (int64_t)ldc, k, k, nrm, (double*)Cm, (int64_t)ldc);
    HIPCHK(vec_scal(nrm, (double*)S, kfull, c->stream));
    *disc_norm *= nrm;
    return MPSK_OK;
  }
  return tsplit_core(c, m, n, theta, ldt, max_keep, trunc_err, AL, ldal, Cm, ldc, AR, ldar, S, kept, disc_norm);
}

static int tsplit_core(mpsk_ctx* c, int m, int n, const void* theta, int ldt, int max_keep, double trunc_err,
                       void* AL, int ldal, void* Cm, int ldc, void* AR, int ldar, void* S, int* kept, double* disc_norm) {
  REQUIRE(c && theta && AL && Cm && AR && S && kept && disc_norm, "NULL argument");
  REQUIRE(m > 0 && n > 0, "dimensions must be positive");
  const int mm = m < n ? n : m, nn = m < n ? m : n, transposed = m < n ? 1 : 0;
  REQUIRE(nn > 64, "mpsk_tsplit needs min(m, n) > 64 (use mpsk_tsvd for small tensors)");
  REQUIRE(ldt >= m && ldal >= m && ldc >= 1 && ldar >= 1, "leading dimension too small");
  REQUIRE(trunc_err >= 0.0, "trunc_err must be >= 0");
  REQUIRE(!c->pend.active, "a deferred factorization is pending (mpsk_qr_commit first)");
  c->defer_next = false;                       // (the inner mpsk_qrpos / mpsk_lqpos calls complete at once)
  HIPCHK(hipSetDevice(c->device));
  const auto ev = [](size_t v) { return (v + 1) & ~(size_t)1; };
  // truncation-aware stage: worth it while r columns are well below nn (the Jacobi part scales ~ r^2, the iteration ~ q r)
  static const double sub_over = getenv("MPSK_SPLIT_OVERSAMPLE") ? atof(getenv("MPSK_SPLIT_OVERSAMPLE")) : 0.5;
  static const double sub_tol = getenv("MPSK_SPLIT_TOL") ? atof(getenv("MPSK_SPLIT_TOL")) : 1.0e-12;
  int r_sub = 0;
  if (c->split_skip > 0) --c->split_skip;          // (backing off after the stage gave up: see below)
  else if (c->svd_precondition == 3 && max_keep > 0 && ldt == m) {
    // r = k + max(64, k / 2) while that stays within 5/8 of the columns (d >= 3 chains: k = n / d), else k + max(64, k / 4)
    // (d = 2: k = n / 2, r = 5n / 8: 2048^2 -> 1024 50 ms against 68, 4096^2 -> 2048 180 against 255; more iterations but
    // the Jacobi stage, ~ r^2, is what counts), else no stage
    static const double sub_frac = getenv("MPSK_SPLIT_MAXFRAC") ? atof(getenv("MPSK_SPLIT_MAXFRAC")) : 0.625;
    for (double ov : {sub_over, 0.5 * sub_over}) {
      int over = (int)std::ceil(ov * max_keep);
      if (over < 64) over = 64;
      r_sub = (max_keep + over + 63) / 64 * 64;
      if (r_sub <= (int)(nn * sub_frac) && r_sub > 64) break;
      r_sub = 0;
    }
  }
  const size_t a_d = transposed ? ev((size_t)mm * nn) : 0, q_d = ev((size_t)mm * nn), r_d = ev((size_t)nn * nn);
  const size_t t_d = ev((size_t)mm * nn);
  const size_t s_d = r_sub ? 2 * ev((size_t)nn * r_sub) + 2 * ev((size_t)mm * r_sub) : 0;
  const size_t need = sizeof(double) * (a_d + q_d + 2 * r_d + t_d + s_d + 8);
  if (c->ws3_bytes < need) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->stream2));
    if (c->ws3) HIPCHK(hipFree(c->ws3));
    c->ws3 = nullptr; c->ws3_bytes = 0;
    if (hipMalloc(&c->ws3, need) != hipSuccess) return fail(MPSK_ERR_NOMEM, "mpsk_tsplit: workspace hipMalloc failed");
    c->ws3_bytes = need;
  }
  double* At = (double*)c->ws3;            // theta^T when m < n
  double* Qb = At + a_d;                   // Q of the preconditioning QR of the tall orientation A' (mm x nn)
  double* Rb = Qb + q_d;                   // R (nn x nn); R1 of the second pass
  double* Y = Rb + r_d;                    // sorted, normalised singular vectors from the Jacobi iteration (nn x nn)
  double* T = Y + r_d;                     // scratch: R^T, then theta V_k / U_k^T theta
  SplitSub sb;
  sb.r = r_sub;
  sb.Yb = T + t_d; sb.Zb = sb.Yb + ev((size_t)nn * r_sub); sb.Wb = sb.Zb + ev((size_t)mm * r_sub);
  sb.Bp = sb.Wb + ev((size_t)nn * r_sub);
  const double* Ap = (const double*)theta;
  int lda = ldt;
  if (transposed) {
    HIPCHK(transpose((const double*)theta, ldt, m, n, At, mm, c->stream));
    Ap = At; lda = mm;
  }
  // every factorization below runs in c->ws.  Its size is NOT monotone in the shape (the in-step solve / Gram of the small
  // regime adds CQ_GS npad^2 doubles: a 1280 x 1280 QRpos needs 3x the workspace of a 2048 x 2048 one), so take the maximum
  // over all shapes this call can factor: A' (mm x nn), R^T (nn x nn) and, with the subspace stage, Y (nn x r), A'Y / B'
  // (mm x r), R_B^T (r x r)
  size_t qwd = std::max(qr_ws_doubles(mm, nn), qr_ws_doubles(nn, nn));
  if (r_sub) qwd = std::max(std::max(qwd, qr_ws_doubles(nn, r_sub)), std::max(qr_ws_doubles(mm, r_sub), qr_ws_doubles(r_sub, r_sub)));
  const size_t qws = sizeof(double) * qwd, sws = tsvd_workspace_bytes(nn, nn);
  if (int rc = ensure_ws(c, (qws > sws ? qws : sws) + 256)) return rc;
  const bool dbl = c->svd_precondition >= 2;
  c->last_split_iters = 0; c->last_split_resid = 0.0; c->last_split_path = 0;

  // attempt 0: the truncation-aware stage (when configured); attempt 1 (or the only one): the full iteration
  // test hook (tests/test_gpu_ops.py): one iteration and the residual check waved through, so that the dominance probe is
  // the only line of defence -- it must then send the call to the full iteration
  const bool dbg_skip_check = getenv("MPSK_SPLIT_DEBUG_SKIP_CHECK") != nullptr;
  const bool dbg_trace = getenv("MPSK_SPLIT_TRACE") != nullptr;      // sync + one stderr line per stage (fault localisation)
#define SPLIT_STAGE(name) do { if (dbg_trace) { hipError_t e_ = hipStreamSynchronize(c->stream); \
    fprintf(stderr, "[mpsk_tsplit] stage %-28s %s\n", name, hipGetErrorString(e_)); fflush(stderr); } } while (0)
  double theta_nrm = 0.0, rho_prev = 0.0;
  int q_total = 0, q_prev = 0, n_checks = 0;
  bool sub_ready = false;                  // sb.Yb holds a basis to continue from
  for (int attempt = 0; attempt < 8; ++attempt) {
    const bool sub = r_sub > 0 && c->last_split_path != 2;
    const int nc = sub ? r_sub : nn;       // columns of the matrix the Jacobi split runs on
    const double* Bsrc = Ap;
    int ldb = lda;
    double te = trunc_err;
    if (sub) {
      if (!sub_ready) {
        if (int rc = mpsk_vnrm2(c, (int64_t)m * n, theta, &theta_nrm)) return rc;
        hipLaunchKernelGGL(split_fill_kernel, dim3(1024), dim3(256), 0, c->stream, sb.Yb, (int64_t)nn * r_sub, 0x5eedu);
        const auto hit = c->split_q_hint.find(std::make_pair(nn, r_sub));
        int q0 = hit == c->split_q_hint.end() ? 8 : hit->second;
        if (q0 < 2) q0 = 2;
        if (dbg_skip_check) q0 = 1;
        if (int rc = split_iterate(c, mm, nn, Ap, lda, sb, q0)) return rc;
        SPLIT_STAGE("iterate");
        q_total += q0;
        sub_ready = true;
      }
      // W = QRpos(Y) to working accuracy (R is not needed: Rb is scratch here), B' = A' W
      if (int rc = qrpos_dispatch(c, nn, r_sub, sb.Yb, nn, sb.Wb, nn, Rb, r_sub, (double*)c->ws)) return rc;
      SPLIT_STAGE("QRpos(Y)");
      GemmArgs gb = mk(Ap, sb.Wb, sb.Bp, mm, r_sub, nn, lda, nn, mm);
      HIPCHK(gemm_f64(gb, c->stream));
      Bsrc = sb.Bp; ldb = mm;
      SPLIT_STAGE("B' = A' W");
      if (trunc_err > 0.0) {               // weight outside the subspace counts as discarded in the truncerr rule
        double bn = 0.0;
        if (int rc = mpsk_vnrm2(c, (int64_t)mm * r_sub, sb.Bp, &bn)) return rc;
        const double out2 = std::max(theta_nrm * theta_nrm - bn * bn, 0.0);
        te = std::sqrt(std::max(trunc_err * trunc_err - out2, 0.0));
        if (te == 0.0) te = 1.0e-300;
      }
    }
    if (int rc = qrpos_dispatch(c, mm, nc, Bsrc, ldb, Qb, mm, Rb, nc, (double*)c->ws)) return rc;
    SPLIT_STAGE("QRpos(B')");
    // Double preconditioning (svd mode 2, Drmac-Veselic: "QR of R^T"): R^T = Q1 R1, Jacobi on the columns of R1^T.
    // A' = Qb R = Qb R1^T Q1^T, and R1^T W = G = Y Sigma at convergence, so  A' = (Qb Y) Sigma (Q1 W)^T : the V-free
    // iteration now yields the LEFT singular vectors Qb Y of the tall orientation (orthonormal to rounding as a product of
    // orthonormal factors) and costs fewer sweeps (9 -> 6 at n = 512, 10 -> 7 at n = 1024 in the block-Jacobi model with
    // exact inner solves; measured sweep counts: profiles/r02_svd_modes.log) for one extra n x n QRpos.
    if (dbl) {
      HIPCHK(transpose(Rb, nc, nc, nc, T, nc, c->stream));
      if (int rc = qrpos_dispatch(c, nc, nc, T, nc, Y, nc, Rb, nc, (double*)c->ws)) return rc;     // Y = Q1 (not needed), Rb = R1
      SPLIT_STAGE("QRpos(R^T)");
    }
    std::string err;
    hipError_t e = tsvd(nc, nc, Rb, nc, Y, nc, (double*)S, nullptr, 1, max_keep, te, kept, disc_norm, c->ws, c->stream,
                        &err, &c->last_svd_sweeps, nullptr, 0, 0, 0, c->xstreams, 3, /*vfree=*/1);
    if (e == hipErrorNotReady && sub) {      // the Jacobi stage on B' did not converge: nothing is lost, take the full iteration
      c->last_split_path = 2;
      continue;
    }
    if (e != hipSuccess) return fail(MPSK_ERR_HIP, std::string("mpsk_tsplit: ") + (err.empty() ? hipGetErrorString(e) : err.c_str()));
    const int k = *kept;
    REQUIRE(ldc >= k && ldar >= k, "leading dimension of C / AR smaller than the kept rank");
    SPLIT_STAGE("jacobi");
    // the factor that carries theta: M' = U_k'^T A' (k x nn) as T (k x n) when A' = theta, as T^T (m x k) when A' = theta^T
    double* Vk = At;                       // (transposed) theta^T is no longer needed once the iteration is over: see below
    if (!dbl) {
      if (!transposed) {
        // theta (m x n): Y_k = right singular vectors.  B = theta Y_k = U_k S_k ; AL C = QRpos(B) ; AR = Y_k^T
        GemmArgs g = mk((const double*)theta, Y, T, m, k, n, ldt, nn, m);
        HIPCHK(gemm_f64(g, c->stream));
        if (int rc = qrpos_f64(c, m, k, T, m, AL, ldal, Cm, ldc)) return rc;
        HIPCHK(transpose(Y, nn, n, k, (double*)AR, ldar, c->stream));
      } else {
        // theta (m x n), m < n: Y_k = left singular vectors = AL.  M = Y_k^T theta = S_k V_k^T ; C AR = LQpos(M)
        HIPCHK(hipMemcpy2DAsync(AL, sizeof(double) * ldal, Y, sizeof(double) * nn, sizeof(double) * m, k,
                                hipMemcpyDeviceToDevice, c->stream));
        GemmArgs g = mk(Y, (const double*)theta, T, k, n, m, nn, ldt, k, 1, 0);
        HIPCHK(gemm_f64(g, c->stream));
        if (int rc = lqpos_f64(c, k, n, T, k, Cm, ldc, AR, ldar)) return rc;
      }
      return MPSK_OK;
    }
    if (!transposed) {
      // A' = theta: AL = Qb Y_k (left singular vectors).  M = AL^T theta = S_k V_k^T ; C AR = LQpos(M)
      GemmArgs g = mk(Qb, Y, (double*)AL, m, k, nc, mm, nc, ldal);
      HIPCHK(gemm_f64(g, c->stream));
      GemmArgs g2 = mk((const double*)AL, (const double*)theta, T, k, n, m, ldal, ldt, k, 1, 0);
      HIPCHK(gemm_f64(g2, c->stream));
    } else {
      // A' = theta^T (n x m): V_k = Qb Y_k = right singular vectors of theta.  B = theta V_k = U_k S_k ; AL C = QRpos(B) ; AR = V_k^T
      if (sub) Vk = sb.Zb;                 // At (theta^T) is still needed if the check sends us back into the iteration
      GemmArgs g = mk(Qb, Y, Vk, n, k, nc, mm, nc, n);
      HIPCHK(gemm_f64(g, c->stream));
      GemmArgs g2 = mk((const double*)theta, Vk, T, m, k, n, ldt, n, m);
      HIPCHK(gemm_f64(g2, c->stream));
    }
    if (sub) {
      // the check: rho = || M' (I - W W^T) ||_F / ||theta||_F  (M' = T, k x nn, or T^T = B, nn x k)
      double* P1 = sb.Yb;                  // r x k (or k x r) scratch: Yb is rebuilt below if the iteration continues
      double* Rs = transposed ? sb.Bp : sb.Zb;     // residual, k x nn / nn x k  (Bp / Zb are free now; Vk may live in Zb)
      double rho = 0.0;
      if (!transposed) {
        GemmArgs p = mk(T, sb.Wb, P1, k, r_sub, nn, k, nn, k);                    // P1 = M' W            (k x r)
        HIPCHK(gemm_f64(p, c->stream));
        HIPCHK(hipMemcpyAsync(Rs, T, sizeof(double) * (size_t)k * nn, hipMemcpyDeviceToDevice, c->stream));
        GemmArgs p2 = mk(P1, sb.Wb, Rs, k, nn, r_sub, k, nn, k, 0, 1);            // Rs = M' - P1 W^T
        p2.alpha = -1.0; p2.beta = 1.0;
        HIPCHK(gemm_f64(p2, c->stream));
      } else {
        GemmArgs p = mk(sb.Wb, T, P1, r_sub, k, nn, nn, nn, r_sub, 1, 0);         // P1 = W^T B           (r x k)
        HIPCHK(gemm_f64(p, c->stream));
        HIPCHK(hipMemcpyAsync(Rs, T, sizeof(double) * (size_t)k * nn, hipMemcpyDeviceToDevice, c->stream));
        GemmArgs p2 = mk(sb.Wb, P1, Rs, nn, k, r_sub, nn, r_sub, nn);             // Rs = B - W P1
        p2.alpha = -1.0; p2.beta = 1.0;
        HIPCHK(gemm_f64(p2, c->stream));
      }
      if (int rc = mpsk_vnrm2(c, (int64_t)k * nn, Rs, &rho)) return rc;
      SPLIT_STAGE("check");
      rho = theta_nrm > 0.0 ? rho / theta_nrm : rho;
      if (dbg_skip_check) rho = 0.0;
      c->last_split_iters = q_total; c->last_split_resid = rho;
      ++n_checks;
      // Ritz ratio sigma~_r / sigma~_k: an upper bound of the convergence factor sigma_{r+1} / sigma_k per half iteration
      std::vector<double> hs(r_sub);
      HIPCHK(hipMemcpyAsync(hs.data(), S, sizeof(double) * r_sub, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      const double ratio = (hs[k - 1] > 0.0 && std::isfinite(rho)) ? hs[r_sub - 1] / hs[k - 1] : 1.0;
      if (getenv("MPSK_SVD_DEBUG"))
        fprintf(stderr, "[mpsk_tsplit] subspace stage: r = %d of %d, %d iterations, residual %.3e (tol %.1e), Ritz ratio s_r / s_k = %.3f, %d Jacobi sweeps\n",
                r_sub, nn, q_total, rho, sub_tol, ratio, c->last_svd_sweeps);
      // (the check is blind to a collapsed basis -- U_k = 0 has residual 0 -- so the kept left vectors must also be an isometry)
      {
        double un = 0.0;
        if (!transposed) { if (int rc = mpsk_vnrm2(c, (int64_t)m * k, AL, &un)) return rc; }
        else { if (int rc = mpsk_vnrm2(c, (int64_t)n * k, Vk, &un)) return rc; }
        if (!(std::fabs(un * un - (double)k) <= 1.0e-8 * k) && !(ldal != m && !transposed)) rho = INFINITY;
      }
      if (!(rho <= sub_tol)) {
        // not there: iterations still needed from the measured check value
        // (a Ritz ratio taken before convergence UNDERestimates sigma_{r+1} / sigma_k -- sigma~_r is still far below
        // sigma_r --, so from the second check on the OBSERVED decay of the check value per iteration is used instead)
        int q_more = 1 << 20;
        double fac = ratio < 0.97 ? ratio * ratio : 1.0;              // error factor per iteration
        if (rho_prev > 0.0 && rho < rho_prev && q_total > q_prev) fac = std::pow(rho / rho_prev, 1.0 / (q_total - q_prev));
        else fac = std::sqrt(fac);                                    // first check: assume half the predicted rate
        if (fac < 0.95 && std::isfinite(rho)) q_more = (int)std::ceil(std::log(0.1 * sub_tol / rho) / std::log(fac));
        if (q_more < 2) q_more = 2;
        rho_prev = rho; q_prev = q_total;
        // budget: one iteration ~ 4 n^2 r flops + two Cholesky chains; the full Jacobi iteration ~ 10 sweeps of 6 n^3 / ... :
        // in measured terms (4096^2, r = 1536) 3.7 ms against ~200 ms, i.e. ~50 iterations; scaled by r / nn it stays ~40
        const int q_cap = 40;
        if (attempt >= 3 || q_total + q_more > q_cap) {
          c->last_split_path = 2;          // fall through to the full iteration; flat spectra come in runs (the early
          c->split_q_hint.erase(std::make_pair(nn, r_sub));   // sweeps of a chain), so the next calls skip the stage: 4, 8, ... 64 of them
          c->split_backoff = c->split_backoff ? std::min(2 * c->split_backoff, 64) : 4;
          c->split_skip = c->split_backoff;
          continue;
        }
        // continue from the r left Ritz vectors: Z = Qb Y (mm x r), Y <- orth1(A'^T Z), then q_more iterations
        GemmArgs gz = mk(Qb, Y, sb.Zb, mm, r_sub, r_sub, mm, r_sub, mm);
        HIPCHK(gemm_f64(gz, c->stream));
        GemmArgs gy = mk(Ap, sb.Zb, sb.Wb, nn, r_sub, mm, lda, mm, nn, 1, 0);
        HIPCHK(gemm_f64(gy, c->stream));
        if (int rc = split_orth1(c, nn, r_sub, sb.Wb, sb.Yb)) return rc;
        if (int rc = split_iterate(c, mm, nn, Ap, lda, sb, q_more)) return rc;
        q_total += q_more + 1;
        continue;
      }
      c->last_split_path = 1;
      c->split_backoff = 0;
      // next call's first guess (neighbouring bonds of a chain have similar spectra): what this one took, one less when
      // the first check already passed with two digits to spare (a failed check costs a second Jacobi stage, ~15 iterations'
      // worth; an iteration too many costs one)
      c->split_q_hint[std::make_pair(nn, r_sub)] = (n_checks == 1 && rho <= 1.0e-2 * sub_tol && q_total > 3) ? q_total - 1 : q_total;
      // S: the r leading values are Ritz values of the converged subspace; the rest is not computed
      if (nn > r_sub)
        hipLaunchKernelGGL(split_fill_nan_kernel, dim3((nn - r_sub + 255) / 256), dim3(256), 0, c->stream, (double*)S + r_sub, nn - r_sub);
      // discarded weight: || theta - AL M ||_F formed directly (theta_nrm^2 - |M|^2 cancels when little is discarded);
      // Qb (mm x nn) is free once AL / V_k exist
      double* Dm = Qb;
      HIPCHK(hipMemcpy2DAsync(Dm, sizeof(double) * m, theta, sizeof(double) * ldt, sizeof(double) * m, n, hipMemcpyDeviceToDevice, c->stream));
      if (!transposed) {
        GemmArgs gd = mk((const double*)AL, T, Dm, m, n, k, ldal, k, m);
        gd.alpha = -1.0; gd.beta = 1.0;
        HIPCHK(gemm_f64(gd, c->stream));
      } else {
        GemmArgs gd = mk(T, Vk, Dm, m, n, k, m, n, m, 0, 1);                      // theta - (theta V_k) V_k^T
        gd.alpha = -1.0; gd.beta = 1.0;
        HIPCHK(gemm_f64(gd, c->stream));
      }
      if (int rc = mpsk_vnrm2(c, (int64_t)m * n, Dm, disc_norm)) return rc;
      SPLIT_STAGE("remainder");
      // dominance probe.  The check above certifies that span(AL) is INVARIANT (a singular subspace of theta); that it is
      // the DOMINANT one rests on the random start having a component along every leading direction (probability one, and
      // r - k >= 64 spare columns).  Cheap insurance against the measure-zero case: a few power iterations on the
      // remainder D = theta - AL M, whose largest singular value must be sigma_{k+1} <= S[k-1]; a left-out direction well
      // above the cut would dominate D and show after 6 steps ((2/3)^12 < 1e-2 for sigma_miss = 1.5 S[k-1]).
      {
        double* xv = sb.Bp;                // n, then m doubles (Bp, mm x r, is free by now)
        double* yv = sb.Bp + ev((size_t)n);
        hipLaunchKernelGGL(split_fill_kernel, dim3(64), dim3(256), 0, c->stream, xv, (int64_t)n, 0xd0a1u);
        double est = 0.0;
        for (int it = 0; it < 6; ++it) {
          double nx = 0.0;
          if (int rc = mpsk_vnrm2(c, n, xv, &nx)) return rc;
          if (!(nx > 0.0)) break;
          HIPCHK(vec_scal(1.0 / nx, xv, n, c->stream));
          GemmArgs gy = mk(Dm, xv, yv, m, 1, n, m, n, m);
          HIPCHK(gemm_f64(gy, c->stream));
          if (int rc = mpsk_vnrm2(c, m, yv, &est)) return rc;
          GemmArgs gx = mk(Dm, yv, xv, n, 1, m, m, m, n, 1, 0);
          HIPCHK(gemm_f64(gx, c->stream));
        }
        SPLIT_STAGE("probe");
        if (est > 1.02 * hs[k - 1] + 1.0e-13 * theta_nrm) {
          if (getenv("MPSK_SVD_DEBUG"))
            fprintf(stderr, "[mpsk_tsplit] dominance probe: |D x| = %.6e above S[k-1] = %.6e -> full iteration\n", est, hs[k - 1]);
          c->last_split_path = 2;
          continue;
        }
      }
    }
    SPLIT_STAGE("before LQpos / QRpos");
    if (!transposed) {
      if (int rc = lqpos_f64(c, k, n, T, k, Cm, ldc, AR, ldar)) return rc;
    } else {
      if (int rc = qrpos_f64(c, m, k, T, m, AL, ldal, Cm, ldc)) return rc;
      HIPCHK(transpose(Vk, n, n, k, (double*)AR, ldar, c->stream));
    }
    SPLIT_STAGE("done");
    return MPSK_OK;
  }
  return fail(MPSK_ERR_HIP, "mpsk_tsplit: internal error (no path finished)");
#undef SPLIT_STAGE
}

// ---- complex128 two-site split (ctx dtype MPSK_C128) -------------------------------------------------------------------
// theta (m x n complex, interleaved) ~ AL (m x k) C (k x k) AR (k x n), all complex, AL / AR isometries, C lower triangular
// with a real positive diagonal, S the k kept COMPLEX singular values.  Runs on the real embedding E (2m x 2n), whose
// singular values are those of theta, each twice: the real split of E returns an arbitrary basis inside every doubled
// value, so what is taken from it is the kept SUBSPACE (2k + 16 leading left vectors through the truncation-aware
// real split), made an embedding again by projecting structured random vectors on it -- the construction of
// cplx.split_two_site (mpskit.jl_amd/cplx.py), including a J-invariant choice inside a cluster that straddles the cut.
// Truncation by max_keep only (truncdim); trunc_err > 0 is refused.
static int tsplit_c128(mpsk_ctx* c, int m, int n, const void* theta, int ldt, int max_keep, double trunc_err,
                       void* AL, int ldal, void* Cm, int ldc, void* AR, int ldar, void* S, int* kept, double* disc_norm) {
  REQUIRE(c && theta && AL && Cm && AR && S && kept && disc_norm, "NULL argument");
  REQUIRE(m > 0 && n > 0, "dimensions must be positive");
  REQUIRE(trunc_err == 0.0, "complex mpsk_tsplit truncates by max_keep only (trunc_err must be 0)");
  const int kfull = m < n ? m : n;
  const int k = (max_keep > 0 && max_keep < kfull) ? max_keep : kfull;
  REQUIRE(ldt >= m && ldal >= m && ldc >= k && ldar >= k, "leading dimension too small");
  HIPCHK(hipSetDevice(c->device));
  const int m2 = 2 * m, n2 = 2 * n, kE = 2 * kfull, K2 = 2 * k;
  const auto ev = [](size_t v) { return (v + 1) & ~(size_t)1; };
  const size_t e_d = ev((size_t)m2 * n2), al_d = ev((size_t)m2 * kE), c_d = ev((size_t)kE * kE), ar_d = ev((size_t)kE * n2);
  const size_t b_d = ev((size_t)m2 * K2), x_d = ev((size_t)m2 * K2), g_d = ev((size_t)kE * K2), me_d = ev((size_t)K2 * n2);
  const size_t mh_d = ev((size_t)2 * k * n), r_d = ev((size_t)K2 * K2);
  double* buf = nullptr;
  if (int rc = cx_scratch(c, 2, sizeof(double) * (e_d + al_d + c_d + ar_d + ev((size_t)kE) + 2 * b_d + 2 * x_d + g_d + me_d + mh_d + r_d), &buf))
    return rc;
  double *E = buf, *ALe = E + e_d, *Ce = ALe + al_d, *ARe = Ce + c_d, *Se = ARe + ar_d, *B = Se + ev((size_t)kE), *W = B + b_d;
  double *Xh = W + b_d, *Xe = Xh + x_d, *G = Xe + x_d, *Me = G + g_d, *Mh = Me + me_d, *Rs = Mh + mh_d;
  hipLaunchKernelGGL(cx_embed_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)theta, (int64_t)2 * ldt, m, n, E, (int64_t)m2);
  std::vector<double> hs;
  int lo = K2, hi = K2, have = 0;
  for (int pad = 16;; pad *= 4) {
    int req = K2 + pad;
    if (req > kE) req = kE;
    int kk = 0;
    double dn = 0.0;
    if (kE > 64) {
      if (int rc = tsplit_f64(c, m2, n2, E, m2, req, 0.0, ALe, m2, Ce, kE, ARe, kE, Se, &kk, &dn)) return rc;
    } else {        // small tensors: the full decomposition (mpsk_tsvd has no size floor); only its left vectors are used
      if (int rc = mpsk_tsvd(c, m2, n2, E, m2, ALe, m2, Se, ARe, kE, req, 0.0, &kk, &dn)) return rc;
    }
    have = kk;
    hs.resize(have);
    HIPCHK(hipMemcpyAsync(hs.data(), Se, sizeof(double) * have, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // [lo, hi): the even-aligned run of singular values equal to hs[K2 - 1] to 1e-8 -- the cluster the cut may split
    const double sk = hs[K2 - 1], tolc = 1.0e-8 * sk + 1.0e-14 * hs[0];
    lo = K2; while (lo > 0 && std::fabs(hs[lo - 1] - sk) <= tolc) --lo;
    hi = K2; while (hi < have && std::fabs(hs[hi] - sk) <= tolc) ++hi;
    lo -= lo & 1; hi += hi & 1;
    if (hi > have) hi = have;
    if (hi < have || have >= kE || pad > 4096) break;     // the cluster ends inside what was computed (or nothing is left)
  }
  // B: orthonormal basis of the kept, J-invariant subspace
  if (hi == K2) {
    HIPCHK(hipMemcpy2DAsync(B, sizeof(double) * m2, ALe, sizeof(double) * m2, sizeof(double) * m2, K2, hipMemcpyDeviceToDevice, c->stream));
  } else {
    if (lo > 0) HIPCHK(hipMemcpy2DAsync(B, sizeof(double) * m2, ALe, sizeof(double) * m2, sizeof(double) * m2, lo, hipMemcpyDeviceToDevice, c->stream));
    const int r2 = K2 - lo, nc = hi - lo;
    const double* Uc = ALe + (size_t)m2 * lo;
    hipLaunchKernelGGL(split_fill_kernel, dim3(1024), dim3(256), 0, c->stream, Xh, (int64_t)m2 * (r2 / 2), 0xc1u);
    hipLaunchKernelGGL(cx_embed_kernel, dim3(1024), dim3(256), 0, c->stream, Xh, (int64_t)m2, m, r2 / 2, Xe, (int64_t)m2);
    GemmArgs g1 = mk(Uc, Xe, G, nc, r2, m2, m2, m2, nc, 1, 0);            // Uc^T Xc
    HIPCHK(gemm_f64(g1, c->stream));
    GemmArgs g2 = mk(Uc, G, W, m2, r2, nc, m2, nc, m2);                   // Uc (Uc^T Xc)
    HIPCHK(gemm_f64(g2, c->stream));
    if (int rc = qrpos_f64(c, m2, r2, W, m2, B + (size_t)m2 * lo, m2, Rs, r2)) return rc;
  }
  // embedded orthonormal basis of that subspace: structured random vectors projected on it, QRpos
  hipLaunchKernelGGL(split_fill_kernel, dim3(1024), dim3(256), 0, c->stream, Xh, (int64_t)m2 * k, 0xc2u);
  hipLaunchKernelGGL(cx_embed_kernel, dim3(1024), dim3(256), 0, c->stream, Xh, (int64_t)m2, m, k, Xe, (int64_t)m2);
  GemmArgs g3 = mk(B, Xe, G, K2, K2, m2, m2, m2, K2, 1, 0);
  HIPCHK(gemm_f64(g3, c->stream));
  GemmArgs g4 = mk(B, G, W, m2, K2, K2, m2, K2, m2);
  HIPCHK(gemm_f64(g4, c->stream));
  if (int rc = qrpos_f64(c, m2, K2, W, m2, B, m2, Rs, K2)) return rc;          // B <- al_E (embedded, m2 x K2)
  hipLaunchKernelGGL(cx_half_kernel, dim3(1024), dim3(256), 0, c->stream, B, (int64_t)m2, m, k, (double*)AL, (int64_t)2 * ldal,
                     (double*)nullptr, (int64_t)0, 0);
  // M = AL^H theta (k x n complex) through the embedded product, then C AR = LQpos(M)
  GemmArgs g5 = mk(B, E, Me, K2, n2, m2, m2, m2, K2, 1, 0);
  HIPCHK(gemm_f64(g5, c->stream));
  hipLaunchKernelGGL(cx_half_kernel, dim3(1024), dim3(256), 0, c->stream, Me, (int64_t)K2, k, n, Mh, (int64_t)2 * k,
                     (double*)nullptr, (int64_t)0, 0);
  double tn = 0.0, mn = 0.0;
  if (int rc = mpsk_vnrm2(c, (int64_t)m2 * n2, E, &tn)) return rc;           // |E|^2 = 2 |theta|^2
  if (int rc = mpsk_vnrm2(c, (int64_t)2 * k * n, Mh, &mn)) return rc;
  if (int rc = lqpos_c128(c, k, n, Mh, k, Cm, ldc, AR, ldar)) return rc;
  std::vector<double> sc(kfull, std::nan(""));
  for (int j = 0; j < k; ++j) sc[j] = hs[2 * j];
  HIPCHK(hipMemcpyAsync(S, sc.data(), sizeof(double) * kfull, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));                                    // sc is a host temporary
  *kept = k;
  *disc_norm = std::sqrt(std::max(0.5 * tn * tn - mn * mn, 0.0));
  return MPSK_OK;
}

// conversions between an interleaved complex matrix (m x n complex = 2m x n doubles, ldh in DOUBLES) and its real embedding
// (2m x 2n doubles): one launch each.  mpsk_cx_half returns the STRUCTURED PART of a nearly embedded matrix
// (re = (E00 + E11) / 2, im = (E10 - E01) / 2 on every 2 x 2 block), which is the complex matrix whose embedding is closest.
int mpsk_cx_embed(mpsk_ctx* c, int m, int n, const void* H, int64_t ldh, void* E, int64_t lde) {
  REQUIRE(c && H && E, "NULL argument");
  REQUIRE(m > 0 && n > 0 && ldh >= 2 * (int64_t)m && lde >= 2 * (int64_t)m, "bad dimensions");
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(cx_embed_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)H, ldh, m, n, (double*)E, lde);
  return MPSK_OK;
}
int mpsk_cx_half(mpsk_ctx* c, int m, int n, const void* E, int64_t lde, void* H, int64_t ldh) {
  REQUIRE(c && H && E, "NULL argument");
  REQUIRE(m > 0 && n > 0 && ldh >= 2 * (int64_t)m && lde >= 2 * (int64_t)m, "bad dimensions");
  HIPCHK(hipSetDevice(c->device));
  hipLaunchKernelGGL(cx_half_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)E, lde, m, n, (double*)H, ldh,
                     (double*)nullptr, (int64_t)0, 0);
  return MPSK_OK;
}

// the ABI entry: fp64, or complex128 when the ctx dtype says so (mpsk_ctx_set_dtype)
int mpsk_tsplit(mpsk_ctx* c, int m, int n, const void* theta, int ldt, int max_keep, double trunc_err,
                void* AL, int ldal, void* Cm, int ldc, void* AR, int ldar, void* S, int* kept, double* disc_norm) {
  REQUIRE(c, "ctx is NULL");
  return c->dtype == MPSK_C128 ? tsplit_c128(c, m, n, theta, ldt, max_keep, trunc_err, AL, ldal, Cm, ldc, AR, ldar, S, kept, disc_norm)
                               : tsplit_f64(c, m, n, theta, ldt, max_keep, trunc_err, AL, ldal, Cm, ldc, AR, ldar, S, kept, disc_norm);
}

int mpsk_ctx_set_svd_mode(mpsk_ctx* c, int precondition) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(precondition >= 0 && precondition <= 3, "svd mode must be 0 (none), 1 (QR), 2 (QR + QR of R^T), 3 (2 + truncation-aware mpsk_tsplit)");
  c->svd_precondition = precondition;
  return MPSK_OK;
}
int mpsk_ctx_svd_stats(mpsk_ctx* c, int* last_sweeps) {
  REQUIRE(c && last_sweeps, "NULL argument");
  *last_sweeps = c->last_svd_sweeps;
  return MPSK_OK;
}
int mpsk_ctx_split_stats(mpsk_ctx* c, int* path, int* iterations, double* residual) {
  REQUIRE(c, "ctx is NULL");
  if (path) *path = c->last_split_path;
  if (iterations) *iterations = c->last_split_iters;
  if (residual) *residual = c->last_split_resid;
  return MPSK_OK;
}

// complex128 (ctx dtype MPSK_C128): C = alpha op(A) op(B) + beta C on interleaved complex matrices, op = conjugate transpose,
// alpha / beta real.  An interleaved complex matrix IS the even-column half of its real embedding, and only those columns
// of the result are wanted:  C_half = op(E_A) . (op(B))_half  -- ONE real GEMM with A embedded (2M x 2K) and B (or its
// conjugate transpose, materialised) as is: 8 M N K real flops, the complex optimum.
static int gemm_c128(mpsk_ctx* c, int transA, int transB, int M, int N, int K, double alpha, const void* A, int64_t lda,
                     const void* B, int64_t ldb, double beta, void* C, int64_t ldc) {
  HIPCHK(hipSetDevice(c->device));
  const int ar = transA ? K : M, ac = transA ? M : K;                  // A as stored: ar x ac complex
  const size_t ea = (size_t)4 * ar * ac, bt = transB ? (size_t)2 * K * N : 0;
  double* buf = nullptr;
  if (int rc = cx_scratch(c, 1, sizeof(double) * (ea + bt), &buf)) return rc;
  double* EA = buf;
  hipLaunchKernelGGL(cx_embed_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)A, 2 * lda, ar, ac, EA, (int64_t)2 * ar);
  const double* Bh = (const double*)B;
  int64_t ldbh = 2 * ldb;
  if (transB) {                                                        // B stored N x K complex -> B^H (K x N)
    double* Bt = buf + ea;
    hipLaunchKernelGGL(cx_ctranspose_kernel, dim3(1024), dim3(256), 0, c->stream, (const double*)B, 2 * ldb, N, K, Bt, (int64_t)2 * K);
    Bh = Bt; ldbh = (int64_t)2 * K;
  }
  GemmArgs g = mk(EA, Bh, (double*)C, 2 * M, N, 2 * K, (int64_t)2 * ar, ldbh, 2 * ldc, transA ? 1 : 0, 0);
  g.alpha = alpha; g.beta = beta;
  HIPCHK(gemm_f64(g, c->stream));
  return MPSK_OK;
}

int mpsk_gemm(mpsk_ctx* c, int transA, int transB, int M, int N, int K, double alpha, const void* A, int64_t lda,
              const void* B, int64_t ldb, double beta, void* C, int64_t ldc) {
  REQUIRE(c && A && B && C, "NULL argument");
  REQUIRE(M > 0 && N > 0 && K > 0, "dimensions must be positive");
  if (c->dtype == MPSK_C128) return gemm_c128(c, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
  HIPCHK(hipSetDevice(c->device));
  GemmArgs g = mk((const double*)A, (const double*)B, (double*)C, M, N, K, lda, ldb, ldc, transA ? 1 : 0, transB ? 1 : 0);
  g.alpha = alpha; g.beta = beta;
  HIPCHK(gemm_f64(g, c->stream));
  return MPSK_OK;
}

int mpsk_copy2d(mpsk_ctx* c, int rows, int cols, const void* src, int64_t lds, void* dst, int64_t ldd) {
  REQUIRE(c && src && dst, "NULL argument");
  REQUIRE(rows > 0 && cols > 0 && lds >= rows && ldd >= rows, "bad dimensions");
  HIPCHK(hipMemcpy2DAsync(dst, sizeof(double) * ldd, src, sizeof(double) * lds, sizeof(double) * rows, cols,
                          hipMemcpyDeviceToDevice, c->stream));
  return MPSK_OK;
}

// --------------------------------------------------------------------------------------------
// vectors
// --------------------------------------------------------------------------------------------
static int fetch_scalars(mpsk_ctx* c, int k, double* host_out) {
  HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal, sizeof(double) * k, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::memcpy(host_out, c->h_scal, sizeof(double) * k);
  return MPSK_OK;
}

int mpsk_vmultidot(mpsk_ctx* c, int64_t n, int k, const void* const* xs, const void* y, double* host_out) {
  REQUIRE(c && xs && y && host_out, "NULL argument");
  REQUIRE(k > 0 && k <= MAXK && n > 0, "bad k or n");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(vec_multidot((const double* const*)xs, k, (const double*)y, n, c->d_scal, c->d_partial, c->stream));
  return fetch_scalars(c, k, host_out);
}

int mpsk_vdot(mpsk_ctx* c, int64_t n, const void* x, const void* y, double* host_out) {
  const void* xs[1] = {x};
  return mpsk_vmultidot(c, n, 1, xs, y, host_out);
}

int mpsk_vnrm2(mpsk_ctx* c, int64_t n, const void* x, double* host_out) {
  double v = 0.0;
  if (int rc = mpsk_vdot(c, n, x, x, &v)) return rc;
  *host_out = std::sqrt(v);
  return MPSK_OK;
}

int mpsk_vgs_step(mpsk_ctx* c, int64_t n, int k, const void* const* xs, void* y, double* host_out) {
  REQUIRE(c && xs && y && host_out, "NULL argument");
  REQUIRE(k > 0 && k <= MAXK && n > 0, "bad k or n");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(vec_multidot((const double* const*)xs, k, (const double*)y, n, c->d_scal, c->d_partial, c->stream));
  HIPCHK(vec_multiaxpy((const double* const*)xs, c->d_scal, k, -1.0, (double*)y, n, c->stream));
  return fetch_scalars(c, k, host_out);
}

// One Krylov orthogonalisation step with ONE host sync: twice-iterated classical Gram-Schmidt of y
// against xs[0..k), then y <- y / ||y||.  host_h[j] = total coefficient on xs[j], *host_beta = ||y||
// before normalisation.  (KrylovKit's ModifiedGramSchmidt2 + normalize, 4 calls / 3 syncs fused.)
int mpsk_vorth_step(mpsk_ctx* c, int64_t n, int k, const void* const* xs, void* y, double* host_h, double* host_beta) {
  REQUIRE(c && xs && y && host_h && host_beta, "NULL argument");
  REQUIRE(k > 0 && 2 * k + 1 <= MAXK && n > 0, "bad k or n");
  HIPCHK(hipSetDevice(c->device));
  const double* const* X = (const double* const*)xs;
  double* yy = (double*)y;
  HIPCHK(vec_cgs2(X, k, yy, n, c->d_scal, c->d_partial, c->stream));
  HIPCHK(vec_scal_rsqrt_dev(c->d_scal + 2 * k, yy, n, c->stream));
  double tmp[MAXK];
  if (int rc = fetch_scalars(c, 2 * k + 1, tmp)) return rc;
  for (int j = 0; j < k; ++j) host_h[j] = tmp[j] + tmp[k + j];
  *host_beta = tmp[2 * k] > 0.0 ? std::sqrt(tmp[2 * k]) : 0.0;
  return MPSK_OK;
}

// Same fused step with the 2k+1 scalars left on the DEVICE (dev_out: first-pass dots [k], second-pass dots [k],
// squared norm of the remainder): no host synchronisation, so a fixed-length Krylov recurrence can be enqueued
// back to back and its small projected matrix is read once at the end.
int mpsk_vorth_step_dev(mpsk_ctx* c, int64_t n, int k, const void* const* xs, void* y, void* dev_out) {
  REQUIRE(c && xs && y && dev_out, "NULL argument");
  REQUIRE(k > 0 && 2 * k + 1 <= MAXK && n > 0, "bad k or n");
  HIPCHK(hipSetDevice(c->device));
  const double* const* X = (const double* const*)xs;
  double* yy = (double*)y;
  double* out = (double*)dev_out;
  HIPCHK(vec_cgs2(X, k, yy, n, out, c->d_partial, c->stream));
  HIPCHK(vec_scal_rsqrt_dev(out + 2 * k, yy, n, c->stream));
  return MPSK_OK;
}

int mpsk_vlincomb(mpsk_ctx* c, int64_t n, int k, const void* const* xs, const double* host_coefs, void* y) {
  REQUIRE(c && xs && y && host_coefs, "NULL argument");
  REQUIRE(k > 0 && k <= MAXK && n > 0, "bad k or n");
  HIPCHK(hipSetDevice(c->device));
  // asynchronous: the coefficients travel through their own pinned buffer, which is only rewritten once the upload that
  // last read it has completed (an event wait, normally already satisfied) -- no stream synchronisation on either side
  if (c->coef_pending) { HIPCHK(hipEventSynchronize(c->ev_coef)); c->coef_pending = false; }
  std::memcpy(c->h_coef, host_coefs, sizeof(double) * k);
  HIPCHK(hipMemcpyAsync(c->d_coef, c->h_coef, sizeof(double) * k, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->ev_coef, c->stream));
  c->coef_pending = true;
  HIPCHK(hipMemsetAsync(y, 0, sizeof(double) * n, c->stream));
  HIPCHK(vec_multiaxpy((const double* const*)xs, c->d_coef, k, 1.0, (double*)y, n, c->stream));
  return MPSK_OK;
}

// ys[j] = sum_i host_coefs[i + k j] xs[i], j < m: the basis rotation of a thick restart in one pass over the vectors
int mpsk_vmultilincomb(mpsk_ctx* c, int64_t n, int k, const void* const* xs, int m, void* const* ys, const double* host_coefs) {
  REQUIRE(c && xs && ys && host_coefs, "NULL argument");
  REQUIRE(k > 0 && k <= 32 && m > 0 && m <= 32 && n > 0, "needs 0 < k, m <= 32 and n > 0");
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < k; ++i) REQUIRE(ys[j] != xs[i], "outputs must not alias inputs");
  HIPCHK(hipSetDevice(c->device));
  if (c->coef_pending) { HIPCHK(hipEventSynchronize(c->ev_coef)); c->coef_pending = false; }
  std::memcpy(c->h_coef, host_coefs, sizeof(double) * (size_t)k * m);
  HIPCHK(hipMemcpyAsync(c->d_coef, c->h_coef, sizeof(double) * (size_t)k * m, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->ev_coef, c->stream));
  c->coef_pending = true;
  HIPCHK(vec_multilincomb((const double* const*)xs, k, (double* const*)ys, m, c->d_coef, n, c->stream));
  return MPSK_OK;
}

// Ritz coefficients of a fixed-budget solve from the scalars mpsk_vorth_step_dev collected (see ritz_small_kernel)
int mpsk_vritz_dev(mpsk_ctx* c, int m, int stride, const void* dev_slot, void* dev_coef, void* dev_info) {
  REQUIRE(c && dev_slot && dev_coef, "NULL argument");
  REQUIRE(m > 0 && m <= 32 && stride >= 2 * m + 1, "needs 0 < m <= 32 and stride >= 2 m + 1");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(vec_ritz_small((const double*)dev_slot, m, stride, (double*)dev_coef, (double*)dev_info, c->stream));
  return MPSK_OK;
}

// y = sum_j dev_coefs[j] xs[j] with the coefficients already on the device
int mpsk_vlincomb_dev(mpsk_ctx* c, int64_t n, int k, const void* const* xs, const void* dev_coefs, void* y) {
  REQUIRE(c && xs && y && dev_coefs, "NULL argument");
  REQUIRE(k > 0 && k <= MAXK && n > 0, "bad k or n");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemsetAsync(y, 0, sizeof(double) * n, c->stream));
  HIPCHK(vec_multiaxpy((const double* const*)xs, (const double*)dev_coefs, k, 1.0, (double*)y, n, c->stream));
  return MPSK_OK;
}

// y = x / |x| without a host synchronisation; |x|^2 is left in dev_n2 (device memory, one double) when given
int mpsk_vnormalize_dev(mpsk_ctx* c, int64_t n, const void* x, void* y, void* dev_n2) {
  REQUIRE(c && x && y, "NULL argument");
  REQUIRE(n > 0, "bad n");
  HIPCHK(hipSetDevice(c->device));
  double* n2 = dev_n2 ? (double*)dev_n2 : c->d_scal + (MAXK - 1);
  const double* xs[1] = {(const double*)x};
  HIPCHK(vec_multidot(xs, 1, (const double*)x, n, n2, c->d_partial, c->stream));
  if (y != x) HIPCHK(vec_scal_rsqrt_dev_oop(n2, (const double*)x, (double*)y, n, c->stream));
  else HIPCHK(vec_scal_rsqrt_dev(n2, (double*)y, n, c->stream));
  return MPSK_OK;
}

// dev_out[0] = |x|^2 (device memory), no host synchronisation
int mpsk_vnrm2_dev(mpsk_ctx* c, int64_t n, const void* x, void* dev_out) {
  REQUIRE(c && x && dev_out, "NULL argument");
  REQUIRE(n > 0, "bad n");
  HIPCHK(hipSetDevice(c->device));
  const double* xs[1] = {(const double*)x};
  HIPCHK(vec_multidot(xs, 1, (const double*)x, n, (double*)dev_out, c->d_partial, c->stream));
  return MPSK_OK;
}

int mpsk_vaxpby(mpsk_ctx* c, int64_t n, double alpha, const void* x, double beta, void* y) {
  REQUIRE(c && x && y, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(vec_axpby(alpha, (const double*)x, beta, (double*)y, n, c->stream));
  return MPSK_OK;
}
int mpsk_vscal(mpsk_ctx* c, int64_t n, double alpha, void* x) {
  REQUIRE(c && x, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(vec_scal(alpha, (double*)x, n, c->stream));
  return MPSK_OK;
}
int mpsk_vtimes_i(mpsk_ctx* c, int64_t n, const void* x, void* y) {
  REQUIRE(c && x && y, "NULL argument");
  REQUIRE(n % 2 == 0 && x != y, "needs an even length (interleaved re/im row pairs) and out-of-place operands");
  REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "operands must be 16-byte aligned");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(vec_times_i((const double*)x, (double*)y, n, c->stream));
  return MPSK_OK;
}
int mpsk_vcopy(mpsk_ctx* c, int64_t n, const void* x, void* y) {
  REQUIRE(c && x && y, "NULL argument");
  HIPCHK(hipMemcpyAsync(y, x, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
  return MPSK_OK;
}
int mpsk_vzero(mpsk_ctx* c, int64_t n, void* x) {
  REQUIRE(c && x, "NULL argument");
  HIPCHK(hipMemsetAsync(x, 0, sizeof(double) * n, c->stream));
  return MPSK_OK;
}

}  // extern "C"
