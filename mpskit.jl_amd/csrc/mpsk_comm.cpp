// libmpsk_comm: RCCL behind the C ABI of include/mpsk_comm.h (host code only; the kernels are RCCL's).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include <string>
#include "mpsk_comm.h"

static thread_local std::string g_cerr;
static int cfail(const std::string& m) { g_cerr = m; return MPSK_ERR_HIP; }
#define NCCLCHK(expr)                                                                   \
  do {                                                                                  \
    ncclResult_t r_ = (expr);                                                           \
    if (r_ != ncclSuccess) return cfail(std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)

struct mpsk_comm {
  mpsk_ctx* ctx;
  ncclComm_t nc;
  int world, rank, device;
};

static_assert(sizeof(ncclUniqueId) <= MPSK_COMM_ID_BYTES, "ncclUniqueId does not fit MPSK_COMM_ID_BYTES");

extern "C" {

const char* mpsk_comm_last_error(void) { return g_cerr.c_str(); }

int mpsk_comm_unique_id(void* id_out) {
  if (!id_out) return cfail("mpsk_comm_unique_id: NULL argument");
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  std::memset(id_out, 0, MPSK_COMM_ID_BYTES);
  std::memcpy(id_out, &id, sizeof(id));
  return MPSK_OK;
}

int mpsk_comm_create(mpsk_ctx* ctx, int world, int rank, const void* id, mpsk_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return cfail("mpsk_comm_create: bad argument");
  int dev = 0;
  if (mpsk_ctx_get_device(ctx, &dev) != MPSK_OK) return cfail("mpsk_comm_create: mpsk_ctx_get_device failed");
  if (hipSetDevice(dev) != hipSuccess) return cfail("mpsk_comm_create: hipSetDevice failed");
  ncclUniqueId nid;
  std::memcpy(&nid, id, sizeof(nid));
  auto* c = new mpsk_comm{ctx, nullptr, world, rank, dev};
  ncclResult_t r = ncclCommInitRank(&c->nc, world, nid, rank);
  if (r != ncclSuccess) { delete c; return cfail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
  *out = c;
  return MPSK_OK;
}

int mpsk_comm_destroy(mpsk_comm* c) {
  if (!c) return MPSK_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();          // (not through c->ctx: the ctx may already have been destroyed by the caller)
  (void)ncclCommDestroy(c->nc);
  delete c;
  return MPSK_OK;
}

int mpsk_comm_info(const mpsk_comm* c, int* world, int* rank) {
  if (!c) return cfail("mpsk_comm_info: comm is NULL");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  return MPSK_OK;
}

static int stream_of(mpsk_comm* c, hipStream_t* s) {
  void* p = nullptr;
  if (mpsk_ctx_get_stream(c->ctx, &p) != MPSK_OK) return cfail("mpsk_ctx_get_stream failed");
  *s = (hipStream_t)p;
  return MPSK_OK;
}

int mpsk_comm_allgather(mpsk_comm* c, const void* send, void* recv, size_t count) {
  if (!c || !send || !recv) return cfail("mpsk_comm_allgather: NULL argument");
  hipStream_t s;
  if (int rc = stream_of(c, &s)) return rc;
  NCCLCHK(ncclAllGather(send, recv, count, ncclDouble, c->nc, s));
  return MPSK_OK;
}

int mpsk_comm_allreduce_sum(mpsk_comm* c, void* buf, size_t count) {
  if (!c || !buf) return cfail("mpsk_comm_allreduce_sum: NULL argument");
  hipStream_t s;
  if (int rc = stream_of(c, &s)) return rc;
  NCCLCHK(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->nc, s));
  return MPSK_OK;
}

int mpsk_comm_reduce_scatter_sum(mpsk_comm* c, const void* send, void* recv, size_t count) {
  if (!c || !send || !recv) return cfail("mpsk_comm_reduce_scatter_sum: NULL argument");
  hipStream_t s;
  if (int rc = stream_of(c, &s)) return rc;
  NCCLCHK(ncclReduceScatter(send, recv, count, ncclDouble, ncclSum, c->nc, s));
  return MPSK_OK;
}

int mpsk_comm_hac_apply(mpsk_comm* c, mpsk_hac* h, const void* xblk, void* yblk, int Dl, int d, int Dr) {
  if (!c || !h || !xblk || !yblk) return cfail("mpsk_comm_hac_apply: NULL argument");
  if (Dl <= 0 || d <= 0 || Dr <= 0 || Dl % c->world != 0) return cfail("mpsk_comm_hac_apply: world must divide Dl");
  const size_t blk = (size_t)(Dl / c->world) * d * Dr;
  double* mine = (double*)yblk + (size_t)c->rank * blk;
  if (int rc = mpsk_hac_apply(h, xblk, c->world, mine)) return cfail(std::string("mpsk_hac_apply: ") + mpsk_last_error()), rc;
  return mpsk_comm_allgather(c, mine, yblk, blk);
}

}  // extern "C"
