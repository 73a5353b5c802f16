/* libmpsk_comm -- the collectives of the bond-sharded sweep behind a C ABI.
 *
 * The reference is single-process Julia; a multi-GPU drop-in runs ONE Julia process per GPU (MPI.jl or Distributed
 * workers) and needs three collectives on device buffers, on the stream of its mpsk_ctx:
 *   - all-gather     : completes the output of the sharded matvec (one per H_AC application) and materialises a
 *                      column-sharded right environment once per site visit (ONE call: rank-major staging buffer);
 *   - reduce-scatter : completes a left-environment update whose contraction index a' is sharded AND whose output bond is
 *                      sharded: every rank receives exactly the row block it stores (half the bytes of an all-reduce);
 *   - all-reduce     : the same update when the output bond is too small to shard (chain edges).
 * This library is a thin wrapper over RCCL (ncclAllGather / ncclReduceScatter / ncclAllReduce over xGMI).  The Python host of this repo
 * makes the same two calls through torch.distributed (backend "nccl" == RCCL; mpskit.jl_amd/dist.py `Comm`); a Julia
 * host binds these entry points with ccall.  Rendezvous: rank 0 calls mpsk_comm_unique_id and ships
 * the MPSK_COMM_ID_BYTES to the other ranks by whatever transport the host has (MPI.Bcast, a shared file, a socket);
 * every rank then calls mpsk_comm_create.
 *
 * All functions return 0 on success; on failure mpsk_comm_last_error() gives the message.
 */
#ifndef MPSK_COMM_H
#define MPSK_COMM_H
#include <stddef.h>
#include "mpsk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpsk_comm mpsk_comm;
#define MPSK_COMM_ID_BYTES 128

const char* mpsk_comm_last_error(void);
int mpsk_comm_unique_id(void* id_out /* MPSK_COMM_ID_BYTES */);
/* collective over `world` ranks; binds to ctx's device, every later call is enqueued on ctx's CURRENT stream */
int mpsk_comm_create(mpsk_ctx* ctx, int world, int rank, const void* id, mpsk_comm** out);
int mpsk_comm_destroy(mpsk_comm* comm);
int mpsk_comm_info(const mpsk_comm* comm, int* world, int* rank);
/* recv[r * count : (r + 1) * count] = rank r's send (fp64).  send == recv + rank * count is the in-place form the
 * matvec uses: the local kernel has written this rank's block straight into the destination vector. */
int mpsk_comm_allgather(mpsk_comm* comm, const void* send, void* recv, size_t count);
int mpsk_comm_allreduce_sum(mpsk_comm* comm, void* buf, size_t count);
/* recv[0 : count] = sum over ranks of their send[rank * count : (rank + 1) * count]  (send holds world * count doubles:
 * the partial left environment re-ordered into rank-major row blocks [world][W][Drb / world, Dr], dist.py) */
int mpsk_comm_reduce_scatter_sum(mpsk_comm* comm, const void* send, void* recv, size_t count);
/* One sharded application of a prepared effective Hamiltonian (mpsk_hac_create with GL = this rank's rows, Dlo = Dl / world):
 * xblk, yblk in the blocked layout of mpsk_dAC_blocked ([world][Dl / world, d, Dr]).  The local rows go into block
 * `rank` of yblk, the in-place all-gather completes it on every rank. */
int mpsk_comm_hac_apply(mpsk_comm* comm, mpsk_hac* h, const void* xblk, void* yblk, int Dl, int d, int Dr);

#ifdef __cplusplus
}
#endif
#endif /* MPSK_COMM_H */
