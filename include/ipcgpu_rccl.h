/* ipcgpu_rccl.h -- the multi-GPU glue on the CALLER's side: RCCL (the ROCm build of NCCL) bound to a libipcgpu context from C.
 *
 * One process per GPU.  Every rank creates its context, rank 0 draws a unique id and hands its 128 bytes to the other ranks by
 * whatever the host program already has (MPI_Bcast inside ipc-sim/IPC's main.cpp, torch.distributed in bench.py, a file), then
 * every rank attaches:
 *
 *     char id[IPCGPU_RCCL_ID_BYTES];
 *     if (rank == 0) ipcgpu_rccl_unique_id(id);
 *     bcast(id);                                    // the caller's bootstrap
 *     ipcgpu_rccl_attach(ctx, rank, world, id);     // ncclCommInitRank + ipcgpu_opt_set_allreduce_stream + ipcgpu_opt_set_exchange_stream
 *     ipcgpu_linsys_set_shard(ctx, rank, world);    // and / or ipcgpu_ctx_set_shard
 *
 * From then on every exchange of the library is enqueued on the context's own HIP stream, no Python, no host synchronisation: the nodal gradient,
 * energies, step bounds, the pivot flag and the solution vector as ncclAllReduce; the update matrices / vectors of the sharded Cholesky and the
 * solution entries of the separators above the cut as ONE group of ncclSend / ncclRecv per level of the cut (round 5: point to point to the ranks
 * that need them -- xGMI links every pair of GPUs of a node directly).
 * Source: include/adapters/ipcgpu_rccl.cpp (links -lrccl -lipcgpu), built by ipc_amd/build.py into ipc_amd/libipcgpu_rccl.so.
 */
#ifndef IPCGPU_RCCL_H
#define IPCGPU_RCCL_H
#include "ipcgpu.h"
#ifdef __cplusplus
extern "C" {
#endif
#define IPCGPU_RCCL_ID_BYTES 128
/* rank 0: a fresh ncclUniqueId (128 bytes) */
int ipcgpu_rccl_unique_id(void* id128);
/* all ranks (collective): communicator on the context's device, all-reduce hook installed */
int ipcgpu_rccl_attach(ipcgpu_ctx* ctx, int rank, int world, const void* id128);
/* destroys the communicator and removes the hook (the context and its solver stop using it at once: a host hook set with
 * ipcgpu_opt_set_allreduce before the attach is in charge again).  Call it BEFORE ipcgpu_ctx_destroy: the binding lives on the caller's side and the
 * library does not know about it. */
int ipcgpu_rccl_detach(ipcgpu_ctx* ctx);
/* all-reduces `count` doubles of a scratch device buffer filled with (rank + 1) through the installed hook and returns element 0:
 * world * (world + 1) / 2 for op 0 (sum), 1 for op 1 (min).  A smoke test of the binding. */
int ipcgpu_rccl_selftest(ipcgpu_ctx* ctx, int rank, long long count, int op, double* result);
/* ring shift of `count` doubles filled with (rank + 1) through the point-to-point hook (ncclSend / ncclRecv in one group): returns what arrived from
 * rank - 1 (mod world), i.e. ((rank - 1 + world) % world) + 1. */
int ipcgpu_rccl_selftest_p2p(ipcgpu_ctx* ctx, int rank, int world, long long count, double* result);
const char* ipcgpu_rccl_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
