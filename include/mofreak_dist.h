/*
 * mofreak_dist.h -- C ABI of libmofreak_dist.so: the N-GPU side of the MoFREAK extraction path, one process per GPU,
 * over RCCL (rccl.h) with no Python and no torch in between.
 *
 * The reference has no counterpart: computeMoFREAKFiles (src/MoFREAK/main.cpp:854-924) walks the dataset sequentially on
 * one CPU.  The path shards with no data-path collective -- a video depends on nothing but itself -- so this library holds
 * only what the N > 1 run adds: the assignment of videos to ranks (longest processing time first) and
 * the ONE exchange step, the variable-length gather of 32-byte .mofreak rows to the root rank: ncclAllGather of the ranks'
 * row counts, then grouped ncclSend / ncclRecv, every peer straight to the root over its own xGMI link (not a ring).
 * mofreak_amd/host/MoFREAKUtilities (computeMoFREAKFromFilesSharded) and `facade_main files --ranks N` use it.
 *
 * Every function returns MOFREAK_OK or a negative MOFREAK_ERR_* (mofreak_hip.h); mofreak_dist_last_error() has the text.
 */
#ifndef MOFREAK_DIST_H
#define MOFREAK_DIST_H

#include <stdint.h>

#include "mofreak_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MOFREAK_DIST_ABI_VERSION 1
#define MOFREAK_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */

int mofreak_dist_abi_version(void);
const char *mofreak_dist_last_error(void);

/* Longest-processing-time-first assignment of n videos with the given costs (frame counts, file sizes) to `world` ranks:
 * rank_of_out[i] = the rank video i goes to.  Deterministic (ties: lower index, lower rank): every rank computes the same
 * table.  Host only. */
int mofreak_shard_lpt(const int64_t *costs, int n, int world, int32_t *rank_of_out);

typedef struct mofreak_comm mofreak_comm;
/* ncclGetUniqueId: rank 0 makes the id and hands its 128 bytes to the other ranks (a file, a pipe, MPI ...). */
int mofreak_comm_unique_id(void *id128_out);
/* ncclCommInitRank on `device` (hipSetDevice) plus a stream of its own for the exchanges. */
int mofreak_comm_create(const void *id128, int rank, int world, int device, mofreak_comm **out);
/* An existing communicator (ncclComm_t) and stream (hipStream_t) of the caller's; not destroyed with the handle. */
int mofreak_comm_wrap(void *nccl_comm, void *hip_stream, int rank, int world, mofreak_comm **out);
void mofreak_comm_destroy(mofreak_comm *c);
int mofreak_comm_rank(const mofreak_comm *c);
int mofreak_comm_world(const mofreak_comm *c);

/* ncclAllGather of every rank's row count: counts_out[world] (host) on every rank.  Synchronises the exchange stream. */
int mofreak_gather_counts(mofreak_comm *c, int64_t n_rows, int64_t *counts_out);
/* ncclAllReduce(sum) of n int64 values given and returned in host memory (the per-video row counts of a round: every
 * video belongs to exactly one rank, so the sum is the count).  Synchronises. */
int mofreak_allreduce_sum_i64(mofreak_comm *c, int64_t *values, int n);
/* The gather itself.  d_rows: this rank's counts[rank] rows in DEVICE memory; on the root d_out (device, room for the sum
 * of counts) receives rank 0's rows, then rank 1's, ...: the root copies its own rows device to device and posts one
 * ncclRecv per peer with rows, every peer one ncclSend, all inside one ncclGroupStart / ncclGroupEnd.  d_out is ignored
 * on the other ranks.  Synchronises the exchange stream. */
int mofreak_gather_rows(mofreak_comm *c, const mofreak_row *d_rows, const int64_t *counts, int root, mofreak_row *d_out);
/* A grouped ncclSend + ncclRecv of n_bytes from this rank to itself on device buffers of the library's own, compared
 * afterwards: the point-to-point path on a box that has one GPU.  MOFREAK_OK if the bytes arrived. */
int mofreak_comm_self_exchange(mofreak_comm *c, int64_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MOFREAK_DIST_H */
