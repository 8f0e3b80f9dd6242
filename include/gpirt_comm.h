/*
 * gpirt_comm.h -- the collective of the item-sharded respondent step on RCCL over xGMI,
 * exported by libgpirt_rccl.so (kept apart from libgpirt_hip.so so that single-GPU callers
 * never load RCCL).  No torch, no MPI: the host only has to carry 128 bytes from rank 0 to
 * the other ranks once.
 *
 * The reference has no counterpart (it is single-process R and holds no
 * collective anywhere); what this serves is gpirt_config.allreduce of include/gpirt.h, the
 * one exchange of the sharded sweep: per lock-step proposal round of the respondent step
 * (utils/sample_score.m's role in main.m:63), the sum over ranks of each rank's per-respondent
 * partial log-likelihoods.
 *
 * One process per GPU:
 *     gpirt_set_device(local_rank);
 *     if (rank == 0) gpirt_comm_unique_id(id);        // then send id[128] to every rank
 *     gpirt_comm_create(id, rank, world, &comm);      // collective: all ranks call it
 *     gpirt_config_defaults(&cfg); ... cfg.item_begin/item_end = this rank's items ...
 *     gpirt_comm_bind(comm, &cfg);                    // rank, world, hook, stream-ordered flag
 *     gpirt_sampler_create(&cfg, ..., &s);
 *     gpirt_comm_set_stream(comm, gpirt_sampler_stream(s));
 *     gpirt_sampler_sweep(s, k); ...
 *     gpirt_sampler_destroy(s); gpirt_comm_destroy(comm);
 *
 * The sum is an ncclAllGather of the ranks' vectors followed by a local sum in rank order,
 * both enqueued on the sampler's stream (no host round trip per round): every rank adds the
 * same numbers in the same order, so all ranks hold the same bits and take the same accept /
 * shrink decisions, and a run repeats bit for bit.  Across different numbers of ranks the
 * partial sums are re-associated (differences at rounding level, 1e-13 relative).
 */
#ifndef GPIRT_COMM_H
#define GPIRT_COMM_H

#include "gpirt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GPIRT_COMM_ID_BYTES 128

typedef struct gpirt_comm gpirt_comm;

/* Text of the last failure of a gpirt_comm_* call on this thread. */
const char *gpirt_comm_last_error(void);
/* Rank 0: fill id[GPIRT_COMM_ID_BYTES] (an ncclUniqueId); the host carries it to every rank. */
int gpirt_comm_unique_id(void *id);
/* All ranks, each with its own device current: join the communicator. */
int gpirt_comm_create(const void *id, int32_t rank, int32_t world, gpirt_comm **out);
/* Write rank, world, the hook below, its user pointer and GPIRT_ALLREDUCE_ON_STREAM into cfg, and
 * allocate the all-gather's staging for the largest vector a sweep of THIS configuration exchanges
 * (call it with n, m, C, n_grid, theta_mode and flags final), so that no exchange inside a sweep
 * allocates or drains the stream. */
int gpirt_comm_bind(gpirt_comm *c, gpirt_config *cfg);
/* The HIP stream (hipStream_t) the collectives are enqueued on: gpirt_sampler_stream(s). */
int gpirt_comm_set_stream(gpirt_comm *c, void *stream);
/* The gpirt_allreduce_fn itself: sum `count` doubles at the DEVICE pointer d_buf over the ranks,
 * in place, stream-ordered (returns once enqueued).  `comm` is the gpirt_comm. */
int gpirt_comm_allreduce(double *d_buf, int64_t count, void *comm);
/* Number of collectives enqueued so far. */
int64_t gpirt_comm_calls(const gpirt_comm *c);
/* Ranks in the communicator as RCCL itself reports them (ncclCommCount); 0 on failure. */
int gpirt_comm_world(const gpirt_comm *c);
/* Doubles per rank the all-gather's staging holds now. */
int64_t gpirt_comm_capacity(const gpirt_comm *c);
void gpirt_comm_destroy(gpirt_comm *c);

#ifdef __cplusplus
}
#endif
#endif /* GPIRT_COMM_H */
