/* triro_rccl.h -- C ABI of libtriro_rccl.so: ONE call that enqueues a whole pipelined step of a ray-sharded closest-hit
 * query on this rank's GPU (round 6; SURVEY.md 7.1 "csrc/gather_rccl.cpp", 8(e)).
 *
 * The reference is single-GPU (triro/backend/base.cpp:15-17); there is no interface of it to replace.  What this replaces
 * is the Python driver of the same step (triro/ray/sharded.py: ShardedRayMeshIntersector.closest_of_shard_async with
 * records="slot"), which costs the host ~200 us of Python + torch.distributed per step against ~0.2 ms of GPU time:
 *
 *     every rank but the destination, per chunk k of its shard:
 *         tr_intersects_closest_slots (4 bytes per ray)  ->  ncclSend to the destination, on the caller's stream
 *     the destination, per chunk k:
 *         tr_intersects_closest of its own chunk, dense, straight into its rows of the outputs (caller's stream)
 *         on the side stream, behind that chunk's trace:  ncclGroupStart; ncclRecv x (world - 1); ncclGroupEnd  into the
 *         peers' rows of the record buffer, then tr_closest_from_slots on those rows (the rays of the whole batch are on
 *         the destination: the reference's call hands the whole batch to one process, ray_optix.py:121-146)
 *     the destination records `done_event` on the side stream; the caller orders its stream behind it.
 *
 * 7 peers -> 7 point-to-point receives over 7 distinct xGMI links, no ring, no reduction.  Results are those of
 * tr_intersects_closest on the whole batch, bit for bit (replicas of the hierarchy are bit-identical: tr_bvh_replica_hash).
 *
 * The library finds RCCL at run time: the copy that is already loaded (PyTorch's), else librccl.so.1 -- it has no link-time
 * dependency on it, and libtriro_hip.so has none on this library.  Python keeps the ladder and the preflight
 * (sharded.py); this is the top rung ("native"), opt-in: TRIRO_NATIVE_STEP=1 / set_exchange_mode("native").
 */
#ifndef TRIRO_RCCL_H
#define TRIRO_RCCL_H

#include "triro_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TR_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

typedef struct tr_comm tr_comm;

/* message of the last failure of an entry point of THIS library on the calling thread */
const char *tr_rccl_last_error(void);
/* 0 when RCCL could be found and every entry point resolved (tr_rccl_last_error says what was missing) */
int tr_rccl_available(void);
/* rank 0: a fresh communicator id (ncclGetUniqueId); the caller ships the bytes to the other ranks (any transport) */
int tr_comm_unique_id(uint8_t id[TR_COMM_ID_BYTES]);
/* every rank: ncclCommInitRank on `device` (collective over the `world` ranks that hold the same id) */
int tr_comm_create(const uint8_t id[TR_COMM_ID_BYTES], int world, int rank, int device, tr_comm **out);
int tr_comm_destroy(tr_comm *comm);
/* From any thread: ncclCommAbort -- ends a transfer that nobody answers (the kernels leave their streams) and frees the
 * communicator; later steps on the handle fail, tr_comm_destroy still takes it.  What a preflight's watchdog calls. */
int tr_comm_abort(tr_comm *comm);

enum {
    TR_STEP_NO_EXCHANGE = 1, /* the peers' records are already in d_records (bench.py --emulate-world): no send / recv   */
    TR_STEP_LOOPBACK = 2,    /* test hook for ONE rank: this rank plays every rank in turn -- the peers' chunks are traced
                                into a staging buffer and travel through ncclSend / ncclRecv to itself (one group)      */
    TR_STEP_TEST_DROP_SEND = 4 /* with LOOPBACK: the sends are left out, the receives wait for a peer that never answers --
                                the situation tr_comm_abort exists for (tests/native_step_world1.py)                     */
};

typedef struct tr_shard_step {
    int64_t n_total;       /* rays of the whole batch                                                                   */
    int32_t world, rank;   /* as in the communicator (LOOPBACK: the pretended world, rank = dst)                        */
    int32_t dst;           /* rank that receives the results                                                            */
    int32_t chunks;        /* every shard is cut into this many chunks (>= 1; every rank must pass the same)            */
    int64_t per_row;       /* > 1: image batch, shards and chunks are whole rows of this many rays; else flat           */
    const int64_t *bounds; /* [world][2]: lo, hi of every rank's shard in flat ray indices (multiples of per_row)       */
    const tr_rays *my_rays;  /* this rank's shard: [m, 3] or, image, [rows, per_row, 3] (any strides)                    */
    const tr_rays *all_rays; /* destination: the whole batch, [n_total, 3] or [rows, per_row, 3]; NULL elsewhere         */
    int32_t *d_records;    /* destination: [n_total] slots of every ray (its own rows stay unused); others: [m]          */
    int32_t *d_staging;    /* LOOPBACK only: [largest peer shard] scratch                                               */
    uint8_t *d_hit;        /* destination: the five outputs of the WHOLE batch, dense per flat ray index                 */
    uint8_t *d_front;
    int32_t *d_tri;
    float *d_loc3;
    float *d_uv2;
    void *stream;          /* hipStream_t the caller's work is on                                                        */
    void *side_stream;     /* destination: hipStream_t of the receives and of the peers' rows; a peer: of its sends (NULL: `stream`) */
    void *done_event;      /* destination: hipEvent_t recorded on side_stream when every row is final                    */
    int32_t flags;
} tr_shard_step;

/* Enqueues the step and returns; nothing is synchronised.  A rank whose trace fails still takes part in the exchange
 * with records that say "miss" and returns the error afterwards, so that the others do not hang in their receives. */
int tr_sharded_closest_step(const tr_bvh *bvh, tr_comm *comm, const tr_shard_step *step);

#ifdef __cplusplus
}
#endif
#endif
