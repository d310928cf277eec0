// gather_rccl.cpp -- libtriro_rccl.so: one pipelined step of a ray-sharded closest-hit query in ONE C call
// (include/triro_rccl.h; SURVEY.md 7.1 / 8(e): "ncclGroupStart; ncclRecv x (G-1) / ncclSend; ncclGroupEnd").
//
// The GPU work is libtriro_hip.so's (tr_intersects_closest, tr_intersects_closest_slots, tr_closest_from_slots); this file
// only orders it and the point-to-point transfers on two HIP streams, the way triro/ray/sharded.py does from Python --
// at a few microseconds of host time per launch instead of ~200 us per step.  RCCL is looked up at run time (the copy
// PyTorch has loaded, else librccl.so.1): no link-time dependency, no second copy of RCCL in a process.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/triro_rccl.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[TR_COMM_ID_BYTES]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclInt32 = 2 };
struct Rccl {
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
Rccl& rccl() {
    static Rccl R = [] {
        Rccl r;
        void* h = nullptr;
        // the copy that is already in the process (PyTorch links one), else the system's
        if (dlsym(RTLD_DEFAULT, "ncclSend")) h = RTLD_DEFAULT;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h) { r.why = "RCCL not found (no ncclSend in the process, librccl.so.1 not loadable)"; return r; }
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) r.why += std::string(r.why.empty() ? "missing " : ", ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.ok = r.why.empty();
        return r;
    }();
    return R;
}
int nccl_fail(const char* what, int rc) {
    const char* s = rccl().GetErrorString ? rccl().GetErrorString(rc) : "?";
    return fail(TR_ERR_HIP, std::string(what) + ": " + (s ? s : "?"));
}
#define NCCL_TRY(what, expr) do { int _rc = (expr); if (_rc != ncclSuccess) return nccl_fail(what, _rc); } while (0)
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(TR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorName(_e)); } while (0)

// rows [a, z) of the LEADING dimension of a ray descriptor ([m, 3] or [rows, w, 3]; strides in elements)
tr_rays slice_rays(const tr_rays& r, int ndim, int64_t a, int64_t z) {
    tr_rays s = r;
    const int lead = TR_MAX_SIZE_LENGTH - ndim;            // shapes / strides are right-aligned
    s.d_origins = r.d_origins + a * r.ostride[lead];
    s.d_directions = r.d_directions + a * r.dstride[lead];
    s.shape[lead] = z - a;
    int64_t inner = 1;
    for (int k = lead + 1; k < TR_MAX_SIZE_LENGTH - 1; k++) inner *= r.shape[k];
    s.nray = (z - a) * inner;
    return s;
}
// contiguous part [lo, hi) of n items for piece k of K: the Python front end's shard_bounds
void piece(int64_t n, int64_t K, int64_t k, int64_t* lo, int64_t* hi) {
    const int64_t base = n / K, rem = n % K;
    *lo = k * base + std::min(k, rem);
    *hi = *lo + base + (k < rem ? 1 : 0);
}

}  // namespace

struct tr_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
    std::vector<hipEvent_t> events;      // chunk events of the steps (reused: an event may be re-recorded once consumed)
    size_t next_event = 0;
    hipEvent_t event() {
        if (events.size() < 64) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            events.push_back(e);
            return e;
        }
        return events[next_event++ % events.size()];
    }
};

extern "C" {

const char* tr_rccl_last_error(void) { return g_err.c_str(); }

int tr_rccl_available(void) {
    if (rccl().ok) return TR_OK;
    return fail(TR_ERR_INVALID_ARG, rccl().why);
}

int tr_comm_unique_id(uint8_t id[TR_COMM_ID_BYTES]) {
    if (!id) return fail(TR_ERR_INVALID_ARG, "id == NULL");
    if (!rccl().ok) return fail(TR_ERR_INVALID_ARG, rccl().why);
    ncclUniqueId u;
    NCCL_TRY("ncclGetUniqueId", rccl().GetUniqueId(&u));
    memcpy(id, u.internal, TR_COMM_ID_BYTES);
    return TR_OK;
}

int tr_comm_create(const uint8_t id[TR_COMM_ID_BYTES], int world, int rank, int device, tr_comm** out) {
    if (!id || !out) return fail(TR_ERR_INVALID_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(TR_ERR_INVALID_ARG, "rank / world out of range");
    if (!rccl().ok) return fail(TR_ERR_INVALID_ARG, rccl().why);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId u;
    memcpy(u.internal, id, TR_COMM_ID_BYTES);
    tr_comm* c = new tr_comm();
    c->world = world; c->rank = rank; c->device = device;
    const int rc = rccl().CommInitRank(&c->comm, world, u, rank);
    if (rc != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return TR_OK;
}

int tr_comm_destroy(tr_comm* c) {
    if (!c) return TR_OK;
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    int rc = ncclSuccess;
    if (c->comm && rccl().ok) rc = rccl().CommDestroy(c->comm);
    delete c;
    return rc == ncclSuccess ? TR_OK : nccl_fail("ncclCommDestroy", rc);
}

int tr_comm_abort(tr_comm* c) {
    // From ANY thread: ends the communicator's kernels (a receive nobody answers spins on its stream for good) and
    // frees it.  The handle stays valid for tr_comm_destroy; steps on it fail from here on.
    if (!c) return TR_OK;
    ncclComm_t h = c->comm;
    c->comm = nullptr;
    if (!h || !rccl().ok) return TR_OK;
    const int rc = rccl().CommAbort(h);
    return rc == ncclSuccess ? TR_OK : nccl_fail("ncclCommAbort", rc);
}

int tr_sharded_closest_step(const tr_bvh* bvh, tr_comm* comm, const tr_shard_step* s) {
    if (!bvh || !s || !s->bounds || !s->my_rays) return fail(TR_ERR_INVALID_ARG, "null argument");
    const bool no_x = (s->flags & TR_STEP_NO_EXCHANGE) != 0, loop = (s->flags & TR_STEP_LOOPBACK) != 0;
    const int world = s->world, rank = s->rank, dst = s->dst;
    if (world < 1 || rank < 0 || rank >= world || dst < 0 || dst >= world || s->chunks < 1) return fail(TR_ERR_INVALID_ARG, "rank / world / dst / chunks out of range");
    const bool exchange = world > 1 && !no_x;
    if (exchange && (!comm || !rccl().ok)) return fail(TR_ERR_INVALID_ARG, comm ? rccl().why : "comm == NULL");
    if (exchange && !comm->comm) return fail(TR_ERR_INVALID_ARG, "the communicator was aborted");
    if (exchange && !loop && (comm->world != world || comm->rank != rank)) return fail(TR_ERR_INVALID_ARG, "step.world / rank differ from the communicator's");
    if (loop && (rank != dst || !s->d_staging)) return fail(TR_ERR_INVALID_ARG, "LOOPBACK: rank must be dst and d_staging set");
    const int64_t q = s->per_row > 1 ? s->per_row : 1;
    const int ndim = q > 1 ? 3 : 2;
    for (int r = 0; r < world; r++) {
        const int64_t a = s->bounds[2 * r], z = s->bounds[2 * r + 1];
        if (a < 0 || z < a || z > s->n_total || a % q || z % q) return fail(TR_ERR_INVALID_ARG, "bounds must be ordered multiples of per_row within the batch");
    }
    const bool want = rank == dst;
    hipStream_t cur = (hipStream_t)s->stream, side = (hipStream_t)s->side_stream;
    if (want && (!s->all_rays || !s->d_records || !s->d_hit || !s->d_front || !s->d_tri || !s->d_loc3 || !s->d_uv2 || !s->done_event))
        return fail(TR_ERR_INVALID_ARG, "the destination needs all_rays, d_records, the five outputs and done_event");
    if (!want && !s->d_records && s->bounds[2 * rank + 1] > s->bounds[2 * rank]) return fail(TR_ERR_INVALID_ARG, "d_records == NULL");
    // every rank cuts its shard into the SAME number of chunks: bounded by the smallest non-empty shard
    int64_t K = s->chunks;
    for (int r = 0; r < world; r++) {
        const int64_t rows = (s->bounds[2 * r + 1] - s->bounds[2 * r]) / q;
        if (rows > 0) K = std::min(K, rows);
    }
    K = std::max<int64_t>(K, 1);
    const int64_t lo = s->bounds[2 * rank];
    int deferred = TR_OK;
    std::string deferred_msg;
    auto note = [&](int rc, const char* what) {
        if (rc != TR_OK && deferred == TR_OK) { deferred = rc; deferred_msg = std::string(what) + ": " + tr_last_error(); }
        return rc;
    };
    if (want && side) {
        // the side stream starts behind what the caller's stream holds (the outputs were allocated there)
        hipEvent_t e0 = comm ? comm->event() : nullptr;
        hipEvent_t tmp = nullptr;
        if (!e0) { HIP_TRY(hipEventCreateWithFlags(&tmp, hipEventDisableTiming)); e0 = tmp; }
        HIP_TRY(hipEventRecord(e0, cur));
        HIP_TRY(hipStreamWaitEvent(side, e0, 0));
        if (tmp) (void)hipEventDestroy(tmp);
    }
    hipStream_t xs = (want && side) ? side : cur;         // stream of the receives and of the peers' rows
    for (int64_t k = 0; k < K; k++) {
        // chunk k of every rank (every rank can compute everybody's)
        std::vector<int64_t> ca(world), cz(world);
        for (int r = 0; r < world; r++) {
            const int64_t rlo = s->bounds[2 * r], rows = (s->bounds[2 * r + 1] - rlo) / q;
            int64_t a, z;
            piece(rows, K, k, &a, &z);
            ca[r] = rlo + a * q; cz[r] = rlo + z * q;
        }
        const int64_t a = ca[rank] - lo, z = cz[rank] - lo;      // within this rank's shard
        if (want) {
            // ---- the destination: its own chunk dense, straight into its rows of the outputs
            if (z > a) {
                const tr_rays mine = slice_rays(*s->my_rays, ndim, a / q, z / q);
                note(tr_intersects_closest(bvh, &mine, s->d_hit + ca[rank], s->d_front + ca[rank], s->d_tri + ca[rank],
                                           s->d_loc3 + 3 * ca[rank], s->d_uv2 + 2 * ca[rank], cur), "trace");
            }
            if (exchange && !loop) {
                // receives start behind this chunk's trace: the peers' chunk k is ready about when ours is, and a receive
                // kernel posted earlier would spin on CUs through the whole trace
                if (side) {
                    hipEvent_t ek = comm->event();
                    if (!ek) return fail(TR_ERR_HIP, "hipEventCreate");
                    HIP_TRY(hipEventRecord(ek, cur));
                    HIP_TRY(hipStreamWaitEvent(side, ek, 0));
                }
                NCCL_TRY("ncclGroupStart", rccl().GroupStart());
                for (int r = 0; r < world; r++)
                    if (r != rank && cz[r] > ca[r])
                        NCCL_TRY("ncclRecv", rccl().Recv(s->d_records + ca[r], (size_t)(cz[r] - ca[r]), ncclInt32, r, comm->comm, xs));
                NCCL_TRY("ncclGroupEnd", rccl().GroupEnd());
            } else if (exchange && loop) {
                // ONE rank plays everybody: a peer's chunk is traced into the staging buffer and sent to ourselves
                for (int r = 0; r < world; r++) {
                    if (r == rank || cz[r] <= ca[r]) continue;
                    const tr_rays theirs = slice_rays(*s->all_rays, ndim, ca[r] / q, cz[r] / q);
                    note(tr_intersects_closest_slots(bvh, &theirs, s->d_staging, xs), "trace (loopback)");
                    NCCL_TRY("ncclGroupStart", rccl().GroupStart());
                    if (!(s->flags & TR_STEP_TEST_DROP_SEND))
                        NCCL_TRY("ncclSend", rccl().Send(s->d_staging, (size_t)(cz[r] - ca[r]), ncclInt32, 0, comm->comm, xs));
                    NCCL_TRY("ncclRecv", rccl().Recv(s->d_records + ca[r], (size_t)(cz[r] - ca[r]), ncclInt32, 0, comm->comm, xs));
                    NCCL_TRY("ncclGroupEnd", rccl().GroupEnd());
                }
            }
            // ---- the peers' rows of chunk k: ray + slot -> the five outputs (adjacent ranges: one launch)
            int64_t ra = -1, rz = -1;
            auto flush = [&]() {
                if (rz > ra && ra >= 0) {
                    const tr_rays rows = slice_rays(*s->all_rays, ndim, ra / q, rz / q);
                    note(tr_closest_from_slots(bvh, &rows, s->d_records + ra, q > 1 ? q : 0, s->d_hit + ra, s->d_front + ra,
                                               s->d_tri + ra, s->d_loc3 + 3 * ra, s->d_uv2 + 2 * ra, xs), "finish");
                }
                ra = rz = -1;
            };
            for (int r = 0; r < world; r++) {
                if (r == rank || cz[r] <= ca[r]) continue;
                if (rz == ca[r]) rz = cz[r];
                else { flush(); ra = ca[r]; rz = cz[r]; }
            }
            flush();
        } else if (z > a) {
            // ---- a peer: chunk k as 4-byte records, then on its way (stream-ordered behind the trace)
            const tr_rays mine = slice_rays(*s->my_rays, ndim, a / q, z / q);
            if (note(tr_intersects_closest_slots(bvh, &mine, s->d_records + a, cur), "trace") != TR_OK)
                (void)hipMemsetAsync(s->d_records + a, 0xff, (size_t)(z - a) * 4, cur);      // records that say "miss"
            if (exchange) {
                // with a side stream the send leaves the caller's stream free for the next chunk's / step's trace (the
                // caller keeps d_records alive and untouched until the side stream has passed the send)
                hipStream_t ss = cur;
                if (side) {
                    hipEvent_t ek = comm->event();
                    if (!ek) return fail(TR_ERR_HIP, "hipEventCreate");
                    HIP_TRY(hipEventRecord(ek, cur));
                    HIP_TRY(hipStreamWaitEvent(side, ek, 0));
                    ss = side;
                }
                NCCL_TRY("ncclSend", rccl().Send(s->d_records + a, (size_t)(z - a), ncclInt32, dst, comm->comm, ss));
            }
        }
    }
    if (want) HIP_TRY(hipEventRecord((hipEvent_t)s->done_event, xs));
    if (deferred != TR_OK) return fail(deferred, deferred_msg);
    return TR_OK;
}

}  // extern "C"
