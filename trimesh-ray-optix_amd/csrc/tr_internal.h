// tr_internal.h -- host-side shared declarations of libtriro_hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>

#include "../../include/triro_hip.h"
#include "tr_bvh.h"
#include "tr_wide.h"

// error plumbing ----------------------------------------------------------------------
void tr_set_error(const std::string& msg);
int tr_fail(int code, const std::string& msg);

#define TR_HIP_TRY(expr)                                                                  \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            return tr_fail(_e == hipErrorOutOfMemory ? TR_ERR_OUT_OF_MEMORY : TR_ERR_HIP, \
                           std::string(#expr) + ": " + hipGetErrorName(_e) + " (" +       \
                               hipGetErrorString(_e) + ")");                              \
        }                                                                                 \
    } while (0)

#define TR_TRY(expr)            \
    do {                        \
        int _s = (expr);        \
        if (_s != TR_OK) return _s; \
    } while (0)

// adaptive launch order: per (handle, stream) measured block costs and the order derived from
// them.  buf = cost[TR_SCHED_MAX] | order 0 + its stamp | launch scratch | costs of the last sort | order 1 + its stamp
// (launch_policy.inc)
constexpr int TR_SCHED_MAX = 131072;  // blocks (x256 rays) up to which the order is learned
constexpr int TR_SCHED_SLOTS = 16;   // (stream, class) pairs per handle: 8 streams x {plain, split} orders
struct tr_sched_slot {
    hipStream_t stream = nullptr;
    int cls = 0;            // 1: orders with split blocks (launch shapes that steal), 0: plain
    uint32_t* buf = nullptr;
    int64_t nblocks = 0;    // block count the current order was measured for (0 = none)
    int64_t split = 0;      // ... and the number of split blocks per XCD the order was written with
    int lgh = 0;            // ... and the rows-per-tile exponent of its block -> ray map
    int64_t launches = 0;   // launches with this block count so far
    // shape of the launch whose costs the last k_sched_sort kept (k_sched_rescale lends them to the next batch shape)
    bool prev_valid = false;
    int64_t prev_nblocks = 0, prev_w = 0, prev_h = 0;
    int prev_lgh = 0;
    bool used = false;
    // Two order buffers: launches read order_buf(cur) while a sort writes the other one.  A sort that was DEFERRED
    // (`pending`: the costs of the last measuring launch are still unsorted) rides in the next launch of the same shape
    // as an extra workgroup (tr_sort_job, query_direct_body) or, if that launch cannot carry it, runs as k_sched_sort
    // before it; p_* = what that sort needs.
    int cur = 0;
    bool pending = false;
    int p_nblocks = 0, p_xc = 0, p_split = 0, p_split4 = 0, p_outlier8 = 0, p_floor = 0;
    // per-lane stack overflow rows of the wide streaming launch on this stream (k_query_wide; grown on demand)
    int32_t* wspill = nullptr;
    size_t wspill_elems = 0;
};

// the opaque handle ---------------------------------------------------------------------
struct tr_bvh {
    int device = 0;
    int64_t num_tris = 0;
    int64_t num_nodes = 0;
    int32_t depth = 0;
    int32_t key_mode = 0;
    void* arena = nullptr;      // one hipMalloc: nodes | links | tris | qnodes
    int64_t arena_bytes = 0;
    int64_t capacity_tris = 0;  // arena was sized for this many triangles
    tr_node* nodes = nullptr;
    tr_link* links = nullptr;
    tr_tri* tris = nullptr;
    tr_qnode* qnodes = nullptr;   // 32-byte grid nodes of the unordered schedule (same arena)
    tr_qframe frame = {{0, 0, 0}, {1, 1, 1}};   // their grid: a function of the bounds below
    tr_qframe* frame_dev = nullptr;             // ... and its copy in device memory, which the kernels READ (tr_view_live): a launch
                                                // captured in a HIP graph must see the frame of a later refit, not the one its arguments froze
    float aabb_min[3] = {0, 0, 0};
    float aabb_max[3] = {0, 0, 0};
    void* refit_temp = nullptr;   // boxes + flags of tr_bvh_refit, kept between calls (animation loops)
    size_t refit_temp_bytes = 0;
    // adaptive launch order (speed only, see traverse.hip).  One slot per stream that has
    // queried this handle, so launches on different streams never share hint buffers.
    tr_sched_slot sched[TR_SCHED_SLOTS];
    std::mutex* sched_mutex = nullptr;
    // 8-wide compressed nodes of the streaming launch (tr_wide.h): derived data, built by the first streaming
    // query after a build / refit / load on that query's stream (under sched_mutex); other streams wait for
    // wide_event
    tr_wnode* wnodes = nullptr;
    int64_t wcap = 0, wcount = 0;
    uint8_t* wflag = nullptr;     // per binary node: round in which it was found to be the root of a wide node, 0 = none
    int64_t* widx = nullptr;      // exclusive scan of the flags (+ one word for the total)
    int64_t wtmp_cap = 0;
    bool wide_valid = false;
    bool wide_unavailable = false;     // the last attempt to build them for THIS hierarchy failed (too large, out of memory): not retried until the next build / refit / load
    hipEvent_t wide_event = nullptr;
    hipStream_t wide_stream = nullptr;
    tr_launch_info last_launch = {};     // tr_bvh_last_launch (written under sched_mutex)
    bool have_last_launch = false;
};

// bvh->frame -> bvh->frame_dev (a small synchronous copy; build / refit / load have just synchronised anyway)
int tr_bvh_sync_frame(tr_bvh* bvh);

// per-device runtime state (tr_init) ------------------------------------------------------
struct tr_device_state {
    bool ready = false;
    int device = 0;
    int num_cus = 0;
    // topology, read from the device (round 6; until then 8 XCDs and 24 resident waves per CU were constants of the MI355X in SPX mode)
    int num_xcd = 8;                // hipDeviceAttributeNumberOfXccs (8 when the query fails)
    int64_t l2_bytes = 0;           // hipDeviceProp_t::l2CacheSize as reported (0 = unknown)
    int waves_per_cu = 0;           // resident waves of the stealing closest launch on one CU, from the occupancy query (tr_policy_units, launch_policy.inc); 0 = not asked yet
    int* counters = nullptr;        // ring of scratch words (coherence probe, work counter of the streaming launch) for handles without a free scheduling slot
    unsigned next_counter = 0;
    // builder temporaries (sort buffers, boxes, hierarchy), kept between builds so that a
    // rebuild (`update_raw`) costs no hipMalloc/hipFree.  One build per device at a time.
    void* build_temp = nullptr;
    size_t build_temp_bytes = 0;
    std::mutex build_mutex;
    // side stream of the builder (under build_mutex): the node-layout rounds (top-down) run beside the
    // refit rounds (bottom-up) -- both depend only on the Karras hierarchy, the emit joins them
    hipStream_t build_side = nullptr;
    hipEvent_t build_fork = nullptr, build_join = nullptr;
};
constexpr int TR_NUM_COUNTERS = 4096;

int tr_get_device_state(int device, tr_device_state** out);
// returns with st->build_mutex HELD on success; tr_build_temp_release unlocks (and frees the
// buffer when option build_cache == 0).  The caller must have drained its stream by then.
int tr_build_temp_acquire(tr_device_state* st, size_t bytes, void** out);
int tr_build_temp_release(tr_device_state* st);

// builder (bvh_build.hip) -----------------------------------------------------------------
int tr_build_impl(tr_bvh* bvh, const float* d_vertices, int64_t nv, const int32_t* d_faces,
                  int64_t nf, hipStream_t stream);

int tr_arena_alloc(tr_bvh* bvh, int64_t nf);
// bytes of the arena that `nf` triangles actually use (<= arena_bytes, the capacity)
int64_t tr_arena_used_bytes(int64_t nf);
// forget the current hierarchy: queries on the handle return misses (used when a build or
// refit fails half way, so that nothing ever traverses a half-written arena)
void tr_bvh_reset(tr_bvh* bvh);
int tr_refit_impl(tr_bvh* bvh, const float* d_vertices, int64_t nv, const int32_t* d_faces,
                  int64_t nf, hipStream_t stream);
// the 8-wide nodes of a handle that already has them (bvh->wnodes != NULL) again, now, on `stream` (after update / refit);
// a handle that never walked them builds them on first use (traverse.hip: ensure_wide)
void tr_wide_rebuild(tr_bvh* bvh, hipStream_t stream);

// Makes `device` current for the lifetime of the guard (every entry point that launches on or
// copies from a handle's arena runs under one: a C caller may be on another current device).
struct tr_device_guard {
    int prev = -1;
    bool changed = false;
    int enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) return TR_ERR_NO_DEVICE;
        if (prev != device) {
            if (hipSetDevice(device) != hipSuccess) return TR_ERR_NO_DEVICE;
            changed = true;
        }
        return TR_OK;
    }
    ~tr_device_guard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

// options ---------------------------------------------------------------------------------
// Process-wide tuning knobs (tr_set_option).  Stored as relaxed atomics; every entry point takes
// ONE snapshot (tr_opts()) and works from that copy, so a concurrent tr_set_option can never be
// seen half-way through a launch decision.
struct tr_options {
    int adaptive = 1;     // start the blocks that were most expensive in the previous launch first
    int compact = 1;      // allow the 32-bit trail / 32-bit offset kernels when the BVH permits
    int xcd_chunk = 128;    // direct kernel: blocks per XCD-local chunk (0 = identity map)
    int steal = 1;        // intra-wave work stealing: 0 off, 1 auto (closest/first/any up to 4 M rays), >= 2 forced with that trip threshold
    int tile = 1;         // image-shaped batches: waves take 8x8 pixel tiles (0 never, 1 from 4 M rays on, 2 always)
    int tile_small = 4;   // image-shaped batches below the `tile` threshold: 0 rows of 64 pixels, 1 = 2x32, 2 = 4x16, 3 = 8x8 tiles, 4 = by triangles per ray
    int node_layout = 1;  // order of the traversal nodes in memory (build time): 0 Karras numbering, 1 treelets of 3 levels, depth first
    int build_cache = 1;  // keep the builder's temporaries (about 130 B/triangle) per device between builds
    int stream = 1;       // streaming launch with wave-level ray refill: 0 never, 1 large non-image batches, 2 always
    int stream_rays = 256;    // rays per range of the streaming launch (512 was the optimum of the static map)
    int stream_refill = 0;    // idle lanes that trigger a refill (0 = by query: 28 closest / first, 20 any / count)
    int stream_dynamic = 1;   // ranges handed out by a work counter to a resident-sized grid (0: one static range per wave)
    int grid_nodes = 1;   // stealing closest / first / any launches on the 32-byte grid nodes: 0 never (the exact 64-byte nodes), 1 / 2 yes
    int split = 1;        // block splitting: 0 off, 1 auto, N >= 2: the nblocks >> N most expensive blocks of the previous launch get two launch slots
    int split_steal = 8;  // ... and give subtrees away from this trip on
    int split_outlier = 1;    // ... but only blocks that cost at least N eighths of the mean block cost (0: all of them, 1: N by how full the chip is)
    int split_floor = 40;     // ... and at least this many microseconds (device clock) per wave
    int leaf_vote = 32;   // unordered schedule: lanes with a queued leaf that fire a leaf phase
    int order_transfer = 1;   // a batch of a new image shape starts from the previous shape's block costs, resampled (0: from the static order)
    int sort_inline = 1;      // the steady-state sort of the block costs rides in the next launch as a workgroup (0: a kernel behind every measuring launch)
    int wide = 2;         // the streaming launch walks 8-wide compressed nodes (tr_wide.h; built on first use): 0 never, 1 always, 2 where measured faster (from 1 M triangles on)
    int wide_direct = 1;  // the DIRECT launch on the 8-wide nodes (k_query_direct_wide): 0 never, 1 location launches on meshes >= 500 k triangles (where measured faster), 2 count and location, 3 every query
    int wide_stack = 12;  // ... entries of a lane's node stack kept in LDS (<= 12; the rest lives in a global spill row; tests lower it)
    int expand_cus = 0;   // tr_closest_expand: at most this many workgroups per CU, grid-stride beyond (0: one workgroup per 1024 rays)
    int expand_tiles = 1; // tr_closest_expand_slots_rows: 8x8 pixel tiles per wave on image-shaped rows (0: rows of 256 pixels)
    int usteal = 1;       // unordered count launches hand owed subtrees over between lanes and use split launch slots: 0 off, 1 on, >= 2 forced trip threshold
};
tr_options tr_opts();   // snapshot by value
