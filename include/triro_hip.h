/*
 * triro_hip.h -- C ABI of libtriro_hip.so: the MI355X (gfx950) ray / triangle-mesh
 * intersection path behind triro.ray.ray_optix.RayMeshIntersector.
 *
 * This is the drop-in boundary.  Each entry point replaces one function of the reference's
 * pybind11 module "triro" (triro/backend/binding.cpp:31-59); plain pointers and sizes only,
 * no torch types.  All pointers named d_* are DEVICE pointers on the device the BVH was
 * built on; every call enqueues work on `stream` (a hipStream_t passed as void*, NULL =
 * the null stream) and returns without synchronising unless stated otherwise.  Outputs are
 * allocated by the caller (the reference allocates them inside the extension with
 * torch::empty, ray.cpp:239-259).
 *
 * Return value: 0 (TR_OK) or a tr_status code; tr_last_error() gives a thread-local
 * message.  Nothing here ever calls exit() (the reference does: optix8.h:41-60).
 */
#ifndef TRIRO_HIP_H
#define TRIRO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TR_ABI_VERSION 10
#define TR_MAX_ANYHIT_SIZE 8 /* LaunchParams.h:8  (per-ray cap of intersects_location) */
#define TR_MAX_SIZE_LENGTH 4 /* LaunchParams.h:9  (ray tensors have <= 4 dims)         */
#define TR_MAX_HITS_CAP 32   /* largest `cap` tr_intersects_location_fill accepts      */

typedef enum tr_status {
    TR_OK = 0,
    TR_ERR_INVALID_ARG = 1,
    TR_ERR_HIP = 2,           /* a HIP runtime call failed; message has the hipError name */
    TR_ERR_NO_DEVICE = 3,
    TR_ERR_OUT_OF_MEMORY = 4,
    TR_ERR_INTERNAL = 5
} tr_status;

/* Opaque acceleration structure (replaces OptixAccelStructureWrapperCPP, ray.h:11-16). */
typedef struct tr_bvh tr_bvh;

/*
 * Ray batch = the reference's RayInput (LaunchParams.h:11-28) as filled by ray.cpp:151-159,
 * 177-179: shape and strides RIGHT-ALIGNED in 4 slots; unused leading shape slots hold
 * INT64_MAX, unused leading stride slots 0; strides in ELEMENTS (floats); the last dim is
 * 3 (x,y,z, element stride = stride[3]).  origins and directions share `shape`
 * (= origins.sizes()) and have independent strides (stride 0 = broadcast is legal).
 * nray = product of the leading dims.
 */
typedef struct tr_rays {
    const float *d_origins;
    const float *d_directions;
    int64_t nray;
    int64_t shape[TR_MAX_SIZE_LENGTH];
    int64_t ostride[TR_MAX_SIZE_LENGTH];
    int64_t dstride[TR_MAX_SIZE_LENGTH];
} tr_rays;

typedef struct tr_bvh_info {
    int32_t device;        /* HIP device ordinal the arena lives on                 */
    int64_t num_tris;      /* F                                                       */
    int64_t num_nodes;     /* internal nodes (F-1 for F >= 2, else 0)               */
    int32_t depth;         /* height of the hierarchy in internal-node levels       */
    int32_t key_mode;      /* 0: 63-bit Morton keys, 1: depth-bounded fallback keys */
    int64_t arena_bytes;   /* device memory owned by the handle (nodes, links, tris, grid nodes) */
    int64_t node_bytes;    /* bytes of the exact traversal nodes (64 B each)        */
    int64_t tri_bytes;     /* bytes of leaf triangle records (48 B each)            */
    float aabb_min[3];     /* mesh bounds                                            */
    float aabb_max[3];
} tr_bvh_info;

/* Per-launch traversal statistics (diagnostic build of the kernels; tr_trace_stats). */
typedef struct tr_trace_stats {
    uint64_t rays;
    uint64_t node_visits;   /* internal nodes fetched (64 B each)   */
    uint64_t tri_tests;     /* leaf triangles fetched (48 B each)   */
    uint64_t climb_steps;   /* parent-link loads while backtracking */
} tr_trace_stats;

/* How the last direct-launch query of a handle was shaped (diagnostics; none of it changes results). */
typedef struct tr_launch_info {
    int64_t rays;           /* rays of the batch                                              */
    int64_t blocks;         /* ray blocks of 128 rays                                         */
    int64_t slots;          /* launch slots = blocks + extra slots of split blocks            */
    int32_t query;          /* TR_Q_* of the launch                                           */
    int32_t shape;          /* 0 plain, 1 stealing, 2 unordered two-phase schedule, 3 unordered + stealing, 4 direct launch on the 8-wide nodes */
    int32_t tile_rows_lg;   /* 0: 64 rays of one row per wave, 1: 2x32, 2: 4x16, 3: 8x8 tiles  */
    int32_t split_blocks;   /* blocks per XCD that were dealt to 2 / 4 launch slots           */
    int32_t learned_order;  /* 1: the launch used a learned launch order                      */
    int32_t grid_nodes;     /* 1: the launch walked the 32-byte grid nodes, 0: the exact ones */
    int32_t addressing;     /* 0: 64-bit, 1: 32-bit offsets + 32-bit trail, 2: 32-bit offsets + 64-bit trail */
    int32_t sort_carried;   /* 1: the launch carried the deferred sort of the last measured block costs (option sort_inline) */
} tr_launch_info;

/* -- runtime bring-up: replaces initOptix/createOptixContext/createOptixModule/
 *    createOptixPipelines/buildSBT (base.cpp:31-157, binding.cpp:41-48).  Idempotent,
 *    thread-safe; device < 0 selects the current HIP device.                             */
int tr_init(int device);
int tr_abi_version(void);
const char *tr_last_error(void);

/* -- acceleration structure: replaces buildAccelStructure / freeAccelStructure
 *    (ray.cpp:27-102, binding.cpp:35-38).  d_vertices: [nv,3] float32 dense, d_faces:
 *    [nf,3] int32 dense.  The handle keeps its own copy of the triangle data (as the GAS
 *    does with ALLOW_RANDOM_VERTEX_ACCESS, ray.cpp:37), so the inputs may be freed after
 *    the call returns.  Synchronises `stream` once (tree height read-back).              */
int tr_bvh_build(const float *d_vertices, int64_t nv, const int32_t *d_faces, int64_t nf,
                 void *stream, tr_bvh **out);
/*    rebuild in place for RayMeshIntersector.update_raw (ray_optix.py:55-69)             */
int tr_bvh_update(tr_bvh *bvh, const float *d_vertices, int64_t nv, const int32_t *d_faces,
                  int64_t nf, void *stream);
/*    refit: same faces (topology), new vertex positions -- keeps the hierarchy and recomputes
 *    every box (cheaper than a rebuild for deforming meshes; the reference always rebuilds,
 *    ray_optix.py:55-69).  d_faces must be the array the BVH was built from.               */
int tr_bvh_refit(tr_bvh *bvh, const float *d_vertices, int64_t nv, const int32_t *d_faces,
                 int64_t nf, void *stream);
int tr_bvh_destroy(tr_bvh *bvh);
/*    (de)serialisation to a HOST buffer: header + the traversal arena (the reference rebuilds
 *    its GAS in every process).  Both calls synchronise `stream`.                          */
int64_t tr_bvh_serialized_size(const tr_bvh *bvh);
int tr_bvh_serialize(const tr_bvh *bvh, void *h_buffer, int64_t size, void *stream);
int tr_bvh_deserialize(const void *h_buffer, int64_t size, void *stream, tr_bvh **out);
int tr_bvh_get_info(const tr_bvh *bvh, tr_bvh_info *info);
/*    diagnostics: the shape of the last query of this handle that took the direct launch (a batch the
 *    coherence probe handed to the streaming launch still reports the direct shape it enqueued).
 *    TR_ERR_INVALID_ARG before the first such query.                                          */
int tr_bvh_last_launch(const tr_bvh *bvh, tr_launch_info *info);

/* What the launch policy knows about the device, and the ray-count boundaries it derives from it (ABI 10; no counterpart in
 * the reference).  The boundaries between the launch shapes are multiples of the device's RESIDENT LANES -- CUs x resident
 * waves per CU of the stealing closest launch (from the occupancy calculator) x 64 --: 32/3 rounds for stealing / tiles,
 * 64/3 for the 8-wide nodes, 128/3 for streaming counts.  MI355X, SPX: 256 CUs, 8 XCDs, 24 waves -> 393 216 lanes ->
 * 4 194 304 / 8 388 608 / 16 777 216 rays, the values they were measured at. */
typedef struct tr_topology {
    int32_t num_cus, num_xcd, waves_per_cu, reserved;
    int64_t l2_bytes;               /* hipDeviceProp_t::l2CacheSize as reported (0: unknown) */
    int64_t resident_lanes;
    int64_t steal_max_rays, wide_min_rays, count_stream_min_rays;
} tr_topology;
int tr_device_topology(int device, tr_topology *out);
/*    test hook: copy the traversal arrays to HOST buffers (any may be NULL).
 *    nodes: num_nodes*16 words (64 B), links: num_nodes*2 int32, tris: num_tris*12 words. */
int tr_bvh_download(const tr_bvh *bvh, void *h_nodes, void *h_links, void *h_tris,
                    void *stream);
/*    test hook: the second node array (num_nodes*8 words: 32-byte nodes with both child boxes on
 *    a 16-bit grid, used by the unordered count / location schedule) and its grid
 *    (h_frame6 = base[3], scale[3]; plane q of axis k lies at fma(q, scale[k], base[k])).   */
int tr_bvh_download_qnodes(const tr_bvh *bvh, void *h_qnodes, float *h_frame6, void *stream);

/* -- queries: replace intersectsAny/First/Closest/Count/Location (ray.cpp:161-378,
 *    binding.cpp:49-58).  Output element i belongs to flat ray index i.
 *    any     : hit[i] = 1/0                                   (shaders.cu:67-89)
 *    first   : tri[i] = closest triangle or -1                (shaders.cu:93-116)
 *    closest : hit, front (uint8 0/1), tri (int32, -1 on miss), loc [n,3], uv [n,2];
 *              miss values 0                                  (shaders.cu:120-172)
 *    count   : number of triangles hit in [0, 1e7], uncapped  (shaders.cu:176-194)      */
int tr_intersects_any(const tr_bvh *bvh, const tr_rays *rays, uint8_t *d_hit, void *stream);
int tr_intersects_first(const tr_bvh *bvh, const tr_rays *rays, int32_t *d_tri, void *stream);
int tr_intersects_closest(const tr_bvh *bvh, const tr_rays *rays, uint8_t *d_hit,
                          uint8_t *d_front, int32_t *d_tri, float *d_loc, float *d_uv,
                          void *stream);
int tr_intersects_count(const tr_bvh *bvh, const tr_rays *rays, int32_t *d_count, void *stream);

/* -- closest hit in 12 bytes per ray (ABI 6; no counterpart in the reference, which is single-GPU:
 *    base.cpp:15-17).  A ray-sharded run gathers results over xGMI; the five dense outputs of
 *    intersectsClosest (ray.cpp:231-289) are 26 B/ray, but everything in them is a function of
 *    (triangle, front flag, w0, w1) and the mesh:  tri = face index | front << 30, 0xffffffff on a miss
 *    (face indices must be below 2^30); w0, w1 = barycentric weights of face vertices 0 and 1 = the uv the
 *    reference returns ((1-u-v, u) of OptiX' barycentrics, shaders.cu:149) -- since ABI 10 (round 6: both are
 *    float32 roundings of float64 quotients; until ABI 9 the record held the weights of vertices 1 and 2).
 *    A miss: any record with bit 31 of tri set (the kernels write 0xffffffff, 0, 0); its w0, w1 are ignored.
 *    tr_closest_expand turns such records back into hit / front / tri / loc / uv with the operations
 *    tr_intersects_closest itself uses (w2 = (1-w0)-w1, loc = w0*V0 + w1*V1 + w2*V2, uv = (w0, w1):
 *    shaders.cu:143-149) on d_vertices [nv,3] / d_faces [nf,3] -- the arrays the BVH was built
 *    from -- so expand(packed) is bit-identical to the dense outputs.  Any output may be NULL.
 *    Works on whatever device the pointers live on (no BVH handle involved).                    */
typedef struct tr_packed_hit {
    uint32_t tri;
    float w0, w1;
} tr_packed_hit;
int tr_intersects_closest_packed(const tr_bvh *bvh, const tr_rays *rays, tr_packed_hit *d_packed,
                                 void *stream);
int tr_closest_expand(const tr_packed_hit *d_packed, int64_t n, const float *d_vertices, int64_t nv,
                      const int32_t *d_faces, int64_t nf, uint8_t *d_hit, uint8_t *d_front,
                      int32_t *d_tri, float *d_loc, float *d_uv, void *stream);

/* ... the same with the ARENA SLOT of the hit triangle in place of the face index (ABI 7): the destination rank of a
 *    sharded run then reads ONE 48-byte triangle record per hit (vertices + face index, Morton-ordered: rays that hit
 *    neighbouring triangles read neighbouring records) instead of a face row and three vertex rows in four different
 *    cache lines -- the expansion is bound by the lines its gathers pull through the fabric (profiles/
 *    r04_expand_micro.jsonl).  Slots are only meaningful for the hierarchy that produced them or a bit-identical
 *    replica of it: same mesh, same build options (the builder is deterministic: DESIGN.md 4.1).  Same outputs, bit
 *    for bit, as tr_closest_expand on the face form.                                                        */
int tr_intersects_closest_packed_slots(const tr_bvh *bvh, const tr_rays *rays, tr_packed_hit *d_packed,
                                       void *stream);
int tr_closest_expand_slots(const tr_bvh *bvh, const tr_packed_hit *d_packed, int64_t n, uint8_t *d_hit,
                            uint8_t *d_front, int32_t *d_tri, float *d_loc3, float *d_uv2, void *stream);
/* ... for records that are ROWS OF AN IMAGE (row_length pixels each, a multiple of 32; a multiple of 8 rows, or any
 *    number from 64 rows on): a wave expands blocks of 8 rows x 32 pixels, so that the neighbouring rays that hit the
 *    same triangle read its record once.  Any other shape (row_length 0 included) takes the linear kernel.  Same outputs. */
int tr_closest_expand_slots_rows(const tr_bvh *bvh, const tr_packed_hit *d_packed, int64_t n, int64_t row_length,
                                 uint8_t *d_hit, uint8_t *d_front, int32_t *d_tri, float *d_loc3, float *d_uv2,
                                 void *stream);

/* ... and the 4-BYTE record (ABI 8), for a destination that HOLDS THE RAYS (the reference's call hands the whole batch
 *    to one process: ray_optix.py:121-146; a sharded front end sees it on every rank): the traversal writes only the
 *    arena slot of the nearest triangle, -1 for a miss (tr_intersects_closest_slots: d_slot[nray] int32), and
 *    tr_closest_from_slots finishes the query from (ray, slot) exactly as the dense call does at its end -- (det, U, V)
 *    of the ray against that one triangle, then the barycentric outputs (shaders.cu:139-151) -- so hit / front / tri /
 *    loc / uv are the dense call's bits.  `rays` describes the rays the slots belong to, in the same order (any
 *    strides, nray = number of slots); row_length as for tr_closest_expand_slots_rows (0 = no image rows).  Same
 *    replica rule as the slot form above.  4 B per ray cross the links instead of 12.                      */
int tr_intersects_closest_slots(const tr_bvh *bvh, const tr_rays *rays, int32_t *d_slot, void *stream);
int tr_closest_from_slots(const tr_bvh *bvh, const tr_rays *rays, const int32_t *d_slot, int64_t row_length,
                          uint8_t *d_hit, uint8_t *d_front, int32_t *d_tri, float *d_loc3, float *d_uv2, void *stream);

/* ... and what makes "a bit-identical replica" checkable (ABI 9): an exact 64-bit hash of the triangle arena -- for every
 *    slot its position and the twelve words of its record (three vertices, face index) -- plus the triangle count.  Two
 *    handles with the same hash name the same triangles by the same slots and hold the same vertices for them (up to a
 *    2^-64 collision), so slot-form records of one can be finished on the other; a sharded front end compares the ranks'
 *    hashes before it lets such records travel (triro/ray/sharded.py).  One pass over the arena (63 MB at 1.31 M
 *    triangles), synchronises `stream` (refused with TR_ERR_INVALID_ARG on a stream that is being captured).  The hash
 *    covers the slot -> triangle identity ONLY -- what slot-form records need --, not the node arrays, the key mode or the
 *    node layout.  The reference has no counterpart (single GPU: base.cpp:15-17).        */
int tr_bvh_replica_hash(const tr_bvh *bvh, uint64_t *h_hash, void *stream);

/* -- multi-hit (intersectsLocation, ray.cpp:324-378):
 *    tr_hits_scan replaces the torch glue of ray.cpp:333-342: d_offsets[i] = exclusive
 *    prefix sum of min(d_count[i], cap); the grand total is written to *d_total (device)
 *    and, if h_total != NULL, copied to the host after synchronising `stream` (the
 *    reference's .item<int>(), ray.cpp:339).  d_offsets has n elements (int64).
 *    tr_intersects_location_fill is the second pass (shaders.cu:196-246): for ray i it
 *    writes its min(count,cap) nearest hits, ordered by (distance, triangle index), at
 *    rows d_offsets[i].. of loc [h,3], ray_idx [h] (= i + ray_base), tri_idx [h].        */
int tr_hits_scan(const int32_t *d_count, int64_t n, int32_t cap, int64_t *d_offsets,
                 int64_t *d_total, int64_t *h_total, void *stream);
int tr_intersects_location_fill(const tr_bvh *bvh, const tr_rays *rays, int32_t cap,
                                const int64_t *d_offsets, float *d_loc, int32_t *d_ray_idx,
                                int32_t *d_tri_idx, int64_t ray_base, void *stream);

/*    Fused variant (one traversal instead of two): tr_intersects_count_topk writes the uncapped
 *    count AND, for each ray, its `cap` nearest hits as UNSORTED entries {t_key, arena slot}
 *    (d_hits: [n, cap] tr_hit_entry = 8 bytes each; the first min(count, cap) of a ray are
 *    valid); after tr_hits_scan, tr_location_fill_slots ranks them by (distance, triangle
 *    index) and produces exactly what tr_intersects_location_fill produces, without
 *    traversing again.                                                                      */
typedef struct tr_hit_entry {
    float t_key;
    int32_t slot;
} tr_hit_entry;
int tr_intersects_count_topk(const tr_bvh *bvh, const tr_rays *rays, int32_t cap,
                             int32_t *d_count, tr_hit_entry *d_hits, void *stream);
int tr_location_fill_slots(const tr_bvh *bvh, const tr_rays *rays, int32_t cap,
                           const int32_t *d_count, const int64_t *d_offsets,
                           const tr_hit_entry *d_hits, float *d_loc, int32_t *d_ray_idx,
                           int32_t *d_tri_idx, int64_t ray_base, void *stream);

/* -- stream compaction of closest-hit results (ray_optix.py:142-144, 219-223): keeps the
 *    rows with hit != 0, in ray order.  d_offsets (n int64) = exclusive scan of hit, from
 *    tr_mask_scan (same total/h_total convention as tr_hits_scan).  Any output may be NULL. */
int tr_mask_scan(const uint8_t *d_hit, int64_t n, int64_t *d_offsets, int64_t *d_total,
                 int64_t *h_total, void *stream);
int tr_compact_closest(const uint8_t *d_hit, const int64_t *d_offsets, int64_t n,
                       const uint8_t *d_front, const int32_t *d_tri, const float *d_loc,
                       const float *d_uv, int64_t ray_base, uint8_t *d_front_out,
                       int32_t *d_ray_idx_out, int32_t *d_tri_out, float *d_loc_out,
                       float *d_uv_out, void *stream);

/* -- diagnostics: run closest-hit with the instrumented kernel and return counters
 *    (synchronises).  Not on the hot path.                                               */
int tr_trace_stats_closest(const tr_bvh *bvh, const tr_rays *rays, tr_trace_stats *h_stats,
                           void *stream);
/*    the same for any query: query = 0 any, 1 first, 2 closest, 3 count, 4 location (cap 8)   */
int tr_trace_stats_query(const tr_bvh *bvh, const tr_rays *rays, int query, tr_trace_stats *h_stats,
                   void *stream);

/* -- tuning knobs (process-wide; none of them changes results).  Returns TR_ERR_INVALID_ARG for unknown names or values
 *    out of range.  Launch shapes:
 *      "adaptive" (0/1: learn the launch order -- most expensive blocks first, per XCD -- from the measured block costs of the
 *        previous launch of the same batch shape on the same (handle, stream)), "order_transfer" (0/1: the first launch of a
 *        new image resolution starts from the previous resolution's costs, resampled), "sort_inline" (0/1: in the steady
 *        state the sort of the measured costs rides in the next launch of the shape as one workgroup instead of running as
 *        a kernel behind the measuring launch), "xcd_chunk" (blocks of 128 rays per
 *        XCD-local chunk of the block -> ray map, 0 = identity), "compact" (0/1: 32-bit offsets / trail words where the
 *        hierarchy permits),
 *      "steal" (intra-wave work stealing: 0 off / 1 closest, first and any up to 4 M rays -- a ray gives subtrees away from its
 *        64th trip on, in a wave of unrelated rays from the first look -- / N >= 2 forced, N = the trip for every wave), "usteal" (the same for count launches: 0 / 1 / N forced),
 *      "split" (0 off / 1 auto / N >= 2: the nblocks >> N most expensive blocks of the learned order run as two -- the first
 *        quarter as four -- launch slots of half / quarter lane density whose idle lanes steal from trip "split_steal" on;
 *        really split are the blocks that cost at least "split_outlier" eighths of the mean block cost (0 all, 1 by how full
 *        the chip is) and at least "split_floor" microseconds per wave),
 *      "tile" (0 never / 1 auto / 2 always: image-shaped batches [..., H, W, 3] with W % 8 == 0 are traced in 8x8 pixel tiles
 *        per wave), "tile_small" (0..4: footprint of a wave for images below the tile threshold: rows, 2x32, 4x16, 8x8, auto),
 *      "leaf_vote" (1..64 lanes with a queued leaf that trigger a leaf phase of count / location launches),
 *      "grid_nodes" (0: stealing closest / first / any launches walk the exact 64-byte nodes; 1 (default) / 2: the 32-byte
 *        grid nodes -- two 16-byte loads per visit, one fused multiply-add per box plane),
 *      "stream" (0 never / 1 auto: batches above 4 M rays (count: from 16 M on) that a probe on the device finds incoherent / 2 always: the
 *        streaming launch with wave-level ray refill), "stream_rays", "stream_refill", "stream_dynamic" (rays per range,
 *        idle lanes that trigger a refill -- 0 = by query: 28 closest / first, 20 any / count --, ranges handed out by a work counter),
 *      "wide" (0 never / 1 always / 2 (default) on meshes from 1 M triangles on AND batches from 8 M rays on: the streaming launch walks 8-wide nodes with 8-bit child boxes,
 *        built on the first query that wants them), "wide_direct" (0 never / 1 multi-hit list launches on meshes from 500 k
 *        triangles on / 2 count and location / 3 every query: the direct launch on the 8-wide nodes), "wide_stack" (1..12:
 *        entries of a lane's node stack kept in LDS; the rest spills to global memory),
 *      "expand_cus" (0 = one workgroup per 1024 records; N = at most N workgroups per CU: an expansion that runs beside a
 *        trace), "expand_tiles" (0/1: tr_closest_expand_slots_rows takes blocks of 8 rows x 32 pixels per wave).
 *    Builder: "build_cache" (0/1: keep the builder's temporaries, about 130 B/triangle per device, between builds),
 *      "node_layout" (0/1, read at BUILD time: nodes in Karras numbering / in treelets of three levels, depth first).
 *    Retired names are accepted and ignored (their launch shapes were measured out: DESIGN_experiments.md): "persistent",
 *      "blocks_per_cu", "block_size", "scramble", "unordered", "occ8", "lds_top", "expand4", "tail_split", "refill",
 *      "refill_min", "xcd_segments", "leaf_min".                                                                 */
int tr_set_option(const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* TRIRO_HIP_H */
