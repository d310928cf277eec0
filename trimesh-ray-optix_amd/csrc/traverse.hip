// traverse.hip -- query kernels and their C-ABI launchers (replace the five OptiX pipelines
// of triro/backend/shaders.cu:67-246 and the launch wrappers of ray.cpp:161-378).
//
// One ray per lane, wave64; the per-ray state machine is tr_fused_step (tr_bvh.h): stackless
// trail + LDS far-child ring, one node visit and one queued leaf test per trip.  Launch shapes:
//   direct     : grid = ceil(n/BS) workgroups of BS = 64/128/256 rays (default, k_query_direct).
//                Which ray block a workgroup takes is a scheduling choice: the measured
//                per-XCD cost order of the previous launch (k_sched_sort), else the
//                XCD-chunked, scrambled static map.
//   persistent : grid = CUs x blocks_per_cu; each wave pulls 64-ray batches from a global
//                work counter (option, not faster at the measured sizes).  The per-lane refill
//                variant of round 1 ("active-ray repacking": 0.8 vs 3.6 Grays/s) was removed in
//                round 2 after the randomised sweep found a mismatch in it.
// Kernel parameters travel by value (no per-call malloc/memcpy/free as in ray.cpp:279-287).
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "tr_internal.h"

namespace {

struct RayFetch {
    const float* o;
    const float* d;
    int64_t n;
    int64_t s0, s1, s2;   // leading dims (right-aligned; unused = INT64_MAX)
    int64_t os[4], ds[4]; // element strides
    int omode, dmode;     // 0 dense [n,3], 1 broadcast (all leading strides 0), 2 general
};

struct QueryOut {
    uint8_t* hit;
    uint8_t* front;
    int32_t* tri;
    float* loc;
    float* uv;
    int32_t* count;
    tr_hit_entry* hits;   // TR_Q_LOCATION: [n, cap] unsorted nearest hits (tr_topk<0>)
    int32_t cap;
    tr_packed_hit* packed;   // TR_Q_CLOSEST: when set, 12 bytes per ray instead of the five arrays
    int packed_slots;        // ... with the arena slot of the triangle instead of its face index (tr_intersects_closest_packed_slots)
};

// strided fetch of ray `idx`: the reference's getRay/getIndices (shaders.cu:27-63) with
// 64-bit index math and fast paths for dense and broadcast tensors
__device__ __forceinline__ void fetch_ray(const RayFetch& rf, int64_t idx, float* o, float* d) {
    int64_t i0 = 0, i1 = 0, i2 = 0;
    if (rf.omode == 2 || rf.dmode == 2) {
        uint64_t r = (uint64_t)idx;
        i2 = (int64_t)(r % (uint64_t)rf.s2); r /= (uint64_t)rf.s2;
        i1 = (int64_t)(r % (uint64_t)rf.s1); r /= (uint64_t)rf.s1;
        i0 = (int64_t)(r % (uint64_t)rf.s0);
    }
    {
        int64_t off, s3 = rf.os[3];
        if (rf.omode == 0) { off = idx * 3; s3 = 1; }
        else if (rf.omode == 1) off = 0;
        else off = i0 * rf.os[0] + i1 * rf.os[1] + i2 * rf.os[2];
        o[0] = rf.o[off]; o[1] = rf.o[off + s3]; o[2] = rf.o[off + 2 * s3];
    }
    {
        int64_t off, s3 = rf.ds[3];
        if (rf.dmode == 0) { off = idx * 3; s3 = 1; }
        else if (rf.dmode == 1) off = 0;
        else off = i0 * rf.ds[0] + i1 * rf.ds[1] + i2 * rf.ds[2];
        d[0] = rf.d[off]; d[1] = rf.d[off + s3]; d[2] = rf.d[off + 2 * s3];
    }
}

template <int Q>
__device__ __forceinline__ void write_result(const tr_bvh_view& b, const QueryOut& out, int64_t i,
                                             const tr_ray& r, const tr_result& res) {
    if (Q == TR_Q_ANY) {
        out.hit[i] = res.best_face >= 0 ? 1 : 0;
    } else if (Q == TR_Q_FIRST) {
        out.tri[i] = res.best_face;
    } else if (Q == TR_Q_COUNT || Q == TR_Q_LOCATION) {
        out.count[i] = res.count;
    } else if (Q == TR_Q_CLOSEST) {
        float loc[3] = {0.f, 0.f, 0.f}, uv[2] = {0.f, 0.f};
        uint8_t hit = 0, front = 0;
        if (out.packed && out.packed_slots == 2) {      // tr_intersects_closest_slots: the slot and nothing else
            reinterpret_cast<int32_t*>(out.packed)[i] = res.best_face >= 0 ? res.best_slot : -1;
            return;
        }
        if (out.packed) {
            // packed form (tr_intersects_closest_packed): {face | front << 30, u, v}; tr_closest_expand
            // applies tr_bary_outputs to the same (u, v) and the same vertices -> the same bits
            tr_packed_hit ph = {0xffffffffu, 0.f, 0.f};
            if (res.best_face >= 0) {
                tr_counters* nc = nullptr;
                tr_tri t = tr_load_tri<false>(b, res.best_slot, nc);
                tr_hit h; h.t = res.best_t;
                tr_tri_duv(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, h.det, h.U, h.V);
                tr_hit_bary(h, ph.u, ph.v);
                ph.tri = (uint32_t)(out.packed_slots ? res.best_slot : res.best_face) | (h.det > 0.f ? 0x40000000u : 0u);
            }
            out.packed[i] = ph;
            return;
        }
        if (res.best_face >= 0) {
            tr_counters* nc = nullptr;
            tr_tri t = tr_load_tri<false>(b, res.best_slot, nc);
            // (det, U, V) are recomputed from the winning triangle instead of being carried
            // through the traversal loop (three registers and their moves on every hit update)
            tr_hit h; h.t = res.best_t;
            tr_tri_duv(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, h.det, h.U, h.V);
            tr_hit_outputs(h, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, loc, uv);
            hit = 1; front = h.det > 0.f ? 1 : 0;
        }
        out.hit[i] = hit;
        out.front[i] = front;
        out.tri[i] = res.best_face;
        out.loc[3 * i] = loc[0]; out.loc[3 * i + 1] = loc[1]; out.loc[3 * i + 2] = loc[2];
        out.uv[2 * i] = uv[0]; out.uv[2 * i + 1] = uv[1];
    }
}

// single-triangle / empty meshes: no hierarchy exists; evaluate the predicate directly
template <int Q>
__device__ __forceinline__ void brute_one(const tr_bvh_view& b, const tr_ray& r, bool valid,
                                          tr_result& res) {
    res.best_t = TR_TMAX; res.best_face = -1; res.best_slot = -1;
    res.U = 0.f; res.V = 0.f; res.det = 1.f; res.count = 0;
    if (!valid || b.num_tris < 1) return;
    tr_counters* nc = nullptr;
    tr_tri t = tr_load_tri<false>(b, 0, nc);
    tr_hit h;
    if (tr_tri_hit(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, h)) {
        res.best_t = h.t; res.best_face = t.face; res.best_slot = 0;
        res.U = h.U; res.V = h.V; res.det = h.det; res.count = 1;
    }
}

// Wave-level traversal of one ray per lane with the fused, software-pipelined trip
// (tr_fused_step): every lane advances on every trip; results do not depend on the schedule
// (the hit predicate is order-independent, tr_math.h).
template <bool C, bool DEEP = false> struct tr_word { typedef uint64_t T; };
template <> struct tr_word<true, false> { typedef uint32_t T; };

// COMPACT = both arrays are below 4 GiB: SGPR-base + 32-bit-offset loads; and, unless DEEP (the
// hierarchy is more than 32 levels high), 32-bit trail / owned words.  Chosen on the host per BVH.
// COMPACT + DEEP is what meshes of a few million triangles and more get (5.2 M-triangle sphere: 34
// levels): 80 instead of 82 VGPRs in the stealing closest kernel, i.e. 6 instead of 5 waves/SIMD.
#ifndef TR_STREAM_QN
#define TR_STREAM_QN true     // the streaming launch walks the 32-byte grid nodes (tr_rec_q)
#endif
#ifndef TR_ALTERNATE
#define TR_ALTERNATE 1        // every second trip runs without the leaf block (tr_fused_step<..., TEST>)
#endif
template <int Q, int K, bool STATS, bool COMPACT = false, bool UNI = false, bool DEEP = false>
__device__ __forceinline__ void wave_traverse(const tr_bvh_view& b, const tr_ray& r, bool go,
                                              tr_result& res, tr_topk<K>& top, tr_counters* cnt,
                                              const tr_ring ring) {
    tr_result_init(res);
    if (Q == TR_Q_LOCATION) top.init();
    // fused, software-pipelined schedule: every lane advances on every trip
    typedef typename tr_word<COMPACT, DEEP>::T W;
    tr_state_t<W> fs;
    tr_state_init(fs);
    if (!go) fs.node = -1;
    while (!tr_done(fs)) {
        tr_fused_step<Q, K, STATS, COMPACT, W, UNI, true>(b, r, fs, res, top, cnt, ring);
        TR_CONVERGE();
#pragma unroll
        for (int a = 0; a < TR_ALTERNATE; a++) {
            tr_fused_step<Q, K, STATS, COMPACT, W, UNI, false>(b, r, fs, res, top, cnt, ring);   // no-op for a finished lane
            TR_CONVERGE();
        }
    }
}

// ---- intra-wave work stealing ------------------------------------------------------------------
// A wave runs until its slowest ray is done; on the headline batch that is 334 trips for one
// grazing ray while the other 63 lanes finished after 30-60.  Here an idle lane takes the
// SHALLOWEST owed far child (the biggest untouched subtree) of a busy lane together with that
// lane's ray and traverses it as an independent sub-traversal (empty trail: it ends when the
// subtree is exhausted); the per-ray results are merged at the end of the wave by the same
// (t_key, face) minimum (closest / first) or by summation (count).  Any partition of the tree
// among lanes examines the same set of candidate triangles except for culling, so results are
// bit-identical.  Donors are rays that are still busy at their `steal_min`-th trip (64): by then a
// primary ray has normally found its hit (the thief inherits that bound), and what is left are
// the grazing rays that make the long waves.  Splitting earlier costs culling (-6 % at 48, -15 %
// at 32 on the headline).  wl = 6*64 ints of LDS scratch per wave.
#ifndef TR_STEAL_EVERY
#define TR_STEAL_EVERY 3u     // hand-overs are attempted on every (TR_STEAL_EVERY+1)-th trip ...
#endif
#ifndef TR_STEAL_SHARE
#define TR_STEAL_SHARE 1      // lanes working on the same ray exchange their best hit at every look
#endif
#ifndef TR_STEAL_IDLE
#define TR_STEAL_IDLE 1       // ... when at least this many lanes are idle
#endif
__device__ __forceinline__ int lane_rank(unsigned long long mask) {   // set bits of mask below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// LT: LDS-staged node packets (north_star).  `toplds` holds the grid nodes of the top TR_TOP_LEVELS levels
// in heap order (copied once per workgroup, k_query_direct).  A wave starts at the root with all of its
// rays and walks the first levels in lockstep: as long as EVERY lane that visits a node this trip is still
// on its first descent inside the table (hp != 0: its heap index), the records come from LDS
// (two ds_read_b128) and the trip issues no node gather at all; the first trip on which a lane has
// backtracked, taken stolen work or left the table ends it for the wave.  Same records, same arithmetic.
// SLIM: 192 instead of 384 ints of LDS scratch per wave -- the hand-over lists its donors with
// ds_permute / ds_bpermute and moves node and depth through registers (as wave_count_unordered_steal
// does), so that only the per-ray merge keys and winner slots stay in LDS: ring + scratch = 9.5 KiB per
// 128-thread workgroup, 16 workgroups = 8 waves per SIMD (k_query_direct_occ8).
template <int Q, bool STATS, bool COMPACT, bool DEEP = false, bool QN = false, bool LT = false, bool SLIM = false>
__device__ __forceinline__ bool wave_traverse_steal(const tr_bvh_view& b, tr_ray& r, bool go,
                                                    tr_result& res, tr_counters* cnt,
                                                    const tr_ring ring, int32_t* wl, int lane,
                                                    uint32_t steal_min, const tr_i4* toplds = nullptr) {
    typedef typename tr_word<COMPACT, DEEP>::T W;
    tr_result_init(res);
    tr_topk<1> top;
    tr_state_t<W, !QN> fs;
    tr_state_init(fs);
    if (!go) fs.node = -1;
    uint32_t hp = (LT && go) ? 1u : 0u;      // heap index of the node this lane is at (0: not in the table)
    bool table_live = LT;                    // wave-uniform: the lockstep descent through the table is still on
    // one trip; with LT the record comes from the LDS table while the whole wave is inside it
    auto trip_step = [&](auto test_tag) {
        constexpr bool TEST = decltype(test_tag)::value;
        if constexpr (LT && QN) {
            if (table_live) {
                const int32_t room = TEST ? fs.p2 : fs.p1;
                const bool has_node = fs.node >= 0 && room < 0;
                table_live = __ballot(has_node && hp == 0u) == 0ull && __ballot(has_node) != 0ull;
                if (table_live) {
                    typedef __attribute__((address_space(3))) const int32_t lds_ci32;
                    lds_ci32* tp = (lds_ci32*)toplds + 8u * (has_node ? hp : 1u);
                    tr_rec_q rec;
                    rec.w0.x = tp[0]; rec.w0.y = tp[1]; rec.w0.z = tp[2]; rec.w0.w = tp[3];
                    rec.w1.x = tp[4]; rec.w1.y = tp[5]; rec.w1.z = tp[6]; rec.w1.w = tp[7];
                    if (!tr_done(fs)) {
                        const uint32_t d0 = fs.depth;
                        tr_fused_body<Q, 1, STATS, COMPACT, W, TEST>(b, r, fs, res, top, cnt, ring, has_node, rec);
                        if (has_node) {
                            // still descending?  c0 -> 2h, c1 -> 2h+1; anything else (backtrack, end) leaves the table
                            const bool down = fs.depth == d0 + 1u && fs.depth < (uint32_t)TR_TOP_LEVELS;
                            hp = !down ? 0u : (fs.node == rec.w1.z ? 2u * hp : (fs.node == rec.w1.w ? 2u * hp + 1u : 0u));
                        }
                    }
                    return;
                }
            }
        }
        if (!tr_done(fs)) tr_fused_step<Q, 1, STATS, COMPACT, W, false, TEST, QN>(b, r, fs, res, top, cnt, ring);
    };
    int owner = lane;          // lane whose ray this lane is working on
    bool split = false;        // wave-uniform: some ray is (or was) traversed by more than one lane
    uint32_t trip = 0;
#ifdef TR_TIMELINE
    int tl_handovers = 0;
#endif
    // explicit LDS pointers: volatile accesses through generic pointers would compile to flat
    // loads/stores with 64-bit addresses held in VGPRs for the whole loop
    typedef __attribute__((address_space(3))) volatile int32_t lds_i32;
    typedef __attribute__((address_space(3))) volatile unsigned long long lds_u64;
    lds_i32* const lw = (lds_i32*)wl;
    lds_i32* const list = lw;                    // [64] donor lane of pair k          (not with SLIM)
    lds_i32* const xnode = lw + 64;              // [64] node handed over by donor lane
    lds_i32* const xdepth = lw + 128;            // [64] its depth
    // per-ray accumulators at the owner's index: partial results are deposited whenever a lane
    // finishes a piece of work (before it takes the next one) and once more at the end
    constexpr int ACC = SLIM ? 0 : 192;          // SLIM: the accumulators are all the scratch there is
    int32_t* sum = wl + ACC;                                                        // count
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(wl + ACC);   // closest / first
    lds_i32* const vsum = lw + ACC;
    lds_u64* const vkeys = (lds_u64*)(lw + ACC);
    lds_i32* const vslots = lw + ACC + 128;
    auto deposit = [&]() {
        if (Q == TR_Q_COUNT) {
            if (res.count) atomicAdd(&sum[owner], res.count);
        } else if (Q == TR_Q_ANY) {
            if (res.best_face >= 0) vsum[owner] = 1;      // any hit of any worker
        } else {
            const bool have = res.best_slot >= 0;
            // t_key >= 0, but tr_tri_mt can return -0.0f (a ray that starts exactly in a triangle's
            // plane): its bit pattern 0x80000000 would sort above every positive distance while
            // tr_closer treats it as equal to +0.0f.  Dropping the sign bit is exact for t_key >= 0
            // and makes the integer order of the key the (t_key, face) order of tr_closer.
            const unsigned long long key = ((unsigned long long)(__float_as_uint(res.best_t) & 0x7fffffffu) << 32) |
                                           (unsigned)res.best_face;
            if (have) atomicMin(&keys[owner], key);
            __builtin_amdgcn_wave_barrier();
            if (have && vkeys[owner] == key) vslots[owner] = res.best_slot;   // faces are distinct: one winner
        }
    };
    for (;;) {
        // TR_STEAL_EVERY+1 plain trips (idle lanes sit them out under the exec mask: a trip takes
        // longer the more lanes take part in its loads), then one look at the wave
#pragma unroll 1
#if TR_ALTERNATE
        for (uint32_t k = 0; k <= TR_STEAL_EVERY; k += 1 + TR_ALTERNATE) {
            trip_step(std::true_type{});
            TR_CONVERGE();
#pragma unroll
            for (int a = 0; a < TR_ALTERNATE; a++) {
                trip_step(std::false_type{});
                TR_CONVERGE();
            }
        }
#else
        for (uint32_t k = 0; k <= TR_STEAL_EVERY; k++) {
            if (!tr_done(fs)) tr_fused_step<Q, 1, STATS, COMPACT, W>(b, r, fs, res, top, cnt, ring);
            TR_CONVERGE();
        }
#endif
        trip += TR_ALTERNATE ? (TR_STEAL_EVERY / (1u + TR_ALTERNATE) + 1u) * (1u + TR_ALTERNATE) : TR_STEAL_EVERY + 1u;
        if (TR_STEAL_SHARE && split) {
            // Rays that are traversed by several lanes share what they have found: a lane's bound is
            // the best hit of ANY lane working on its ray (closest / first), and an any-hit ray ends
            // for all of them with the first hit.  A bound that is a real hit of the same ray culls
            // exactly what the lane's own hit at that distance would cull, and the result is the
            // minimum (t_key, face) over all lanes either way -- but the subtrees given away early no
            // longer lose the culling the donor's later hits would have brought.
            if (Q == TR_Q_ANY) {
                if (res.best_face >= 0) vsum[owner] = 1;
                __builtin_amdgcn_wave_barrier();
                if (vsum[owner] != 0 && !tr_done(fs)) {
                    fs.node = -1; fs.p0 = -1; fs.p1 = -1;
#if TR_LEAF_QUEUE
                    fs.p2 = -1;
#endif
                }
            } else if (Q == TR_Q_CLOSEST || Q == TR_Q_FIRST) {
                const bool have = res.best_slot >= 0;
                const unsigned long long mine = ((unsigned long long)(__float_as_uint(res.best_t) & 0x7fffffffu) << 32) |
                                                (unsigned)(res.best_face < 0 ? 0x7fffffff : res.best_face);
                if (have) atomicMin(&keys[owner], mine);
                __builtin_amdgcn_wave_barrier();
                const unsigned long long k = vkeys[owner];
                if (have && k == mine) vslots[owner] = res.best_slot;
                if (k < mine) {     // another lane's hit: a bound, not a result of this lane
                    res.best_t = __uint_as_float((unsigned)(k >> 32));
                    res.best_face = (int32_t)(unsigned)k;
                    res.best_slot = -1;
                }
            }
        }
        const bool done = tr_done(fs);
        const unsigned long long idle = __ballot(done);
        if (idle == ~0ull) break;
        if (__popcll(idle) >= TR_STEAL_IDLE) {
            const W cand = fs.trail & fs.owned;      // owed far children that are still in the ring
            const bool can_give = !done && cand != 0 && trip >= steal_min;
            const unsigned long long donors = __ballot(can_give);
            const int ni = __popcll(idle), nd = __popcll(donors);
            const int np = ni < nd ? ni : nd;
            if (np > 0) {
#ifdef TR_TIMELINE
                tl_handovers += np;
#endif
                if (!split) {   // first hand-over in this wave: set the accumulators up
                    if (Q == TR_Q_COUNT || Q == TR_Q_ANY) vsum[lane] = 0;
                    else vkeys[lane] = ~0ull;
                    split = true;
                    __builtin_amdgcn_wave_barrier();
                }
                const int drank = lane_rank(donors), irank = lane_rank(idle);
                const bool give = can_give && drank < np;
                int gnode = 0, gdepth = 0;
                if (give) {
                    const uint32_t j = (uint32_t)__builtin_ctzll((unsigned long long)cand);
                    gnode = tr_ring_get(ring, j & (TR_RING - 1));
                    gdepth = (int32_t)(j + 1);
                    if (!SLIM) { list[drank] = lane; xnode[lane] = gnode; xdepth[lane] = gdepth; }
                    fs.trail &= ~(W(1) << j);
                    fs.owned &= ~(W(1) << j);
                }
                __builtin_amdgcn_wave_barrier();
                const bool take = done && irank < np;
                int src;
                if (SLIM) {
                    // giver g sends its lane id to lane g (everybody else to distinct lanes from the top); the
                    // read-back is issued by EVERY lane: ds_bpermute returns 0 for a masked-off source lane
                    const unsigned long long givers = __ballot(give);
                    const int tgt = give ? drank : 63 - lane_rank(~givers);
                    const int lst = __builtin_amdgcn_ds_permute(tgt << 2, lane);
                    const int pick = __shfl(lst, irank & 63);
                    src = take ? pick : lane;
                } else {
                    src = take ? list[irank] : lane;
                }
                // a lane that takes new work first hands in what it has found so far
                if (take) {
                    deposit();
                    tr_result_init(res);
                }
                // the ray (and its owner / current bound) moves with the subtree
                r.ox = __shfl(r.ox, src); r.oy = __shfl(r.oy, src); r.oz = __shfl(r.oz, src);
                r.dx = __shfl(r.dx, src); r.dy = __shfl(r.dy, src); r.dz = __shfl(r.dz, src);
                r.ix = __shfl(r.ix, src); r.iy = __shfl(r.iy, src); r.iz = __shfl(r.iz, src);
                r.sel_n = (uint32_t)__shfl((int)r.sel_n, src); r.sel_f = (uint32_t)__shfl((int)r.sel_f, src); r.sel_z = (uint32_t)__shfl((int)r.sel_z, src);
                if constexpr (QN) {      // the fused box-test constants of the grid nodes (tr_ray_fuse)
                    r.qax = __shfl(r.qax, src); r.qay = __shfl(r.qay, src); r.qaz = __shfl(r.qaz, src);
                    r.qnx = __shfl(r.qnx, src); r.qny = __shfl(r.qny, src); r.qnz = __shfl(r.qnz, src);
                    r.qfx = __shfl(r.qfx, src); r.qfy = __shfl(r.qfy, src); r.qfz = __shfl(r.qfz, src);
                    r.qaz2 = r.qaz;
                }
                const int own2 = __shfl(owner, src);
                const float bt = __shfl(res.best_t, src);
                const int n2 = SLIM ? __shfl(gnode, src) : 0, d2 = SLIM ? __shfl(gdepth, src) : 0;
                if (take) {
                    owner = own2;
                    tr_state_init(fs);
                    fs.node = SLIM ? n2 : xnode[src];
                    fs.depth = (uint32_t)(SLIM ? d2 : xdepth[src]);
                    res.best_t = bt;
                    hp = 0u;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (split) {
        // everybody hands in what it still holds; every owner lane reads its ray's total
        deposit();
        __builtin_amdgcn_wave_barrier();
        if (Q == TR_Q_COUNT) {
            res.count = vsum[lane];
        } else if (Q == TR_Q_ANY) {
            res.best_face = vsum[lane] ? 0 : -1;
        } else {
            const unsigned long long k = vkeys[lane];
            tr_result_init(res);
            if (k != ~0ull) {
                res.best_t = __uint_as_float((unsigned)(k >> 32));
                res.best_face = (int32_t)(unsigned)k;
                res.best_slot = vslots[lane];
            }
        }
    }
#ifdef TR_TIMELINE
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { lw[0] = (int32_t)trip; lw[1] = tl_handovers; }
#endif
    return split;
}

// All 64 lanes of a wave call this together (`in_range` = the lane owns ray i).
template <int Q, bool STATS, bool COMPACT = false, bool UNI = false, bool DEEP = false>
__device__ __forceinline__ void process_ray(const tr_bvh_view& b, const RayFetch& rf,
                                            const QueryOut& out, int64_t i, bool in_range,
                                            tr_counters* cnt, const tr_ring ring) {
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    if (in_range) fetch_ray(rf, i, o, d);
    tr_ray r;
    const bool valid = tr_ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]) && in_range;
    tr_result res;
    if (Q == TR_Q_LOCATION) {
        // fused multi-hit: uncapped count + the ray's `cap` nearest hits as unsorted entries
        tr_topk<0> top;
        top.ent = out.hits + (in_range ? i : 0) * out.cap;
        top.tris = b.tris;
        top.cap = out.cap;
        if (b.num_tris >= 2) {
            wave_traverse<Q, 0, STATS, COMPACT, UNI, DEEP>(b, r, valid, res, top, cnt, ring);
        } else {
            top.init();
            brute_one<Q>(b, r, valid, res);
            if (res.count) top.insert(res.best_t, res.best_face, 0);
        }
    } else {
        tr_topk<1> top;
        if (b.num_tris >= 2) wave_traverse<Q, 1, STATS, COMPACT, UNI, DEEP>(b, r, valid, res, top, cnt, ring);
        else brute_one<Q>(b, r, valid, res);
    }
    if (in_range) write_result<Q>(b, out, i, r, res);
}

template <int Q, bool STATS, bool COMPACT, bool DEEP = false, bool QN = false, bool LT = false, bool SLIM = false>
__device__ __forceinline__ void process_ray_steal(const tr_bvh_view& b, const RayFetch& rf,
                                                  const QueryOut& out, int64_t i, bool in_range,
                                                  tr_counters* cnt, const tr_ring ring, int32_t* wl,
                                                  uint32_t steal_min, const tr_i4* toplds = nullptr) {
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    if (in_range) fetch_ray(rf, i, o, d);
    tr_ray r;
    const bool valid = tr_ray_setup_q(r, b.frame, o[0], o[1], o[2], d[0], d[1], d[2]) && in_range;
    tr_result res;
    bool split = false;
    if (b.num_tris >= 2) split = wave_traverse_steal<Q, STATS, COMPACT, DEEP, QN, LT, SLIM>(b, r, valid, res, cnt, ring, wl, (int)(threadIdx.x & 63), steal_min, toplds);
    else brute_one<Q>(b, r, valid, res);   // no hierarchy below two triangles
    if (split && in_range) {   // this lane may hold another lane's ray now: take its own again
        fetch_ray(rf, i, o, d);
        tr_ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]);
    }
    if (in_range) write_result<Q>(b, out, i, r, res);
}

// ---- unordered two-phase schedule (tr_unord_step): any / count / location --------------------
// Wave-level vote per trip: the leaf phase (three triangle loads + the full predicate) runs only
// when a lane's queue is nearly full ("parked": it could not take both children of its node),
// when at least `leaf_min` lanes have something queued, or when no lane has a node left.
template <int Q, int K, bool STATS, bool COMPACT, bool DEEP = false>
__device__ __forceinline__ void wave_traverse_unordered(const tr_bvh_view& b, const tr_ray& r, bool go,
                                                        tr_result& res, tr_topk<K>& top, tr_counters* cnt,
                                                        const tr_ring ring, const tr_leafq lq, int leaf_min) {
    typedef typename tr_word<COMPACT, DEEP>::T W;
    tr_result_init(res);
    if (Q == TR_Q_LOCATION) top.init();
    tr_ustate_t<W> st;
    tr_ustate_init(st);
    if (!go) st.node = -1;
    for (;;) {
        const bool can_node = tr_ucan_node(st);
        const unsigned long long mn = __ballot(can_node), ml = __ballot(st.nq > 0);
        if ((mn | ml) == 0ull) break;
        const bool parked = st.node >= 0 && !can_node;
        const bool leaf_phase = mn == 0ull || __ballot(parked) != 0ull || (int)__popcll(ml) >= (int)leaf_min;
        tr_unord_step<Q, K, STATS, COMPACT, W>(b, r, can_node, leaf_phase, st, res, top, cnt, ring, lq);
        TR_CONVERGE();
    }
}

#ifdef TR_USTEAL_DEBUG
__device__ unsigned g_usteal_debug[4];
#endif
// ---- the unordered schedule with work stealing (count) --------------------------------------------
// A count launch has no culling, but on a silhouette image the grazing rays are outliers here too
// (headline image: bulk done at 377 us of 556), and a launch that leaves wave slots of the chip empty
// ends with its longest waves.  Both are answered by SPLIT launch slots (k_sched_sort): a slot owns
// every 2nd / 4th ray of a block and its other lanes take owed subtrees of those rays from the first
// trips on: headline count 0.545 -> 0.383 ms, C4 at 262 k rays 0.436 -> 0.348, the interior scene at
// 230 k rays 0.254 -> 0.164 (profiles/r03_sweep_usteal.jsonl).  A launch of several rounds of EQUALLY
// long waves (C4 at 1 M rays) gains nothing, also not from splitting the blocks of its last round
// (tried: the ramp-down is half a wave's time whatever the last waves are, and split waves are not
// 2-4x shorter; r03_sweep_usteal_uniform.jsonl).  The multi-hit list query stays on the plain unordered
// schedule: a list that several lanes append to needs its fill count in one place and its overflow
// path (a ray with more than `cap` hits replaces its farthest entry) serialised -- built and measured:
// headline location -25 %, but C4 +5 % and the interior scene, where rays have up to 13 hits, 5x SLOWER
// (profiles/r03_location_steal_experiment.jsonl).  Hand-over as in wave_traverse_steal, with two differences: (1) no LDS scratch --
// the kernel sits exactly at the LDS budget of 7 waves per SIMD (ring + leaf queue), so donor lanes
// are listed with ds_permute / ds_bpermute and counts are handed in through v_readlane loops; (2) the
// merge is a sum, order-free.  Any partition of a tree among lanes visits the same leaves, so the
// counts are bit-identical (tests force thresholds 0 ... 64 on every scene family).
template <int Q, bool STATS, bool COMPACT, bool DEEP = false>
__device__ __forceinline__ int wave_count_unordered_steal(const tr_bvh_view& b, tr_ray& r, bool go, tr_counters* cnt,
                                                          const tr_ring ring, const tr_leafq lq, int leaf_min,
                                                          int lane, uint32_t steal_min) {
    typedef typename tr_word<COMPACT, DEEP>::T W;
    tr_result res;
    tr_result_init(res);
    tr_topk<1> top;
    tr_ustate_t<W> st;
    tr_ustate_init(st);
    if (!go) st.node = -1;
    int owner = lane;      // lane whose ray this lane is working on
    int acc = 0;           // hits of THIS lane's ray that have been handed in (by itself or by others)
    uint32_t trip = 0;
    // lanes with `want` hand the count they hold to the owner of the ray they worked on
    auto hand_in = [&](bool want) {
        unsigned long long m = __ballot(want && res.count != 0);
        while (m != 0ull) {
            const int t = (int)__builtin_ctzll(m);
            const int o = __builtin_amdgcn_readlane(owner, t), c = __builtin_amdgcn_readlane(res.count, t);
            if (lane == o) acc += c;
            m &= m - 1ull;
        }
        if (want) res.count = 0;
    };
    for (;;) {
        bool live = true;      // wave-uniform: some lane still has a node or a queued leaf
#pragma unroll 1
        for (int k = 0; k < 4 && live; k++) {
            const bool can_node = tr_ucan_node(st);
            const unsigned long long mn = __ballot(can_node), ml = __ballot(st.nq > 0);
            live = (mn | ml) != 0ull;
            if (live) {
                const bool parked = st.node >= 0 && !can_node;
                const bool leaf_phase = mn == 0ull || __ballot(parked) != 0ull || (int)__popcll(ml) >= (int)leaf_min;
                tr_unord_step<Q, 1, STATS, COMPACT, W>(b, r, can_node, leaf_phase, st, res, top, cnt, ring, lq);
            }
            TR_CONVERGE();
        }
        trip += 4;
        const bool done = tr_udone(st);
        const unsigned long long idle = __ballot(done);
        if (idle == ~0ull) break;
#ifdef TR_USTEAL_DEBUG
        if (trip > (1u << 16)) { if (lane == 0) atomicAdd(&g_usteal_debug[0], 1u); if (!done) atomicAdd(&g_usteal_debug[1], 1u); break; }
#endif
        if (idle == 0ull) continue;
        const W cand = st.trail & st.owned;          // owed far children that are still in the ring
        const bool can_give = !done && cand != 0 && trip >= steal_min;
        const unsigned long long donors = __ballot(can_give);
        const int ni = __popcll(idle), nd = __popcll(donors);
        const int np = ni < nd ? ni : nd;
        if (np == 0) continue;
#ifdef TR_USTEAL_DEBUG
        if (lane == 0) atomicAdd(&g_usteal_debug[2], (unsigned)np);
#endif
        const int drank = lane_rank(donors), irank = lane_rank(idle);
        const bool give = can_give && drank < np;
        const bool take = done && irank < np;
        // compact list of the giving lanes without LDS: giver g sends its lane id to lane g (the other
        // lanes send theirs to distinct lanes from the top, so that every lane is written exactly once)
        const unsigned long long givers = __ballot(give);
        const int tgt = give ? drank : 63 - lane_rank(~givers);
        const int list = __builtin_amdgcn_ds_permute(tgt << 2, lane);
        // (the read-back is executed by EVERY lane: ds_bpermute returns 0 for a source lane that is
        // masked off, and list entry g lives in lane g, which need not be a taker itself)
        const int pick = __shfl(list, irank & 63);
        const int src = take ? pick : lane;
        int gnode = 0, gdepth = 0;
        if (give) {
            const uint32_t j = (uint32_t)__builtin_ctzll((unsigned long long)cand);     // the shallowest: the biggest subtree
            gnode = tr_ring_get(ring, j & (TR_RING - 1));
            gdepth = (int)(j + 1);
            st.trail &= ~(W(1) << j);
            st.owned &= ~(W(1) << j);
        }
        // a lane that takes new work first hands in what it counted for the ray it is leaving
        if (take && owner == lane) { acc += res.count; res.count = 0; }
        hand_in(take && owner != lane);
        r.ox = __shfl(r.ox, src); r.oy = __shfl(r.oy, src); r.oz = __shfl(r.oz, src);
        r.dx = __shfl(r.dx, src); r.dy = __shfl(r.dy, src); r.dz = __shfl(r.dz, src);
        r.ix = __shfl(r.ix, src); r.iy = __shfl(r.iy, src); r.iz = __shfl(r.iz, src);
        r.sel_n = (uint32_t)__shfl((int)r.sel_n, src); r.sel_f = (uint32_t)__shfl((int)r.sel_f, src); r.sel_z = (uint32_t)__shfl((int)r.sel_z, src);
        r.qax = __shfl(r.qax, src); r.qay = __shfl(r.qay, src); r.qaz = __shfl(r.qaz, src);      // (tr_ray_fuse)
        r.qnx = __shfl(r.qnx, src); r.qny = __shfl(r.qny, src); r.qnz = __shfl(r.qnz, src);
        r.qfx = __shfl(r.qfx, src); r.qfy = __shfl(r.qfy, src); r.qfz = __shfl(r.qfz, src);
        r.qaz2 = r.qaz;
        const int own2 = __shfl(owner, src), n2 = __shfl(gnode, src), d2 = __shfl(gdepth, src);
        if (take) {
            owner = own2;
            tr_ustate_init(st);
            st.node = n2;
            st.depth = (uint32_t)d2;
        }
        TR_CONVERGE();
    }
    if (owner == lane) { acc += res.count; res.count = 0; }
    hand_in(owner != lane);
    return acc;
}

template <int Q, bool STATS, bool COMPACT, bool DEEP = false, bool USTEAL = false>
__device__ __forceinline__ void process_ray_unordered(const tr_bvh_view& b, const RayFetch& rf,
                                                      const QueryOut& out, int64_t i, bool in_range,
                                                      tr_counters* cnt, const tr_ring ring,
                                                      const tr_leafq lq, int leaf_min, uint32_t steal_min = 0) {
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    if (in_range) fetch_ray(rf, i, o, d);
    tr_ray r;
    const bool valid = tr_ray_setup_q(r, b.frame, o[0], o[1], o[2], d[0], d[1], d[2]) && in_range;
    tr_result res;
    if (Q == TR_Q_LOCATION) {
        tr_topk<0> top;
        top.ent = out.hits + (in_range ? i : 0) * out.cap;
        top.tris = b.tris;
        top.cap = out.cap;
        wave_traverse_unordered<Q, 0, STATS, COMPACT, DEEP>(b, r, valid, res, top, cnt, ring, lq, leaf_min);
    } else if (USTEAL && Q == TR_Q_COUNT) {
        tr_result_init(res);
        res.count = wave_count_unordered_steal<Q, STATS, COMPACT, DEEP>(b, r, valid, cnt, ring, lq, leaf_min,
                                                                       (int)(threadIdx.x & 63), steal_min);
    } else {
        tr_topk<1> top;
        wave_traverse_unordered<Q, 1, STATS, COMPACT, DEEP>(b, r, valid, res, top, cnt, ring, lq, leaf_min);
    }
    if (in_range) write_result<Q>(b, out, i, r, res);
}

template <bool STATS>
__device__ __forceinline__ void flush_stats(const tr_counters& c, unsigned long long* stats) {
    if (!STATS) return;
    unsigned long long a = c.nodes, t = c.tris, k = c.climbs;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off); t += __shfl_xor(t, off); k += __shfl_xor(k, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[1], a); atomicAdd(&stats[2], t); atomicAdd(&stats[3], k);
    }
}

#include "traverse_wide.inc"

#ifdef TR_TIMELINE
// experiment build only (not part of the ABI): TR_TIMELINE = number of wave records kept
__device__ unsigned long long g_timeline[4 * TR_TIMELINE];
#endif

// MODE: 0 fused ordered trip, 1 fused trip + intra-wave work stealing, 2 unordered two-phase
// schedule (any / count / location on hierarchies of at least two triangles)
template <int Q, bool STATS, bool COMPACT, int BS, int MODE, bool DEEP, bool QN, bool LT, bool SLIM>
__device__ __forceinline__ void query_direct_body(const tr_bvh_view& b, const RayFetch& rf, const QueryOut& out,
                                                  int xcd_map, int scramble, int tile_w, int steal_min,
                                                  const uint32_t* __restrict__ order, int order_split,
                                                  uint32_t* __restrict__ cost,
                                                  unsigned long long* stats,
                                                  const int* __restrict__ sel,
                                                  const tr_wide_args& wa = tr_wide_args{nullptr, nullptr, 0, 0}) {
    // dual launch (k_probe_coherence): this launch shape is the one for coherent batches (id 0)
    if (sel && *sel != 0) return;
#ifdef TR_LDS_PAD
    // experiment (scripts/exp_lds_budget.sh): what would a per-workgroup LDS table of TR_LDS_PAD
    // bytes (e.g. the top levels of the tree staged once per workgroup) cost in occupancy alone?
    __shared__ volatile int32_t pad_lds[TR_LDS_PAD / 4];
    pad_lds[threadIdx.x] = (int32_t)blockIdx.x;      // volatile: the allocation must survive
#endif
#ifdef TR_TIMELINE
    // experiment (scripts/exp_timeline.py): per-wave start / end / placement of the launch
    const unsigned long long tl_start = wall_clock64();
#endif
    const unsigned long long t_start = cost ? wall_clock64() : 0ull;
    __shared__ int32_t ring_lds[MODE == 4 ? 1 : TR_RING * BS];
    const tr_ring ring = {ring_lds + threadIdx.x, BS};
    // LDS-staged node packets: the top levels of the tree, once per workgroup (4 KiB, L2-resident source)
    __shared__ tr_i4 top_lds[LT ? 2 * TR_TOP_SLOTS : 1];
    if (LT) {
        for (int k = threadIdx.x; k < 2 * TR_TOP_SLOTS; k += BS) top_lds[k] = reinterpret_cast<const tr_i4*>(b.top)[k];
        __syncthreads();
    }
    // XCD-aware block -> ray-tile map: workgroups are dealt round-robin over the 8 XCDs
    // (blocks b and b+8 share one).  The ray range is cut into chunks of `xcd_map` blocks and
    // chunk c goes to XCD c % 8, so each XCD's private L2 works on compact pieces of the image
    // (and of the BVH) while expensive regions are still spread over all XCDs.  Placement only
    // affects speed.
    int64_t blk = blockIdx.x;
    int part = 0, parts_lg = 0;           // block splitting (k_sched_sort): this launch slot's share
    int64_t nblk = gridDim.x;             // ray blocks of the launch (the grid may have extra slots)
    if (order) {
        // The order buffer belongs to the (handle, stream) and is rewritten by the sort behind every
        // measuring launch: a launch replayed from a HIP graph -- or any launch, after such a replay
        // -- may find an order that was written for ANOTHER grid.  The sort stamps what it wrote
        // (block count, split blocks per XCD) behind the array; anything else is ignored and this
        // launch runs in the static order (its extra slots have nothing to do).
        nblk -= 8 * ((int64_t)order_split + 2 * (order_split >> 2));
        const uint32_t* hdr = order + TR_SCHED_MAX;
        if (hdr[0] != (uint32_t)nblk || hdr[1] != (uint32_t)order_split) {
            order = nullptr;
            if ((int64_t)blockIdx.x >= nblk) return;
        }
    }
    if (order) {
        // measured order: most expensive blocks first.  Bits 30-31 of an entry = lg of the number of
        // launch slots the block's rays were dealt to, bits 28-29 = this slot's part.
        const uint32_t e = order[blockIdx.x];
        blk = e & 0x07ffffffu;
        if (blk >= nblk) return;          // a launch slot the sort left unused (fewer blocks split than the grid allows)
        parts_lg = (int)(e >> 30);
        part = (int)((e >> 28) & 3u);
        if (MODE != 1 && MODE != 3) {     // (split orders are only written for the shapes that steal)
            if (part) return;
            parts_lg = 0;
        }
    } else if (xcd_map > 0) {
        const int64_t T = xcd_map, span = 8 * T;
        const int64_t nfull = nblk / span * span;                 // blocks covered by whole spans
        if (blk < nfull) {
            const int64_t x = blk & 7;                            // XCD label
            int64_t k = blk >> 3;                                 // index within the XCD
            // no measured order yet: visit the XCD's blocks in a scrambled order (k -> k*P mod
            // count, P prime) so that an expensive image region is spread over the whole
            // launch instead of being started last
            if (scramble > 1) k = (k * scramble) % (nfull >> 3);
            blk = ((k / T) * 8 + x) * T + (k % T);
        }
    }
    int64_t i = blk * BS + threadIdx.x;
    if (tile_w > 0) {
        // image-shaped batch: a wave takes a tile of 2^lgh rows x 2^(6-lgh) pixels (8x8, 4x16 or
        // 2x32; lgh in bits 28-29 of the argument) instead of 64 pixels of one row
        const int lgh = (tile_w >> 28) & 3, lgw = 6 - lgh;
        const int64_t width = tile_w & 0x0fffffff;
        const int64_t tile = i >> 6, tpr = width >> lgw;
        const int lane = (int)(i & 63);
        const int64_t ty = tile / tpr, tx = tile - ty * tpr;
        i = ((ty << lgh) + (lane >> lgw)) * width + (tx << lgw) + (lane & ((1 << lgw) - 1));
    }
    tr_counters cnt = {0, 0, 0};
#ifdef TR_TIMELINE
    unsigned long long tl_extra = 0;
#endif
    if (MODE == 4) {
        // 8-wide compressed nodes, one ray per lane (wave_traverse_wide): count / location / closest / first / any
        __shared__ int32_t wstack_lds[(TR_WNODES + TR_WLEAVES) * BS];
        const bool in_range = i < rf.n;
        float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
        if (in_range) fetch_ray(rf, i, o, d);
        tr_ray r;
        const bool valid = tr_ray_setup_q(r, b.frame, o[0], o[1], o[2], d[0], d[1], d[2]) && in_range;
        tr_result res;
        if constexpr (Q == TR_Q_LOCATION) {
            tr_topk<0> top;
            top.ent = out.hits + (in_range ? i : 0) * out.cap;
            top.tris = b.tris;
            top.cap = out.cap;
            wave_traverse_wide<Q, 0, STATS, BS>(b, wa, r, valid, res, top, cnt, wstack_lds);
        } else {
            tr_topk<1> top;
            wave_traverse_wide<Q, 1, STATS, BS>(b, wa, r, valid, res, top, cnt, wstack_lds);
        }
        if (in_range) write_result<Q>(b, out, i, r, res);
    } else if (MODE == 2) {
        // the steal_min argument carries the leaf-phase vote threshold of this schedule
        __shared__ int32_t leafq_lds[TR_LEAFQ * BS];
        const tr_leafq lq = {leafq_lds + threadIdx.x, BS};
        process_ray_unordered<Q, STATS, COMPACT, DEEP>(b, rf, out, i, i < rf.n, &cnt, ring, lq, steal_min);
    } else if (MODE == 3) {
        // unordered schedule + stealing: leaf vote (bits 0-7) | trip from which a ray gives subtrees away
        // (bits 8-19) | the same for split blocks (bits 20-31)
        __shared__ int32_t leafq_lds[TR_LEAFQ * BS];
        const tr_leafq lq = {leafq_lds + threadIdx.x, BS};
        const bool mine = (((int)threadIdx.x ^ part) & ((1 << parts_lg) - 1)) == 0;
        const uint32_t smin = parts_lg ? ((uint32_t)steal_min >> 20) & 0xfffu : ((uint32_t)steal_min >> 8) & 0xfffu;
        process_ray_unordered<Q, STATS, COMPACT, DEEP, true>(b, rf, out, i, i < rf.n && mine, &cnt, ring, lq,
                                                            steal_min & 0xff, smin);
    } else if (MODE == 1) {
        constexpr int SCR = SLIM ? 192 : 384;          // ints of stealing scratch per wave
        __shared__ alignas(8) int32_t steal_lds[(BS / 64) * SCR];
        // A split block (one of the most expensive of the previous launch): this slot owns the rays
        // of every 2^parts_lg-th lane; the other lanes start idle and take subtrees of those rays
        // from the trip in the upper half of the argument on (the lower half: everybody else)
        const bool mine = (((int)threadIdx.x ^ part) & ((1 << parts_lg) - 1)) == 0;
        const uint32_t smin = parts_lg ? (uint32_t)steal_min >> 16 : (uint32_t)steal_min & 0xffffu;
        process_ray_steal<Q, STATS, COMPACT, DEEP, QN, LT, SLIM>(b, rf, out, i, i < rf.n && mine, &cnt, ring,
                                             steal_lds + (threadIdx.x >> 6) * SCR, smin, top_lds);
#ifdef TR_TIMELINE
        tl_extra = (unsigned)(steal_lds[(threadIdx.x >> 6) * SCR] & 0xffff) |
                   ((unsigned long long)(steal_lds[(threadIdx.x >> 6) * SCR + 1] & 0xffff) << 16);
#endif
    } else {
        // the plain shape is what large coherent batches get (small ones steal, incoherent ones
        // stream): look for wave-uniform trips (tr_fused_step)
        process_ray<Q, STATS, COMPACT, !STATS && BS == 128 && Q != TR_Q_LOCATION, DEEP>(b, rf, out, i, i < rf.n, &cnt, ring);
    }
    if (cost && (threadIdx.x & 63) == 0) {
        // 100 MHz ticks.  A split block records twice what it would have cost in one piece (roughly):
        // it has to stay among the expensive ones, or the split set alternates between two groups of
        // blocks from one measurement to the next (and every other group of launches has a long tail)
        const unsigned long long dt = (wall_clock64() - t_start) << (parts_lg ? parts_lg + 1 : 0);
        atomicMax(&cost[blk], (uint32_t)(dt > 0x7ffffull ? 0x7ffffull : dt));
    }
#ifdef TR_TIMELINE
    if ((threadIdx.x & 63) == 0) {
        const uint64_t w = (uint64_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
        if (w < TR_TIMELINE) {
            g_timeline[w * 4 + 0] = tl_start;
            g_timeline[w * 4 + 1] = wall_clock64();
            g_timeline[w * 4 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32) |
                                    (unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_ID | XCC_ID
            unsigned long long extra = 0;
            if (MODE == 1) {   // trips of the wave | hand-overs (wave_traverse_steal)
                extra = tl_extra;
            }
            g_timeline[w * 4 + 3] = (unsigned long long)blk | (extra << 32);
        }
    }
#endif
    flush_stats<STATS>(cnt, stats);
}

// MODE: 0 fused ordered trip, 1 fused trip + intra-wave work stealing, 2 unordered two-phase schedule,
// 3 unordered + stealing (query_direct_body)
#ifdef TR_DIRECT_WAVES
#define TR_DIRECT_OCC __attribute__((amdgpu_waves_per_eu(TR_DIRECT_WAVES, TR_DIRECT_WAVES)))
#else
#define TR_DIRECT_OCC
#endif
template <int Q, bool STATS, bool COMPACT, int BS, int MODE = 0, bool DEEP = false, bool QN = false, bool LT = false>
__global__ __launch_bounds__(BS) TR_DIRECT_OCC void k_query_direct(tr_bvh_view b, RayFetch rf, QueryOut out,
                                                      int xcd_map, int scramble, int tile_w, int steal_min,
                                                      const uint32_t* __restrict__ order, int order_split,
                                                      uint32_t* __restrict__ cost,
                                                      unsigned long long* stats,
                                                      const int* __restrict__ sel) {
    query_direct_body<Q, STATS, COMPACT, BS, MODE, DEEP, QN, LT, false>(b, rf, out, xcd_map, scramble, tile_w, steal_min, order,
                                                                        order_split, cost, stats, sel);
}
// The stealing closest / first launch on the grid nodes at EIGHT waves per SIMD: 64 registers (the
// compiler is asked for them; the kernel needs 67 unconstrained) and 9.5 KiB of LDS per workgroup (the
// slim hand-over, wave_traverse_steal<..., SLIM>).  Pays where the launch is large -- 4 M rays -3.8 % --
// and costs where it is small or the waves share lines (262 k ... 590 k rays +1...+8 %, C2 / C4 / interior on
// forced grid nodes +4...+6 %: profiles/r03_ab_occ8.txt).  Option occ8: 0 never, 1 from 2 M rays on, 2 always.
// (Round 4: the 64-register cap is gone -- the sign-selected slab test of tr_qnode_slabs needs three more registers than
// it allowed, and pays more than the eighth wave did; what remains is the slim hand-over at the natural register count,
// option occ8 default 0.)
template <int Q>
__global__ __launch_bounds__(128)
void k_query_direct_occ8(tr_bvh_view b, RayFetch rf, QueryOut out, int xcd_map, int scramble, int tile_w, int steal_min,
                         const uint32_t* __restrict__ order, int order_split, uint32_t* __restrict__ cost,
                         unsigned long long* stats, const int* __restrict__ sel) {
    query_direct_body<Q, false, true, 128, 1, false, true, false, true>(b, rf, out, xcd_map, scramble, tile_w, steal_min, order,
                                                                        order_split, cost, stats, sel);
}

// The direct launch on the 8-wide compressed nodes (query_direct_body MODE 4; option wide_direct): the block -> ray
// map, the tiles and the learned launch order of the direct launch, the per-lane two-stack walk of k_query_wide.
template <int Q, bool STATS>
__global__ __launch_bounds__(128) void k_query_direct_wide(tr_bvh_view b, RayFetch rf, QueryOut out, int xcd_map, int scramble,
                                                           int tile_w, const uint32_t* __restrict__ order, int order_split,
                                                           uint32_t* __restrict__ cost, unsigned long long* stats,
                                                           const int* __restrict__ sel, tr_wide_args wa) {
    query_direct_body<Q, STATS, true, 128, 4, false, false, false, false>(b, rf, out, xcd_map, scramble, tile_w, 0, order, order_split,
                                                                          cost, stats, sel, wa);
}

// A batch of a new SHAPE (another image resolution of the same scene) need not start from nothing: the block costs
// measured at the previous shape are resampled onto the new launch's blocks -- block b of the new launch covers some
// piece of the image, the old block that covered that piece lends its cost -- and sorted into a launch order before the
// first launch of the new shape (VERDICT r03 "next" #7b: a resolution change fell back to the static order, 0.33-0.43 ms
// for the headline batch).  Shapes: image width / height and the rows-per-tile exponent of the block -> ray map
// (0 = rows of 64 pixels); blocks hold 128 rays.  Speed only: any order is a correct order.
__device__ __forceinline__ void sched_block_pixel(int64_t b, int64_t w, int lgh, int64_t* x, int64_t* y) {
    const int64_t i = b * 128;
    if (lgh == 0) { *y = i / w; *x = i - *y * w; return; }
    const int lgw = 6 - lgh;
    const int64_t tile = i >> 6, tpr = w >> lgw, ty = tile / tpr, tx = tile - ty * tpr;
    *y = ty << lgh; *x = tx << lgw;
}
__global__ __launch_bounds__(256) void k_sched_rescale(const uint32_t* __restrict__ prev, int64_t pn, int64_t pw, int64_t ph, int plgh,
                                                       uint32_t* __restrict__ cost, int64_t nn, int64_t w, int64_t h, int lgh) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nn) return;
    int64_t x, y;
    sched_block_pixel(b, w, lgh, &x, &y);
    // the block's footprint: 128 pixels of a row, or two tiles side by side; sampled at 4 x 2 points, the lenders' costs
    // averaged (a block of 8 rows x 16 pixels that borrows from blocks of 1 row x 128 pixels meets eight of them)
    const int64_t fw = lgh ? (int64_t)(128 >> lgh) : 128, fh = lgh ? (int64_t)(1 << lgh) : 1;
    unsigned long long sum = 0;
    int cnt = 0;
#pragma unroll
    for (int sy = 0; sy < 4; sy++)
#pragma unroll
        for (int sx = 0; sx < 2; sx++) {
            const double fx = ((double)x + (double)fw * (0.25 + 0.5 * sx)) / (double)w;
            const double fy = ((double)y + (double)fh * (0.125 + 0.25 * sy)) / (double)h;
            int64_t xo = (int64_t)(fx * (double)pw), yo = (int64_t)(fy * (double)ph);
            xo = xo < 0 ? 0 : (xo >= pw ? pw - 1 : xo);
            yo = yo < 0 ? 0 : (yo >= ph ? ph - 1 : yo);
            int64_t io;
            if (plgh == 0) io = yo * pw + xo;
            else {
                const int plgw = 6 - plgh;
                io = (((yo >> plgh) * (pw >> plgw) + (xo >> plgw)) << 6);
            }
            const int64_t bo = io >> 7;
            if (bo < pn) { sum += prev[bo]; cnt++; }
        }
    cost[b] = cnt ? (uint32_t)(sum / (unsigned)cnt) : 0u;
}

// Order the blocks of the last launch by measured cost, most expensive first: one workgroup,
// counting sort on the cost quantised to 256 levels (max-reduce, LDS histogram, scan, scatter;
// the order inside a level is arbitrary -- any permutation is a correct launch order).
// Resets the cost array for the next measurement.
__global__ __launch_bounds__(1024) void k_sched_sort(uint32_t* __restrict__ cost,
                                                      uint32_t* __restrict__ order, int nblocks,
                                                      int xcd_map, int split, int split4,
                                                      int outlier8, int floor_ticks,
                                                      const int* __restrict__ sel, uint32_t* __restrict__ prev) {
    // dual launch (k_probe_coherence): the direct launch whose costs this would sort returned at its first
    // instruction -- nothing was measured, the order (if any) stays as it is (round 3 sorted an all-zero cost
    // array of 97 656 blocks behind every streamed 12.5 M-ray launch: 215 us of serial work per call)
    if (sel && *sel != 0) return;
    // list x: blocks whose home in the XCD-chunked map is XCD x (see k_query_direct; the blocks
    // past the last whole span are dealt round-robin there, so their home is i % 8).  Launch
    // slot j*8+x runs on XCD x, so list x fills the slots of XCD x in cost order: expensive
    // blocks first AND every block stays on the XCD (L2) that its neighbours in the image use.
    // |list x| = number of slots of XCD x because the whole spans are multiples of 8 blocks.
    __shared__ uint32_t bins[8][256];
    __shared__ uint32_t smax;
    __shared__ unsigned long long ssum;
    __shared__ uint32_t nsplit[8];       // blocks of list x that are really split (<= split)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < 8 * 256; k += 1024) (&bins[0][0])[k] = 0;
    if (tid == 0) { smax = 1; ssum = 0ull; }
    __syncthreads();
    const int T = xcd_map;
    const int nfull = T > 0 ? nblocks / (8 * T) * (8 * T) : 0;
    uint32_t m = 0;
    unsigned long long sum = 0;
    for (int i = tid; i < nblocks; i += 1024) { const uint32_t c = cost[i]; m = c > m ? c : m; sum += c; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(m, off); m = o > m ? o : m;
        sum += __shfl_xor(sum, off);
    }
    if (lane == 0) { atomicMax(&smax, m); atomicAdd(&ssum, sum); }
    __syncthreads();
    // any monotone quantisation will do: level 0 = most expensive (costs are < 2^19 ticks)
    const float scale = 255.0f / (float)smax;
    for (int i = tid; i < nblocks; i += 1024) {
        const uint32_t q = 255u - min(255u, (uint32_t)((float)cost[i] * scale));
        const int x = i < nfull ? (i / T) & 7 : i & 7;
        atomicAdd(&bins[x][q], 1u);
    }
    __syncthreads();
    // Which blocks are WORTH splitting is decided here, from the costs: a block gets extra launch slots
    // only if it sticks out -- at least outlier8 / 8 times the mean block cost -- and is long enough for
    // hand-overs to pay (floor_ticks: waves of a few dozen trips end before a thief has done anything).
    // The grid has room for `split` blocks per XCD; what is not used stays empty (sentinel entries).
    // An interior scene with evenly expensive rays splits nothing (0.050 -> 0.036 ms at 230 k rays), a
    // silhouette image splits its silhouette (profiles/r03_policy_matrix.jsonl, r03_sweep_outlier.jsonl).
    uint32_t thr = 0;
    if (outlier8 > 0) {
        const float mean = (float)ssum / (float)nblocks;
        const float t = fmaxf(mean * (float)outlier8 * 0.125f, (float)floor_ticks);
        thr = t >= 4.0e9f ? 0xffffffffu : (uint32_t)t;
    }
    // (the level of the threshold: blocks of a level are kept or dropped together)
    const uint32_t thr_level = thr > smax ? 0u : 255u - min(255u, (uint32_t)((float)thr * scale));
    if (wave < 8) {   // exclusive scan of list `wave`: 4 bins per lane
        uint32_t v[4], sum4 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = bins[wave][4 * lane + k]; sum4 += v[k]; }
        uint32_t inc = sum4;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
        uint32_t run = inc - sum4;
        // blocks at levels < thr_level (strictly more expensive than the threshold's level) qualify
        uint32_t above = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bins[wave][4 * lane + k] = run;
            if ((uint32_t)(4 * lane + k) == thr_level) above = run;
            run += v[k];
        }
        // exactly one lane holds the prefix count at thr_level
        const unsigned long long who = __ballot((uint32_t)(4 * lane) <= thr_level && thr_level < (uint32_t)(4 * lane + 4));
        const uint32_t cnt = __shfl(above, (int)__builtin_ctzll(who));
        if (lane == 0) nsplit[wave] = thr > smax ? 0u : min((uint32_t)split, outlier8 > 0 ? cnt : (uint32_t)split);
    }
    __syncthreads();
    // unused extra slots of every XCD: sentinel entries (the query kernel returns at once)
    {
        const uint32_t spmax = (uint32_t)split, q4max = (uint32_t)split4;
        const uint32_t extra_max = spmax + 2u * q4max;
        for (uint32_t k = tid; k < 8u * extra_max; k += 1024u) {
            const uint32_t x = k & 7u, e = k >> 3;
            const uint32_t sp = nsplit[x], q4 = sp >> 2;
            const uint32_t len = (uint32_t)((nblocks >> 3) + (((uint32_t)nblocks & 7u) > x ? 1 : 0));
            if (e >= sp + 2u * q4) order[(len + e) * 8u + x] = 0x07ffffffu;
        }
    }
    for (int i = tid; i < nblocks; i += 1024) {
        const uint32_t q = 255u - min(255u, (uint32_t)((float)cost[i] * scale));
        const int x = i < nfull ? (i / T) & 7 : i & 7;
        const uint32_t j = atomicAdd(&bins[x][q], 1u);
        // block splitting: the `sp` most expensive blocks of XCD x get two launch slots
        // each (halves of their rays, see k_query_direct); the launch has 8 * split slots more
        // (the first quarter of them four: quarters of their rays)
        const uint32_t sp = nsplit[x], q4 = sp >> 2;
        if (j < q4) {
            for (uint32_t k = 0; k < 4u; k++)
                order[(4u * j + k) * 8u + (uint32_t)x] = (uint32_t)i | (2u << 30) | (k << 28);
        } else if (j < sp) {
            const uint32_t p = 4u * q4 + 2u * (j - q4);
            order[p * 8u + (uint32_t)x] = (uint32_t)i | (1u << 30);
            order[(p + 1u) * 8u + (uint32_t)x] = (uint32_t)i | (1u << 30) | (1u << 28);
        } else {
            order[(j + sp + 2u * q4) * 8u + (uint32_t)x] = (uint32_t)i;
        }
        if (prev) prev[i] = cost[i];      // kept for k_sched_rescale: the next batch SHAPE starts from these
        cost[i] = 0u;
    }
    if (tid == 0) { order[TR_SCHED_MAX] = (uint32_t)nblocks; order[TR_SCHED_MAX + 1] = (uint32_t)split; }
}

template <int Q, bool STATS>
__global__ __launch_bounds__(256) void k_query_persistent(tr_bvh_view b, RayFetch rf, QueryOut out,
                                                          unsigned long long* counter,
                                                          unsigned long long* stats) {
    __shared__ int32_t ring_lds[TR_RING * 256];
    const tr_ring ring = {ring_lds + threadIdx.x, 256};
    const int lane = threadIdx.x & 63;
    tr_counters cnt = {0, 0, 0};
    for (;;) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(counter, 64ull);
        base = __shfl(base, 0);
        if ((int64_t)base >= rf.n) break;
        int64_t i = (int64_t)base + lane;
        process_ray<Q, STATS>(b, rf, out, i, i < rf.n, &cnt, ring);
    }
    flush_stats<STATS>(cnt, stats);
}

// ---- streaming launch with wave-level ray refill ("active-ray repacking") -------------------------
// For incoherent batches a wave of the direct launch runs until its slowest ray is done while most
// of its lanes finished long ago: VALU lane utilisation is 10 of 64 on C3 and on a C5(ii) shard
// (profiles/r02_c3any_summary.md, r02_c5s_summary.md) against 35-49 on coherent images.  Here
// every wave owns a contiguous range of `rays_per_wave` rays and keeps its lanes busy: whenever at
// least `refill_min` lanes are idle it stores their finished results and hands them the next rays
// of ITS OWN range -- no atomics, no work counters (the round-1 refill kernel paid three dependent
// round trips per refill: flush, atomic, ray fetch), just a wave-uniform cursor and a prefix rank
// (v_mbcnt) among the idle lanes.  One refill costs about one trip and serves >= refill_min rays.
// Results do not depend on the schedule: the per-ray state machine is the fused trip of the direct
// launch (tr_fused_step), only the lane <-> ray assignment changes.
// Which of the two launch shapes suits a large flat batch?  Coherent rays (a flattened image) run
// 1.6-1.9x faster in the direct launch (XCD-local image pieces, measured launch order, lanes that
// share nodes), incoherent ones 1.3-1.9x faster in the streaming launch
// (profiles/r02_sweep_stream*.jsonl).  One workgroup samples 256 pairs of NEIGHBOURING rays
// spread over the batch: a pair is coherent when the directions are within ~2.5 degrees and the
// origins within 1 % of the scene diagonal.  *sel = 0 (direct) when at least 3/4 of the pairs
// are, else 1 (stream).  Both kernels are then enqueued and the one not selected returns at its
// first instruction -- no host round trip; a wrong guess costs speed, never correctness.
__global__ __launch_bounds__(256) void k_probe_coherence(RayFetch rf, float scene_diag, int* __restrict__ sel) {
    __shared__ int votes;
    if (threadIdx.x == 0) votes = 0;
    __syncthreads();
    const int64_t stride = rf.n / 256 > 0 ? rf.n / 256 : 1;
    const int64_t i = (int64_t)threadIdx.x * stride;
    if (i + 1 < rf.n) {
        float o0[3], d0[3], o1[3], d1[3];
        fetch_ray(rf, i, o0, d0);
        fetch_ray(rf, i + 1, o1, d1);
        const float dd = d0[0] * d1[0] + d0[1] * d1[1] + d0[2] * d1[2];
        const float n0 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2];
        const float n1 = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
        const float ox = o1[0] - o0[0], oy = o1[1] - o0[1], oz = o1[2] - o0[2];
        const bool par = dd > 0.f && dd * dd >= 0.998f * n0 * n1;                    // cos^2 >= 0.998
        const bool near = ox * ox + oy * oy + oz * oz <= 1e-4f * scene_diag * scene_diag;
        if (par && near) atomicAdd(&votes, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *sel = votes >= 192 ? 0 : 1;
        sel[2] = 0; sel[3] = 0;        // the streaming launch's work counter (2nd 64-bit word of the slot)
    }
}

template <int Q, bool STATS, bool COMPACT, int BS, bool DEEP>
__device__ __forceinline__ void query_stream_body(const tr_bvh_view& b, const RayFetch& rf, const QueryOut& out,
                                                  int rays_per_wave, int refill_min, int xcd_map,
                                                  unsigned long long* stats,
                                                  const int* __restrict__ sel,
                                                  unsigned long long* work) {
    if (sel && *sel != 1) return;      // dual launch: this is the shape for incoherent batches (id 1)
#ifdef TR_TIMELINE
    const unsigned long long tl_start = wall_clock64();
    unsigned tl_trips = 0, tl_refills = 0;
#endif
    typedef typename tr_word<COMPACT, DEEP>::T W;
    __shared__ int32_t ring_lds[TR_RING * BS];
    const tr_ring ring = {ring_lds + threadIdx.x, BS};
    // the XCD-chunked block -> range map of the direct launch: consecutive ranges stay on one
    // XCD's L2 in chunks of `xcd_map` blocks (matters for coherent batches; speed only)
    int64_t blk = blockIdx.x;
    if (xcd_map > 0) {
        const int64_t T = xcd_map, span = 8 * T;
        const int64_t nfull = (int64_t)gridDim.x / span * span;
        if (blk < nfull) {
            const int64_t x = blk & 7, k = blk >> 3;
            blk = ((k / T) * 8 + x) * T + (k % T);
        }
    }
    const int64_t wave = blk * (BS / 64) + (threadIdx.x >> 6);
    int64_t next = wave * rays_per_wave;                 // wave-uniform cursor into the wave's range
    int64_t end = next + rays_per_wave;
    if (end > rf.n) end = rf.n;
    // work != NULL: ranges are handed out by a work counter instead (one atomic per range): a wave
    // that has used its range up takes the next one and keeps refilling, so only the very last
    // range of every wave is drained and the launch ends within one range's time for all waves
    // (the first range of a wave is still the static one: no burst of atomics at the start)
    bool exhausted = work == nullptr;
    const unsigned long long first_dynamic = (unsigned long long)gridDim.x * (BS / 64) * (unsigned long long)rays_per_wave;
    if (next >= rf.n) { next = 0; end = 0; }
    tr_counters cnt = {0, 0, 0};
    int64_t rid = -1;          // ray this lane holds (-1 none); its result is stored when the lane is refilled
    bool busy = false;         // still traversing
    tr_ray r;
    tr_state_t<W, !TR_STREAM_QN> fs;     // grid nodes: no intervals in the leaf FIFO (tr_fold_leaf)
    tr_result res;
    tr_topk<1> top;
    tr_state_init(fs);
    tr_result_init(res);
    tr_ray_setup_q(r, b.frame, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    for (;;) {
        if (!exhausted && next >= end) {
            unsigned long long base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(work, (unsigned long long)rays_per_wave) + first_dynamic;
            base = __shfl(base, 0);
            if (base >= (unsigned long long)rf.n) {
                exhausted = true;
            } else {
                next = (int64_t)base;
                end = next + rays_per_wave;
                if (end > rf.n) end = rf.n;
            }
        }
        const unsigned long long idle = __ballot(!busy);
        const int nidle = __popcll(idle);
        const bool more = next < end;
        if (!more && nidle == 64) break;
        if (more && (nidle >= refill_min || nidle == 64)) {
            if (!busy) {
                if (rid >= 0) write_result<Q>(b, out, rid, r, res);      // the finished ray of this lane
                const int64_t cand = next + lane_rank(idle);
                rid = -1;
                if (cand < end) {
                    rid = cand;
                    float o[3], d[3];
                    fetch_ray(rf, cand, o, d);
                    const bool valid = tr_ray_setup_q(r, b.frame, o[0], o[1], o[2], d[0], d[1], d[2]);
                    tr_state_init(fs);
                    tr_result_init(res);
                    if (b.num_tris >= 2) busy = valid;
                    else brute_one<Q>(b, r, valid, res);                 // no hierarchy below two triangles
                }
            }
            next += nidle;     // lanes past `end` took nothing; the cursor only has to reach `end`
#ifdef TR_TIMELINE
            tl_refills++;
#endif
        }
        // trips until the next refill is due (or, once the range is used up, until all lanes are
        // done): a plain single-exit loop like the direct launch's, with a wave-uniform exit test
        const int stop = (next < end || !exhausted) ? refill_min : 64;
        int idle_now;
        do {
            if (busy) {
                tr_fused_step<Q, 1, STATS, COMPACT, W, false, true, TR_STREAM_QN>(b, r, fs, res, top, &cnt, ring);
                busy = !tr_done(fs);
            }
            TR_CONVERGE();
#pragma unroll
            for (int a = 0; a < TR_ALTERNATE; a++) {
                if (busy) {
                    tr_fused_step<Q, 1, STATS, COMPACT, W, false, false, TR_STREAM_QN>(b, r, fs, res, top, &cnt, ring);
                    busy = !tr_done(fs);
                }
                TR_CONVERGE();
            }
            idle_now = __popcll(__ballot(!busy));
#ifdef TR_TIMELINE
            tl_trips++;
#endif
        } while (idle_now < stop);
    }
    if (rid >= 0) write_result<Q>(b, out, rid, r, res);
#ifdef TR_TIMELINE
    if ((threadIdx.x & 63) == 0 && wave < TR_TIMELINE) {
        g_timeline[wave * 4 + 0] = tl_start;
        g_timeline[wave * 4 + 1] = wall_clock64();
        g_timeline[wave * 4 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32) |
                                   (unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11));
        g_timeline[wave * 4 + 3] = (unsigned long long)(wave & 0x0fffffff) | ((unsigned long long)(tl_trips & 0xffff) << 32) |
                                   ((unsigned long long)(tl_refills & 0xffff) << 48);
    }
#endif
    flush_stats<STATS>(cnt, stats);
}
// (Round 3 ran the compact instantiations at 8 waves per SIMD, 64 registers: -2...-7 %.  Round 4's sign-selected slab
// test needs three registers more and is worth about as much on these fabric-bound launches -- C3 any -1.7 %, C5(ii)
// shard -1 %, count +1.5 % at 7 waves: profiles/r04_ab_qsign.txt -- so every instantiation keeps the compiler's budget.)
// Round 5: the fused box test's per-ray constants (tr_ray_fuse) put the kernel's refill path -- every lane's traversal
// state live across a ray fetch, a set-up and a result write -- at 85-89 registers, five waves per SIMD; asked for six
// (80 registers) the compiler parks 2-5 kernel-lifetime values in scratch (stored once in the prologue, read once per
// refill, nothing inside the trips: tests/test_round4_cpu.py) and the launch is as fast or faster than both the
// five-wave build and round 4's 72-register kernel without the fused test (C3 any 0.892 -> 0.847 ms, C5(ii) shard
// 1.845 -> 1.851, shard count 2.505 -> 2.427; seven waves: slower again -- profiles/r05_ab_stream_waves.txt).
#ifndef TR_STREAM_WAVES
#define TR_STREAM_WAVES 6
#endif
#if TR_STREAM_WAVES > 0
#define TR_STREAM_OCC __attribute__((amdgpu_waves_per_eu(TR_STREAM_WAVES, TR_STREAM_WAVES)))
#else
#define TR_STREAM_OCC
#endif
template <int Q, bool COMPACT, int BS, bool DEEP = false>
__global__ __launch_bounds__(BS) TR_STREAM_OCC void k_query_stream(tr_bvh_view b, RayFetch rf, QueryOut out,
                                                     int rays_per_wave, int refill_min, int xcd_map,
                                                     unsigned long long* stats, const int* __restrict__ sel,
                                                     unsigned long long* work) {
    query_stream_body<Q, false, COMPACT, BS, DEEP>(b, rf, out, rays_per_wave, refill_min, xcd_map, stats, sel, work);
}
// the instrumented launch (tr_trace_stats_query: three more live counters): the compiler's own register budget
template <int Q, bool COMPACT, int BS, bool DEEP = false>
__global__ __launch_bounds__(BS) void k_query_stream_stats(tr_bvh_view b, RayFetch rf, QueryOut out,
                                                           int rays_per_wave, int refill_min, int xcd_map,
                                                           unsigned long long* stats, const int* __restrict__ sel,
                                                           unsigned long long* work) {
    query_stream_body<Q, true, COMPACT, BS, DEEP>(b, rf, out, rays_per_wave, refill_min, xcd_map, stats, sel, work);
}
// ---- multi-hit second pass (shaders.cu:196-246) ----------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_location(tr_bvh_view b, RayFetch rf, int32_t cap,
                                                  const int64_t* __restrict__ offsets,
                                                  float* __restrict__ loc,
                                                  int32_t* __restrict__ ray_idx,
                                                  int32_t* __restrict__ tri_idx, int64_t ray_base) {
    __shared__ int32_t ring_lds[TR_RING * 256];
    const tr_ring ring = {ring_lds + threadIdx.x, 256};
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rf.n) return;
    float o[3], d[3];
    fetch_ray(rf, i, o, d);
    tr_ray r;
    bool valid = tr_ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]);
    tr_result res;
    tr_topk<K> top;
    tr_counters* nc = nullptr;
    if (b.num_tris >= 2) {
        wave_traverse<TR_Q_LOCATION, K, false>(b, r, valid, res, top, nc, ring);
    } else {
        top.init();
        brute_one<TR_Q_LOCATION>(b, r, valid, res);
        if (res.count) top.insert(res.best_t, res.best_face, 0);
    }
    int32_t nout = res.count < cap ? res.count : cap;
    int64_t g = offsets[i];
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (k < nout) {
            tr_tri t = tr_load_tri<false>(b, top.slot[k], nc);
            tr_hit h;
            // recompute (U, V, det) of the kept hit: same arithmetic, same values
            tr_tri_hit(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, h);
            float l3[3], uv[2];
            tr_hit_outputs(h, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, l3, uv);
            loc[3 * (g + k)] = l3[0]; loc[3 * (g + k) + 1] = l3[1]; loc[3 * (g + k) + 2] = l3[2];
            ray_idx[g + k] = (int32_t)(i + ray_base);
            tri_idx[g + k] = t.face;
        }
    }
}

// ---- fused multi-hit (memory-resident hit list, tr_topk<0>): the traversal is
// k_query_direct<TR_Q_LOCATION>; k_fill_list ranks each ray's entries and writes the rows.
// one thread per (ray, k): rank entry k among the ray's stored entries by (t_key, face), then
// re-evaluate its triangle (same arithmetic, same values) and write row offsets[i] + rank
__global__ __launch_bounds__(256) void k_fill_list(tr_bvh_view b, RayFetch rf, int32_t cap,
                                                   const int32_t* __restrict__ count,
                                                   const int64_t* __restrict__ offsets,
                                                   const tr_hit_entry* __restrict__ entries,
                                                   float* __restrict__ loc,
                                                   int32_t* __restrict__ ray_idx,
                                                   int32_t* __restrict__ tri_idx, int64_t ray_base) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = g / cap;
    const int32_t k = (int32_t)(g - i * cap);
    if (i >= rf.n) return;
    // everything that only depends on (i, k) is requested before the first use, so that the kernel
    // is two dependent memory round trips deep (entry -> triangle) instead of four
    // (count -> entry -> triangle -> offset); slots of unused entries are not dereferenced
    const tr_hit_entry* e = entries + i * cap;
    const int32_t c = count[i];
    const tr_hit_entry me = e[k];
    const int64_t off = offsets[i];
    float o[3], d[3];
    fetch_ray(rf, i, o, d);
    const int32_t ns = c < cap ? c : cap;
    if (k >= ns) return;
    tr_counters* nc = nullptr;
    const tr_tri t = tr_load_tri<false>(b, me.slot, nc);
    int32_t rank = 0;
    for (int32_t j = 0; j < ns; j++) {
        if (j == k) continue;
        const tr_hit_entry ej = e[j];
        if (ej.t_key < me.t_key || (ej.t_key == me.t_key && b.tris[ej.slot].face < t.face)) rank++;
    }
    tr_ray r;
    tr_ray_setup(r, o[0], o[1], o[2], d[0], d[1], d[2]);
    tr_hit h;
    tr_tri_hit(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, h);
    float l3[3], uv[2];
    tr_hit_outputs(h, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, l3, uv);
    const int64_t w = off + rank;
    loc[3 * w] = l3[0]; loc[3 * w + 1] = l3[1]; loc[3 * w + 2] = l3[2];
    ray_idx[w] = (int32_t)(i + ray_base);
    tri_idx[w] = t.face;
}

// ---- scans (replace the torch glue of ray.cpp:333-342 and ray_optix.py:142-144) ---------------
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK = 1024;
constexpr int SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

template <typename T>
__device__ __forceinline__ int64_t scan_value(const T* in, int64_t i, int64_t n, int32_t cap) {
    if (i >= n) return 0;
    int64_t v = (int64_t)in[i];
    if (sizeof(T) == 1) return v != 0 ? 1 : 0;
    return v < cap ? v : cap;
}

__device__ __forceinline__ int64_t block_exclusive(int64_t s, int64_t* total_out) {
    __shared__ int64_t wsum[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int64_t wpre = 0, tot = 0;
    for (int w = 0; w < SCAN_BLOCK / 64; w++) {
        int64_t x = wsum[w];
        if (w < wave) wpre += x;
        tot += x;
    }
    *total_out = tot;
    return wpre + inc - s;
}

template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partial(const T* __restrict__ in, int64_t n,
                                                             int32_t cap,
                                                             int64_t* __restrict__ partial) {
    int64_t i0 = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) s += scan_value(in, i0 + k, n, cap);
    int64_t tot;
    block_exclusive(s, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partials(int64_t* __restrict__ partial,
                                                              int64_t nblocks,
                                                              int64_t* __restrict__ total) {
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblocks; base += SCAN_BLOCK) {
        int64_t i = base + threadIdx.x;
        int64_t v = i < nblocks ? partial[i] : 0;
        int64_t tot;
        int64_t ex = block_exclusive(v, &tot);
        int64_t carry = carry_s;
        if (i < nblocks) partial[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_final(const T* __restrict__ in, int64_t n,
                                                           int32_t cap,
                                                           const int64_t* __restrict__ partial,
                                                           int64_t* __restrict__ offsets) {
    int64_t i0 = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = scan_value(in, i0 + k, n, cap); s += v[k]; }
    int64_t tot;
    int64_t ex = block_exclusive(s, &tot) + partial[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (i0 + k < n) offsets[i0 + k] = ex;
        ex += v[k];
    }
}

__global__ __launch_bounds__(256) void k_compact_closest(
    const uint8_t* __restrict__ hit, const int64_t* __restrict__ offsets, int64_t n,
    const uint8_t* __restrict__ front, const int32_t* __restrict__ tri,
    const float* __restrict__ loc, const float* __restrict__ uv, int64_t ray_base,
    uint8_t* __restrict__ front_o, int32_t* __restrict__ ray_o, int32_t* __restrict__ tri_o,
    float* __restrict__ loc_o, float* __restrict__ uv_o) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !hit[i]) return;
    int64_t j = offsets[i];
    if (front_o) front_o[j] = front[i];
    if (ray_o) ray_o[j] = (int32_t)(i + ray_base);
    if (tri_o) tri_o[j] = tri[i];
    if (loc_o) { loc_o[3 * j] = loc[3 * i]; loc_o[3 * j + 1] = loc[3 * i + 1]; loc_o[3 * j + 2] = loc[3 * i + 2]; }
    if (uv_o) { uv_o[2 * j] = uv[2 * i]; uv_o[2 * j + 1] = uv[2 * i + 1]; }
}

// ---- packed closest-hit results -> the five dense outputs (tr_closest_expand) ---------------------
// One thread per ray: {face | front << 30, u, v} -> hit, front, tri, loc, uv with tr_bary_outputs on
// the mesh's own vertex / face arrays (the arena's triangle records are verbatim copies of them), so
// the outputs carry the bits tr_intersects_closest would have written.  Any output may be NULL.
// R rays per thread, 256 apart (R = 4, round 4): the kernel is three DEPENDENT round trips deep (record -> face row ->
// vertex rows -> stores), and with one ray per thread the chip's resident threads hold 0.5 M rays of a 7 M-ray
// expansion at a time: 14 rounds x 3 round trips = 112 us = 2.5 TB/s.  Four independent rays per thread put four
// times as many loads in flight per round trip; the accesses of a wave stay as coalesced as before (lane t
// touches rays t, t + 256, ...).
template <int R>
__global__ __launch_bounds__(256) void k_closest_expand(const tr_packed_hit* __restrict__ packed, int64_t n,
                                                        const float* __restrict__ verts, int64_t nv,
                                                        const int32_t* __restrict__ faces, int64_t nf,
                                                        uint8_t* __restrict__ hit, uint8_t* __restrict__ front,
                                                        int32_t* __restrict__ tri, float* __restrict__ loc,
                                                        float* __restrict__ uv) {
    const int64_t i0 = (int64_t)blockIdx.x * (256 * R) + threadIdx.x;
    // 12-byte rows are fetched as ONE 96-bit gather each (a struct copy; three scalar element reads compile to a
    // dwordx2 + a dword: 8 instead of 4 gather instructions per ray, and the gather instructions -- one lookup per
    // distinct line each -- are what bounds this kernel: 1.8 -> TB/s, profiles/r04_emulate_run2.jsonl).  An empty
    // mesh has no row 0 to read for the misses: any valid 12 bytes will do.
    struct row3i { int32_t a, b, c; };
    struct row3f { float x, y, z; };
    const row3i* frow = reinterpret_cast<const row3i*>(nf > 0 ? (const void*)faces : (const void*)packed);
    const row3f* vrow = reinterpret_cast<const row3f*>(nv > 0 ? (const void*)verts : (const void*)packed);
    tr_packed_hit ph[R];
    row3i fi[R];
    bool ok[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int64_t i = i0 + 256 * k;
        ph[k] = packed[i < n ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint32_t face = ph[k].tri & 0x3fffffffu;
        ok[k] = !(ph[k].tri & 0x80000000u) && (int64_t)face < nf && i0 + 256 * k < n;
        fi[k] = frow[ok[k] ? face : 0u];
    }
    float va[R][9];
#pragma unroll
    for (int k = 0; k < R; k++) {
        ok[k] = ok[k] && (uint32_t)fi[k].a < (uint64_t)nv && (uint32_t)fi[k].b < (uint64_t)nv && (uint32_t)fi[k].c < (uint64_t)nv;
        const row3f a = vrow[ok[k] ? fi[k].a : 0], bb = vrow[ok[k] ? fi[k].b : 0], c = vrow[ok[k] ? fi[k].c : 0];
        va[k][0] = a.x; va[k][1] = a.y; va[k][2] = a.z; va[k][3] = bb.x; va[k][4] = bb.y; va[k][5] = bb.z;
        va[k][6] = c.x; va[k][7] = c.y; va[k][8] = c.z;
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int64_t i = i0 + 256 * k;
        if (i >= n) continue;
        float l3[3] = {0.f, 0.f, 0.f}, u2[2] = {0.f, 0.f};
        uint8_t h = 0, fr = 0;
        int32_t t = -1;
        if (ok[k]) {
            tr_bary_outputs(ph[k].u, ph[k].v, va[k][0], va[k][1], va[k][2], va[k][3], va[k][4], va[k][5], va[k][6], va[k][7], va[k][8], l3, u2);
            h = 1; fr = (ph[k].tri >> 30) & 1u; t = (int32_t)(ph[k].tri & 0x3fffffffu);
        }
        if (hit) hit[i] = h;
        if (front) front[i] = fr;
        if (tri) tri[i] = t;
        if (loc) { loc[3 * i] = l3[0]; loc[3 * i + 1] = l3[1]; loc[3 * i + 2] = l3[2]; }
        if (uv) { uv[2 * i] = u2[0]; uv[2 * i + 1] = u2[1]; }
    }
}

// (option expand4 = 1, the default)  R rays per thread, 256 apart, the face and vertex rows through BUFFER loads:
// a record that is a miss (45 % of the headline image, more of a batch of unrelated rays) gets an offset beyond the
// buffer, and a buffer load out of range returns zeros WITHOUT touching memory -- so the loads stay unconditional
// (all R rays' gathers of a thread are in flight together) and the misses cost nothing.  (With plain loads the
// misses either branch around the gathers -- one ray per thread, 2.4 TB/s -- or all read row 0: one cache line
// hammered by every CU, 2.0 TB/s; profiles/r04_expand_variants.txt.)  Grid-stride loop: `expand_cus` caps the
// grid, so that the expansion the destination rank of a sharded run does BESIDE its own trace holds a few waves per
// CU for longer instead of competing for every wave slot (triro/ray/sharded.py).
template <int R>
__global__ __launch_bounds__(256) void k_closest_expand_buf(const tr_packed_hit* __restrict__ packed, int64_t n,
                                                            const float* __restrict__ verts, int64_t nv,
                                                            const int32_t* __restrict__ faces, int64_t nf,
                                                            uint8_t* __restrict__ hit, uint8_t* __restrict__ front,
                                                            int32_t* __restrict__ tri, float* __restrict__ loc,
                                                            float* __restrict__ uv) {
    typedef int tr_v3i __attribute__((ext_vector_type(3)));
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)faces, 0, (int)(nf * 12), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)verts, 0, (int)(nv * 12), 0x00020000);
    for (int64_t i0 = (int64_t)blockIdx.x * (256 * R) + threadIdx.x; i0 < n; i0 += (int64_t)gridDim.x * (256 * R)) {
        tr_packed_hit ph[R];
        tr_v3i fi[R];
        bool ok[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = i0 + 256 * k;
            ph[k] = packed[i < n ? i : i0];
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint32_t face = ph[k].tri & 0x3fffffffu;
            ok[k] = !(ph[k].tri & 0x80000000u) && (int64_t)face < nf && i0 + 256 * k < n;
            fi[k] = __builtin_amdgcn_raw_buffer_load_b96(frs, ok[k] ? face * 12u : 0xffffffffu, 0, 0);
        }
        float va[R][9];
#pragma unroll
        for (int k = 0; k < R; k++) {
            ok[k] = ok[k] && (uint32_t)fi[k].x < (uint64_t)nv && (uint32_t)fi[k].y < (uint64_t)nv && (uint32_t)fi[k].z < (uint64_t)nv;
            const tr_v3i a = __builtin_amdgcn_raw_buffer_load_b96(vrs, ok[k] ? (uint32_t)fi[k].x * 12u : 0xffffffffu, 0, 0);
            const tr_v3i b = __builtin_amdgcn_raw_buffer_load_b96(vrs, ok[k] ? (uint32_t)fi[k].y * 12u : 0xffffffffu, 0, 0);
            const tr_v3i c = __builtin_amdgcn_raw_buffer_load_b96(vrs, ok[k] ? (uint32_t)fi[k].z * 12u : 0xffffffffu, 0, 0);
            va[k][0] = __int_as_float(a.x); va[k][1] = __int_as_float(a.y); va[k][2] = __int_as_float(a.z);
            va[k][3] = __int_as_float(b.x); va[k][4] = __int_as_float(b.y); va[k][5] = __int_as_float(b.z);
            va[k][6] = __int_as_float(c.x); va[k][7] = __int_as_float(c.y); va[k][8] = __int_as_float(c.z);
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = i0 + 256 * k;
            if (i >= n) continue;
            float l3[3] = {0.f, 0.f, 0.f}, u2[2] = {0.f, 0.f};
            uint8_t h = 0, fr = 0;
            int32_t t = -1;
            if (ok[k]) {
                tr_bary_outputs(ph[k].u, ph[k].v, va[k][0], va[k][1], va[k][2], va[k][3], va[k][4], va[k][5], va[k][6], va[k][7], va[k][8], l3, u2);
                h = 1; fr = (ph[k].tri >> 30) & 1u; t = (int32_t)(ph[k].tri & 0x3fffffffu);
            }
            if (hit) hit[i] = h;
            if (front) front[i] = fr;
            if (tri) tri[i] = t;
            if (loc) { loc[3 * i] = l3[0]; loc[3 * i + 1] = l3[1]; loc[3 * i + 2] = l3[2]; }
            if (uv) { uv[2 * i] = u2[0]; uv[2 * i + 1] = u2[1]; }
        }
    }
}

// Slot form (tr_closest_expand_slots): the record names the arena slot of the triangle; ONE 48-byte triangle record
// (three 16-byte buffer loads, out of range = a miss = no memory access) holds the vertices and the face index.
// RAYS (tr_closest_from_slots): the record is ONLY the slot (4 bytes, negative = miss) and the kernel has the rays: it
// finishes the query the way write_result does -- (det, U, V) from the ray and the winning triangle (tr_tri_duv), then
// tr_hit_outputs -- so the bits are those of a dense trace by construction; 4 instead of 12 bytes per ray cross the links.
template <bool RAYS>
__device__ __forceinline__ tr_packed_hit tr_expand_record(const void* __restrict__ rec, int64_t i) {
    if constexpr (RAYS) {
        const int32_t sl = reinterpret_cast<const int32_t*>(rec)[i];
        return tr_packed_hit{sl < 0 ? 0x80000000u : (uint32_t)sl, 0.f, 0.f};
    } else {
        return reinterpret_cast<const tr_packed_hit*>(rec)[i];
    }
}
// outputs of ray i from its record and its triangle record (q0 q1 q2: ax ay az bx | by bz cx cy | cz face . .)
template <bool RAYS, typename V4>
__device__ __forceinline__ void tr_expand_outputs(const RayFetch& rf, int64_t i, const tr_packed_hit& ph, const V4& q0, const V4& q1,
                                                  const V4& q2, float* l3, float* u2, uint8_t& fr) {
    const float ax = __int_as_float(q0.x), ay = __int_as_float(q0.y), az = __int_as_float(q0.z), bx = __int_as_float(q0.w);
    const float by = __int_as_float(q1.x), bz = __int_as_float(q1.y), cx = __int_as_float(q1.z), cy = __int_as_float(q1.w);
    const float cz = __int_as_float(q2.x);
    if constexpr (RAYS) {
        float o[3], d[3];
        fetch_ray(rf, i, o, d);
        tr_ray r;
        r.ox = o[0]; r.oy = o[1]; r.oz = o[2]; r.dx = d[0]; r.dy = d[1]; r.dz = d[2];
        tr_hit h; h.t = 0.f;
        tr_tri_duv(r, ax, ay, az, bx, by, bz, cx, cy, cz, h.det, h.U, h.V);
        tr_hit_outputs(h, ax, ay, az, bx, by, bz, cx, cy, cz, l3, u2);
        fr = h.det > 0.f ? 1 : 0;
    } else {
        tr_bary_outputs(ph.u, ph.v, ax, ay, az, bx, by, bz, cx, cy, cz, l3, u2);
        fr = (ph.tri >> 30) & 1u;
    }
}
template <int R, bool RAYS = false>
__global__ __launch_bounds__(256) void k_closest_expand_slots(const void* __restrict__ packed, int64_t n,
                                                              const tr_tri* __restrict__ tris, int64_t nt,
                                                              uint8_t* __restrict__ hit, uint8_t* __restrict__ front,
                                                              int32_t* __restrict__ tri, float* __restrict__ loc,
                                                              float* __restrict__ uv, RayFetch rf) {
    typedef int tr_v4i __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)tris, 0, (int)(nt * (int64_t)sizeof(tr_tri)), 0x00020000);
    for (int64_t i0 = (int64_t)blockIdx.x * (256 * R) + threadIdx.x; i0 < n; i0 += (int64_t)gridDim.x * (256 * R)) {
        tr_packed_hit ph[R];
        bool ok[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = i0 + 256 * k;
            ph[k] = tr_expand_record<RAYS>(packed, i < n ? i : i0);
        }
        tr_v4i q0[R], q1[R], q2[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint32_t slot = ph[k].tri & 0x3fffffffu;
            ok[k] = !(ph[k].tri & 0x80000000u) && (int64_t)slot < nt && i0 + 256 * k < n;
            const uint32_t off = ok[k] ? slot * (uint32_t)sizeof(tr_tri) : 0xffffffffu;
            q0[k] = __builtin_amdgcn_raw_buffer_load_b128(trs, off, 0, 0);
            q1[k] = __builtin_amdgcn_raw_buffer_load_b128(trs, ok[k] ? off + 16u : 0xffffffffu, 0, 0);
            q2[k] = __builtin_amdgcn_raw_buffer_load_b128(trs, ok[k] ? off + 32u : 0xffffffffu, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = i0 + 256 * k;
            if (i >= n) continue;
            float l3[3] = {0.f, 0.f, 0.f}, u2[2] = {0.f, 0.f};
            uint8_t h = 0, fr = 0;
            int32_t t = -1;
            if (ok[k]) {
                tr_expand_outputs<RAYS>(rf, i, ph[k], q0[k], q1[k], q2[k], l3, u2, fr);
                h = 1; t = q2[k].y;
            }
            if (hit) hit[i] = h;
            if (front) front[i] = fr;
            if (tri) tri[i] = t;
            if (loc) { loc[3 * i] = l3[0]; loc[3 * i + 1] = l3[1]; loc[3 * i + 2] = l3[2]; }
            if (uv) { uv[2 * i] = u2[0]; uv[2 * i + 1] = u2[1]; }
        }
    }
}

// Slot form on IMAGE-shaped rows (tr_closest_expand_slots with a row length): a wave takes 8x8 pixel tiles (four of
// them, side by side) instead of 256 pixels of one row.  A triangle of the headline image covers ~5 pixels -- about
// 2 x 2 -- so in row order every triangle record is fetched again by the waves of the rows above and below (other
// workgroups, other XCDs, other L2s: 120 bytes of fabric traffic per hit for a 48-byte record that five rays share);
// in tile order the rays that share a record sit in the same wave.  Records are read and outputs written in
// segments of 8 pixels (96 / 8 / 32 / 96 / 64 bytes): partial lines that the L2 merges.
template <bool RAYS = false, bool RAGGED = false>
__global__ __launch_bounds__(256) void k_closest_expand_slots_tiled(const void* __restrict__ packed, int64_t n, int64_t width,
                                                                    const tr_tri* __restrict__ tris, int64_t nt,
                                                                    uint8_t* __restrict__ hit, uint8_t* __restrict__ front,
                                                                    int32_t* __restrict__ tri, float* __restrict__ loc,
                                                                    float* __restrict__ uv, RayFetch rf) {
    typedef int tr_v4i __attribute__((ext_vector_type(4)));
    constexpr int R = 4;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)tris, 0, (int)(nt * (int64_t)sizeof(tr_tri)), 0x00020000);
    const int lane = threadIdx.x & 63;
    // a wave takes a block of 8 rows x 32 pixels; its R = 4 rays per lane are four strips of 2 rows x 32 pixels, so
    // that one memory instruction of the wave touches two runs of 32 pixels (hit 32 B, tri 128 B, loc 384 B, uv 256 B
    // each) -- with one 8x8 tile per instruction the runs were 8 pixels long and the pure-stream part of the kernel ran
    // at 0.072 instead of 0.046 ms -- while the block still holds the rays that share triangle records
    const uint32_t gpr = (uint32_t)(width >> 5);                      // blocks per row of blocks (32-bit: images below 2^31 pixels wide)
    const int64_t ngroups = ((n / width + 7) >> 3) * (int64_t)gpr;    // (the last row of blocks may hold fewer than 8 rows:
                                                                      // strips beyond the batch are skipped, index >= n)
    const int64_t stride = (int64_t)gridDim.x * 4;
    // wave w of the grid takes block w, then the block `stride` further on: a persistent grid (the host sizes it to what
    // is resident -- 28 672 waves of 3 us each, one per 256 rays, kept 3 of a CU's 20 wave slots busy:
    // profiles/r04_expand_pmc_tiles_vs_rows.txt), software-pipelined: the records of the NEXT block are requested
    // before the triangle records of this one are used
    auto ray_index = [&](int64_t g, int k) {
        const uint32_t gy = (uint32_t)(g / gpr), gx = (uint32_t)(g - (int64_t)gy * gpr);
        return ((int64_t)((gy << 3) + (uint32_t)(2 * k) + (uint32_t)(lane >> 5))) * width + (int64_t)((gx << 5) + (uint32_t)(lane & 31));
    };
    int64_t t0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t0 >= ngroups) return;
    int64_t idx[R];
    tr_packed_hit ph[R];
    // RAGGED (the row count is not a multiple of 8): a strip beyond the batch reads the last record -- an unconditional
    // load: a guarded one costs a branch and a full wait per strip --, fetches no triangle and stores nothing.  Its own
    // instantiation: the extra selects cost the exact shape 18 % (0.068 -> 0.080 ms on 7.3 M records).
    auto record = [&](int64_t i) { return tr_expand_record<RAYS>(packed, RAGGED && i >= n ? n - 1 : i); };
#pragma unroll
    for (int k = 0; k < R; k++) { idx[k] = ray_index(t0, k); ph[k] = record(idx[k]); }
    for (;;) {
        tr_v4i q0[R], q1[R], q2[R];
        bool ok[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const uint32_t slot = ph[k].tri & 0x3fffffffu;
            ok[k] = !(ph[k].tri & 0x80000000u) && (int64_t)slot < nt && (!RAGGED || idx[k] < n);
            const uint32_t off = ok[k] ? slot * (uint32_t)sizeof(tr_tri) : 0xffffffffu;
            q0[k] = __builtin_amdgcn_raw_buffer_load_b128(trs, off, 0, 0);
            q1[k] = __builtin_amdgcn_raw_buffer_load_b128(trs, ok[k] ? off + 16u : 0xffffffffu, 0, 0);
            q2[k] = __builtin_amdgcn_raw_buffer_load_b128(trs, ok[k] ? off + 32u : 0xffffffffu, 0, 0);
        }
        // the next group's records (wave-uniform condition)
        const int64_t t1 = t0 + stride;
        const bool more = t1 < ngroups;
        int64_t nidx[R];
        tr_packed_hit nph[R];
        if (more) {
#pragma unroll
            for (int k = 0; k < R; k++) { nidx[k] = ray_index(t1, k); nph[k] = record(nidx[k]); }
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const int64_t i = idx[k];
            if (RAGGED && i >= n) continue;
            float l3[3] = {0.f, 0.f, 0.f}, u2[2] = {0.f, 0.f};
            uint8_t h = 0, fr = 0;
            int32_t t = -1;
            if (ok[k]) {
                tr_expand_outputs<RAYS>(rf, i, ph[k], q0[k], q1[k], q2[k], l3, u2, fr);
                h = 1; t = q2[k].y;
            }
            if (hit) hit[i] = h;
            if (front) front[i] = fr;
            if (tri) tri[i] = t;
            if (loc) { loc[3 * i] = l3[0]; loc[3 * i + 1] = l3[1]; loc[3 * i + 2] = l3[2]; }
            if (uv) { uv[2 * i] = u2[0]; uv[2 * i + 1] = u2[1]; }
        }
        if (!more) break;
        t0 = t1;
#pragma unroll
        for (int k = 0; k < R; k++) { idx[k] = nidx[k]; ph[k] = nph[k]; }
    }
}

// (option expand4 = 2; measured SLOWER than one ray per thread -- 1.49 against 2.49 TB/s on 7.3 M rays,
// profiles/r04_emulate_run1.jsonl: a wave's 16-byte accesses at a 48-byte stride touch every line three times and
// the non-temporal hints keep them from merging -- kept for the record and for A/B runs.)
// Four rays per thread, every global access 16 bytes wide (round 4).  The destination rank of a ray-sharded
// run expands the records of ALL its peers (7 x the rays it traces itself at 8 GPUs) beside its own trace, so
// this kernel has to run near the memory system's rate: 48 B of records in, 104 B of outputs per thread as
// dword / dwordx4 stores (the scalar kernel above issues 1-byte and 12-byte-strided stores), streamed with
// non-temporal hints (neither the records nor the outputs are touched again by this launch, and the trace
// running next to it lives on what the L2s hold of the hierarchy); only the face / vertex gathers stay
// scalar (12-byte rows).  Needs n % 4 == 0 and 16-byte aligned rows (the host launches the scalar kernel for
// whatever does not fit); same arithmetic (tr_bary_outputs), same bits.
typedef uint32_t tr_u4 __attribute__((ext_vector_type(4)));
typedef float tr_fl4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_closest_expand4(const tr_u4* __restrict__ packed4, int64_t n4,
                                                         const float* __restrict__ verts, int64_t nv,
                                                         const int32_t* __restrict__ faces, int64_t nf,
                                                         uint32_t* __restrict__ hit4, uint32_t* __restrict__ front4,
                                                         tr_u4* __restrict__ tri4, tr_fl4* __restrict__ loc4,
                                                         tr_fl4* __restrict__ uv4) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n4) return;
    const tr_u4 p0 = __builtin_nontemporal_load(packed4 + 3 * g);
    const tr_u4 p1 = __builtin_nontemporal_load(packed4 + 3 * g + 1);
    const tr_u4 p2 = __builtin_nontemporal_load(packed4 + 3 * g + 2);
    const uint32_t w[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
    float l[12], u[8];
    uint32_t hm = 0, fm = 0;
    int32_t t[4];
    // the four face rows first, then the twelve vertex rows: the gathers of a thread are in flight together
    int32_t vi[12];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t tri = w[3 * k], face = tri & 0x3fffffffu;
        ok[k] = !(tri & 0x80000000u) && (int64_t)face < nf;
        const int32_t* fp = faces + 3 * (int64_t)(ok[k] ? face : 0u);
        vi[3 * k] = nf > 0 ? fp[0] : 0; vi[3 * k + 1] = nf > 0 ? fp[1] : 0; vi[3 * k + 2] = nf > 0 ? fp[2] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int32_t i0 = vi[3 * k], i1 = vi[3 * k + 1], i2 = vi[3 * k + 2];
        ok[k] = ok[k] && (uint32_t)i0 < (uint64_t)nv && (uint32_t)i1 < (uint64_t)nv && (uint32_t)i2 < (uint64_t)nv;
        float l3[3] = {0.f, 0.f, 0.f}, u2[2] = {0.f, 0.f};
        t[k] = -1;
        if (ok[k]) {
            const float* a = verts + 3 * (int64_t)i0; const float* b = verts + 3 * (int64_t)i1; const float* c = verts + 3 * (int64_t)i2;
            tr_bary_outputs(__uint_as_float(w[3 * k + 1]), __uint_as_float(w[3 * k + 2]), a[0], a[1], a[2], b[0], b[1], b[2],
                            c[0], c[1], c[2], l3, u2);
            hm |= 1u << (8 * k);
            fm |= ((w[3 * k] >> 30) & 1u) << (8 * k);
            t[k] = (int32_t)(w[3 * k] & 0x3fffffffu);
        }
        l[3 * k] = l3[0]; l[3 * k + 1] = l3[1]; l[3 * k + 2] = l3[2];
        u[2 * k] = u2[0]; u[2 * k + 1] = u2[1];
    }
    if (hit4) __builtin_nontemporal_store(hm, hit4 + g);
    if (front4) __builtin_nontemporal_store(fm, front4 + g);
    if (tri4) __builtin_nontemporal_store(tr_u4{(uint32_t)t[0], (uint32_t)t[1], (uint32_t)t[2], (uint32_t)t[3]}, tri4 + g);
    if (loc4) {
        __builtin_nontemporal_store(tr_fl4{l[0], l[1], l[2], l[3]}, loc4 + 3 * g);
        __builtin_nontemporal_store(tr_fl4{l[4], l[5], l[6], l[7]}, loc4 + 3 * g + 1);
        __builtin_nontemporal_store(tr_fl4{l[8], l[9], l[10], l[11]}, loc4 + 3 * g + 2);
    }
    if (uv4) {
        __builtin_nontemporal_store(tr_fl4{u[0], u[1], u[2], u[3]}, uv4 + 2 * g);
        __builtin_nontemporal_store(tr_fl4{u[4], u[5], u[6], u[7]}, uv4 + 2 * g + 1);
    }
}

// (option expand4 = 3)  1024 rays per workgroup, every global access a fully coalesced 16-byte access: the records
// are staged into LDS with consecutive lanes loading consecutive 16 bytes, thread t expands rays t, t + 256, t + 512,
// t + 768 of the tile (12-byte rows at a 12-byte stride: no bank conflicts), the outputs go back through LDS and
// leave as consecutive 16-byte stores.  Full tiles with 16-byte aligned rows only (the host sends the rest to the
// per-ray kernel).
__global__ __launch_bounds__(256) void k_closest_expand_tile(const tr_u4* __restrict__ packed16, int64_t ntiles,
                                                             const float* __restrict__ verts, int64_t nv,
                                                             const int32_t* __restrict__ faces, int64_t nf,
                                                             tr_u4* __restrict__ hit16, tr_u4* __restrict__ front16,
                                                             tr_u4* __restrict__ tri16, tr_u4* __restrict__ loc16,
                                                             tr_u4* __restrict__ uv16) {
    __shared__ tr_u4 s_rec[768];      // 1024 x 12 B: records in, loc out
    __shared__ tr_u4 s_uv[512];       // 1024 x 8 B
    __shared__ tr_u4 s_tri[256];      // 1024 x 4 B
    __shared__ tr_u4 s_hit[64], s_front[64];
    const int t = threadIdx.x;
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const tr_u4* src = packed16 + tile * 768;
    s_rec[t] = src[t]; s_rec[t + 256] = src[t + 256]; s_rec[t + 512] = src[t + 512];
    __syncthreads();
    const uint32_t* rec = reinterpret_cast<const uint32_t*>(s_rec);
    uint32_t w[4][3];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = t + 256 * k;
        w[k][0] = rec[3 * r]; w[k][1] = rec[3 * r + 1]; w[k][2] = rec[3 * r + 2];
    }
    __syncthreads();                  // all records are in registers: s_rec is free for the locations
    struct row3i { int32_t a, b, c; };
    struct row3f { float x, y, z; };
    const row3i* frow = reinterpret_cast<const row3i*>(nf > 0 ? (const void*)faces : (const void*)packed16);
    const row3f* vrow = reinterpret_cast<const row3f*>(nv > 0 ? (const void*)verts : (const void*)packed16);
    row3i fi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t face = w[k][0] & 0x3fffffffu;
        ok[k] = !(w[k][0] & 0x80000000u) && (int64_t)face < nf;
        fi[k] = frow[ok[k] ? face : 0u];
    }
    float va[4][9];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ok[k] = ok[k] && (uint32_t)fi[k].a < (uint64_t)nv && (uint32_t)fi[k].b < (uint64_t)nv && (uint32_t)fi[k].c < (uint64_t)nv;
        const row3f a = vrow[ok[k] ? fi[k].a : 0], bb = vrow[ok[k] ? fi[k].b : 0], c = vrow[ok[k] ? fi[k].c : 0];
        va[k][0] = a.x; va[k][1] = a.y; va[k][2] = a.z; va[k][3] = bb.x; va[k][4] = bb.y; va[k][5] = bb.z;
        va[k][6] = c.x; va[k][7] = c.y; va[k][8] = c.z;
    }
    float* o_loc = reinterpret_cast<float*>(s_rec);
    float* o_uv = reinterpret_cast<float*>(s_uv);
    int32_t* o_tri = reinterpret_cast<int32_t*>(s_tri);
    uint8_t* o_hit = reinterpret_cast<uint8_t*>(s_hit);
    uint8_t* o_front = reinterpret_cast<uint8_t*>(s_front);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = t + 256 * k;
        float l3[3] = {0.f, 0.f, 0.f}, u2[2] = {0.f, 0.f};
        uint8_t h = 0, fr = 0;
        int32_t tf = -1;
        if (ok[k]) {
            tr_bary_outputs(__uint_as_float(w[k][1]), __uint_as_float(w[k][2]), va[k][0], va[k][1], va[k][2], va[k][3], va[k][4],
                            va[k][5], va[k][6], va[k][7], va[k][8], l3, u2);
            h = 1; fr = (w[k][0] >> 30) & 1u; tf = (int32_t)(w[k][0] & 0x3fffffffu);
        }
        o_loc[3 * r] = l3[0]; o_loc[3 * r + 1] = l3[1]; o_loc[3 * r + 2] = l3[2];
        o_uv[2 * r] = u2[0]; o_uv[2 * r + 1] = u2[1];
        o_tri[r] = tf; o_hit[r] = h; o_front[r] = fr;
    }
    __syncthreads();
    if (loc16) { tr_u4* d = loc16 + tile * 768; d[t] = s_rec[t]; d[t + 256] = s_rec[t + 256]; d[t + 512] = s_rec[t + 512]; }
    if (uv16) { tr_u4* d = uv16 + tile * 512; d[t] = s_uv[t]; d[t + 256] = s_uv[t + 256]; }
    if (tri16) tri16[tile * 256 + t] = s_tri[t];
    if (t < 64) {
        if (hit16) hit16[tile * 64 + t] = s_hit[t];
        if (front16) front16[tile * 64 + t] = s_front[t];
    }
}

// ---- host side ----------------------------------------------------------------------------------
int make_fetch(const tr_rays* rays, RayFetch* rf) {
    if (!rays) return tr_fail(TR_ERR_INVALID_ARG, "rays == NULL");
    if (rays->nray < 0) return tr_fail(TR_ERR_INVALID_ARG, "nray < 0");
    if (rays->shape[3] != 3) return tr_fail(TR_ERR_INVALID_ARG, "last ray dimension must be 3");
    int64_t prod = 1;
    for (int k = 0; k < 3; k++) {
        int64_t s = rays->shape[k];
        if (s == INT64_MAX) continue;
        if (s < 0) return tr_fail(TR_ERR_INVALID_ARG, "negative ray dimension");
        prod *= s;
    }
    if (prod != rays->nray) return tr_fail(TR_ERR_INVALID_ARG, "nray != product of leading dims");
    if (rays->nray > 0 && (!rays->d_origins || !rays->d_directions))
        return tr_fail(TR_ERR_INVALID_ARG, "null ray pointer");
    rf->o = rays->d_origins; rf->d = rays->d_directions; rf->n = rays->nray;
    rf->s0 = rays->shape[0]; rf->s1 = rays->shape[1]; rf->s2 = rays->shape[2];
    auto classify = [&](const int64_t* st) {
        bool bcast = true, dense = st[3] == 1;
        int64_t expect = 3;
        for (int k = 2; k >= 0; k--) {
            int64_t s = rays->shape[k];
            if (s == INT64_MAX || s == 1) continue;
            if (st[k] != 0) bcast = false;
            if (st[k] != expect) dense = false;
            expect *= s;
        }
        return dense ? 0 : (bcast ? 1 : 2);
    };
    for (int k = 0; k < 4; k++) { rf->os[k] = rays->ostride[k]; rf->ds[k] = rays->dstride[k]; }
    rf->omode = classify(rays->ostride);
    rf->dmode = classify(rays->dstride);
    if (rf->s0 == INT64_MAX) rf->s0 = 1;   // keep the general path's divisions cheap and safe
    if (rf->s1 == INT64_MAX) rf->s1 = 1;
    if (rf->s2 == INT64_MAX) rf->s2 = 1;
    if (rf->s0 == 0 || rf->s1 == 0 || rf->s2 == 0) { rf->s0 = rf->s1 = rf->s2 = 1; }
    return TR_OK;
}

tr_bvh_view make_view(const tr_bvh* bvh) {
    tr_bvh_view v;
    v.nodes = bvh->nodes; v.links = bvh->links; v.tris = bvh->tris; v.num_tris = bvh->num_tris;
    v.qnodes = bvh->qnodes; v.frame = bvh->frame;
    v.top = bvh->top_table;
    return v;
}

// Every query runs on the device that owns the BVH arena, whatever device is current in the
// calling thread (tr_device_guard restores it), and refuses rays that live on another GPU:
// the kernel would dereference them (or the arena) across devices -- a memory fault that kills
// the process, or silent peer traffic.  The pointer query is skipped when the caller is already
// on the handle's device (the common, checked-by-the-binding case costs nothing extra).
int enter_bvh_device(const tr_bvh* bvh, const tr_rays* rays, tr_device_guard* guard) {
    if (guard->enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    if (guard->changed && rays && rays->nray > 0) {
        for (const float* p : {rays->d_origins, rays->d_directions}) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (attr.type == hipMemoryTypeDevice && attr.device != bvh->device)
                return tr_fail(TR_ERR_INVALID_ARG, "rays are on device " + std::to_string(attr.device) +
                                                       " but the BVH lives on device " + std::to_string(bvh->device));
        }
    }
    return TR_OK;
}

// Adaptive launch order: the blocks of the previous launch on the same (handle, stream) with the
// same block count are started most-expensive-first, so the longest rays of a batch -- its
// critical path -- do not start last.  Hints never affect results; each stream has its own
// buffers, so overlapping launches cannot see a half-written order.  Returns cost != NULL when
// this launch should record block costs (and be followed by k_sched_sort), order != NULL when a
// measured order exists for this block count.
// the slot of (stream, class) of this handle, created on first use; sched_mutex must be held.
// buf = cost[TR_SCHED_MAX] | order[TR_SCHED_MAX] | stamp of the order (block count, split blocks,
// 2 spare words) | 8 words of per-stream launch scratch (coherence-probe result, work counter)
constexpr size_t TR_SCHED_WORDS = 3 * (size_t)TR_SCHED_MAX + 4 + 8;     // cost | order | stamp | scratch | costs of the last sort
constexpr size_t TR_SCHED_PREV = 2 * (size_t)TR_SCHED_MAX + 4 + 8;
tr_sched_slot* sched_slot(tr_bvh* mb, hipStream_t stream, int cls) {
    for (int k = 0; k < TR_SCHED_SLOTS; k++)
        if (mb->sched[k].used && mb->sched[k].stream == stream && mb->sched[k].cls == cls) return &mb->sched[k];
    for (int k = 0; k < TR_SCHED_SLOTS; k++)
        if (!mb->sched[k].used) {
            uint32_t* buf = nullptr;
            if (hipMalloc((void**)&buf, sizeof(uint32_t) * TR_SCHED_WORDS) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            if (hipMemsetAsync(buf, 0, sizeof(uint32_t) * TR_SCHED_WORDS, stream) != hipSuccess) { (void)hipFree(buf); return nullptr; }
            mb->sched[k].used = true; mb->sched[k].stream = stream; mb->sched[k].cls = cls; mb->sched[k].buf = buf; mb->sched[k].nblocks = 0;
            return &mb->sched[k];
        }
    return nullptr;
}

// Eight words that belong to (handle, stream): the coherence probe's result and the streaming
// launch's work counter.  Launches of one stream are ordered, so the words are never shared by two
// launches in flight -- also not when one of them is a graph replay (a slot of the per-device ring,
// the fallback when the handle has no free slot, is reused after 512 launches on ANY stream).
uint32_t* stream_scratch(const tr_bvh* bvh, hipStream_t stream) {
    tr_bvh* mb = const_cast<tr_bvh*>(bvh);
    if (!mb->sched_mutex) return nullptr;
    std::lock_guard<std::mutex> lock(*mb->sched_mutex);
    tr_sched_slot* slot = sched_slot(mb, stream, 0);
    return slot ? slot->buf + 2 * (size_t)TR_SCHED_MAX + 4 : nullptr;
}

struct sched_shape {        // what k_sched_sort / k_sched_rescale need to know about a launch
    int64_t w, h;           // image width / height of the batch (0: not image-shaped)
    int lgh;                // rows-per-tile exponent of the block -> ray map (0: rows of 64 pixels)
    int xc;
    int64_t split, split4;
    int outlier8, floor_ticks;
};
// The launch shape a batch runs in once it has a learned order (`want`: tiles + split blocks) is a poor shape WITHOUT
// one: 8x8 tiles pack the expensive silhouette rays into the same waves, and only the split slots -- which need measured
// costs -- take those waves apart again (headline batch: 0.42 ms for tiles without an order against 0.33 ms for rows).
// So a batch shape's FIRST launch on a (handle, stream) runs in the plain shape (`plain`: the tile rule of launches
// without split blocks, no split slots), its measured block costs -- uninflated: nothing was split -- are resampled
// onto the blocks of the wanted shape (k_sched_rescale) and sorted into the order of the second launch.  And when the
// (handle, stream) has costs of ANOTHER image shape (a change of resolution), those are resampled onto the plain shape of
// the new one, so that already its first launch starts its expensive blocks first (option order_transfer).
// Returns true when this launch takes the `plain` shape.
bool sched_acquire(const tr_bvh* bvh, const tr_options& opt, hipStream_t stream, int64_t nblocks,
                   const sched_shape& want, const sched_shape& plain, const uint32_t** order, uint32_t** cost) {
    *order = nullptr;
    *cost = nullptr;
    tr_bvh* mb = const_cast<tr_bvh*>(bvh);
    if (!opt.adaptive || !mb->sched_mutex || nblocks < 64 || nblocks > TR_SCHED_MAX) return true;
    // launches with split blocks (the stealing shapes) and launches without learn separate orders:
    // their costs differ, and a plain shape would only skip the extra slots of a split order
    const int cls = want.split != 0;
    std::lock_guard<std::mutex> lock(*mb->sched_mutex);
    tr_sched_slot* slot = sched_slot(mb, stream, cls);
    if (!slot) return true;
    auto matches = [&](const sched_shape& sh) { return slot->nblocks == nblocks && slot->split == sh.split && slot->lgh == sh.lgh; };
    const sched_shape* use = &want;
    if (matches(want)) {
        *order = slot->buf + TR_SCHED_MAX;
        slot->launches++;
    } else {
        slot->launches = 0;
        const bool same_shape = want.split == plain.split && want.lgh == plain.lgh;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
        if (capturing) (void)hipGetLastError();
        // Second launch of this batch shape (the first ran `plain`, the slot is stamped with it): on to `want`; anything
        // else -- a fresh slot, another resolution -- starts in `plain`.  (Round 4 made this step depend on `lend` below:
        // a batch that is not image-shaped -- a flat [n, 3] batch with split slots -- or a launch issued under stream
        // capture never left the plain shape, never got an order and re-measured behind every launch: ADVICE r04.)
        const bool second = !same_shape && matches(plain);
        use = (second || same_shape) ? &want : &plain;
        const bool can_sort = opt.order_transfer && slot->prev_valid && !capturing;
        // the costs of the last sort can be resampled when both shapes are images and nothing is being captured
        const bool lend = can_sort && slot->prev_w > 0 && want.w > 0;
        // ... and taken as they are when the wanted shape has the very blocks they were measured on: the second launch
        // of a batch whose plain and wanted shapes share the block -> ray map (a flat batch: only the split slots differ)
        const bool reuse = can_sort && !lend && second && slot->prev_nblocks == nblocks && slot->prev_w == want.w &&
                           slot->prev_h == want.h && slot->prev_lgh == want.lgh;
        uint32_t* prev = slot->buf + TR_SCHED_PREV;
        if (lend) {
            hipLaunchKernelGGL(k_sched_rescale, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, stream, prev, slot->prev_nblocks,
                               slot->prev_w, slot->prev_h, slot->prev_lgh, slot->buf, nblocks, use->w, use->h, use->lgh);
            // borrowed costs only ORDER the launch: no block is split on their word (outlier threshold out of reach: the
            // extra launch slots stay sentinels).  The split set is sticky by design -- a split block records its cost
            // doubled so that it stays split -- and a set chosen from resampled costs stayed, and cost the terrain 18 %
            // and the shells 4 % for good (profiles/r04_first_launch_policy_debug.txt); the sort behind THIS launch picks
            // it from costs measured on this launch's own blocks.
            hipLaunchKernelGGL(k_sched_sort, dim3(1), dim3(1024), 0, stream, slot->buf, slot->buf + TR_SCHED_MAX, (int)nblocks, use->xc,
                               (int)use->split, (int)use->split4, 1 << 20, use->floor_ticks, (const int*)nullptr, (uint32_t*)nullptr);
            if (hipGetLastError() == hipSuccess) *order = slot->buf + TR_SCHED_MAX;
        } else if (reuse) {
            // measured on these blocks, nothing split, nothing resampled: they order the launch AND pick its split set
            if (hipMemcpyAsync(slot->buf, prev, sizeof(uint32_t) * (size_t)nblocks, hipMemcpyDeviceToDevice, stream) == hipSuccess) {
                hipLaunchKernelGGL(k_sched_sort, dim3(1), dim3(1024), 0, stream, slot->buf, slot->buf + TR_SCHED_MAX, (int)nblocks, use->xc,
                                   (int)use->split, (int)use->split4, use->outlier8, use->floor_ticks, (const int*)nullptr, (uint32_t*)nullptr);
                if (hipGetLastError() == hipSuccess) *order = slot->buf + TR_SCHED_MAX;
            } else {
                (void)hipGetLastError();
            }
        }
        slot->prev_valid = false;      // (the costs kept are used up; the sort behind this launch keeps new ones)
    }
    slot->nblocks = nblocks;   // the sort enqueued after the launch makes it valid for the next one
    slot->split = use->split;
    slot->lgh = use->lgh;
    // measure + re-sort after each of the first launches of a batch size, then every 4th: the
    // costs of a scene change slowly and the sort (8 us) is serial work behind every launch
    if (slot->launches < 3 || (slot->launches & 3) == 3) {
        *cost = slot->buf;
        // the sort behind this launch keeps a copy of what it sorted, in this launch's shape
        slot->prev_valid = true;
        slot->prev_nblocks = nblocks; slot->prev_w = use->w; slot->prev_h = use->h; slot->prev_lgh = use->lgh;
    }
    return use == &plain;
}

// Node flavour of a stealing closest / first launch under grid_nodes = 1.  Whether the 32-byte grid
// nodes beat the exact ones depends on how many distinct nodes the lanes of a wave are on (headline
// image -3...-5 %, 21 M triangles -8 %, the shell scene +1.5...+4 %, the interior scene +20 %), which the
// host cannot know -- so it is measured.  The two flavours differ by a few percent, so the measurement
// has to be better than that: launches 4 ... 17 of a (batch size, query) ALTERNATE between the flavours
// (even: exact, odd: grid -- clock ramps and the learning of the launch order hit both alike), the QUERY
// KERNELS of four launches of each flavour are bracketed by events (4, 6, 8, 10 and 5, 9, 13, 17: never a
// launch that also records block costs, every 4th), and when all eight have completed the grid nodes
// stay if their four took less than 98.5 % of the exact nodes' four.  The measurement is repeated 64
// launches later and is final once two agree (a third, 128 launches on, breaks a tie); a rebuild or refit
// starts over.  Never blocks: until the events are done, and while a stream is being captured, launches
// use the previous decision (exact nodes the first time).  Speed only.  *ev_before / *ev_after: events
// to record immediately before / after this launch's query kernel.
int gn_pick(const tr_bvh* bvh, hipStream_t stream, int cls, int64_t key, hipEvent_t* ev_before, hipEvent_t* ev_after) {
    *ev_before = nullptr; *ev_after = nullptr;
    tr_bvh* mb = const_cast<tr_bvh*>(bvh);
    if (!mb->sched_mutex) return 0;
    std::lock_guard<std::mutex> lock(*mb->sched_mutex);
    tr_sched_slot* t = sched_slot(mb, stream, cls);
    if (!t) return 0;
    if (t->gn_key != key) { t->gn_reset(); t->gn_key = key; }
    if (t->gn_choice >= 0) {
        if (t->gn_final) return t->gn_choice;
        if (++t->gn_since < (t->gn_rounds == 1 ? 64 : 128)) return t->gn_choice;
        t->gn_prev = t->gn_choice; t->gn_choice = -1; t->gn_count = 0;      // measure again
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return t->gn_prev > 0; }
    if (!t->gn_events) {
        for (int k = 0; k < 16; k++)
            if (hipEventCreate(&t->gn_ev[k]) != hipSuccess) { (void)hipGetLastError(); t->gn_choice = 0; t->gn_final = true; return 0; }
        t->gn_events = true;
    }
    const int c = t->gn_count++;
    if (c < 4) return t->gn_prev > 0;
    if (c <= 17) {
        const int flavour = c & 1;
        int sample = -1;
        if (!flavour && c <= 10) sample = (c - 4) >> 1;                    // exact: 4, 6, 8, 10
        if (flavour && (c & 3) == 1) sample = 4 + ((c - 5) >> 2);          // grid: 5, 9, 13, 17
        if (sample >= 0) { *ev_before = t->gn_ev[2 * sample]; *ev_after = t->gn_ev[2 * sample + 1]; }
        return flavour;
    }
    if (hipEventQuery(t->gn_ev[7]) == hipSuccess && hipEventQuery(t->gn_ev[15]) == hipSuccess) {
        float ms[8];
        bool ok = true;
        for (int k = 0; k < 8; k++) { ms[k] = 0.f; ok = ok && hipEventElapsedTime(&ms[k], t->gn_ev[2 * k], t->gn_ev[2 * k + 1]) == hipSuccess && ms[k] > 0.f; }
        const float exact = ms[0] + ms[1] + ms[2] + ms[3], grid = ms[4] + ms[5] + ms[6] + ms[7];
        const int measured = ok && grid < 0.985f * exact ? 1 : 0;
        if (!ok) (void)hipGetLastError();
        t->gn_rounds++;
        t->gn_final = !ok || t->gn_rounds >= 3 || (t->gn_rounds == 2 && measured == t->gn_prev);
        t->gn_choice = measured;
        t->gn_since = 0;
        return t->gn_choice;
    }
    (void)hipGetLastError();     // hipErrorNotReady is not an error of this call
    return t->gn_prev > 0;       // undecided: the previous measurement's flavour (exact nodes the first time)
}

// the 8-wide nodes of the handle, built if necessary (defined behind the scan kernels' host wrapper); NULL = not
// available for this launch (the caller keeps the binary streaming launch)
const tr_wnode* ensure_wide(const tr_bvh* bvh, hipStream_t stream);
int32_t* wide_spill(const tr_bvh* bvh, hipStream_t stream, size_t elems);

template <int Q, bool STATS>
int launch_query(const tr_bvh* bvh, const tr_rays* rays, const QueryOut& out,
                 unsigned long long* d_stats, hipStream_t stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    RayFetch rf;
    TR_TRY(make_fetch(rays, &rf));
    if (rf.n == 0) return TR_OK;
    tr_device_guard guard;
    TR_TRY(enter_bvh_device(bvh, rays, &guard));
    tr_device_state* st;
    TR_TRY(tr_get_device_state(bvh->device, &st));
    tr_bvh_view view = make_view(bvh);
    const tr_options opt = tr_opts();   // one snapshot per call
    // lds_top (LDS-staged node packets): 1 = at 128-thread blocks, 2 = at 256-thread blocks (where the table
    // fits beside the far-child ring without costing a wave); closest / first launches that steal on the
    // grid nodes only
    // (only the compact instantiation exists: 32-bit offsets and trail words, i.e. arrays below 4 GiB and at
    // most 32 levels; everything else keeps the ordinary launch and the caller's block size)
    const bool lt_compact = opt.compact && bvh->depth <= 32 && bvh->num_nodes * (int64_t)sizeof(tr_node) < ((int64_t)1 << 32) &&
                            bvh->num_tris * (int64_t)sizeof(tr_tri) < ((int64_t)1 << 32);
    const bool lt_query = (Q == TR_Q_CLOSEST || Q == TR_Q_FIRST) && !STATS && opt.lds_top > 0 && bvh->top_table != nullptr &&
                          bvh->num_tris >= 2 && lt_compact && !opt.persistent && (opt.block_size == 128 || opt.lds_top == 2);
    const int bs = (lt_query && opt.lds_top == 2) ? 256 : opt.block_size;
    // Image-shaped batches whose row count is not a multiple of 8 (from 64 rows on): the block -> ray map of the direct
    // launch is laid over the batch PADDED to whole 8-row tiles -- a tile row beyond the batch maps to ray indices >= n,
    // which every kernel treats as out of range -- so that such a batch keeps the tile shapes (a 1050-row band of the
    // headline image: 0.241 ms in rows of 64 pixels, 0.206 in 8x8 tiles)
    const bool rows_padded = rf.s1 > 1 && rf.s2 % 8 == 0 && rf.s2 >= 8 && rf.s2 < (1 << 28) && rf.n % rf.s2 == 0 &&
                             rf.n % (8 * rf.s2) != 0 && rf.n >= 64 * rf.s2;
    const int64_t n_map = rows_padded ? (rf.n / rf.s2 + 7) / 8 * 8 * rf.s2 : rf.n;
    const int64_t nblocks_direct = (n_map + bs - 1) / bs;
    int64_t pgrid = (int64_t)st->num_cus * opt.blocks_per_cu;
    if (opt.persistent && Q != TR_Q_LOCATION) {   // the multi-hit list query has only the direct shape
        // size the persistent grid by what is actually resident (4 waves per block = 1 per SIMD)
        static std::atomic<int> occ_a{0};   // per instantiation <Q, STATS>
        int occ = occ_a.load(std::memory_order_relaxed);
        if (occ == 0) {
            int nb = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_query_persistent<Q, STATS>, 256, 0);
            occ = (e == hipSuccess && nb > 0) ? nb : 4;
            occ_a.store(occ, std::memory_order_relaxed);
        }
        int bpc = opt.blocks_per_cu < occ ? opt.blocks_per_cu : occ;
        pgrid = (int64_t)st->num_cus * bpc;
    }
    if (opt.persistent && nblocks_direct > pgrid) {
        unsigned slot = __atomic_fetch_add(&st->next_counter, 1u, __ATOMIC_RELAXED) % (TR_NUM_COUNTERS / 8);
        unsigned long long* counter = reinterpret_cast<unsigned long long*>(st->counters) + 8 * slot;
        TR_HIP_TRY(hipMemsetAsync(counter, 0, 8 * sizeof(unsigned long long), stream));
        hipLaunchKernelGGL((k_query_persistent<Q, STATS>), dim3((unsigned)pgrid), dim3(256), 0, stream,
                           view, rf, out, counter, d_stats);
    } else {
        // 32-bit offsets when both arrays are below 4 GiB; 32-bit trail words when the hierarchy is at
        // most 32 levels high (`compact` = both, `deep` = offsets only; 128-thread blocks)
        const bool addr32 = opt.compact && bvh->num_nodes * (int64_t)sizeof(tr_node) < ((int64_t)1 << 32) &&
                            bvh->num_tris * (int64_t)sizeof(tr_tri) < ((int64_t)1 << 32);
        const bool compact = addr32 && bvh->depth <= 32;
        const bool deep = addr32 && !compact && bs == 128;
        // Streaming launch with ray refill for large incoherent batches (stream: 0 never, 1 auto,
        // 2 always).  Auto: every batch of at least 2 M rays gets BOTH launch shapes enqueued behind a
        // coherence probe that selects one on the device (k_probe_coherence): a camera image -- flat
        // or [H, W, 3] -- keeps the direct launch (and its 8x8 tiles), a batch of unrelated rays
        // takes the streaming launch whatever its tensor shape.  The multi-hit list query keeps the
        // direct launch (its per-ray list pointer belongs to a launch slot).
        const int* sel = nullptr;
        if constexpr (Q != TR_Q_LOCATION) {
            const bool auto_stream = opt.stream == 1 && rf.n >= ((int64_t)1 << 21) && bvh->num_tris >= 2;
            if (opt.stream == 2 || auto_stream) {
                unsigned long long* d_work = nullptr;
                unsigned long long* scratch = reinterpret_cast<unsigned long long*>(stream_scratch(bvh, stream));
                if (!scratch) {
                    unsigned slot = __atomic_fetch_add(&st->next_counter, 1u, __ATOMIC_RELAXED) % (TR_NUM_COUNTERS / 8);
                    scratch = reinterpret_cast<unsigned long long*>(st->counters) + 8 * slot;
                }
                if (auto_stream) {
                    int* d_sel = reinterpret_cast<int*>(scratch);
                    d_work = scratch + 1;                                    // zeroed by the probe
                    float diag2 = 0.f;
                    for (int k = 0; k < 3; k++) { const float e = bvh->aabb_max[k] - bvh->aabb_min[k]; diag2 += e * e; }
                    hipLaunchKernelGGL(k_probe_coherence, dim3(1), dim3(256), 0, stream, rf, sqrtf(diag2), d_sel);
                    sel = d_sel;
                }
                const int rpw = opt.stream_rays;
                const int64_t nwaves = (rf.n + rpw - 1) / rpw;
                unsigned grid = (unsigned)((nwaves + 1) / 2);
                int sxc = opt.xcd_chunk > 0 ? 16 : 0;       // blocks (2 ranges each) per XCD-local chunk
                while (sxc > 0 && (int64_t)sxc * 32 > grid) sxc >>= 1;
                // stream_dynamic (default): the ranges come from a work counter (the 2nd word of the
                // probe's slot, zeroed by the probe; memset when the launch is forced) and the grid is
                // what can be resident -- the static map gives every wave exactly one range
                unsigned long long* work = nullptr;
                if (opt.stream_dynamic) {
                    if (d_work == nullptr) {
                        d_work = scratch + 1;
                        TR_HIP_TRY(hipMemsetAsync(d_work, 0, sizeof(unsigned long long), stream));
                    }
                    work = d_work;
                    const unsigned resident = (unsigned)st->num_cus * 16u;
                    if (grid > resident) grid = resident;
                    sxc = 0;
                }
                bool wide_launched = false;
                {
                    // 8-wide compressed nodes (option wide): a third of the dependent fetches of the binary walk
                    // wide: 0 never, 1 always, 2 (default) where it was measured faster than the binary grid nodes: meshes
                    // from 3 M triangles on (5.2 M triangles: closest -10 %) and count launches from 1 M triangles on
                    // (C5(ii) shard count -9 %); on the 1.31 M-triangle headline mesh closest / any are equal within
                    // 1 %, on the 82 k-triangle C2 mesh the binary walk wins by 5...22 % (profiles/r04_ab_wide.txt)
                    // (round 5, with the fused box test and five waves per SIMD: the wide walk wins from 1 M triangles on for
                    // every query -- C5(ii) shard closest 1.845 -> 1.786 ms, count 2.419 -> 2.242 -- and still loses on the
                    // 82 k-triangle C2 / C3 mesh, any-hit 0.841 -> 0.929: profiles/r05_ab_wide_waves.txt)
                    const bool wide_auto = bvh->num_tris >= 1000000;
                    const tr_wnode* wn = ((opt.wide == 1 || (opt.wide == 2 && wide_auto)) && addr32) ? ensure_wide(bvh, stream) : nullptr;
                    if (wn) {
                        static std::atomic<int> wocc_a{0};   // per instantiation <Q, STATS>
                        int wocc = wocc_a.load(std::memory_order_relaxed);
                        if (wocc == 0) {
                            int nb = 0;
                            hipError_t e = STATS ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_query_wide_stats<Q>, 128, 0)
                                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_query_wide<Q>, 128, 0);
                            wocc = (e == hipSuccess && nb > 0) ? nb : 8;
                            wocc_a.store(wocc, std::memory_order_relaxed);
                        }
                        unsigned wgrid = (unsigned)((nwaves + 1) / 2);
                        if (work) { const unsigned resident = (unsigned)st->num_cus * (unsigned)wocc; if (wgrid > resident) wgrid = resident; }
                        // node-stack bound: 7 pending siblings per wide level + the 8 children of the node in hand
                        const int need = 7 * ((bvh->depth + 2) / 3) + 9;
                        const int lds_cap = opt.wide_stack < TR_WSTACK ? opt.wide_stack : TR_WSTACK;
                        const int spill_cap = need > lds_cap ? need - lds_cap : 0;
                        int32_t* spill = spill_cap > 0 ? wide_spill(bvh, stream, (size_t)wgrid * 128 * (size_t)spill_cap) : nullptr;
                        if (spill_cap == 0 || spill) {
                            if constexpr (STATS)
                                hipLaunchKernelGGL((k_query_wide_stats<Q>), dim3(wgrid), dim3(128), 0, stream, view, wn, rf, out, rpw,
                                                   opt.stream_refill, sel, work, spill, spill_cap, lds_cap, d_stats);
                            else
                                hipLaunchKernelGGL((k_query_wide<Q>), dim3(wgrid), dim3(128), 0, stream, view, wn, rf, out, rpw,
                                                   opt.stream_refill, sel, work, spill, spill_cap, lds_cap, d_stats);
                            wide_launched = true;
                        }
                    }
                }
                if (wide_launched) {
                } else {
#define TR_LAUNCH_STREAM(KERNEL)                                                                                        \
    do {                                                                                                                \
        if (compact) hipLaunchKernelGGL((KERNEL<Q, true, 128>), dim3(grid), dim3(128), 0, stream, view, rf, out, rpw,   \
                                        opt.stream_refill, sxc, d_stats, sel, work);                                   \
        else if (addr32) hipLaunchKernelGGL((KERNEL<Q, true, 128, true>), dim3(grid), dim3(128), 0, stream, view, rf,   \
                                            out, rpw, opt.stream_refill, sxc, d_stats, sel, work);                     \
        else hipLaunchKernelGGL((KERNEL<Q, false, 128>), dim3(grid), dim3(128), 0, stream, view, rf, out, rpw,          \
                                opt.stream_refill, sxc, d_stats, sel, work);                                           \
    } while (0)
                    if constexpr (STATS) TR_LAUNCH_STREAM(k_query_stream_stats);
                    else TR_LAUNCH_STREAM(k_query_stream);
#undef TR_LAUNCH_STREAM
                }
                TR_HIP_TRY(hipGetLastError());
                if (!sel) return TR_OK;
            }
        }
        // chunk size of the XCD map: the option is in units of 256 rays; at least 4 chunks per
        // XCD so that the XCDs' shares of an uneven image stay comparable
        int xc = opt.xcd_chunk * (256 / bs);
        while (xc > 0 && (int64_t)xc * 32 > nblocks_direct) xc >>= 1;
        int tile_w = 0;
        // Image-shaped batches ([..., H, W, 3] with W % 8 == 0 and a multiple of 8 rows in total):
        // a wave can take an 8x8 pixel tile instead of 64 pixels of one row.  Tiles make the
        // lanes of a wave more alike (12 % fewer wave-trips on the headline image, +19 % at
        // 16.7 M rays) but they also pack the expensive silhouette rays into waves whose 64
        // lanes all stay active through hundreds of trips, each trip then gathering 64
        // distinct nodes: the critical path of a small launch gets longer (-25 % at 1 M rays).
        // Hence, on their own, tiles only from 4 M rays on; below that together with block splitting
        // (further down), which takes those waves apart (option tile: 0 never, 1 auto, 2 always).
        // Queries without distance pruning (count, location) have no such critical path -- every ray
        // of a tile costs about the same -- and take tiles at any size: C4 count 1.12 -> 0.93 ms,
        // location 1.38 -> 1.18 ms at 1 M rays (profiles/r02_sweep_c4.jsonl).
        const bool tile_any_size = opt.tile == 2 || ((Q == TR_Q_COUNT || Q == TR_Q_LOCATION) && opt.unordered);
        if (opt.tile && (tile_any_size || rf.n >= ((int64_t)1 << 22)) && rf.s1 > 1 && rf.s2 % 8 == 0 && rf.s2 >= 8 && rf.s2 < (1 << 28) && n_map % (8 * rf.s2) == 0)
            tile_w = (int)rf.s2 | (3 << 28);
        // Smaller image-shaped batches of the pruning queries: flatter tiles where the triangles are
        // large on screen (option tile_small: 0 rows, 1 = 2x32, 2 = 4x16, 3 = 8x8, 4 = auto).  Host
        // simulation of the headline image (scripts/exp_tree_shape.py): 2x32 / 4x16 / 8x8 tiles need
        // 11 / 15 / 16 % fewer wave-trips than rows of 64 pixels, but with a triangle per pixel a
        // compact tile also packs the expensive silhouette rays into the same waves (1 M rays on
        // 1.31 M triangles: 0.270 / 0.276 / 0.321 ms for rows / 2x32 / 4x16).  With triangles of many
        // pixels that does not happen: 1 M rays on 82 k triangles 0.161 -> 0.149 ms (2x32), on 20 k
        // triangles 0.130 -> 0.109 ms (4x16) (profiles/r02_sweep_tile_density.jsonl).
        if (opt.tile && !tile_w && opt.tile_small > 0 && rf.s1 > 1 && rf.s2 < (1 << 28)) {
            int lgh = opt.tile_small;
            if (lgh == 4) lgh = rf.n >= 32 * bvh->num_tris ? 2 : (rf.n >= 8 * bvh->num_tris ? 1 : 0);
            const int w = 64 >> lgh, h = 1 << lgh;
            if (lgh > 0 && rf.s2 % w == 0 && n_map % ((int64_t)h * rf.s2) == 0) tile_w = (int)rf.s2 | (lgh << 28);
        }
        // intra-wave work stealing (wave_traverse_steal).  steal = 1 (default): closest / first / any
        // launches of up to 4 M rays, donors from their 64th trip on -- +11 % on the headline, +45 % at
        // 262 k rays, +25 % on the 4-shell scene, -3 % on 1 M incoherent rays; larger launches are
        // throughput-bound (-7 % at 10 M incoherent rays) and count loses 4 % (no culling to
        // protect, but its waves are balanced enough).  steal >= 2 forces it on, with that trip
        // threshold, for closest / first / any / count at any size (tests).
        const int steal_min = opt.steal > 1 ? opt.steal : 64;
        const bool steal = !STATS && (bs == 128 || (lt_query && bs == 256)) &&
                           ((opt.steal == 1 && rf.n <= ((int64_t)1 << 22) && (Q == TR_Q_CLOSEST || Q == TR_Q_FIRST || Q == TR_Q_ANY)) ||
                            (opt.steal > 1 && (Q == TR_Q_CLOSEST || Q == TR_Q_FIRST || Q == TR_Q_COUNT || Q == TR_Q_ANY)));
        // Unordered two-phase schedule for the queries that do not prune by distance (count, location;
        // any where stealing is not in play).  Hierarchies only (a single triangle has none).
        const bool unord = opt.unordered && !steal && bvh->num_tris >= 2 &&
                           (Q == TR_Q_COUNT || Q == TR_Q_LOCATION || (Q == TR_Q_ANY && opt.unordered > 1));
        // ... with hand-over of owed subtrees between the lanes of a wave and split launch slots (count;
        // wave_count_unordered_steal).  usteal: 0 off, 1 on, >= 2 forced with that trip threshold.
        const bool usteal = unord && Q == TR_Q_COUNT && !STATS && bs == 128 && opt.usteal > 0;
        // The direct launch on the 8-wide compressed nodes (k_query_direct_wide): its own launch shape -- no stealing, no
        // split blocks, the plain learned order.  Option wide_direct: 0 never, 1 (default) where measured faster -- the
        // multi-hit LIST query (the one unordered query that does not steal) on meshes from 500 k triangles on: terrain
        // location 0.58 -> 0.46 ms, the million-triangle cloud 3.24 -> 3.08, C4 0.934 -> 0.892, but the 82 k-triangle C2
        // mesh 0.37 -> 0.40 (profiles/r04_policy_*.txt) --, 2 count and location everywhere, 3 every query (closest / first /
        // any lose 25...50 % without stealing and splitting; count loses 2...20 % to its stealing binary launch)
        const bool wd_query = opt.wide_direct == 3 || (opt.wide_direct == 2 && (Q == TR_Q_COUNT || Q == TR_Q_LOCATION)) ||
                              (opt.wide_direct == 1 && Q == TR_Q_LOCATION && bvh->num_tris >= 500000);
        const bool use_wd = wd_query && addr32 && bs == 128 && bvh->num_tris >= 2 && !opt.persistent;
        // Block splitting: the nblocks >> N most expensive blocks of the previous launch get two launch
        // slots each.  A launch ends with its most expensive waves (scripts/exp_timeline.py: with 8x8
        // tiles everything but ~100 waves of the headline image is done after 215 us of 320), and those
        // are waves whose 64 rays ALL graze the surface, so stealing inside the wave has no idle lane to
        // give work to: half the rays per wave leaves 32 lanes that take subtrees from the first trips on
        // -- which is what makes 8x8 tiles (14 % fewer wave-trips: the launch is bound by VALU issue and
        // by gather instructions, both per trip) affordable below 4 M rays.  The first quarter of the split blocks gets four slots (C4 closest
        // 0.221 -> 0.2015 ms).  Launch shapes that steal only; the others keep their own learned order
        // (sched_acquire).  Speed only.  split: 0 off, 1 auto, N >= 2: nblocks >> N.
        const bool small_tris = rf.n >= 8 * bvh->num_tris;   // triangles of many pixels: flat tiles, balanced waves
        const bool can_tile8 = opt.tile && rf.s1 > 1 && rf.s2 % 8 == 0 && rf.s2 >= 8 && rf.s2 < (1 << 28) && n_map % (8 * rf.s2) == 0;
        int split_shift = 0;
        if ((steal || usteal) && opt.split > 1) split_shift = opt.split;
        else if ((usteal || (steal && Q != TR_Q_COUNT)) && opt.split == 1 && rf.n <= ((int64_t)1 << 22)) {
            // (the fewer waves a launch has, the more of them are worth splitting: at 262 k rays -- 2 048
            // blocks, 0.6 waves per slot of the chip -- a quarter of the blocks, 0.136 -> 0.115 ms; at
            // 410 k rays an eighth, 0.142 -> 0.125 ms: profiles/r03_sweep_small_split.jsonl)
            if (can_tile8 && !small_tris) split_shift = nblocks_direct <= 2048 ? 2 : (nblocks_direct <= 4096 ? 3 : (nblocks_direct <= 8192 ? 4 : (nblocks_direct < 32768 ? 5 : 0)));
            // ... and HALF of them while the launch then still leaves three tenths of the chip's wave slots
            // free (147 k rays of the headline image 0.099 -> 0.091 ms, of the terrain 0.098 -> 0.084, headline
            // count 0.154 -> 0.134; at 200 k rays -- 76 % of the slots -- already +4 %, at 262 k rays half the
            // blocks would fill every slot: +3...+17 %; profiles/r03_ab_split_half.txt)
            if (split_shift == 2) {
                const int64_t s1 = (nblocks_direct >> 1) / 8;
                if (2 * (nblocks_direct + 8 * (s1 + 2 * (s1 >> 2))) <= (int64_t)st->num_cus * 28 * 7 / 10) split_shift = 1;
            }
            else if (!can_tile8 && nblocks_direct <= 2048) split_shift = 4;
        }
        int64_t split = 0;
        if (use_wd) split_shift = 0;
        if (split_shift > 0 && (bs == 128 || lt_query) && nblocks_direct >= (opt.split > 1 ? 64 : 512)) split = (nblocks_direct >> split_shift) / 8;
        if (nblocks_direct + 12 * split > TR_SCHED_MAX || !opt.adaptive) split = 0;   // no learned order, no split
        int64_t split4 = split >> 2;
        int64_t split_key = split;
        // ... of which the sort behind the launch really splits the blocks that stick out of the measured
        // cost distribution (k_sched_sort): at least outlier8 / 8 times the mean block cost.  A launch that
        // leaves wave slots of the chip empty can afford to split whatever is above the mean (262 k rays of
        // the headline image: 0.136 -> 0.112 ms); one of several rounds of waves only its real outliers
        // (2.5 x the mean from 2 rounds on: headline 1 M rays 0.219 -> 0.211 ms, the interior scene at
        // 0.9 M rays 0.097 -> 0.089 ms: profiles/r03_sweep_outlier.jsonl).  split_outlier: 0 every block
        // the grid has room for, 1 this rule, >= 2 eighths.
        int outlier8 = opt.split_outlier;
        if (outlier8 == 1) {
            const double fill = 2.0 * (double)nblocks_direct / ((double)st->num_cus * 28.0);   // rounds of waves
            outlier8 = fill <= 1.0 ? 8 : (fill >= 2.0 ? 20 : 8 + (int)(12.0 * (fill - 1.0)));
        }
        const int steal_arg = steal_min | (opt.split_steal << 16);   // trip thresholds: ordinary | split blocks
        const uint32_t* order = nullptr;
        uint32_t* cost = nullptr;
        hipEvent_t gn_after = nullptr;      // node-flavour tuner: event to record behind this launch
        // ... and with split blocks the pruning queries take 8x8 tiles at any size -- once the (handle, stream) has an
        // order for that shape; the first launch of a batch shape runs without split slots in the tile shape chosen so far
        const int tile_w_plain = tile_w;
        if (split > 0 && can_tile8 && !small_tris && opt.tile_small == 4) tile_w = (int)rf.s2 | (3 << 28);
        const bool image_shaped = rf.s1 > 1 && rf.s2 >= 8 && rf.s2 < (1 << 28) && rf.n % rf.s2 == 0;
        const sched_shape want = {image_shaped ? rf.s2 : 0, image_shaped ? rf.n / rf.s2 : 0, tile_w ? (tile_w >> 28) & 3 : 0, xc,
                                  split, split4, outlier8, opt.split_floor * 100};
        sched_shape plain = want;
        if (opt.order_transfer) { plain.lgh = tile_w_plain ? (tile_w_plain >> 28) & 3 : 0; plain.split = 0; plain.split4 = 0; }
        if (!STATS && sched_acquire(bvh, opt, stream, nblocks_direct, want, plain, &order, &cost) && opt.order_transfer) {
            split = 0; split4 = 0; split_key = 0; tile_w = tile_w_plain;
        }
        const int64_t nslots = nblocks_direct + (order ? 8 * (split + 2 * split4) : 0);
        int scramble = 0;
        if (xc > 0 && opt.scramble) {
            const int64_t cnt = nblocks_direct / (8 * (int64_t)xc) * xc;   // blocks per XCD in whole spans
            for (int p : {7919, 7907, 7901})
                if (cnt > 1 && cnt % p != 0) { scramble = p; break; }
        }
#define TR_LAUNCH_DIRECT(C, B, D)                                                                          \
    hipLaunchKernelGGL((k_query_direct<Q, STATS, C, B, 0, D>), dim3((unsigned)nslots), dim3(B), 0, stream, \
                       view, rf, out, xc, scramble, tile_w, steal_min, order, (int)split_key, cost, d_stats, sel)
        bool qn_used = false;          // set by the stealing branch below
        static const bool debug_launch = getenv("TRIRO_DEBUG_LAUNCH") != nullptr;
        if (debug_launch)
            fprintf(stderr, "[triro] query %d: rays %lld blocks %lld slots %lld tile 0x%x split %lld order %d cost %d steal %d unordered %d compact %d\n",
                    Q, (long long)rf.n, (long long)nblocks_direct, (long long)nslots, (unsigned)tile_w, (long long)split,
                    order != nullptr, cost != nullptr, (int)steal, (int)unord, compact ? 1 : (deep ? 2 : 0));
        bool wd_launched = false;
        if (use_wd) {
            const tr_wnode* wn = ensure_wide(bvh, stream);
            if (wn) {
                const int need = 7 * ((bvh->depth + 2) / 3) + 9;
                const int lds_cap = opt.wide_stack < TR_WSTACK ? opt.wide_stack : TR_WSTACK;
                const int spill_cap = need > lds_cap ? need - lds_cap : 0;
                // (one spill row per lane of the GRID: above 1 GiB -- tens of millions of rays on a deep hierarchy -- the launch
                // keeps the binary shapes, whose state needs no memory)
                const size_t spill_elems = (size_t)nslots * 128 * (size_t)spill_cap;
                int32_t* spill = (spill_cap > 0 && spill_elems <= ((size_t)1 << 28)) ? wide_spill(bvh, stream, spill_elems) : nullptr;
                if (spill_cap == 0 || spill) {
                    const tr_wide_args wa = {wn, spill, spill_cap, lds_cap};
                    hipLaunchKernelGGL((k_query_direct_wide<Q, STATS>), dim3((unsigned)nslots), dim3(128), 0, stream, view, rf, out,
                                       xc, scramble, tile_w, order, (int)split_key, cost, d_stats, sel, wa);
                    wd_launched = true;
                }
            }
        }
        if (wd_launched) {
        } else if (unord) {
            if constexpr (Q == TR_Q_COUNT || Q == TR_Q_LOCATION || Q == TR_Q_ANY) {
                const int leaf_min = opt.leaf_vote;
                if constexpr (Q == TR_Q_COUNT && !STATS) {
                    if (usteal) {
                        // leaf vote | trip threshold of ordinary blocks | of split blocks
                        const int us_min = opt.usteal > 1 ? opt.usteal : 16;
                        const int uarg = (leaf_min & 0xff) | ((us_min & 0xfff) << 8) | ((opt.split_steal & 0xfff) << 20);
                        if (compact)
                            hipLaunchKernelGGL((k_query_direct<Q, false, true, 128, 3, false>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                               view, rf, out, xc, scramble, tile_w, uarg, order, (int)split_key, cost, d_stats, sel);
                        else if (deep)
                            hipLaunchKernelGGL((k_query_direct<Q, false, true, 128, 3, true>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                               view, rf, out, xc, scramble, tile_w, uarg, order, (int)split_key, cost, d_stats, sel);
                        else
                            hipLaunchKernelGGL((k_query_direct<Q, false, false, 128, 3, false>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                               view, rf, out, xc, scramble, tile_w, uarg, order, (int)split_key, cost, d_stats, sel);
                    }
                }
                if (!usteal) {
#define TR_LAUNCH_UNORD(C, B, D)                                                                            \
    hipLaunchKernelGGL((k_query_direct<Q, STATS, C, B, 2, D>), dim3((unsigned)nslots), dim3(B), 0, stream, \
                       view, rf, out, xc, scramble, tile_w, leaf_min, order, (int)split_key, cost, d_stats, sel)
                if (bs == 64) { if (compact) TR_LAUNCH_UNORD(true, 64, false); else TR_LAUNCH_UNORD(false, 64, false); }
                else if (bs == 128) { if (compact) TR_LAUNCH_UNORD(true, 128, false); else if (deep) TR_LAUNCH_UNORD(true, 128, true); else TR_LAUNCH_UNORD(false, 128, false); }
                else { if (compact) TR_LAUNCH_UNORD(true, 256, false); else TR_LAUNCH_UNORD(false, 256, false); }
                }
#undef TR_LAUNCH_UNORD
            }
        } else
        if (steal) {
            bool steal_launched = false;
            // Closest / first can walk the 32-byte grid nodes like the streaming launch does: two gathers
            // per visit instead of four.  It pays where the lanes of a wave are on different deep nodes
            // (headline -2...-5 %, 5.2 M / 21 M triangles -6 / -8 %) and costs where they share lines (C2
            // +14 %, the shell scene +1.5...+4 %); any-hit loses 3 % and keeps the exact nodes.  Option
            // grid_nodes: 0 never, 1 measured per batch (gn_pick), 2 always.
            bool qn = opt.grid_nodes == 2;
            hipEvent_t ev_before = nullptr;
            if (opt.grid_nodes == 1 && (Q == TR_Q_CLOSEST || Q == TR_Q_FIRST) && (compact || deep) && opt.adaptive)
                qn = gn_pick(bvh, stream, split > 0, nblocks_direct * 8 + Q, &ev_before, &gn_after) != 0;
            if (ev_before) (void)hipEventRecord(ev_before, stream);
            qn_used = qn && (Q == TR_Q_CLOSEST || Q == TR_Q_FIRST) && (compact || deep);
            if constexpr ((Q == TR_Q_CLOSEST || Q == TR_Q_FIRST) && !STATS) {
                if (lt_query && compact) {       // the table holds grid nodes: this launch walks them
                    qn_used = true;
                    if (bs == 256)
                        hipLaunchKernelGGL((k_query_direct<Q, false, true, 256, 1, false, true, true>), dim3((unsigned)nslots), dim3(256), 0, stream,
                                           view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
                    else
                        hipLaunchKernelGGL((k_query_direct<Q, false, true, 128, 1, false, true, true>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                           view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
                    steal_launched = true;
                }
            }
            if constexpr (Q == TR_Q_CLOSEST || Q == TR_Q_FIRST) {
                // (8 waves per SIMD only for the compact instantiation: with 64-bit trail words -- hierarchies deeper
                // than 32 levels -- the kernel does not fit 64 registers without spilling; those run at 7 waves)
                if (!steal_launched && qn && (opt.occ8 == 2 || (opt.occ8 == 1 && rf.n >= ((int64_t)1 << 21))) && compact && bs == 128) {
                    hipLaunchKernelGGL((k_query_direct_occ8<Q>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                       view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
                    steal_launched = true;
                }
                if (steal_launched) {
                } else if (qn && compact) {
                    hipLaunchKernelGGL((k_query_direct<Q, false, true, 128, 1, false, true>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                       view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
                    steal_launched = true;
                } else if (qn && deep) {
                    hipLaunchKernelGGL((k_query_direct<Q, false, true, 128, 1, true, true>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                       view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
                    steal_launched = true;
                }
            }
            if (steal_launched) {
            } else if (compact)
                hipLaunchKernelGGL((k_query_direct<Q, false, true, 128, 1>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                   view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
            else if (deep)
                hipLaunchKernelGGL((k_query_direct<Q, false, true, 128, 1, true>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                   view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
            else
                hipLaunchKernelGGL((k_query_direct<Q, false, false, 128, 1>), dim3((unsigned)nslots), dim3(128), 0, stream,
                                   view, rf, out, xc, scramble, tile_w, steal_arg, order, (int)split_key, cost, d_stats, sel);
        } else
        if (bs == 64) { if (compact) TR_LAUNCH_DIRECT(true, 64, false); else TR_LAUNCH_DIRECT(false, 64, false); }
        else if (bs == 128) { if (compact) TR_LAUNCH_DIRECT(true, 128, false); else if (deep) TR_LAUNCH_DIRECT(true, 128, true); else TR_LAUNCH_DIRECT(false, 128, false); }
        else { if (compact) TR_LAUNCH_DIRECT(true, 256, false); else TR_LAUNCH_DIRECT(false, 256, false); }
#undef TR_LAUNCH_DIRECT
        if (gn_after) (void)hipEventRecord(gn_after, stream);      // brackets the query kernel only
        if (cost)
            hipLaunchKernelGGL(k_sched_sort, dim3(1), dim3(1024), 0, stream, cost, cost + TR_SCHED_MAX,
                               (int)nblocks_direct, xc, (int)split, (int)split4, outlier8, opt.split_floor * 100, sel,
                               cost + TR_SCHED_PREV);
        if (!STATS && bvh->sched_mutex) {
            tr_bvh* mb = const_cast<tr_bvh*>(bvh);
            std::lock_guard<std::mutex> lock(*mb->sched_mutex);
            tr_launch_info& li = mb->last_launch;
            li.rays = rf.n; li.blocks = nblocks_direct; li.slots = nslots; li.query = Q;
            li.shape = wd_launched ? 4 : (unord ? (usteal ? 3 : 2) : (steal ? 1 : 0));
            li.tile_rows_lg = tile_w ? (tile_w >> 28) & 3 : 0;
            li.split_blocks = order ? (int32_t)split : 0;
            li.reserved = 0;
            li.learned_order = order != nullptr;
            li.grid_nodes = (qn_used || unord) && !wd_launched;
            li.addressing = compact ? 1 : (deep ? 2 : 0);
            mb->have_last_launch = true;
        }
    }
    TR_HIP_TRY(hipGetLastError());
    return TR_OK;
}

template <typename T>
int scan_impl(const T* d_in, int64_t n, int32_t cap, int64_t* d_offsets, int64_t* d_total,
              int64_t* h_total, hipStream_t stream) {
    if (n < 0) return tr_fail(TR_ERR_INVALID_ARG, "n < 0");
    if (!d_total) return tr_fail(TR_ERR_INVALID_ARG, "d_total == NULL");
    if (n > 0 && (!d_in || !d_offsets)) return tr_fail(TR_ERR_INVALID_ARG, "null scan pointer");
    int dev = 0;
    TR_HIP_TRY(hipGetDevice(&dev));
    tr_device_state* st;
    TR_TRY(tr_get_device_state(dev, &st));
    if (n == 0) {
        TR_HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(int64_t), stream));
    } else {
        int64_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
        // block partials: a stream-ordered allocation, so scans enqueued on different streams
        // (or from different host threads) never share scratch and nothing synchronises
        int64_t* partial = nullptr;
        const size_t bytes = sizeof(int64_t) * (size_t)nblocks;
        bool pooled = hipMallocAsync((void**)&partial, bytes, stream) == hipSuccess;
        if (!pooled) {
            (void)hipGetLastError();
            TR_HIP_TRY(hipMalloc((void**)&partial, bytes));
        }
        hipLaunchKernelGGL((k_scan_partial<T>), dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, d_in, n, cap, partial);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(SCAN_BLOCK), 0, stream, partial, nblocks, d_total);
        hipLaunchKernelGGL((k_scan_final<T>), dim3((unsigned)nblocks), dim3(SCAN_BLOCK), 0, stream, d_in, n, cap, partial, d_offsets);
        hipError_t le = hipGetLastError();
        if (pooled) {
            hipError_t fe = hipFreeAsync(partial, stream);
            if (le == hipSuccess) le = fe;
        } else {
            (void)hipStreamSynchronize(stream);   // no stream-ordered allocator: free after the work
            (void)hipFree(partial);
        }
        TR_HIP_TRY(le);
    }
    if (h_total) {
        TR_HIP_TRY(hipMemcpyAsync(h_total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        TR_HIP_TRY(hipStreamSynchronize(stream));
    }
    return TR_OK;
}

// ---- 8-wide nodes: construction on first use (traverse_wide.inc) ----------------------------------------------
const tr_wnode* ensure_wide(const tr_bvh* cbvh, hipStream_t stream) {
    tr_bvh* bvh = const_cast<tr_bvh*>(cbvh);
    if (!bvh->sched_mutex || bvh->num_nodes < 1 || bvh->num_tris < 2) return nullptr;
    std::lock_guard<std::mutex> lock(*bvh->sched_mutex);
    if (bvh->wide_valid) {
        if (stream != bvh->wide_stream && bvh->wide_event) (void)hipStreamWaitEvent(stream, bvh->wide_event, 0);
        return bvh->wnodes;
    }
    // (a hierarchy whose wide nodes cannot be built -- 4 GiB of records and more, no memory -- used to run
    // the marking rounds and a synchronising scan again behind EVERY streaming query; one failed attempt per build now)
    if (bvh->wide_unavailable) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    struct mark_failed { tr_bvh* b; bool ok = false; ~mark_failed() { if (!ok) b->wide_unavailable = true; } } outcome{bvh};
    const int64_t n = bvh->num_nodes;
    if (bvh->wtmp_cap < n) {
        if (bvh->wflag) (void)hipFree(bvh->wflag);
        if (bvh->widx) (void)hipFree(bvh->widx);
        bvh->wflag = nullptr; bvh->widx = nullptr; bvh->wtmp_cap = 0;
        if (hipMalloc((void**)&bvh->wflag, (size_t)n) != hipSuccess || hipMalloc((void**)&bvh->widx, sizeof(int64_t) * (size_t)(n + 1)) != hipSuccess) {
            (void)hipGetLastError();
            if (bvh->wflag) { (void)hipFree(bvh->wflag); bvh->wflag = nullptr; }
            return nullptr;
        }
        bvh->wtmp_cap = n;
    }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const int rounds = (bvh->depth + 2) / 3 + 1;
    for (int r = 0; r <= rounds; r++)
        hipLaunchKernelGGL(k_wide_mark, dim3(blocks), dim3(256), 0, stream, bvh->nodes, n, bvh->wflag, r);
    int64_t nw = 0;
    if (scan_impl<uint8_t>(bvh->wflag, n, 1, bvh->widx, bvh->widx + n, &nw, stream) != TR_OK || nw < 1) return nullptr;
    if (nw * (int64_t)sizeof(tr_wnode) >= ((int64_t)1 << 32)) return nullptr;      // 32-bit offsets in the kernel
    if (bvh->wcap < nw) {
        if (bvh->wnodes) (void)hipFree(bvh->wnodes);
        bvh->wnodes = nullptr; bvh->wcap = 0;
        if (hipMalloc((void**)&bvh->wnodes, sizeof(tr_wnode) * (size_t)nw) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        bvh->wcap = nw;
    }
    hipLaunchKernelGGL(k_wide_emit, dim3(blocks), dim3(256), 0, stream, bvh->nodes, n, bvh->wflag, bvh->widx, bvh->wnodes);
    if (hipGetLastError() != hipSuccess) return nullptr;
    if (!bvh->wide_event && hipEventCreateWithFlags(&bvh->wide_event, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); bvh->wide_event = nullptr; }
    if (bvh->wide_event) (void)hipEventRecord(bvh->wide_event, stream);
    else (void)hipStreamSynchronize(stream);
    bvh->wide_stream = stream;
    bvh->wcount = nw;
    bvh->wide_valid = true;
    outcome.ok = true;
    return bvh->wnodes;
}

int32_t* wide_spill(const tr_bvh* cbvh, hipStream_t stream, size_t elems) {
    tr_bvh* bvh = const_cast<tr_bvh*>(cbvh);
    if (!bvh->sched_mutex) return nullptr;
    std::lock_guard<std::mutex> lock(*bvh->sched_mutex);
    tr_sched_slot* slot = sched_slot(bvh, stream, 0);
    if (!slot) return nullptr;
    if (slot->wspill_elems < elems) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
        if (slot->wspill) { (void)hipStreamSynchronize(stream); (void)hipFree(slot->wspill); slot->wspill = nullptr; slot->wspill_elems = 0; }
        if (hipMalloc((void**)&slot->wspill, elems * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        slot->wspill_elems = elems;
    }
    return slot->wspill;
}

}  // namespace

void tr_wide_rebuild(tr_bvh* bvh, hipStream_t stream) {
    if (bvh && bvh->wnodes && !bvh->wide_valid && bvh->num_tris >= 2) (void)ensure_wide(bvh, stream);
}

extern "C" {

int tr_intersects_any(const tr_bvh* bvh, const tr_rays* rays, uint8_t* d_hit, void* stream) {
    if (!d_hit && rays && rays->nray > 0) return tr_fail(TR_ERR_INVALID_ARG, "d_hit == NULL");
    QueryOut out = {d_hit, nullptr, nullptr, nullptr, nullptr, nullptr};
    return launch_query<TR_Q_ANY, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

int tr_intersects_first(const tr_bvh* bvh, const tr_rays* rays, int32_t* d_tri, void* stream) {
    if (!d_tri && rays && rays->nray > 0) return tr_fail(TR_ERR_INVALID_ARG, "d_tri == NULL");
    QueryOut out = {nullptr, nullptr, d_tri, nullptr, nullptr, nullptr};
    return launch_query<TR_Q_FIRST, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

int tr_intersects_closest(const tr_bvh* bvh, const tr_rays* rays, uint8_t* d_hit, uint8_t* d_front,
                          int32_t* d_tri, float* d_loc, float* d_uv, void* stream) {
    if (rays && rays->nray > 0 && (!d_hit || !d_front || !d_tri || !d_loc || !d_uv))
        return tr_fail(TR_ERR_INVALID_ARG, "null output pointer");
    QueryOut out = {d_hit, d_front, d_tri, d_loc, d_uv, nullptr};
    return launch_query<TR_Q_CLOSEST, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

int tr_intersects_closest_packed(const tr_bvh* bvh, const tr_rays* rays, tr_packed_hit* d_packed, void* stream) {
    if (rays && rays->nray > 0 && !d_packed) return tr_fail(TR_ERR_INVALID_ARG, "d_packed == NULL");
    if (bvh && bvh->num_tris >= ((int64_t)1 << 30)) return tr_fail(TR_ERR_INVALID_ARG, "packed results hold face indices below 2^30");
    QueryOut out = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, d_packed};
    return launch_query<TR_Q_CLOSEST, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

int tr_closest_expand(const tr_packed_hit* d_packed, int64_t n, const float* d_vertices, int64_t nv,
                      const int32_t* d_faces, int64_t nf, uint8_t* d_hit, uint8_t* d_front, int32_t* d_tri,
                      float* d_loc, float* d_uv, void* stream) {
    if (n < 0 || nv < 0 || nf < 0) return tr_fail(TR_ERR_INVALID_ARG, "negative size");
    if (n == 0) return TR_OK;
    if (!d_packed) return tr_fail(TR_ERR_INVALID_ARG, "d_packed == NULL");
    if (nf > 0 && (!d_vertices || !d_faces)) return tr_fail(TR_ERR_INVALID_ARG, "null mesh pointer");
    // expand4: 0 one ray per thread, 1 (default) four rays per thread 256 apart, 2 four adjacent rays per thread with
    // 16-byte accesses for the aligned body (k_closest_expand4), 3 tiles of 1024 rays staged through LDS, every
    // global access coalesced (k_closest_expand_tile; aligned full tiles, the rest as mode 1)
    const int mode = tr_opts().expand4;
    int64_t done = 0;
    if (mode == 2) {
        const uintptr_t align_bits = (uintptr_t)d_packed | (uintptr_t)d_tri | (uintptr_t)d_loc | (uintptr_t)d_uv |
                                     (((uintptr_t)d_hit | (uintptr_t)d_front) << 2);
        const int64_t n4 = (align_bits & 15) == 0 ? n / 4 : 0;
        if (n4 > 0)
            hipLaunchKernelGGL(k_closest_expand4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const tr_u4*>(d_packed), n4, d_vertices, nv, d_faces, nf,
                               reinterpret_cast<uint32_t*>(d_hit), reinterpret_cast<uint32_t*>(d_front),
                               reinterpret_cast<tr_u4*>(d_tri), reinterpret_cast<tr_fl4*>(d_loc), reinterpret_cast<tr_fl4*>(d_uv));
        done = 4 * n4;
    } else if (mode == 3) {
        const uintptr_t align_bits = (uintptr_t)d_packed | (uintptr_t)d_tri | (uintptr_t)d_loc | (uintptr_t)d_uv |
                                     (uintptr_t)d_hit | (uintptr_t)d_front;
        const int64_t ntiles = (align_bits & 15) == 0 ? n / 1024 : 0;
        if (ntiles > 0)
            hipLaunchKernelGGL(k_closest_expand_tile, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream,
                               reinterpret_cast<const tr_u4*>(d_packed), ntiles, d_vertices, nv, d_faces, nf,
                               reinterpret_cast<tr_u4*>(d_hit), reinterpret_cast<tr_u4*>(d_front),
                               reinterpret_cast<tr_u4*>(d_tri), reinterpret_cast<tr_u4*>(d_loc), reinterpret_cast<tr_u4*>(d_uv));
        done = 1024 * ntiles;
    }
    const int64_t rest = n - done;
    if (rest > 0) {
        const tr_packed_hit* pp = d_packed + done;
        uint8_t* ph = d_hit ? d_hit + done : nullptr; uint8_t* pf = d_front ? d_front + done : nullptr;
        int32_t* pt = d_tri ? d_tri + done : nullptr;
        float* pl = d_loc ? d_loc + 3 * done : nullptr; float* pu = d_uv ? d_uv + 2 * done : nullptr;
        const bool buf_ok = nf * 12 < ((int64_t)1 << 31) && nv * 12 < ((int64_t)1 << 31) && nf > 0 && nv > 0;
        if (mode == 1 && rest >= 4096 && buf_ok) {
            int64_t blocks = (rest + 1023) / 1024;
            const tr_options o2 = tr_opts();
            if (o2.expand_cus > 0) {
                int dev = 0;
                tr_device_state* st = nullptr;
                if (hipGetDevice(&dev) == hipSuccess && tr_get_device_state(dev, &st) == TR_OK && blocks > (int64_t)st->num_cus * o2.expand_cus)
                    blocks = (int64_t)st->num_cus * o2.expand_cus;
            }
            hipLaunchKernelGGL(k_closest_expand_buf<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                               pp, rest, d_vertices, nv, d_faces, nf, ph, pf, pt, pl, pu);
        } else if (mode != 0 && rest >= 4096)
            hipLaunchKernelGGL(k_closest_expand<4>, dim3((unsigned)((rest + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                               pp, rest, d_vertices, nv, d_faces, nf, ph, pf, pt, pl, pu);
        else
            hipLaunchKernelGGL(k_closest_expand<1>, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               pp, rest, d_vertices, nv, d_faces, nf, ph, pf, pt, pl, pu);
    }
    TR_HIP_TRY(hipGetLastError());
    return TR_OK;
}

int tr_intersects_closest_packed_slots(const tr_bvh* bvh, const tr_rays* rays, tr_packed_hit* d_packed, void* stream) {
    if (rays && rays->nray > 0 && !d_packed) return tr_fail(TR_ERR_INVALID_ARG, "d_packed == NULL");
    if (bvh && bvh->num_tris >= ((int64_t)1 << 30)) return tr_fail(TR_ERR_INVALID_ARG, "packed results hold triangle slots below 2^30");
    QueryOut out = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, d_packed, 1};
    return launch_query<TR_Q_CLOSEST, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

extern "C++" {
template <bool RAYS>
static int expand_slots_impl(const tr_bvh* bvh, const void* d_rec, int64_t n, int64_t row_length, const RayFetch& rf, uint8_t* d_hit,
                             uint8_t* d_front, int32_t* d_tri, float* d_loc, float* d_uv, void* stream) {
    if (bvh->num_tris * (int64_t)sizeof(tr_tri) >= ((int64_t)1 << 31)) return tr_fail(TR_ERR_INVALID_ARG, "slot form needs a triangle array below 2 GiB");
    tr_device_state* st;
    TR_TRY(tr_get_device_state(bvh->device, &st));
    const tr_options opt = tr_opts();
    // image-shaped rows (row_length pixels each): blocks of 8 rows x 32 pixels per wave, so that the rays that share a
    // triangle record share a wave (option expand_tiles).  A row count that is not a multiple of 8 leaves the last row
    // of blocks partly empty: taken from 64 rows on.
    const bool tiled = opt.expand_tiles && row_length >= 32 && row_length % 32 == 0 && n % row_length == 0 && n >= 4096 &&
                       (n % (8 * row_length) == 0 || n >= 64 * row_length);
    if (tiled) {
        const int64_t n_map = (n / row_length + 7) / 8 * 8 * row_length;
        int64_t blocks = (n_map / 256 + 3) / 4;               // 4 waves per workgroup, one block of 8 x 32 pixels per wave and pass
        // one wave per block of 8 rows x 32 pixels (expand_cus = N > 0: at most N workgroups per CU, the waves loop;
        // measured on 7.3 M records of the headline image: no cap 0.066 ms, 8 per CU 0.072, what is resident 0.078)
        if (opt.expand_cus > 0 && blocks > (int64_t)st->num_cus * opt.expand_cus) blocks = (int64_t)st->num_cus * opt.expand_cus;
        if (n_map == n)
            hipLaunchKernelGGL((k_closest_expand_slots_tiled<RAYS, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                               d_rec, n, row_length, bvh->tris, bvh->num_tris, d_hit, d_front, d_tri, d_loc, d_uv, rf);
        else
            hipLaunchKernelGGL((k_closest_expand_slots_tiled<RAYS, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                               d_rec, n, row_length, bvh->tris, bvh->num_tris, d_hit, d_front, d_tri, d_loc, d_uv, rf);
    } else if (n >= 4096) {
        int64_t blocks = (n + 1023) / 1024;
        if (opt.expand_cus > 0 && blocks > (int64_t)st->num_cus * opt.expand_cus) blocks = (int64_t)st->num_cus * opt.expand_cus;
        hipLaunchKernelGGL((k_closest_expand_slots<4, RAYS>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           d_rec, n, bvh->tris, bvh->num_tris, d_hit, d_front, d_tri, d_loc, d_uv, rf);
    } else {
        hipLaunchKernelGGL((k_closest_expand_slots<1, RAYS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           d_rec, n, bvh->tris, bvh->num_tris, d_hit, d_front, d_tri, d_loc, d_uv, rf);
    }
    TR_HIP_TRY(hipGetLastError());
    return TR_OK;
}
}   // extern "C++"

int tr_closest_expand_slots_rows(const tr_bvh* bvh, const tr_packed_hit* d_packed, int64_t n, int64_t row_length, uint8_t* d_hit,
                                 uint8_t* d_front, int32_t* d_tri, float* d_loc, float* d_uv, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    if (n < 0 || row_length < 0) return tr_fail(TR_ERR_INVALID_ARG, "negative size");
    if (n == 0) return TR_OK;
    if (!d_packed) return tr_fail(TR_ERR_INVALID_ARG, "d_packed == NULL");
    tr_device_guard guard;
    TR_TRY(enter_bvh_device(bvh, nullptr, &guard));
    RayFetch none{};
    return expand_slots_impl<false>(bvh, d_packed, n, row_length, none, d_hit, d_front, d_tri, d_loc, d_uv, stream);
}

// The 4-byte record form (round 4): the traversal writes only the arena slot of the winning triangle (or -1), and
// whoever holds the RAYS finishes the query from (ray, slot) -- what write_result does at the end of a dense trace.
int tr_intersects_closest_slots(const tr_bvh* bvh, const tr_rays* rays, int32_t* d_slot, void* stream) {
    if (!d_slot && rays && rays->nray > 0) return tr_fail(TR_ERR_INVALID_ARG, "d_slot == NULL");
    if (bvh && bvh->num_tris * (int64_t)sizeof(tr_tri) >= ((int64_t)1 << 31))
        return tr_fail(TR_ERR_INVALID_ARG, "slot form needs a triangle array below 2 GiB");
    QueryOut out = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, reinterpret_cast<tr_packed_hit*>(d_slot), 2};
    return launch_query<TR_Q_CLOSEST, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

int tr_closest_from_slots(const tr_bvh* bvh, const tr_rays* rays, const int32_t* d_slot, int64_t row_length, uint8_t* d_hit,
                          uint8_t* d_front, int32_t* d_tri, float* d_loc, float* d_uv, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    if (row_length < 0) return tr_fail(TR_ERR_INVALID_ARG, "negative size");
    RayFetch rf;
    TR_TRY(make_fetch(rays, &rf));
    if (rf.n == 0) return TR_OK;
    if (!d_slot) return tr_fail(TR_ERR_INVALID_ARG, "d_slot == NULL");
    tr_device_guard guard;
    TR_TRY(enter_bvh_device(bvh, rays, &guard));
    return expand_slots_impl<true>(bvh, d_slot, rf.n, row_length, rf, d_hit, d_front, d_tri, d_loc, d_uv, stream);
}

int tr_closest_expand_slots(const tr_bvh* bvh, const tr_packed_hit* d_packed, int64_t n, uint8_t* d_hit, uint8_t* d_front,
                            int32_t* d_tri, float* d_loc, float* d_uv, void* stream) {
    return tr_closest_expand_slots_rows(bvh, d_packed, n, 0, d_hit, d_front, d_tri, d_loc, d_uv, stream);
}

int tr_intersects_count(const tr_bvh* bvh, const tr_rays* rays, int32_t* d_count, void* stream) {
    if (!d_count && rays && rays->nray > 0) return tr_fail(TR_ERR_INVALID_ARG, "d_count == NULL");
    QueryOut out = {nullptr, nullptr, nullptr, nullptr, nullptr, d_count};
    return launch_query<TR_Q_COUNT, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

int tr_hits_scan(const int32_t* d_count, int64_t n, int32_t cap, int64_t* d_offsets,
                 int64_t* d_total, int64_t* h_total, void* stream) {
    if (cap < 0) return tr_fail(TR_ERR_INVALID_ARG, "cap < 0");
    return scan_impl<int32_t>(d_count, n, cap, d_offsets, d_total, h_total, (hipStream_t)stream);
}

int tr_mask_scan(const uint8_t* d_hit, int64_t n, int64_t* d_offsets, int64_t* d_total,
                 int64_t* h_total, void* stream) {
    return scan_impl<uint8_t>(d_hit, n, 1, d_offsets, d_total, h_total, (hipStream_t)stream);
}

int tr_intersects_location_fill(const tr_bvh* bvh, const tr_rays* rays, int32_t cap,
                                const int64_t* d_offsets, float* d_loc, int32_t* d_ray_idx,
                                int32_t* d_tri_idx, int64_t ray_base, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    if (cap < 0 || cap > TR_MAX_HITS_CAP) return tr_fail(TR_ERR_INVALID_ARG, "cap out of range");
    RayFetch rf;
    TR_TRY(make_fetch(rays, &rf));
    if (rf.n == 0 || cap == 0) return TR_OK;
    if (!d_offsets) return tr_fail(TR_ERR_INVALID_ARG, "d_offsets == NULL");
    tr_device_guard guard;
    TR_TRY(enter_bvh_device(bvh, rays, &guard));
    tr_bvh_view view = make_view(bvh);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((rf.n + 255) / 256)), block(256);
    if (cap <= 8)
        hipLaunchKernelGGL(k_location<8>, grid, block, 0, s, view, rf, cap, d_offsets, d_loc, d_ray_idx, d_tri_idx, ray_base);
    else if (cap <= 16)
        hipLaunchKernelGGL(k_location<16>, grid, block, 0, s, view, rf, cap, d_offsets, d_loc, d_ray_idx, d_tri_idx, ray_base);
    else
        hipLaunchKernelGGL(k_location<32>, grid, block, 0, s, view, rf, cap, d_offsets, d_loc, d_ray_idx, d_tri_idx, ray_base);
    TR_HIP_TRY(hipGetLastError());
    return TR_OK;
}

int tr_intersects_count_topk(const tr_bvh* bvh, const tr_rays* rays, int32_t cap, int32_t* d_count,
                             tr_hit_entry* d_hits, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    if (cap < 1 || cap > TR_MAX_HITS_CAP) return tr_fail(TR_ERR_INVALID_ARG, "cap out of range");
    RayFetch rf;
    TR_TRY(make_fetch(rays, &rf));
    if (rf.n == 0) return TR_OK;
    if (!d_count || !d_hits) return tr_fail(TR_ERR_INVALID_ARG, "null output pointer");
    QueryOut out = {nullptr, nullptr, nullptr, nullptr, nullptr, d_count, d_hits, cap};
    return launch_query<TR_Q_LOCATION, false>(bvh, rays, out, nullptr, (hipStream_t)stream);
}

int tr_location_fill_slots(const tr_bvh* bvh, const tr_rays* rays, int32_t cap, const int32_t* d_count,
                           const int64_t* d_offsets, const tr_hit_entry* d_hits, float* d_loc,
                           int32_t* d_ray_idx, int32_t* d_tri_idx, int64_t ray_base, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    if (cap < 1 || cap > TR_MAX_HITS_CAP) return tr_fail(TR_ERR_INVALID_ARG, "cap out of range");
    RayFetch rf;
    TR_TRY(make_fetch(rays, &rf));
    if (rf.n == 0) return TR_OK;
    if (!d_count || !d_offsets || !d_hits) return tr_fail(TR_ERR_INVALID_ARG, "null input pointer");
    tr_device_guard guard;
    TR_TRY(enter_bvh_device(bvh, rays, &guard));
    tr_bvh_view view = make_view(bvh);
    const int64_t threads = rf.n * cap;
    hipLaunchKernelGGL(k_fill_list, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       view, rf, cap, d_count, d_offsets, d_hits, d_loc, d_ray_idx, d_tri_idx, ray_base);
    TR_HIP_TRY(hipGetLastError());
    return TR_OK;
}

int tr_compact_closest(const uint8_t* d_hit, const int64_t* d_offsets, int64_t n,
                       const uint8_t* d_front, const int32_t* d_tri, const float* d_loc,
                       const float* d_uv, int64_t ray_base, uint8_t* d_front_out,
                       int32_t* d_ray_idx_out, int32_t* d_tri_out, float* d_loc_out,
                       float* d_uv_out, void* stream) {
    if (n < 0) return tr_fail(TR_ERR_INVALID_ARG, "n < 0");
    if (n == 0) return TR_OK;
    if (!d_hit || !d_offsets) return tr_fail(TR_ERR_INVALID_ARG, "null mask/offsets");
    if ((d_front_out && !d_front) || (d_tri_out && !d_tri) || (d_loc_out && !d_loc) || (d_uv_out && !d_uv))
        return tr_fail(TR_ERR_INVALID_ARG, "output requested without its input");
    hipLaunchKernelGGL(k_compact_closest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_hit, d_offsets, n, d_front, d_tri, d_loc, d_uv, ray_base, d_front_out,
                       d_ray_idx_out, d_tri_out, d_loc_out, d_uv_out);
    TR_HIP_TRY(hipGetLastError());
    return TR_OK;
}

int tr_trace_stats_query(const tr_bvh* bvh, const tr_rays* rays, int query, tr_trace_stats* h_stats,
                   void* stream) {
    if (!bvh || !rays || !h_stats) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    if (query < TR_Q_ANY || query > TR_Q_LOCATION) return tr_fail(TR_ERR_INVALID_ARG, "unknown query id");
    hipStream_t s = (hipStream_t)stream;
    tr_device_guard guard;
    TR_TRY(enter_bvh_device(bvh, rays, &guard));
    int64_t n = rays->nray;
    unsigned long long* d_stats = nullptr;
    uint8_t* buf = nullptr;
    // scratch outputs of the instrumented launch: closest 26 B/ray, multi-hit 4 + 8*cap B/ray
    const size_t per_ray = 4 + 8 * (size_t)TR_MAX_ANYHIT_SIZE;
    TR_HIP_TRY(hipMalloc((void**)&d_stats, 64));
    hipError_t e = hipMalloc((void**)&buf, per_ray * (size_t)(n > 0 ? n : 1) + 64);
    if (e != hipSuccess) { (void)hipFree(d_stats); return tr_fail(TR_ERR_OUT_OF_MEMORY, "stats outputs"); }
    size_t nn = (size_t)(n > 0 ? n : 1);
    float* loc = (float*)buf;                       // 12 n
    float* uv = loc + 3 * nn;                       // 8 n
    int32_t* tri = (int32_t*)(uv + 2 * nn);         // 4 n
    uint8_t* hitp = (uint8_t*)(tri + nn);           // n
    uint8_t* front = hitp + nn;                     // n
    int status = TR_OK;
    if (hipMemsetAsync(d_stats, 0, 64, s) != hipSuccess) status = tr_fail(TR_ERR_HIP, "memset stats");
    if (status == TR_OK) {
        QueryOut out = {hitp, front, tri, loc, uv, nullptr, nullptr, 0};
        switch (query) {
            case TR_Q_ANY: status = launch_query<TR_Q_ANY, true>(bvh, rays, out, d_stats, s); break;
            case TR_Q_FIRST: status = launch_query<TR_Q_FIRST, true>(bvh, rays, out, d_stats, s); break;
            case TR_Q_CLOSEST: status = launch_query<TR_Q_CLOSEST, true>(bvh, rays, out, d_stats, s); break;
            case TR_Q_COUNT: out.count = (int32_t*)buf; status = launch_query<TR_Q_COUNT, true>(bvh, rays, out, d_stats, s); break;
            default:
                out.count = (int32_t*)buf;
                out.hits = (tr_hit_entry*)(buf + 4 * ((nn + 1) / 2 * 2));   // 8-byte aligned
                out.cap = TR_MAX_ANYHIT_SIZE;
                status = launch_query<TR_Q_LOCATION, true>(bvh, rays, out, d_stats, s);
        }
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    if (status == TR_OK && hipMemcpyAsync(h, d_stats, 32, hipMemcpyDeviceToHost, s) != hipSuccess)
        status = tr_fail(TR_ERR_HIP, "memcpy stats");
    if (hipStreamSynchronize(s) != hipSuccess && status == TR_OK) status = tr_fail(TR_ERR_HIP, "sync stats");
    (void)hipFree(d_stats);
    (void)hipFree(buf);
    h_stats->rays = (uint64_t)n; h_stats->node_visits = h[1]; h_stats->tri_tests = h[2]; h_stats->climb_steps = h[3];
    return status;
}

int tr_trace_stats_closest(const tr_bvh* bvh, const tr_rays* rays, tr_trace_stats* h_stats, void* stream) {
    return tr_trace_stats_query(bvh, rays, TR_Q_CLOSEST, h_stats, stream);
}

}  // extern "C"

#ifdef TR_USTEAL_DEBUG
extern "C" int tr_debug_usteal(unsigned* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_usteal_debug), 16);
}
#endif
#ifdef TR_TIMELINE
extern "C" int tr_debug_timeline(unsigned long long* host_out, long long n_waves) {
    if (n_waves > TR_TIMELINE) n_waves = TR_TIMELINE;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline), (size_t)n_waves * 32);
}
#endif
#ifdef TR_DEBUG_SCHED
// experiment builds only (not part of the ABI): the scheduling buffer of (handle, stream, class) -> host
extern "C" int tr_debug_sched(tr_bvh* bvh, void* stream, int cls, uint32_t* host_out, long long words) {
    for (int k = 0; k < TR_SCHED_SLOTS; k++)
        if (bvh->sched[k].used && bvh->sched[k].stream == (hipStream_t)stream && bvh->sched[k].cls == cls) {
            (void)hipStreamSynchronize((hipStream_t)stream);
            return (int)hipMemcpy(host_out, bvh->sched[k].buf, sizeof(uint32_t) * (size_t)words, hipMemcpyDeviceToHost);
        }
    return -1;
}
#endif
