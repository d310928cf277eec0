// tr_bvh.h -- traversal-side data layout of the LBVH and the per-ray traversal core.
//
// The core is host+device code on purpose: the HIP kernels (traverse.hip) instantiate it
// per lane, and tests/host_sim compiles the very same functions with g++ to check the
// trail/parent-link state machine against the oracle without a GPU.  It is not a CPU
// fallback: nothing in the C ABI can reach the host instantiation.
//
// Layout in HBM (one arena per mesh, see DESIGN.md "Data layout"):
//   nodes : (F-1) x 64 B   tr_node   both children's boxes + child ids + parent + sibling
//   qnodes: (F-1) x 32 B   tr_qnode  the same boxes on a 16-bit grid + child ids (unordered schedule)
//   links : (F-1) x  8 B   tr_link   {parent, sibling} again, packed 8 per 64-B line, read
//                                    only while backtracking (keeps the climb off the
//                                    64-B node lines)
//   tris  :  F    x 48 B   tr_tri    Morton-ordered copy of the triangle (v0,v1,v2) + the
//                                    original face index
// Child ids: c >= 0 internal node index, c < 0 leaf, slot = ~c into `tris`.
// One triangle per leaf; the leaf box stored in the parent is the triangle's padded box (tr_tri_box).
//
// Traversal is stackless: a 64-bit trail (bit k set = the node at depth k on the current
// path still owes its far child) plus parent/sibling links (Hapala 2011 / Afra &
// Szirmay-Kalos 2014 style backtracking).  The builder guarantees depth <= 64.
#pragma once
#include <type_traits>
#include "tr_math.h"
#include "../../include/triro_hip.h"   // tr_hit_entry

// Leaf schedule of the fused trip.  1: ONE leaf test per trip out of a 3-slot per-lane FIFO (a
// node visit can add two leaves while one is consumed; the node waits only when the FIFO is
// full).  0: both leaves of the previous visit are tested in the same trip (two inlined tests,
// 12 more live registers).  Results are identical; 1 is +5 % at >= 4 M rays, equal at 1 M.
#ifndef TR_LEAF_QUEUE
#define TR_LEAF_QUEUE 1
#endif

struct alignas(16) tr_f4 {
    float x, y, z, w;
};
struct alignas(16) tr_i4 {
    int32_t x, y, z, w;
};

// Child boxes are stored as {lo.x, lo.y | lo.z, hi.z | hi.x, hi.y}: every aligned pair of
// floats meets the SAME pair of ray constants ((ox,oy) or (oz,oz)), so the slab planes of a
// node are 6 packed subtracts + 6 packed multiplies on the register pairs the 16-byte loads
// produce (tr_node_slabs).
struct alignas(64) tr_node {
    float box0[6];  // child 0: lo.x lo.y lo.z hi.z hi.x hi.y
    float box1[6];  // child 1
    int32_t c0, c1;
    int32_t parent;   // -1 at the root
    int32_t sibling;  // other child of parent (same encoding as c0/c1); root: 0
};
static_assert(sizeof(tr_node) == 64, "node must be 64 B");
TR_HD void tr_node_set_box(float* box, const float* lo, const float* hi) {
    box[0] = lo[0]; box[1] = lo[1]; box[2] = lo[2]; box[3] = hi[2]; box[4] = hi[0]; box[5] = hi[1];
}

struct alignas(8) tr_link {
    int32_t parent, sibling;
};

// ---- second node array for the UNORDERED schedule (count / location): 32-byte nodes -------------
// Both child boxes on a 16-bit grid over the mesh bounds + child ids: two 16-byte loads per visit
// instead of four.  The unordered leaf phase evaluates the full predicate from the triangle's
// vertices anyway (tr_tri_hit), so the grid boxes only have to be SUPERSETS of the exact ones: the
// builder searches, with the very decode the traversal uses, the largest grid plane <= lo and the
// smallest >= hi, and the slab arithmetic on the decoded planes is the contract's -- monotone under
// box inclusion, so nothing the exact tree would visit is culled.  Measured: C4 count 0.93 -> 0.80
// ms, location 1.18 -> 1.0 ms.  The ordered (closest / first / any) trip keeps the exact 64-byte
// nodes: it needs the leaf's exact slab interval from its parent, and the decode (12 cvt + 6 packed
// fma per visit) made every ordered configuration slower (experiments/q32_nodes.patch).
// Grid plane q of axis k lies at fma(q, scale[k], base[k]), q = 0 .. 65535; base = the mesh bounds'
// minimum, scale = a power of two with decode(65535) >= the bounds' maximum: a pure function of
// the bounds (tr_qframe_make), so host and device derive identical frames.
struct tr_qframe {
    float base[3];
    float scale[3];
};
TR_HD float tr_qdecode(uint32_t q, float scale, float base) { return fmaf((float)q, scale, base); }
TR_HD void tr_qframe_make(const float* mn, const float* mx, tr_qframe* f) {
    for (int k = 0; k < 3; k++) {
        const float ext = mx[k] - mn[k];
        int e = 0;
        float s = 1.0f;
        if (ext > 0.f && ext <= 3.0e38f) {
            (void)frexpf(ext / 65535.0f, &e);      // ext/65535 = m * 2^e, m in [0.5, 1)
            s = ldexpf(1.0f, e < -126 ? -126 : e);
        }
        // rounding in the two lines above can leave the last plane just short of the maximum
        for (int it = 0; it < 300 && !(tr_qdecode(65535u, s, mn[k]) >= mx[k]); it++) s *= 2.0f;
        f->base[k] = mn[k];
        f->scale[k] = s;
    }
}
// ---- fused conservative box test (round 5) --------------------------------------------------------------------
// The grid nodes (and the 8-wide nodes, tr_wide.h) only have to be tested CONSERVATIVELY: the leaves are decided by
// the full predicate (tr_tri_hit), whose own slab interval lies inside every ancestor's (monotone under box
// inclusion).  The contract's form of a plane's distance is three roundings deep -- p = fma(q, scale, base),
// (p - o), (..) * inv: a packed fma, a packed add and a packed multiply per pair of planes, 18 of the ~44 VALU
// instructions of a node's box test.  Here the ray carries, per axis, A = scale * k and B = (base - o) * k -+ e, and a
// plane costs ONE fma: t' = fma(q, A, B).  What makes that safe:
//   * k = the contract's reciprocal, its magnitude clamped to kmax = the power of two with kmax * M <= 2^100, M = the
//     position scale of the axis (|base|, |top|, |base - o|, twice the extent): every product stays finite, no
//     inf - inf, no 0 * inf.  Only rays (numerically) parallel to a coordinate plane are clamped.
//   * e >= the largest possible difference between t' and the contract's value for any plane of the mesh's box.
//     With u = 2^-24, T = (p - o) inv exact, P = max |plane|, E = 65536 scale (>= the extent):
//       contract:  p^ = fl(q s + b), d^ = fl(p^ - o), t^ = fl(d^ inv):   |t^ - T|        <= u |inv| (|p| + 2 |p - o|)
//       fused:     bo = fl(b - o), b0 = fl(bo k), B = fl(b0 -+ e), t' = fl(q A + B):
//                                                                        |t' - (T -+ e)| <= u |k| (|p - o| + 3 |b - o|) + 2 u e
//       exit pad:  t^ * (1 + 2^-21), rounded (TR_SLAB_PAD, round 6):     + 9 u |inv| |p - o|
//     and |p - o| <= |b - o| + E:  e >= u |inv| (P + 15 |b - o| + 12 E) for the grid nodes; the 8-wide nodes decode in
//     their own frame (planes up to one extent outside the mesh's box, base_n - o rounded per node):
//     e >= u |inv| (P + 14 |b - o| + 27 E).  e = 1.25 u |k| (P + 15 |b - o| + 27 E) covers both.  Entry planes get
//     -e, exit planes +e (sel_n / sel_f already separate them): t'_entry <= t_entry, t'_exit >= t_exit * TR_SLAB_PAD.
//   * a clamped axis additionally gets -+ 1.1e7 (> TR_TLIM, the largest limit of a box test): with |k| < |inv| the fused
//     distance has the sign of the contract's and a smaller magnitude, so an exit plane behind the origin stays behind,
//     one in front is pushed beyond every limit (it cannot cull), an entry plane in front stays smaller, one behind
//     stays <= 0 -- all that tr_slab_hit(max(tn, 0) <= min(tf, limit <= TR_TLIM)) can see.
//   * M beyond 1e30 (or not finite): the axis is ignored (A = 0, B = -+inf).
// So: tn' <= max(tn, 0) and tf' >= min(tf, TR_TLIM) for the contract's (tn, tf) of the same decoded box -- the fused
// test accepts whatever the contract's test accepts (tests/host_sim checks exactly this over the fuzz corpus), the
// traversal visits a superset of nodes, the leaves decide: results are bit-identical.  In position units the margin
// is ~1.3 float spacings of the coordinates plus 7e-7 of the camera distance: a few hundredths of a grid cell for a
// camera within a few extents of a mesh near the origin.
TR_HD void tr_fuse_axis(float o, float inv, float base, float scale, float& k, float& e, float& A, float& Bn, float& Bf) {
    const float top = fmaf(65535.0f, scale, base);
    const float bo = base - o;
    const float M = fmaxf(fabsf(base), fabsf(top)) + fabsf(bo) + 131072.0f * scale;
    const bool usable = M <= 1.0e30f;                             // (false for NaN, too)
    const uint32_t be = (tr_f2u(M) >> 23) & 0xffu;               // M = m * 2^(be - 127), m in [1, 2)  (be = 0: M < 2^-126)
    const uint32_t ke = 353u - be > 254u ? 254u : 353u - be;      // kmax = 2^(226 - be): kmax * M < 2^100
    const float kmax = tr_u2f(ke << 23);
    const bool clamped = fabsf(inv) > kmax;
    const float kk = clamped ? copysignf(kmax, inv) : inv;
    const float Me = fmaxf(fabsf(base), fabsf(top)) + 15.0f * fabsf(bo) + 1769472.0f * scale;             // P + 15 |b - o| + 27 E
    const float ee = fabsf(kk) * Me * 7.450580596923828e-08f + (clamped ? 1.1e7f : 0.0f) + 1.0e-37f;      // 1.25 * 2^-24
    const float b0 = bo * kk;
    // (selects, no branch: an early return here left the compiler with a stack object for the B pairs)
    k = usable ? kk : 0.0f;
    e = usable ? ee : INFINITY;
    A = usable ? scale * kk : 0.0f;
    Bn = usable ? b0 - ee : -INFINITY;
    Bf = usable ? b0 + ee : INFINITY;
}
TR_HD void tr_ray_fuse(tr_ray& r, const tr_qframe& f) {
    tr_fuse_axis(r.ox, r.ix, f.base[0], f.scale[0], r.kx, r.ex, r.qax, r.qnx, r.qfx);
    tr_fuse_axis(r.oy, r.iy, f.base[1], f.scale[1], r.ky, r.ey, r.qay, r.qny, r.qfy);
    tr_fuse_axis(r.oz, r.iz, f.base[2], f.scale[2], r.kz, r.ez, r.qaz, r.qnz, r.qfz);
    r.qaz2 = r.qaz;
}
// the box a ray is anchored to (tr_ray_anchor, tr_math.h): the grid frame of the mesh, base ... plane 65535
TR_HD void tr_ray_anchor_q(const tr_qframe& f, float& ox, float& oy, float& oz, float dx, float dy, float dz) {
    const float hi[3] = {fmaf(65535.0f, f.scale[0], f.base[0]), fmaf(65535.0f, f.scale[1], f.base[1]), fmaf(65535.0f, f.scale[2], f.base[2])};
    tr_ray_anchor(f.base, hi, ox, oy, oz, dx, dy, dz);
}
// ray set-up of the kernels that walk the exact nodes: anchor, then the constants of tr_math.h
TR_HD bool tr_ray_setup_a(tr_ray& r, const tr_qframe& f, float ox, float oy, float oz, float dx, float dy, float dz) {
    tr_ray_anchor_q(f, ox, oy, oz, dx, dy, dz);
    return tr_ray_setup(r, ox, oy, oz, dx, dy, dz);
}
// ray set-up for the kernels that walk grid nodes / 8-wide nodes
TR_HD bool tr_ray_setup_q(tr_ray& r, const tr_qframe& f, float ox, float oy, float oz, float dx, float dy, float dz) {
    tr_ray_anchor_q(f, ox, oy, oz, dx, dy, dz);
    const bool valid = tr_ray_setup(r, ox, oy, oz, dx, dy, dz);
    tr_ray_fuse(r, f);
    return valid;
}

// largest grid plane <= lo / smallest grid plane >= hi, defined through the traversal's own decode
// (monotone in q); lo >= base and hi <= decode(65535) by construction.  Closed-form estimate plus a
// fix-up walk of a step or two (round 2 ran a 16-step binary search per plane: 12 searches per node
// on every build and refit).
TR_HD uint32_t tr_qestimate(float x, float scale, float base) {
    const float e = (x - base) / scale;          // scale is a power of two: the division is exact
    return !(e > 0.f) ? 0u : (e >= 65535.f ? 65535u : (uint32_t)e);
}
// (the walk is bounded: where ulp(base) >> scale -- a mesh far from the origin relative to its extent, a
// degenerate axis -- thousands of consecutive q decode to the same float and a linear walk would take up to
// 65535 steps per plane, 12 planes per node; after 4 steps the binary search of round 2 takes over.  Same
// result: the largest q with decode(q) <= lo / the smallest with decode(q) >= hi, decode being monotone.)
TR_HD uint32_t tr_qfloor(float lo, float scale, float base) {
    uint32_t q = tr_qestimate(lo, scale, base);
    int steps = 0;
    while (q > 0u && !(tr_qdecode(q, scale, base) <= lo) && steps < 4) { q--; steps++; }
    while (q < 65535u && tr_qdecode(q + 1u, scale, base) <= lo && steps < 4) { q++; steps++; }
    if (steps >= 4) {
        // largest q in [0, 65535] with decode(q) <= lo (decode(0) = base <= lo by construction; NaN: 0)
        uint32_t a = 0u, z = 65535u;
        while (a < z) {
            const uint32_t m = (a + z + 1u) >> 1;
            if (tr_qdecode(m, scale, base) <= lo) a = m; else z = m - 1u;
        }
        q = a;
    }
    return q;
}
TR_HD uint32_t tr_qceil(float hi, float scale, float base) {
    uint32_t q = tr_qestimate(hi, scale, base);
    int steps = 0;
    while (q < 65535u && !(tr_qdecode(q, scale, base) >= hi) && steps < 4) { q++; steps++; }
    while (q > 0u && tr_qdecode(q - 1u, scale, base) >= hi && steps < 4) { q--; steps++; }
    if (steps >= 4) {
        // smallest q in [0, 65535] with decode(q) >= hi (decode(65535) >= hi by construction; NaN: 65535)
        uint32_t a = 0u, z = 65535u;
        while (a < z) {
            const uint32_t m = (a + z) >> 1;
            if (tr_qdecode(m, scale, base) >= hi) z = m; else a = m + 1u;
        }
        q = a;
    }
    return q;
}
// Child boxes as 16-bit pairs {lo.x, lo.y | lo.z, hi.z | hi.x, hi.y} (low half first): every pair
// meets the SAME pair of frame and ray constants ((x,y) or (z,z)), so a node decodes with 6 packed
// fma and its 12 slab planes are 6 packed subtracts + 6 packed multiplies (tr_qnode_slabs).
struct alignas(32) tr_qnode {
    uint32_t q[6];   // q[0..2] child 0, q[3..5] child 1
    int32_t c0, c1;
};
static_assert(sizeof(tr_qnode) == 32, "quantised node must be 32 B");
TR_HD void tr_qnode_set_box(uint32_t* q, const float* lo, const float* hi, const tr_qframe& f) {
    const uint32_t lx = tr_qfloor(lo[0], f.scale[0], f.base[0]), ly = tr_qfloor(lo[1], f.scale[1], f.base[1]);
    const uint32_t lz = tr_qfloor(lo[2], f.scale[2], f.base[2]);
    const uint32_t hx = tr_qceil(hi[0], f.scale[0], f.base[0]), hy = tr_qceil(hi[1], f.scale[1], f.base[1]);
    const uint32_t hz = tr_qceil(hi[2], f.scale[2], f.base[2]);
    q[0] = lx | (ly << 16); q[1] = lz | (hz << 16); q[2] = hx | (hy << 16);
}

// TR_TRI_BYTES: 48 (packed: half of the records straddle a 64-byte line) or 64 (one line per record)
#ifndef TR_TRI_BYTES
#define TR_TRI_BYTES 48
#endif
struct alignas(16) tr_tri {
    float ax, ay, az, bx, by, bz, cx, cy, cz;
    int32_t face;  // original triangle index
    float esum;    // tr_tri_scale: |b - a|_1 + |c - a|_1, the triangle's factor of the inside test's error bound (tr_tri_fast)
    int32_t pad1;
#if TR_TRI_BYTES == 64
    int32_t pad2[4];
#endif
};
static_assert(sizeof(tr_tri) == TR_TRI_BYTES, "tri record must be 48 (or, experiment, 64) B");

struct tr_bvh_view {
    const tr_node* nodes;
    const tr_link* links;
    const tr_tri* tris;
    int64_t num_tris;
    const tr_qnode* qnodes;   // 32-byte grid nodes of the unordered schedule (same topology as `nodes`)
    tr_qframe frame;
    const tr_qframe* frame_dev;   // device copy of `frame`, kept current by build / refit / load (NULL: host simulation)
};
// first statement of every kernel that takes a view: the frame as it is NOW (a graph replay after a refit: the arguments
// of the captured launch still hold the old one, and both the grid nodes' decode and the rays' anchor depend on it)
#if defined(__HIP_DEVICE_COMPILE__)
#define TR_VIEW_LIVE(b) do { if ((b).frame_dev) (b).frame = *(b).frame_dev; } while (0)
#else
#define TR_VIEW_LIVE(b) ((void)0)
#endif
enum tr_query { TR_Q_ANY = 0, TR_Q_FIRST = 1, TR_Q_CLOSEST = 2, TR_Q_COUNT = 3, TR_Q_LOCATION = 4 };

struct tr_counters {
    uint32_t nodes, tris, climbs;
#ifdef TR_COUNT_BOTTOM
    uint32_t bottom, bottom_hits;   // host experiment (scripts/exp_pair_leaves.py): visits of nodes whose
                                    // children are both leaves, and how many of those passed the box test
#endif
};

// sorted list of the K nearest hits (by (t, face)); static indexing only
template <int K>
struct tr_topk {
    float t[K];
    int32_t face[K];
    int32_t slot[K];
    TR_HDM void init() {
#pragma unroll
        for (int i = 0; i < K; i++) { t[i] = INFINITY; face[i] = 0x7fffffff; slot[i] = -1; }
    }
    TR_HDM void insert(float nt, int32_t nface, int32_t nslot) {
#pragma unroll
        for (int i = 0; i < K; i++) {
            bool lt = tr_closer(nt, nface, t[i], face[i]);
            float tt = t[i]; int32_t tf_ = face[i], ts = slot[i];
            t[i] = lt ? nt : tt; face[i] = lt ? nface : tf_; slot[i] = lt ? nslot : ts;
            nt = lt ? tt : nt; nface = lt ? tf_ : nface; nslot = lt ? ts : nslot;
        }
    }
};

// K = 0: the multi-hit list lives in memory, unsorted (tr_intersects_count_topk).  A lane carries
// a pointer and a fill count instead of 3*K registers of sorted (t, face, slot): the K = 8
// kernel needed 92 VGPRs (5 waves/SIMD) and ran a 32-select insertion for the whole wave on
// almost every leaf trip.  Appending is one 8-byte store; a ray with more than `cap` hits
// replaces its farthest stored hit when the new one is closer (rare path: loads).  The fill
// pass ranks the <= cap entries of a ray by (t_key, face) -- same final order as the sorted list.
static_assert(sizeof(tr_hit_entry) == 8, "hit entries are 8 bytes (include/triro_hip.h)");
template <>
struct tr_topk<0> {
    tr_hit_entry* ent;    // this ray's `cap` entries (set by the kernel, not by init())
    const tr_tri* tris;   // faces for tie-breaks
    int32_t cap, n;
    TR_HDM void init() { n = 0; }
    TR_HDM void insert(float nt, int32_t nface, int32_t nslot) {
        if (n < cap) {
            ent[n].t_key = nt; ent[n].slot = nslot;
            n++;
            return;
        }
        // more than `cap` hits: keep the `cap` nearest by (t_key, face)
        int32_t mi = 0;
        float mt = ent[0].t_key;
        int32_t mf = tris[ent[0].slot].face;
        for (int32_t j = 1; j < cap; j++) {
            const float tj = ent[j].t_key;
            if (tj < mt) continue;
            const int32_t fj = tris[ent[j].slot].face;
            if (tj > mt || fj > mf) { mi = j; mt = tj; mf = fj; }
        }
        if (tr_closer(nt, nface, mt, mf)) { ent[mi].t_key = nt; ent[mi].slot = nslot; }
    }
};

struct tr_result {
    // closest / first / any
    float best_t;
    int32_t best_face;   // -1 = miss
    int32_t best_slot;
    // count
    int32_t count;
};
TR_HD void tr_set_best_t(tr_result& res, float t) {
    res.best_t = t;
}

// COMPACT addressing: byte offsets fit 32 bits (nodes*64 and tris*48 below 4 GiB), so the
// loads use an SGPR base + 32-bit VGPR offset instead of 64-bit per-lane address arithmetic
template <bool COMPACT>
TR_HD const tr_f4* tr_node_ptr(const tr_bvh_view& b, int32_t node) {
    if (COMPACT)
        return reinterpret_cast<const tr_f4*>(reinterpret_cast<const char*>(b.nodes) + ((uint32_t)node << 6));
    return reinterpret_cast<const tr_f4*>(b.nodes + node);
}
template <bool COMPACT>
TR_HD const tr_i4* tr_qnode_ptr(const tr_bvh_view& b, int32_t node) {
    if (COMPACT)
        return reinterpret_cast<const tr_i4*>(reinterpret_cast<const char*>(b.qnodes) + ((uint32_t)node << 5));
    return reinterpret_cast<const tr_i4*>(b.qnodes + node);
}
template <bool COMPACT>
TR_HD const tr_f4* tr_tri_ptr(const tr_bvh_view& b, int32_t slot) {
    if (COMPACT)
        return reinterpret_cast<const tr_f4*>(reinterpret_cast<const char*>(b.tris) + (uint32_t)slot * (uint32_t)sizeof(tr_tri));
    return reinterpret_cast<const tr_f4*>(b.tris + slot);
}

TR_HD void tr_result_init(tr_result& res) {
    tr_set_best_t(res, TR_TMAX); res.best_face = -1; res.best_slot = -1;
    res.count = 0;
}
// the limit a box's entry distance is culled against (tr_math.h, TR_CULL_SLACK)
template <int Q>
TR_HD float tr_cull_limit(const tr_result& res) {
    // (the product is recomputed per trip: kept in a register beside best_t it measured slower, profiles/r06_pmc_headline.txt)
    return (Q == TR_Q_FIRST || Q == TR_Q_CLOSEST) ? res.best_t * TR_CULL_SLACK : TR_TLIM;
}

template <bool STATS, bool COMPACT = false>
TR_HD tr_tri tr_load_tri(const tr_bvh_view& b, int32_t slot, tr_counters* cnt) {
    const tr_f4* p = tr_tri_ptr<COMPACT>(b, slot);
    tr_f4 q0 = p[0], q1 = p[1], q2 = p[2];
    tr_tri t;
    t.ax = q0.x; t.ay = q0.y; t.az = q0.z; t.bx = q0.w;
    t.by = q1.x; t.bz = q1.y; t.cx = q1.z; t.cy = q1.w;
    t.cz = q2.x;
    union { float f; int32_t i; } u;
    u.f = q2.y;
    t.face = u.i;
    t.esum = q2.z;
    t.pad1 = 0;
    if (STATS) cnt->tris++;
    return t;
}

// Fold one accepted/rejected leaf test into the per-query state.  `live` = the lane really
// owns this leaf (the code runs unpredicated for the whole wave).  Returns true when the ray
// is finished (ANY query, first accepted hit).
// Fold one decided leaf test into the per-query state.  Returns true when the ray is finished (ANY query, first hit).
template <int Q, int K>
TR_HD bool tr_fold_hit(bool hit, float t, int32_t face, int32_t slot, tr_result& res, tr_topk<K>& top) {
    if (Q == TR_Q_ANY) {
        if (hit) res.best_face = face;
        return hit;
    } else if (Q == TR_Q_COUNT) {
        res.count += hit ? 1 : 0;
    } else if (Q == TR_Q_LOCATION) {
        if (hit) { res.count++; top.insert(t, face, slot); }
    } else {
        if (hit && tr_closer(t, face, res.best_t, res.best_face < 0 ? 0x7fffffff : res.best_face)) {
            tr_set_best_t(res, t); res.best_face = face; res.best_slot = slot;
        }
    }
    return false;
}
// The leaf test of the traversal: the float32 part of the predicate (tr_tri_fast) here, inside the trip; a test it
// leaves UNDECIDED (< 1 % of them: the ray passes within rounding of an edge, or grazes the plane) is parked in `pe`
// (the triangle's slot) and decided by the float64 part at the END of the trip (tr_drain_exact) -- a real call that
// owns thirty registers: there the trip's node record and triangle are dead and the call fits under the kernels'
// register budget, inside the leaf block it cost every kernel a wave per SIMD.  Until then the candidate culls
// nothing, which changes no result.  `live` = the lane really owns this leaf (the code runs unpredicated for the
// whole wave).  At most one test per trip and lane, and `pe` is emptied before the next one.
template <int Q, int K>
TR_HD bool tr_fold_leaf(bool live, const tr_ray& r, const tr_tri& t, int32_t slot, tr_result& res, tr_topk<K>& top, int32_t& pe) {
    tr_hit h;
    h.t = 0.f;
    const int c = tr_tri_fast(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, t.esum, h);
#ifndef TR_NO_EXACT      // (-DTR_NO_EXACT: what the float64 part costs -- undecided tests count as misses: WRONG results, timing only)
    if (live && c == TR_UNDECIDED) pe = slot;
#endif
    return tr_fold_hit<Q, K>(live && c == TR_HIT, h.t, t.face, slot, res, top);
}
// decide the parked test (see tr_fold_leaf); wave-uniform skip when no lane has one.
// COLD: the branch is marked as unlikely -- the register allocator then saves the registers the float64 call clobbers
// AROUND THE CALL (four scratch stores and loads on 0.6 % of the leaf tests) instead of keeping six kernel-lifetime
// values in scratch for the whole kernel (a store per lane in the prologue, a load after the loop: 48 MB of L2 <->
// fabric traffic per launch of the headline, L2 misses +30 %).  Headline 0.1838 -> 0.1800 ms, every direct config
// -1...-4 %; the streaming closest kernel on the binary nodes gets 10 spills and +6 % from the same hint and keeps
// the plain branch (profiles/r06_ab_cold_drain.txt).
#ifndef TR_DRAIN_COLD
#define TR_DRAIN_COLD 1     // 0: no branch hint anywhere (A/B and fault-hunting builds)
#endif
template <int Q, int K, bool COMPACT = false, bool COLD = true>
TR_HD bool tr_drain_exact(const tr_bvh_view& b, const tr_ray& r, int32_t& pe, tr_result& res, tr_topk<K>& top) {
    bool fin = false;
    const bool any_parked = TR_WAVE_ANY(pe >= 0);
    if ((COLD && TR_DRAIN_COLD) ? __builtin_expect(any_parked, 0) : any_parked) {
        if (pe >= 0) {
            tr_counters* nc = nullptr;
            const tr_tri t = tr_load_tri<false, COMPACT>(b, pe, nc);
            const tr_exact_res e = tr_tri_exact_t(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz);
            float tt = e.t;
            if (__builtin_expect(e.tie != 0, 0)) {
                // an exact tie (the ray passes through an edge or a vertex): does this triangle own it?  The triangle is
                // FETCHED AGAIN for that question -- kept alive across the call above it would cost every kernel eight registers
#if defined(__HIP_DEVICE_COMPILE__)
                __asm__ volatile("" ::: "memory");
#endif
                const tr_tri u = tr_load_tri<false, COMPACT>(b, pe, nc);
                if (!tr_tie_own(r.dx, r.dy, r.dz, u.ax, u.ay, u.az, u.bx, u.by, u.bz, u.cx, u.cy, u.cz, e.tie)) tt = -1.0f;
            }
            const bool hit = tt >= TR_TMIN && tt <= TR_TMAX;
            fin = tr_fold_hit<Q, K>(hit, tt, t.face, pe, res, top);
            pe = -1;
        }
    }
    return fin;
}

// Far-child ring: the far child pushed at depth k is remembered in slot k % TR_RING of a
// per-lane ring (LDS on the GPU: slot s of lane t lives at base[s * stride], stride = block
// size, so a wave's accesses are bank-conflict free whatever the per-lane depths are).  A
// 64-bit `owned` mask says which depths still own their slot; backtracking reads the slot
// (one LDS read) when owned and falls back to climbing the parent links otherwise.
#define TR_RING 16
#define TR_RING_MASK 0x0001000100010001ull
struct tr_ring {
    int32_t* base;   // nullptr = no ring (always climb)
    int32_t stride;
};
// Ring accesses go through an explicit LDS pointer on the device.  With the generic pointer the compiler
// merged "pop the ring slot (LDS)" and "read the sibling from links[] (global)" -- two 4-byte loads that
// feed the same variable -- into ONE flat_load_dword with a selected address: every backtracking trip
// then sent a 64-lane flat load through the texture addresser (the busiest unit of the kernels) and
// waited for vmcnt AND lgkmcnt, where a ds_read_b32 would do.
TR_HD int32_t tr_ring_get(const tr_ring& ring, uint32_t slot) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TR_RING_GENERIC)      // (-DTR_RING_GENERIC: the old code path, for A/B runs)
    typedef __attribute__((address_space(3))) int32_t tr_lds_i32;
    return ((tr_lds_i32*)ring.base)[slot * ring.stride];
#else
    return ring.base[slot * ring.stride];
#endif
}
TR_HD void tr_ring_put(const tr_ring& ring, uint32_t slot, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TR_RING_GENERIC)
    typedef __attribute__((address_space(3))) int32_t tr_lds_i32;
    ((tr_lds_i32*)ring.base)[slot * ring.stride] = v;
#else
    ring.base[slot * ring.stride] = v;
#endif
}

// Per-lane traversal state between two iterations.  W = uint64_t in general; uint32_t when
// the hierarchy is at most 32 levels high (halves the 64-bit shift/clz work per trip).
template <typename W>
struct tr_state_t {
    int32_t node;     // next internal node to visit, -1 = hierarchy exhausted
    uint32_t depth;
    W trail;          // bit k: the node at depth k on the current path still owes its far child
    W owned;          // bit k: that far child is still in ring slot k % TR_RING
    // leaves found by earlier node visits (tri slots, -1 = none), tested one per trip so that their triangle loads
    // overlap the next node's load (one memory round trip per iteration); up to two arrive per visit
    int32_t p0, p1, p2;
    int32_t pe;       // a leaf test the float32 part left undecided, waiting for tr_drain_exact (-1 = none)
};
typedef tr_state_t<uint64_t> tr_state;
typedef tr_state_t<uint32_t> tr_state32;

TR_HD uint32_t tr_top_bit(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }
TR_HD uint32_t tr_top_bit(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }
TR_HD uint64_t tr_ring_mask(uint64_t) { return TR_RING_MASK; }
TR_HD uint32_t tr_ring_mask(uint32_t) { return 0x00010001u; }

template <typename W>
TR_HD void tr_state_init(tr_state_t<W>& st) {
    st.node = 0; st.depth = 0; st.trail = 0; st.owned = 0;
    st.p0 = -1; st.p1 = -1; st.p2 = -1; st.pe = -1;
}

template <typename W>
TR_HD bool tr_pending(const tr_state_t<W>& st) { return st.p0 >= 0 || st.p1 >= 0; }
template <typename W>
TR_HD bool tr_done(const tr_state_t<W>& st) { return st.node < 0 && st.p0 < 0 && st.p1 < 0 && st.pe < 0; }

#ifndef TR_PK_SLAB
#define TR_PK_SLAB 1
#endif

// Slab intervals of both children of a node held in three 16-byte registers:
// n0 = lo0.x lo0.y lo0.z hi0.z | n1 = hi0.x hi0.y lo1.x lo1.y | n2 = lo1.z hi1.z hi1.x hi1.y.
// On the device the 12 planes are packed FP32 (v_pk_add_f32 / v_pk_mul_f32: the same IEEE
// subtract and multiply per plane as tr_slab, two planes per instruction).
TR_HD void tr_node_slabs(const tr_ray& r, const tr_f4& n0, const tr_f4& n1, const tr_f4& n2,
                         float& tn0, float& tf0, float& tn1, float& tf1) {
#if defined(__HIP_DEVICE_COMPILE__) && TR_PK_SLAB
    typedef float tr_v2 __attribute__((ext_vector_type(2)));
    const tr_v2 oxy = {r.ox, r.oy}, ozz = {r.oz, r.oz};
    const tr_v2 ixy = {r.ix, r.iy}, izz = {r.iz, r.iz};
    const tr_v2 a = (tr_v2{n0.x, n0.y} - oxy) * ixy;   // child 0: x1 y1
    const tr_v2 b = (tr_v2{n0.z, n0.w} - ozz) * izz;   //          z1 z2
    const tr_v2 c = (tr_v2{n1.x, n1.y} - oxy) * ixy;   //          x2 y2
    const tr_v2 d = (tr_v2{n1.z, n1.w} - oxy) * ixy;   // child 1: x1 y1
    const tr_v2 e = (tr_v2{n2.x, n2.y} - ozz) * izz;   //          z1 z2
    const tr_v2 f = (tr_v2{n2.z, n2.w} - oxy) * ixy;   //          x2 y2
    tn0 = fmaxf(fmaxf(fminf(a.x, c.x), fminf(a.y, c.y)), fminf(b.x, b.y));
    tf0 = fminf(fminf(fmaxf(a.x, c.x), fmaxf(a.y, c.y)), fmaxf(b.x, b.y)) * TR_SLAB_PAD;
    tn1 = fmaxf(fmaxf(fminf(d.x, f.x), fminf(d.y, f.y)), fminf(e.x, e.y));
    tf1 = fminf(fminf(fmaxf(d.x, f.x), fmaxf(d.y, f.y)), fmaxf(e.x, e.y)) * TR_SLAB_PAD;
#else
    tr_slab(r, n0.x, n0.y, n0.z, n1.x, n1.y, n0.w, tn0, tf0);
    tr_slab(r, n1.z, n1.w, n2.x, n2.z, n2.w, n2.y, tn1, tf1);
#endif
}

// NODE PHASE: visit st.node (lane must have a node and no queued leaf): fetch the 64-B node,
// test both child boxes, queue hit leaf children in (p0, p1), then move to the next node
// (near child, or the deepest owed far child; -1 when the hierarchy is exhausted).
template <int Q, bool STATS>
TR_HD void tr_node_step(const tr_bvh_view& b, const tr_ray& r, tr_state& st, const tr_result& res,
                        tr_counters* cnt, const tr_ring ring) {
    const tr_f4* np = reinterpret_cast<const tr_f4*>(b.nodes + st.node);
    const tr_f4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    if (STATS) cnt->nodes++;
    // n3 = c0 c1 parent sibling
    float tn0, tf0, tn1, tf1;
    tr_node_slabs(r, n0, n1, n2, tn0, tf0, tn1, tf1);
    union { float f; int32_t i; } u0, u1, u2, u3;
    u0.f = n3.x; u1.f = n3.y; u2.f = n3.z; u3.f = n3.w;
    const int32_t c0 = u0.i, c1 = u1.i;
    int32_t parent = u2.i, sibling = u3.i;
    const float lim = tr_cull_limit<Q>(res);
    bool h0 = tr_slab_hit(tn0, tf0, lim);
    bool h1 = tr_slab_hit(tn1, tf1, lim);
#ifdef TR_COUNT_BOTTOM
    if (STATS && c0 < 0 && c1 < 0) { cnt->bottom++; cnt->bottom_hits += (h0 ? 1u : 0u) + (h1 ? 1u : 0u); }
#endif
    // leaf children are queued for the leaf phase
    if (h0 && c0 < 0) { st.p0 = ~c0; h0 = false; }
    if (h1 && c1 < 0) { st.p1 = ~c1; h1 = false; }
    if (h0 | h1) {
        const bool both = h0 & h1;
        const bool swap = both ? (tn1 < tn0) : h1;   // descend into c1?
        if (both) {
            st.trail |= (1ull << st.depth);
            if (ring.base) {
                const uint32_t slot = st.depth & (TR_RING - 1);
                tr_ring_put(ring, slot, swap ? c0 : c1);
                st.owned = (st.owned & ~(TR_RING_MASK << slot)) | (1ull << st.depth);
            }
        }
        st.node = swap ? c1 : c0;
        st.depth++;
    } else if (st.trail == 0) {
        st.node = -1;
    } else {
        // backtrack to the deepest ancestor that still owes its far child
        const uint32_t j = 63u - (uint32_t)__builtin_clzll(st.trail);
        st.trail &= ~(1ull << j);
        if (ring.base && ((st.owned >> j) & 1ull)) {
            st.node = tr_ring_get(ring, j & (TR_RING - 1));
        } else {
            int32_t node = st.node;
            uint32_t depth = st.depth;
            while (depth > j + 1) {   // climb; `node` is at `depth`, (parent, sibling) its links
                node = parent;
                const tr_link l = b.links[node];
                parent = l.parent; sibling = l.sibling;
                depth--;
                if (STATS) cnt->climbs++;
            }
            st.node = sibling;        // far child at depth j+1 (internal by construction)
        }
        st.depth = j + 1;
    }
}

// LEAF PHASE: test the queued leaves (both triangle loads are issued before either test).
// Runs unpredicated for the calling lanes; lanes without a queued leaf in a slot compute on
// triangle 0 and discard.  Sets st.node = -1 when an ANY query is satisfied.
template <int Q, int K, bool STATS>
TR_HD void tr_leaf_step(const tr_bvh_view& b, const tr_ray& r, tr_state& st, tr_result& res,
                        tr_topk<K>& top, tr_counters* cnt) {
    const int32_t q0 = st.p0, q1 = st.p1;
    tr_counters* nc = nullptr;
    const bool any1 = TR_WAVE_ANY(q1 >= 0);
    const tr_tri t0 = tr_load_tri<false>(b, q0 >= 0 ? q0 : 0, nc);
    tr_tri t1 = t0;
    if (any1) t1 = tr_load_tri<false>(b, q1 >= 0 ? q1 : 0, nc);
    if (STATS && q0 >= 0) cnt->tris++;
    bool fin = tr_fold_leaf<Q, K>(q0 >= 0, r, t0, q0, res, top, st.pe);
    fin = tr_drain_exact<Q, K>(b, r, st.pe, res, top) || fin;
    if (any1) {
        const bool live = q1 >= 0 && !fin;
        if (STATS && live) cnt->tris++;
        fin = tr_fold_leaf<Q, K>(live, r, t1, q1, res, top, st.pe) || fin;
        fin = tr_drain_exact<Q, K>(b, r, st.pe, res, top) || fin;
    }
    st.p0 = -1; st.p1 = -1;
    if (Q == TR_Q_ANY && fin) st.node = -1;
}

// FUSED STEP (software-pipelined schedule): one trip = node fetch for every lane that has a
// node + the test of a leaf queued by an EARLIER trip.  The node loads are issued first, so
// the triangle loads and the node loads are in flight together: one memory round trip per
// trip.  Lanes never sit out.
// Slab intervals of both children of a 32-byte grid node held in two 16-byte registers:
// w0 = q[0..3], w1 = q[4], q[5], c0, c1.  Decode: plane = fma(q, scale, base) per 16-bit half; then
// the contract's subtract and multiply per plane.  On the device both steps are packed FP32.
TR_HD void tr_qnode_slabs_contract(const tr_ray& r, const tr_qframe& f, const tr_i4& w0, const tr_i4& w1,
                                   float& tn0, float& tf0, float& tn1, float& tf1) {
    const uint32_t q0 = (uint32_t)w0.x, q1 = (uint32_t)w0.y, q2 = (uint32_t)w0.z, q3 = (uint32_t)w0.w;
    const uint32_t q4 = (uint32_t)w1.x, q5 = (uint32_t)w1.y;
#if defined(__HIP_DEVICE_COMPILE__) && TR_PK_SLAB && !defined(TR_QNOSIGN)
    // Round 4: the planes the ray ENTERS a child box through and the ones it leaves through are selected per ray with
    // three v_perm_b32 per child (selectors precomputed in tr_ray_setup), so that a child needs one max3 and one min3
    // instead of six min / max before them: 21 instead of 24 instructions per child.  The values are tr_slab's -- min(t1,
    // t2) IS the entry plane's t once the reciprocal's sign is known (rays are NaN-free) -- so nothing changes but the
    // count: headline 0.205 -> 0.199 ms, C4 closest -3 %, terrain -5 %, count -2 % (-DTR_QNOSIGN: the old form, for A/B).
    // It costs three registers: the 8-waves-per-SIMD variants (64 registers) no longer fit and were retired with it
    // (streaming configs +-1 % at 7 waves: profiles/r04_ab_qsign.txt).
    typedef float tr_v2 __attribute__((ext_vector_type(2)));
    const tr_v2 sxy = {f.scale[0], f.scale[1]}, szz = {f.scale[2], f.scale[2]};
    const tr_v2 bxy = {f.base[0], f.base[1]}, bzz = {f.base[2], f.base[2]};
    const tr_v2 oxy = {r.ox, r.oy}, ozz = {r.oz, r.oz};
    const tr_v2 ixy = {r.ix, r.iy}, izz = {r.iz, r.iz};
#define TR_UNPK(w) tr_v2{(float)((w) & 0xffffu), (float)((w) >> 16)}
    {
        const uint32_t nxy = __builtin_amdgcn_perm(q2, q0, r.sel_n), fxy = __builtin_amdgcn_perm(q2, q0, r.sel_f);
        const uint32_t zz = __builtin_amdgcn_perm(q1, q1, r.sel_z);      // (entry z | exit z << 16)
        const tr_v2 a = (__builtin_elementwise_fma(TR_UNPK(nxy), sxy, bxy) - oxy) * ixy;
        const tr_v2 c = (__builtin_elementwise_fma(TR_UNPK(fxy), sxy, bxy) - oxy) * ixy;
        const tr_v2 b = (__builtin_elementwise_fma(TR_UNPK(zz), szz, bzz) - ozz) * izz;
        tn0 = fmaxf(fmaxf(a.x, a.y), b.x);
        tf0 = fminf(fminf(c.x, c.y), b.y) * TR_SLAB_PAD;
    }
    {
        const uint32_t nxy = __builtin_amdgcn_perm(q5, q3, r.sel_n), fxy = __builtin_amdgcn_perm(q5, q3, r.sel_f);
        const uint32_t zz = __builtin_amdgcn_perm(q4, q4, r.sel_z);
        const tr_v2 a = (__builtin_elementwise_fma(TR_UNPK(nxy), sxy, bxy) - oxy) * ixy;
        const tr_v2 c = (__builtin_elementwise_fma(TR_UNPK(fxy), sxy, bxy) - oxy) * ixy;
        const tr_v2 b = (__builtin_elementwise_fma(TR_UNPK(zz), szz, bzz) - ozz) * izz;
        tn1 = fmaxf(fmaxf(a.x, a.y), b.x);
        tf1 = fminf(fminf(c.x, c.y), b.y) * TR_SLAB_PAD;
    }
#undef TR_UNPK
#elif defined(__HIP_DEVICE_COMPILE__) && TR_PK_SLAB
    typedef float tr_v2 __attribute__((ext_vector_type(2)));
    const tr_v2 sxy = {f.scale[0], f.scale[1]}, szz = {f.scale[2], f.scale[2]};
    const tr_v2 bxy = {f.base[0], f.base[1]}, bzz = {f.base[2], f.base[2]};
    const tr_v2 oxy = {r.ox, r.oy}, ozz = {r.oz, r.oz};
    const tr_v2 ixy = {r.ix, r.iy}, izz = {r.iz, r.iz};
#define TR_UNPK(w) tr_v2{(float)((w) & 0xffffu), (float)((w) >> 16)}
    const tr_v2 a = (__builtin_elementwise_fma(TR_UNPK(q0), sxy, bxy) - oxy) * ixy;   // child 0: x1 y1
    const tr_v2 b = (__builtin_elementwise_fma(TR_UNPK(q1), szz, bzz) - ozz) * izz;   //          z1 z2
    const tr_v2 c = (__builtin_elementwise_fma(TR_UNPK(q2), sxy, bxy) - oxy) * ixy;   //          x2 y2
    const tr_v2 d = (__builtin_elementwise_fma(TR_UNPK(q3), sxy, bxy) - oxy) * ixy;   // child 1: x1 y1
    const tr_v2 e = (__builtin_elementwise_fma(TR_UNPK(q4), szz, bzz) - ozz) * izz;   //          z1 z2
    const tr_v2 g = (__builtin_elementwise_fma(TR_UNPK(q5), sxy, bxy) - oxy) * ixy;   //          x2 y2
#undef TR_UNPK
    tn0 = fmaxf(fmaxf(fminf(a.x, c.x), fminf(a.y, c.y)), fminf(b.x, b.y));
    tf0 = fminf(fminf(fmaxf(a.x, c.x), fmaxf(a.y, c.y)), fmaxf(b.x, b.y)) * TR_SLAB_PAD;
    tn1 = fmaxf(fmaxf(fminf(d.x, g.x), fminf(d.y, g.y)), fminf(e.x, e.y));
    tf1 = fminf(fminf(fmaxf(d.x, g.x), fmaxf(d.y, g.y)), fmaxf(e.x, e.y)) * TR_SLAB_PAD;
#else
    const float sx = f.scale[0], sy = f.scale[1], sz = f.scale[2], bx = f.base[0], by = f.base[1], bz = f.base[2];
    tr_slab(r, tr_qdecode(q0 & 0xffffu, sx, bx), tr_qdecode(q0 >> 16, sy, by), tr_qdecode(q1 & 0xffffu, sz, bz),
            tr_qdecode(q2 & 0xffffu, sx, bx), tr_qdecode(q2 >> 16, sy, by), tr_qdecode(q1 >> 16, sz, bz), tn0, tf0);
    tr_slab(r, tr_qdecode(q3 & 0xffffu, sx, bx), tr_qdecode(q3 >> 16, sy, by), tr_qdecode(q4 & 0xffffu, sz, bz),
            tr_qdecode(q5 & 0xffffu, sx, bx), tr_qdecode(q5 >> 16, sy, by), tr_qdecode(q4 >> 16, sz, bz), tn1, tf1);
#endif
}

// The fused form (tr_ray_fuse): per child three v_perm_b32 (entry / exit planes of this ray), six conversions, THREE
// packed fma, one max3, one min3 -- 14 instead of 22 instructions; no pad multiply (the margin e is in B).  Host and
// device evaluate the same fma per plane: identical (tn, tf), so tests/host_sim walks the very path the kernels walk.
#ifndef TR_QFUSE
#define TR_QFUSE 1      // 0: the contract's three-step form (A/B builds)
#endif
TR_HD void tr_qnode_slabs(const tr_ray& r, const tr_qframe& f, const tr_i4& w0, const tr_i4& w1,
                          float& tn0, float& tf0, float& tn1, float& tf1) {
#if !TR_QFUSE
    tr_qnode_slabs_contract(r, f, w0, w1, tn0, tf0, tn1, tf1);
#elif defined(__HIP_DEVICE_COMPILE__)
    const uint32_t q0 = (uint32_t)w0.x, q1 = (uint32_t)w0.y, q2 = (uint32_t)w0.z, q3 = (uint32_t)w0.w;
    const uint32_t q4 = (uint32_t)w1.x, q5 = (uint32_t)w1.y;
    typedef float tr_v2 __attribute__((ext_vector_type(2)));
    const tr_v2 axy = {r.qax, r.qay}, azz = {r.qaz, r.qaz2};
    const tr_v2 nxy = {r.qnx, r.qny}, fxy = {r.qfx, r.qfy}, bz = {r.qnz, r.qfz};
#define TR_UNPK(w) tr_v2{(float)((w) & 0xffffu), (float)((w) >> 16)}
    {
        const uint32_t pn = __builtin_amdgcn_perm(q2, q0, r.sel_n), pf = __builtin_amdgcn_perm(q2, q0, r.sel_f);
        const uint32_t zz = __builtin_amdgcn_perm(q1, q1, r.sel_z);      // (entry z | exit z << 16)
        const tr_v2 a = __builtin_elementwise_fma(TR_UNPK(pn), axy, nxy);
        const tr_v2 c = __builtin_elementwise_fma(TR_UNPK(pf), axy, fxy);
        const tr_v2 b = __builtin_elementwise_fma(TR_UNPK(zz), azz, bz);
        tn0 = fmaxf(fmaxf(a.x, a.y), b.x);
        tf0 = fminf(fminf(c.x, c.y), b.y);
    }
    {
        const uint32_t pn = __builtin_amdgcn_perm(q5, q3, r.sel_n), pf = __builtin_amdgcn_perm(q5, q3, r.sel_f);
        const uint32_t zz = __builtin_amdgcn_perm(q4, q4, r.sel_z);
        const tr_v2 a = __builtin_elementwise_fma(TR_UNPK(pn), axy, nxy);
        const tr_v2 c = __builtin_elementwise_fma(TR_UNPK(pf), axy, fxy);
        const tr_v2 b = __builtin_elementwise_fma(TR_UNPK(zz), azz, bz);
        tn1 = fmaxf(fmaxf(a.x, a.y), b.x);
        tf1 = fminf(fminf(c.x, c.y), b.y);
    }
#undef TR_UNPK
#else
    (void)f;
    const bool nx = r.ix < 0.f, ny = r.iy < 0.f, nz = r.iz < 0.f;      // as sel_n / sel_f / sel_z (tr_ray_setup)
    const uint32_t q[6] = {(uint32_t)w0.x, (uint32_t)w0.y, (uint32_t)w0.z, (uint32_t)w0.w, (uint32_t)w1.x, (uint32_t)w1.y};
    float tn[2], tf[2];
    for (int c = 0; c < 2; c++) {
        const uint32_t lox = q[3 * c] & 0xffffu, loy = q[3 * c] >> 16, loz = q[3 * c + 1] & 0xffffu;
        const uint32_t hiz = q[3 * c + 1] >> 16, hix = q[3 * c + 2] & 0xffffu, hiy = q[3 * c + 2] >> 16;
        const float ax = fmaf((float)(nx ? hix : lox), r.qax, r.qnx), ay = fmaf((float)(ny ? hiy : loy), r.qay, r.qny);
        const float az = fmaf((float)(nz ? hiz : loz), r.qaz, r.qnz);
        const float cx = fmaf((float)(nx ? lox : hix), r.qax, r.qfx), cy = fmaf((float)(ny ? loy : hiy), r.qay, r.qfy);
        const float cz = fmaf((float)(nz ? loz : hiz), r.qaz, r.qfz);
        tn[c] = fmaxf(fmaxf(ax, ay), az);
        tf[c] = fminf(fminf(cx, cy), cz);
    }
    tn0 = tn[0]; tf0 = tf[0]; tn1 = tn[1]; tf1 = tf[1];
#endif
}

// Backtracking without the ring and without parent / sibling in the node record: climb the parent
// links from `node` (at `depth`) to the ancestor at depth j+1 and return its sibling -- the far
// child owed at depth j (internal by construction).  One dependent 8-byte load per level.
template <bool STATS>
TR_HD int32_t tr_climb(const tr_bvh_view& b, int32_t node, uint32_t depth, uint32_t j, tr_counters* cnt) {
    tr_link l = b.links[node];
    if (STATS) cnt->climbs++;
    while (depth > j + 1) {
        node = l.parent;
        l = b.links[node];
        depth--;
        if (STATS) cnt->climbs++;
    }
    return l.sibling;
}

// A fetched node record for the fused trip: the exact 64-byte node (four 16-byte words) or the
// 32-byte grid node (two).  With the full predicate at the leaves (tr_fold_leaf) the trip only needs
// CONSERVATIVE child boxes, so the pruning queries can walk the grid nodes too: two gathers per
// visit instead of four, for a decode of 18 VALU instructions.
struct tr_rec_f { tr_f4 n0, n1, n2, n3; };
struct tr_rec_q { tr_i4 w0, w1; };
TR_HD void tr_rec_slabs(const tr_bvh_view&, const tr_ray& r, const tr_rec_f& n, float& tn0, float& tf0, float& tn1, float& tf1) {
    tr_node_slabs(r, n.n0, n.n1, n.n2, tn0, tf0, tn1, tf1);
}
TR_HD void tr_rec_slabs(const tr_bvh_view& b, const tr_ray& r, const tr_rec_q& n, float& tn0, float& tf0, float& tn1, float& tf1) {
    tr_qnode_slabs(r, b.frame, n.w0, n.w1, tn0, tf0, tn1, tf1);
}
TR_HD void tr_rec_children(const tr_rec_f& n, int32_t& c0, int32_t& c1) {
    union { float f; int32_t i; } u0, u1;
    u0.f = n.n3.x; u1.f = n.n3.y; c0 = u0.i; c1 = u1.i;
}
TR_HD void tr_rec_children(const tr_rec_q& n, int32_t& c0, int32_t& c1) { c0 = n.w1.z; c1 = n.w1.w; }
// parent / sibling of the visited node: the first step of a climb comes with the exact record
// (n3 = c0 c1 parent sibling); the grid node does not carry them (tr_climb reads links[])
TR_HD void tr_rec_links(const tr_rec_f& n, int32_t& parent, int32_t& sibling) {
    union { float f; int32_t i; } u2, u3;
    u2.f = n.n3.z; u3.f = n.n3.w; parent = u2.i; sibling = u3.i;
}
TR_HD void tr_rec_links(const tr_rec_q&, int32_t& parent, int32_t& sibling) { parent = -1; sibling = -1; }

#if !TR_LEAF_QUEUE
#error "the fused trip needs the 3-slot leaf FIFO (TR_LEAF_QUEUE)"
#endif
// everything of a trip after the node record has arrived (n0..n3: in vector registers, or -- on
// trips where the whole wave visits the same node -- in scalar registers)
// TEST = false: a trip WITHOUT the leaf block (tr_fused_step)
template <int Q, int K, bool STATS, bool COMPACT, typename W, bool TEST, bool COLD, typename REC>
TR_HD void tr_fused_body(const tr_bvh_view& b, const tr_ray& r, tr_state_t<W>& st, tr_result& res,
                         tr_topk<K>& top, tr_counters* cnt, const tr_ring ring, const bool has_node,
                         const REC& rec) {
    if (STATS && has_node) cnt->nodes++;
    bool fin = false;
    const int32_t q0 = st.p0;
    if (TEST && TR_WAVE_ANY(q0 >= 0)) {
        tr_counters* nc = nullptr;
        const tr_tri t0 = tr_load_tri<false, COMPACT>(b, q0 >= 0 ? q0 : 0, nc);
        const bool live = q0 >= 0;
        if (STATS && live) cnt->tris++;
        fin = tr_fold_leaf<Q, K>(live, r, t0, q0, res, top, st.pe);
    }
    if (Q == TR_Q_ANY && fin) { st.node = -1; st.p1 = -1; st.p2 = -1; }
    float tn0, tf0, tn1, tf1;
    tr_rec_slabs(b, r, rec, tn0, tf0, tn1, tf1);
    int32_t c0, c1, parent, sibling;
    tr_rec_children(rec, c0, c1);
    tr_rec_links(rec, parent, sibling);
    const float lim = tr_cull_limit<Q>(res);
    const bool go = has_node && st.node >= 0;
    bool h0 = tr_slab_hit(tn0, tf0, lim) && go;
    bool h1 = tr_slab_hit(tn1, tf1, lim) && go;
#if TR_LEAF_QUEUE
    // New FIFO = (carried entries, new leaf of child 0, new leaf of child 1), written as selects
    // (no shift-then-push: every move is a v_cndmask).  The head (p0) was consumed above; b =
    // old p1 and c = old p2 are carried.  When c is valid the node waited (go is false, no new
    // leaves) and c takes the place of "leaf of child 1" in the formulas below.
    if (!TEST) {
        // the head was NOT consumed; a lane that visits its node on such a trip holds at most one
        // queued leaf (p0), so FIFO = (p0, new leaf of child 0, new leaf of child 1) still fits.
        // Lanes that do not visit keep their FIFO as it is.
        const bool l0 = h0 && c0 < 0, l1 = h1 && c1 < 0;
        h0 = h0 && c0 >= 0;
        h1 = h1 && c1 >= 0;
        if (go) {
            const bool hb = st.p0 >= 0, two = l0 && l1;
            const int32_t xi = l0 ? ~c0 : (l1 ? ~c1 : -1);
            st.p2 = (hb && two) ? ~c1 : -1;
            st.p1 = hb ? xi : (two ? ~c1 : -1);
            st.p0 = hb ? st.p0 : xi;
        }
    } else {
        const bool l0 = h0 && c0 < 0;
        const bool wait = st.p2 >= 0;
        const bool l1 = (h1 && c1 < 0) || wait;
        h0 = h0 && c0 >= 0;
        h1 = h1 && c1 >= 0;
        const int32_t i1 = wait ? st.p2 : ~c1;
        const bool hb = st.p1 >= 0, two = l0 && l1;
        const int32_t xi = l0 ? ~c0 : (l1 ? i1 : -1);      // first new entry
        st.p0 = hb ? st.p1 : xi;
        st.p1 = hb ? xi : (two ? i1 : -1);
        st.p2 = (hb && two) ? i1 : -1;
    }
#else
#error "unused"
#endif
    if (go) {
        if (h0 || h1) {
            const bool both = h0 && h1;
            const bool swap = (both && tn1 < tn0) || !h0;   // descend into c1?
            if (both) {
                st.trail |= (W(1) << st.depth);
                if (ring.base) {
                    const uint32_t slot = st.depth & (TR_RING - 1);
                    tr_ring_put(ring, slot, swap ? c0 : c1);
                    st.owned = (st.owned & ~(tr_ring_mask(W(0)) << slot)) | (W(1) << st.depth);
                }
            }
            st.node = swap ? c1 : c0;
            st.depth++;
        } else if (st.trail == 0) {
            st.node = -1;
        } else {
            const uint32_t j = tr_top_bit(st.trail);
            st.trail &= ~(W(1) << j);
            if (ring.base && ((st.owned >> j) & W(1))) {
                st.node = tr_ring_get(ring, j & (TR_RING - 1));
            } else if constexpr (std::is_same<REC, tr_rec_f>::value) {
                int32_t node = st.node;
                uint32_t depth = st.depth;
                while (depth > j + 1) {
                    node = parent;
                    const tr_link l = b.links[node];
                    parent = l.parent; sibling = l.sibling;
                    depth--;
                    if (STATS) cnt->climbs++;
                }
                st.node = sibling;
            } else {
                st.node = tr_climb<STATS>(b, st.node, st.depth, j, cnt);
            }
            st.depth = j + 1;
        }
    }
    // the test this trip's leaf block left undecided, now that the trip's record and triangle are dead (tr_fold_leaf)
    if (TEST) {
        if (tr_drain_exact<Q, K, COMPACT, COLD>(b, r, st.pe, res, top) && Q == TR_Q_ANY) { st.node = -1; st.p0 = -1; st.p1 = -1; st.p2 = -1; }
        st.pe = -1;     // (it is: said once more so that the compiler does not carry it around the loop)
    }
}


// UNI: look for wave-uniform trips (below).  Worth it for large coherent batches (4 M rays: -4 %,
// 16.7 M: -5 %), a loss where the texture path is not the busiest unit or uniform trips are rare
// (1 M rays +-0, 262 k rays +5 %, coarse meshes +3 %, incoherent batches +2 %): the test costs two
// ballots and a readlane on every trip.
//
// TEST: this trip has a leaf block.  The leaf block -- three triangle loads, Moller-Trumbore, a
// division: 3 of the trip's 7 vector-memory instructions and about half of its VALU instructions --
// is executed by the whole wave whenever ANY lane has a queued leaf, i.e. on nearly every trip, while
// only ~8 % of the lane-trips have one.  The callers therefore alternate trips with and without it:
// on a trip without, a lane visits its node if it holds at most one queued leaf (so that two new
// ones still fit).  A leaf is tested at most one trip later than before; same tests, same results.
// QN: walk the 32-byte grid nodes instead of the exact 64-byte ones (see tr_rec_q).
template <int Q, int K, bool STATS, bool COMPACT = false, typename W = uint64_t, bool UNI = false, bool TEST = true,
          bool QN = false, bool COLD = true>
TR_HD void tr_fused_step(const tr_bvh_view& b, const tr_ray& r, tr_state_t<W>& st, tr_result& res,
                         tr_topk<K>& top, tr_counters* cnt, const tr_ring ring) {
    // one leaf test per trip: the head of the lane's FIFO (p0, p1, p2).  The node is visited
    // unless the FIFO is full (it then waits one trip: no fetch, no new leaves).
    const int32_t room = TEST ? st.p2 : st.p1;     // must be empty (-1) for the node to be visited
    const bool has_node = st.node >= 0 && room < 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // Wave-uniform trips: 36-44 % of the node visits of an image-shaped batch happen on trips where
    // every visiting lane is at the SAME node (the top of the tree under an 8x8 pixel tile).  The
    // record is then fetched once, by the scalar unit, instead of by a 64-lane gather of four
    // dwordx4 that all hit the same line: no texture-path work at all for that trip.
    if (UNI) {
        const unsigned long long m = __ballot(has_node);
        if (m != 0ull) {
            const int32_t nu = __builtin_amdgcn_readlane(st.node, (int)__builtin_ctzll(m));
            if (__ballot(has_node && st.node != nu) == 0ull) {
                // constant address space: the node array is read-only while queries run, and a
                // uniform address in that space is what selects the scalar memory path
                if constexpr (QN) {
                    typedef const __attribute__((address_space(4))) tr_i4* tr_ci4p;
                    const tr_ci4p sp = (tr_ci4p)(unsigned long long)tr_qnode_ptr<COMPACT>(b, nu);
                    const tr_rec_q rec = {sp[0], sp[1]};
                    tr_fused_body<Q, K, STATS, COMPACT, W, TEST, COLD>(b, r, st, res, top, cnt, ring, has_node, rec);
                } else {
                    typedef const __attribute__((address_space(4))) tr_f4* tr_cf4p;
                    const tr_cf4p sp = (tr_cf4p)(unsigned long long)tr_node_ptr<COMPACT>(b, nu);
                    const tr_rec_f rec = {sp[0], sp[1], sp[2], sp[3]};
                    tr_fused_body<Q, K, STATS, COMPACT, W, TEST, COLD>(b, r, st, res, top, cnt, ring, has_node, rec);
                }
                return;
            }
        }
    }
#endif
    // node index or 0, from the sign bits (a select here loses the SGPR-base + 32-bit-offset
    // addressing of the four node loads)
    const int32_t nidx = st.node & ~(st.node >> 31) & (room >> 31);
    if constexpr (QN) {
        const tr_i4* np = tr_qnode_ptr<COMPACT>(b, nidx);
        const tr_rec_q rec = {np[0], np[1]};
        tr_fused_body<Q, K, STATS, COMPACT, W, TEST, COLD>(b, r, st, res, top, cnt, ring, has_node, rec);
    } else {
        const tr_f4* np = tr_node_ptr<COMPACT>(b, nidx);
        const tr_rec_f rec = {np[0], np[1], np[2], np[3]};
        tr_fused_body<Q, K, STATS, COMPACT, W, TEST, COLD>(b, r, st, res, top, cnt, ring, has_node, rec);
    }
}


// ---- unordered two-phase schedule (any / count / location) --------------------------------------
// Queries that do not prune by distance gain nothing from near-first order or from testing a
// leaf as soon as it is found -- and the fused trip pays for both: its Moller-Trumbore block and
// its three triangle loads run on (almost) every trip of the wave although only ~5 % of the
// lanes have a leaf to test (profiles/r01j: the texture path, TA/TD, is the busiest unit of the
// launch and costs ~18 cycles per wave-instruction however few lanes take part).  Here box-hit
// leaves are only QUEUED (per-lane LIFO of triangle slots in LDS: slot s of lane t at
// base[s * stride]); the wave tests them in a separate leaf phase that runs when it is worth a
// wave-instruction: a lane's queue is nearly full, enough lanes have something queued, or no lane
// has a node left.  Any order of the leaf tests gives the same count / hit set, so results are
// bit-identical to the fused trip and to the oracle.  The triangle's own slab interval (the
// [tn, tf] of the hit predicate) is recomputed from its vertices (tr_tri_hit: the same
// tr_tri_box + tr_slab the builder stored in the parent), so the queue holds 4 bytes per leaf.
#ifndef TR_LEAFQ
#define TR_LEAFQ 6
#endif
struct tr_leafq {
    int32_t* base;
    int32_t stride;
};

template <typename W>
struct tr_ustate_t {
    int32_t node;     // next internal node to visit, -1 = hierarchy exhausted
    uint32_t depth;
    W trail, owned;   // as in tr_state_t
    int32_t nq;       // queued leaves
    int32_t pe;       // a leaf test waiting for tr_drain_exact (-1 = none; tr_fold_leaf)
};

template <typename W>
TR_HD void tr_ustate_init(tr_ustate_t<W>& st) { st.node = 0; st.depth = 0; st.trail = 0; st.owned = 0; st.nq = 0; st.pe = -1; }
template <typename W>
TR_HD bool tr_udone(const tr_ustate_t<W>& st) { return st.node < 0 && st.nq == 0 && st.pe < 0; }
// a lane may visit its node only while the queue can take both children
template <typename W>
TR_HD bool tr_ucan_node(const tr_ustate_t<W>& st) { return st.node >= 0 && st.nq <= TR_LEAFQ - 2; }

// One trip of the unordered schedule.  `go_node`: this lane visits its node (tr_ucan_node);
// `leaf_phase`: WAVE-UNIFORM, the lanes that have a queued leaf pop and test one.  The node's four
// loads are issued before the leaf phase so that both fetches share one memory round trip.
// Children are entered c0 first; for ANY the nearer one first (a hit found sooner ends the ray
// sooner).
template <int Q, int K, bool STATS, bool COMPACT, typename W>
TR_HD void tr_unord_step(const tr_bvh_view& b, const tr_ray& r, bool go_node, bool leaf_phase,
                         tr_ustate_t<W>& st, tr_result& res, tr_topk<K>& top, tr_counters* cnt,
                         const tr_ring ring, const tr_leafq lq) {
    const int32_t nidx = go_node ? st.node : 0;
    const tr_i4* np = COMPACT ? reinterpret_cast<const tr_i4*>(reinterpret_cast<const char*>(b.qnodes) + ((uint32_t)nidx << 5))
                              : reinterpret_cast<const tr_i4*>(b.qnodes + nidx);
    const tr_i4 w0 = np[0], w1 = np[1];
    if (STATS && go_node) cnt->nodes++;
    bool fin = false;
    if (leaf_phase) {
        // pop one slot, fetch the triangle, evaluate the full predicate (box of the triangle ->
        // slab -> Moller-Trumbore); lanes without a queued leaf compute on record 0 and discard
        const bool gl = st.nq > 0;
        int32_t slot = 0;
        if (gl) { st.nq--; slot = lq.base[st.nq * lq.stride]; }
        tr_counters* nc = nullptr;
        const tr_tri t = tr_load_tri<false, COMPACT>(b, slot, nc);
        if (STATS && gl) cnt->tris++;
        fin = tr_fold_leaf<Q, K>(gl, r, t, slot, res, top, st.pe);   // (ANY: the first accepted hit ends the ray)
    }
    if (Q == TR_Q_ANY && fin) { st.node = -1; st.nq = 0; }
    const bool go = go_node && !fin;
    float tn0, tf0, tn1, tf1;
    tr_qnode_slabs(r, b.frame, w0, w1, tn0, tf0, tn1, tf1);
    const int32_t c0 = w1.z, c1 = w1.w;
    bool h0 = tr_slab_hit(tn0, tf0, TR_TLIM) && go;
    bool h1 = tr_slab_hit(tn1, tf1, TR_TLIM) && go;
    if (h0 && c0 < 0) { lq.base[st.nq * lq.stride] = ~c0; st.nq++; h0 = false; }
    if (h1 && c1 < 0) { lq.base[st.nq * lq.stride] = ~c1; st.nq++; h1 = false; }
    if (go) {
        if (h0 || h1) {
            const bool both = h0 && h1;
            const bool swap = (Q == TR_Q_ANY) ? ((both && tn1 < tn0) || !h0) : !h0;   // descend into c1?
            if (both) {
                st.trail |= (W(1) << st.depth);
                if (ring.base) {
                    const uint32_t slot = st.depth & (TR_RING - 1);
                    tr_ring_put(ring, slot, swap ? c0 : c1);
                    st.owned = (st.owned & ~(tr_ring_mask(W(0)) << slot)) | (W(1) << st.depth);
                }
            }
            st.node = swap ? c1 : c0;
            st.depth++;
        } else if (st.trail == 0) {
            st.node = -1;
        } else {
            const uint32_t j = tr_top_bit(st.trail);
            st.trail &= ~(W(1) << j);
            if (ring.base && ((st.owned >> j) & W(1))) st.node = tr_ring_get(ring, j & (TR_RING - 1));
            else st.node = tr_climb<STATS>(b, st.node, st.depth, j, cnt);
            st.depth = j + 1;
        }
    }
    if (leaf_phase) {      // (wave-uniform) the test the leaf phase left undecided
        if (tr_drain_exact<Q, K, COMPACT>(b, r, st.pe, res, top) && Q == TR_Q_ANY) { st.node = -1; st.nq = 0; }
        st.pe = -1;
    }
}

// one ray: node trips while the queue has room, a leaf test otherwise (host simulation; this is
// the schedule with the LATEST possible leaf tests, so it exercises the queue limits)
template <int Q, int K, bool STATS>
TR_HD void tr_traverse_unordered(const tr_bvh_view& b, const tr_ray& r, bool valid, tr_result& res,
                                 tr_topk<K>& top, tr_counters* cnt, const tr_ring ring, const tr_leafq lq) {
    tr_result_init(res);
    if (Q == TR_Q_LOCATION) top.init();
    if (!valid || b.num_tris < 2) return;
    tr_ustate_t<uint64_t> st;
    tr_ustate_init(st);
    while (!tr_udone(st)) {
        const bool cn = tr_ucan_node(st);
        tr_unord_step<Q, K, STATS, false, uint64_t>(b, r, cn, !cn, st, res, top, cnt, ring, lq);
        TR_CONVERGE();
    }
}

// Full traversal of one ray (generic schedule: leaf phase whenever something is queued).
// The wave-level kernels run the two phases under votes instead (traverse.hip).
template <int Q, int K, bool STATS>
TR_HD void tr_traverse(const tr_bvh_view& b, const tr_ray& r, bool valid, tr_result& res,
                       tr_topk<K>& top, tr_counters* cnt, const tr_ring ring = tr_ring{nullptr, 0}) {
    tr_result_init(res);
    if (Q == TR_Q_LOCATION) top.init();
    if (!valid || b.num_tris < 2) return;   // F < 2 is handled by the brute-force kernel
    tr_state st;
    tr_state_init(st);
    while (!tr_done(st)) {
        if (tr_pending(st)) tr_leaf_step<Q, K, STATS>(b, r, st, res, top, cnt);
        else tr_node_step<Q, STATS>(b, r, st, res, cnt, ring);
        TR_CONVERGE();
    }
}
