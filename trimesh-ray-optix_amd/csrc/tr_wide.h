// tr_wide.h -- 8-wide compressed nodes for the streaming launch (round 4; SURVEY.md 8(f)#3 "wider BVH").
//
// The reference gets a wide, compressed hierarchy from optixAccelBuild (closed: ray.cpp:79-93); here it is
// derived from the binary LBVH by COLLAPSING three levels at a time: a wide node = one binary node r at a
// depth that is a multiple of three + its internal children and grandchildren; its <= 8 children ("exits")
// are the leaves met on those three levels and the internal children of the grandchildren (the roots of the
// next wide nodes).  The topology above and below the collapse is the binary tree's, every exit's box is the
// EXACT child box its binary parent stores -- so a traversal of the wide tree reaches exactly the leaves a
// traversal of the binary tree reaches, up to the looseness of the quantised boxes.
//
// Record (96 bytes = six 16-byte loads; the binary grid nodes of one treelet are up to 7 x 32 = 224 bytes):
//   base[3]      the node's own box minimum (float)
//   e[3], n      per-axis scale as a raw binary32 exponent byte (scale = 2^(e - 127)), number of children
//   q[6][8]      child boxes on the node's 8-bit grid, plane-major: lo.x of children 0..7, lo.y, lo.z, hi.x, hi.y, hi.z
//   child[8]     >= 0 wide node index, < 0 leaf (triangle slot ~c)
// Grid plane q of axis k lies at fma(q, scale[k], base[k]) -- the same decode on the builder and in the
// traversal, monotone in q; the builder takes the largest plane <= lo and the smallest >= hi, so the decoded
// box is a SUPERSET of the exact one and the contract's slab arithmetic on it (monotone under box inclusion,
// tr_math.h) never culls what the exact box would accept.  Leaves are decided by the full predicate
// (tr_tri_hit), as with the 16-bit grid nodes: results are bit-identical to every other launch shape.
#pragma once
#include "tr_bvh.h"

struct alignas(32) tr_wnode {
    float base[3];
    uint8_t e[3];
    uint8_t n;
    uint8_t q[6][8];
    int32_t child[8];
};
static_assert(sizeof(tr_wnode) == 96, "wide node must be 96 B");

TR_HD float tr_wscale(uint32_t e) { return tr_u2f(e << 23); }
TR_HD float tr_wdecode(uint32_t q, float scale, float base) { return fmaf((float)q, scale, base); }

// scale exponent of one axis: the smallest power of two with decode(255) >= hi (found with the decode itself)
TR_HD uint32_t tr_wexp(float lo, float hi) {
    const float ext = hi - lo;
    int e = 0;
    if (ext > 0.f && ext <= 3.0e38f) {
        (void)frexpf(ext / 255.0f, &e);          // ext / 255 = m * 2^e, m in [0.5, 1)
    }
    int be = e + 127;
    if (be < 1) be = 1;
    if (be > 254) be = 254;
    while (be < 254 && !(tr_wdecode(255u, tr_wscale((uint32_t)be), lo) >= hi)) be++;
    return (uint32_t)be;
}
// largest q in [0, 255] with decode(q) <= x (decode(0) = base <= x by construction)
TR_HD uint32_t tr_wfloor(float x, float scale, float base) {
    uint32_t a = 0u, z = 255u;
    while (a < z) {
        const uint32_t m = (a + z + 1u) >> 1;
        if (tr_wdecode(m, scale, base) <= x) a = m; else z = m - 1u;
    }
    return a;
}
// smallest q in [0, 255] with decode(q) >= x (decode(255) >= x by construction)
TR_HD uint32_t tr_wceil(float x, float scale, float base) {
    uint32_t a = 0u, z = 255u;
    while (a < z) {
        const uint32_t m = (a + z) >> 1;
        if (tr_wdecode(m, scale, base) >= x) z = m; else a = m + 1u;
    }
    return a;
}

// The fused conservative box test (tr_ray_fuse, tr_bvh.h) for a node with its own frame: per visit and axis
// A = scale_n * k, B = (base_n - o) * k -+ e (the ray's margin e covers every plane of every node: tr_fuse_axis),
// then ONE fma per plane: t' = fma(q, A, B).  Shared by the kernel (traverse_wide.inc) and tests/host_sim.
TR_HD void tr_wfuse_axis(float base_n, float scale_n, float o, float k, float e, float& A, float& Bn, float& Bf) {
    A = scale_n * k;
    const float t0 = base_n - o;
    Bn = fmaf(t0, k, -e);
    Bf = fmaf(t0, k, e);
}

// the exits of the wide node rooted at binary node r, in left-to-right (Morton) order: box (lo[3], hi[3]) and
// binary child id (>= 0: internal node = root of the next wide node, < 0: leaf).  Returns their number (2..8).
struct tr_wexit {
    float lo[3], hi[3];
    int32_t id;
};
TR_HD void tr_wexit_set(tr_wexit& x, const float* box /* lo.x lo.y lo.z hi.z hi.x hi.y */, int32_t id) {
    x.lo[0] = box[0]; x.lo[1] = box[1]; x.lo[2] = box[2];
    x.hi[2] = box[3]; x.hi[0] = box[4]; x.hi[1] = box[5];
    x.id = id;
}
TR_HD int tr_wexits(const tr_node* nodes, int32_t r, tr_wexit* ex) {
    int ne = 0;
    const tr_node n0 = nodes[r];
    for (int s0 = 0; s0 < 2; s0++) {
        const int32_t c0 = s0 ? n0.c1 : n0.c0;
        const float* b0 = s0 ? n0.box1 : n0.box0;
        if (c0 < 0) { tr_wexit_set(ex[ne++], b0, c0); continue; }
        const tr_node n1 = nodes[c0];
        for (int s1 = 0; s1 < 2; s1++) {
            const int32_t c1 = s1 ? n1.c1 : n1.c0;
            const float* b1 = s1 ? n1.box1 : n1.box0;
            if (c1 < 0) { tr_wexit_set(ex[ne++], b1, c1); continue; }
            const tr_node n2 = nodes[c1];
            tr_wexit_set(ex[ne++], n2.box0, n2.c0);
            tr_wexit_set(ex[ne++], n2.box1, n2.c1);
        }
    }
    return ne;
}

// build the record of one wide node from its exits; `widx` maps a binary node id (the root of a wide node) to
// its wide index
template <typename IDX>
TR_HD void tr_wnode_make(const tr_wexit* ex, int ne, const IDX* widx, tr_wnode* out) {
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        lo[k] = ex[0].lo[k]; hi[k] = ex[0].hi[k];
        for (int j = 1; j < ne; j++) { lo[k] = fminf(lo[k], ex[j].lo[k]); hi[k] = fmaxf(hi[k], ex[j].hi[k]); }
    }
    tr_wnode w;
    float sc[3];
    for (int k = 0; k < 3; k++) {
        w.base[k] = lo[k];
        const uint32_t e = tr_wexp(lo[k], hi[k]);
        w.e[k] = (uint8_t)e;
        sc[k] = tr_wscale(e);
    }
    w.n = (uint8_t)ne;
    for (int j = 0; j < 8; j++) {
        if (j < ne) {
            for (int k = 0; k < 3; k++) {
                w.q[k][j] = (uint8_t)tr_wfloor(ex[j].lo[k], sc[k], lo[k]);
                w.q[3 + k][j] = (uint8_t)tr_wceil(ex[j].hi[k], sc[k], lo[k]);
            }
            w.child[j] = ex[j].id < 0 ? ex[j].id : (int32_t)widx[ex[j].id];
        } else {
            for (int k = 0; k < 6; k++) w.q[k][j] = 0;
            w.child[j] = 0x7fffffff;
        }
    }
    *out = w;
}
