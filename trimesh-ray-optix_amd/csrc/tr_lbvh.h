// tr_lbvh.h -- per-element pieces of the LBVH construction that are pure functions of their
// inputs (Morton code, Karras 2012 node), shared by the HIP builder kernels (bvh_build.hip)
// and by the g++-compiled host simulation under tests/host_sim (logic check without a GPU).
#pragma once
#include "tr_math.h"

TR_HD uint64_t tr_spread21(uint32_t v) {   // 21 bits -> every third bit
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// 63-bit Morton code of a box centre inside the mesh bounds [mn, mx]
TR_HD uint64_t tr_morton63(const float* box /* lo[3], hi[3] */, const float* mn, const float* mx) {
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        float c = 0.5f * box[k] + 0.5f * box[3 + k];
        float ext = mx[k] - mn[k];
        float u = ext > 0.f ? (c - mn[k]) / ext : 0.f;
        u = fminf(fmaxf(u, 0.f), 1.f);
        float s = u * 2097152.0f;   // 2^21
        uint32_t v = (uint32_t)s;
        q[k] = v > 2097151u ? 2097151u : v;
    }
    return (tr_spread21(q[0]) << 2) | (tr_spread21(q[1]) << 1) | tr_spread21(q[2]);
}

// MODE 0: 63-bit Morton keys, ties broken by sorted position (delta up to 64+31)
// MODE 1: depth-bounded unique keys = top 32 Morton bits << 32 | sorted position (delta < 64)
template <int MODE>
TR_HD uint64_t tr_key(const uint64_t* keys, int64_t i) {
    uint64_t k = keys[i];
    if (MODE == 1) k = ((k >> 31) << 32) | (uint64_t)(uint32_t)i;
    return k;
}

template <int MODE>
TR_HD int tr_delta(const uint64_t* keys, int64_t n, int64_t i, uint64_t ki, int64_t j) {
    if (j < 0 || j >= n) return -1;
    uint64_t kj = tr_key<MODE>(keys, j);
    if (ki == kj) return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);   // MODE 0 only (i != j)
    return __builtin_clzll(ki ^ kj);
}

// Karras 2012, one internal node: children of node i over n sorted keys.
// child encoding: >= 0 internal node index, < 0 leaf with slot ~c.
// *span (optional) = internal nodes of the subtree of node i, itself included (= its leaves - 1).
template <int MODE>
TR_HD void tr_karras_node(const uint64_t* keys, int64_t n, int64_t i, int32_t* child_l,
                          int32_t* child_r, int32_t* span = nullptr) {
    const uint64_t ki = tr_key<MODE>(keys, i);
    const int dp = tr_delta<MODE>(keys, n, i, ki, i + 1), dm = tr_delta<MODE>(keys, n, i, ki, i - 1);
    const int64_t d = dp > dm ? 1 : -1;
    const int dmin = dp > dm ? dm : dp;
    int64_t lmax = 2;
    while (tr_delta<MODE>(keys, n, i, ki, i + lmax * d) > dmin) lmax <<= 1;
    int64_t l = 0;
    for (int64_t t = lmax >> 1; t >= 1; t >>= 1)
        if (tr_delta<MODE>(keys, n, i, ki, i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = tr_delta<MODE>(keys, n, i, ki, j);
    int64_t s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (tr_delta<MODE>(keys, n, i, ki, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    *child_l = (lo == gamma) ? ~(int32_t)gamma : (int32_t)gamma;
    *child_r = (hi == gamma + 1) ? ~(int32_t)(gamma + 1) : (int32_t)(gamma + 1);
    if (span) *span = (int32_t)(hi - lo);
}

// ---- node layout in memory: treelets -------------------------------------------------------------
// Karras numbers a node by one end of its key range: siblings are adjacent, but a parent is up to half
// its subtree away from its children.  The traversal arrays are therefore emitted in TREELET order:
// the hierarchy is cut into treelets of TR_TREELET_LEVELS levels (root, children, grandchildren: up to
// 7 nodes = 224 B of grid nodes, two 128-byte lines), a treelet's nodes are stored in level order, and
// the subtrees below it follow one after the other, depth first.  A descent touches a new line every
// ~3 levels instead of every level (measured on the host-permuted arena, scripts/exp_node_layout.py,
// profiles/r03_node_layout.jsonl: headline -4 %, incoherent shards -4 %; a random order costs +6...11 %).
// Positions are a pure function of the topology: the GPU builder and tests/host_sim derive the same
// permutation, replicas on other GPUs stay bit-identical, and every consumer of the arrays only follows
// child / parent ids (node 0 stays the root).
//
// One treelet: root r at position `base`.  Writes pos[] of its members and, for the roots of the
// treelets below it (the internal children of its last level), their bases and the round in which
// they are due.  span[c] = internal nodes of the subtree of c (tr_karras_node).
#define TR_TREELET_LEVELS 3
TR_HD void tr_treelet_assign(const int32_t* cl, const int32_t* cr, const int32_t* span, int32_t r, int32_t base,
                             int32_t next_round, int32_t* pos, int32_t* bases, int32_t* flag) {
    int32_t m[(1 << TR_TREELET_LEVELS) - 1];
    int nm = 0, lo = 0, hi = 1;
    m[nm++] = r;
    for (int lv = 1; lv < TR_TREELET_LEVELS; lv++) {
        for (int k = lo; k < hi; k++) {
            const int32_t a = cl[m[k]], b = cr[m[k]];
            if (a >= 0) m[nm++] = a;
            if (b >= 0) m[nm++] = b;
        }
        lo = hi; hi = nm;
    }
    for (int k = 0; k < nm; k++) pos[m[k]] = base + k;
    int32_t next = base + nm;
    for (int k = lo; k < hi; k++) {          // the last level's internal children: roots of the next treelets
        const int32_t ch[2] = {cl[m[k]], cr[m[k]]};
        for (int e = 0; e < 2; e++)
            if (ch[e] >= 0) { bases[ch[e]] = next; flag[ch[e]] = next_round; next += span[ch[e]]; }
    }
}
