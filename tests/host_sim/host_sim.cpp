// host_sim.cpp -- TEST-ONLY: compiles the product's host+device headers (tr_math.h, tr_lbvh.h,
// tr_bvh.h) with g++ and runs the per-lane logic in a plain loop, so the LBVH construction
// rules (Morton keys, Karras nodes, depth-bounded fallback keys) and the stackless
// trail/parent-link traversal can be checked against the oracle on a machine without a GPU.
// It is not reachable from libtriro_hip.so and is not a CPU fallback.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../trimesh-ray-optix_amd/csrc/tr_bvh.h"
#include "../../trimesh-ray-optix_amd/csrc/tr_lbvh.h"
#include "../../trimesh-ray-optix_amd/csrc/tr_wide.h"

struct SimBvh {
    tr_qframe frame = {{0, 0, 0}, {1, 1, 1}};
    std::vector<tr_qnode> qnodes;   // 32-byte grid nodes of the unordered schedule
    std::vector<tr_node> nodes;
    std::vector<tr_link> links;
    std::vector<tr_tri> tris;
    int depth = 0;
    int key_mode = 0;
};

static int g_node_layout = 1;      // as the product's option node_layout: 1 = treelet order

template <int MODE>
static int build_hierarchy(SimBvh& b, const std::vector<uint64_t>& keys, const std::vector<float>& sbox) {
    const int64_t n = (int64_t)keys.size(), ni = n - 1;
    std::vector<int32_t> cl(ni), cr(ni), par(ni, -1), span(ni, 0);
    for (int64_t i = 0; i < ni; i++) {
        tr_karras_node<MODE>(keys.data(), n, i, &cl[i], &cr[i], &span[i]);
    }
    // node layout (tr_lbvh.h): treelet order, level by level from the root, as the GPU builder does it
    std::vector<int32_t> pos(ni), bases(ni, 0), flag(ni, 0);
    for (int64_t i = 0; i < ni; i++) pos[i] = (int32_t)i;
    if (g_node_layout) {
        std::fill(pos.begin(), pos.end(), -1);
        flag[0] = 1;
        for (int32_t round = 1; round <= 80; round++) {
            bool any = false;
            for (int64_t i = 0; i < ni; i++)
                if (flag[i] == round) { tr_treelet_assign(cl.data(), cr.data(), span.data(), (int32_t)i, bases[i], round + 1, pos.data(), bases.data(), flag.data()); any = true; }
            if (!any) break;
        }
    }
    for (int64_t i = 0; i < ni; i++) {
        if (cl[i] >= 0) par[cl[i]] = (int32_t)i;
        if (cr[i] >= 0) par[cr[i]] = (int32_t)i;
    }
    par[0] = -1;
    // level-synchronous refit, as the GPU builder does it
    std::vector<float> ibox(6 * ni);
    std::vector<int32_t> ready(ni, 0);
    int round = 0;
    while (ready[0] == 0 && round < 200) {
        ++round;
        for (int64_t i = 0; i < ni; i++) {
            if (ready[i]) continue;
            if (cl[i] >= 0 && (ready[cl[i]] == 0 || ready[cl[i]] >= round)) continue;
            if (cr[i] >= 0 && (ready[cr[i]] == 0 || ready[cr[i]] >= round)) continue;
            const float* a = cl[i] < 0 ? &sbox[6 * (int64_t)(~cl[i])] : &ibox[6 * (int64_t)cl[i]];
            const float* c = cr[i] < 0 ? &sbox[6 * (int64_t)(~cr[i])] : &ibox[6 * (int64_t)cr[i]];
            for (int k = 0; k < 3; k++) {
                ibox[6 * i + k] = fminf(a[k], c[k]);
                ibox[6 * i + 3 + k] = fmaxf(a[3 + k], c[3 + k]);
            }
            ready[i] = round;
        }
    }
    if (ready[0] == 0) return -1;
    b.nodes.resize(ni);
    b.links.resize(ni);
    b.qnodes.resize(ni);
    auto at = [&](int32_t c) { return c >= 0 ? pos[c] : c; };      // ids in layout order
    for (int64_t i = 0; i < ni; i++) {
        const float* a = cl[i] < 0 ? &sbox[6 * (int64_t)(~cl[i])] : &ibox[6 * (int64_t)cl[i]];
        const float* c = cr[i] < 0 ? &sbox[6 * (int64_t)(~cr[i])] : &ibox[6 * (int64_t)cr[i]];
        tr_node nd;
        tr_node_set_box(nd.box0, a, a + 3);
        tr_node_set_box(nd.box1, c, c + 3);
        nd.c0 = at(cl[i]); nd.c1 = at(cr[i]);
        int32_t p = par[i], sib = 0;
        if (p >= 0) sib = at((cl[p] == (int32_t)i) ? cr[p] : cl[p]);
        nd.parent = p >= 0 ? pos[p] : -1; nd.sibling = sib;
        const int64_t w = pos[i];
        b.nodes[w] = nd;
        b.links[w].parent = nd.parent; b.links[w].sibling = sib;
        tr_qnode qn;
        tr_qnode_set_box(qn.q, a, a + 3, b.frame);
        tr_qnode_set_box(qn.q + 3, c, c + 3, b.frame);
        qn.c0 = nd.c0; qn.c1 = nd.c1;
        b.qnodes[w] = qn;
    }
    return round;
}

// EXPERIMENT (scripts/exp_tree_shape.py): hierarchy over the Morton-sorted leaves that splits every
// range [a, b] at a chosen index instead of at the highest differing Morton bit.
//   split_mode 2: the middle index (count-balanced, depth = ceil(log2 n))
//   split_mode 3: the Karras split if it is not too lopsided (both sides >= 1/8 of the range), else the middle
static int32_t build_split_rec(SimBvh& b, const std::vector<uint64_t>& keys, const std::vector<float>& sbox,
                               std::vector<float>& ibox, std::vector<int32_t>& cl, std::vector<int32_t>& cr,
                               std::vector<int32_t>& par, int64_t a, int64_t z, int32_t& next, int split_mode,
                               int depth, int& maxdepth) {
    // returns child id of the subtree over leaves [a, z]
    if (a == z) return ~(int32_t)a;
    const int32_t me = next++;
    if (depth + 1 > maxdepth) maxdepth = depth + 1;
    int64_t m = (a + z) / 2;
    if (split_mode == 3) {
        const uint64_t ka = keys[a], kz = keys[z];
        if (ka != kz) {
            const int prefix = __builtin_clzll(ka ^ kz);
            int64_t lo = a, hi = z;                 // last index whose key shares more than `prefix` bits with ka
            while (lo < hi) {
                const int64_t mid = (lo + hi + 1) / 2;
                if (__builtin_clzll(ka ^ keys[mid]) > prefix) lo = mid; else hi = mid - 1;
            }
            const int64_t n = z - a + 1, left = lo - a + 1;
            if (left * 8 >= n && (n - left) * 8 >= n) m = lo;
        }
    }
    if (split_mode == 4 && z - a >= 2) {
        // EXPERIMENT (scripts/round4/exp_tree_quality.py): the split index that minimises the surface-area heuristic
        // area(left) * n_left + area(right) * n_right over ALL positions of the Morton-ordered range (a full sweep)
        const int64_t n = z - a + 1;
        std::vector<float> suf(6 * n);
        float bx[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int64_t i = n - 1; i >= 0; i--) {
            const float* s = &sbox[6 * (a + i)];
            for (int k = 0; k < 3; k++) { bx[k] = fminf(bx[k], s[k]); bx[3 + k] = fmaxf(bx[3 + k], s[3 + k]); }
            for (int k = 0; k < 6; k++) suf[6 * i + k] = bx[k];
        }
        auto area = [](const float* q) { const float x = q[3] - q[0], y = q[4] - q[1], zz = q[5] - q[2]; return x * y + y * zz + zz * x; };
        float pre[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        double best = 1e300;
        for (int64_t i = 0; i + 1 < n; i++) {
            const float* s = &sbox[6 * (a + i)];
            for (int k = 0; k < 3; k++) { pre[k] = fminf(pre[k], s[k]); pre[3 + k] = fmaxf(pre[3 + k], s[3 + k]); }
            const double c = (double)area(pre) * (double)(i + 1) + (double)area(&suf[6 * (i + 1)]) * (double)(n - i - 1);
            if (c < best) { best = c; m = a + i; }
        }
    }
    const int32_t l = build_split_rec(b, keys, sbox, ibox, cl, cr, par, a, m, next, split_mode, depth + 1, maxdepth);
    const int32_t r = build_split_rec(b, keys, sbox, ibox, cl, cr, par, m + 1, z, next, split_mode, depth + 1, maxdepth);
    cl[me] = l; cr[me] = r;
    if (l >= 0) par[l] = me;
    if (r >= 0) par[r] = me;
    const float* A = l < 0 ? &sbox[6 * (int64_t)(~l)] : &ibox[6 * (int64_t)l];
    const float* C = r < 0 ? &sbox[6 * (int64_t)(~r)] : &ibox[6 * (int64_t)r];
    for (int k = 0; k < 3; k++) { ibox[6 * me + k] = fminf(A[k], C[k]); ibox[6 * me + 3 + k] = fmaxf(A[3 + k], C[3 + k]); }
    return me;
}
static int build_hierarchy_split(SimBvh& b, const std::vector<uint64_t>& keys, const std::vector<float>& sbox, int split_mode) {
    const int64_t n = (int64_t)keys.size(), ni = n - 1;
    std::vector<int32_t> cl(ni), cr(ni), par(ni, -1);
    std::vector<float> ibox(6 * ni);
    int32_t next = 0;
    int maxdepth = 0;
    build_split_rec(b, keys, sbox, ibox, cl, cr, par, 0, n - 1, next, split_mode, 0, maxdepth);
    b.nodes.resize(ni);
    b.links.resize(ni);
    b.qnodes.resize(ni);
    for (int64_t i = 0; i < ni; i++) {
        const float* a = cl[i] < 0 ? &sbox[6 * (int64_t)(~cl[i])] : &ibox[6 * (int64_t)cl[i]];
        const float* c = cr[i] < 0 ? &sbox[6 * (int64_t)(~cr[i])] : &ibox[6 * (int64_t)cr[i]];
        tr_node nd;
        tr_node_set_box(nd.box0, a, a + 3);
        tr_node_set_box(nd.box1, c, c + 3);
        nd.c0 = cl[i]; nd.c1 = cr[i];
        int32_t p = par[i], sib = 0;
        if (p >= 0) sib = (cl[p] == (int32_t)i) ? cr[p] : cl[p];
        nd.parent = p; nd.sibling = sib;
        b.nodes[i] = nd;
        b.links[i].parent = p; b.links[i].sibling = sib;
        tr_qnode qn;
        tr_qnode_set_box(qn.q, a, a + 3, b.frame);
        tr_qnode_set_box(qn.q + 3, c, c + 3, b.frame);
        qn.c0 = cl[i]; qn.c1 = cr[i];
        b.qnodes[i] = qn;
    }
    return maxdepth;
}

extern "C" {

// force_mode: -1 = as the GPU builder decides (mode 0, fallback to 1 if height > 64), 0/1 force
// morton_shift: drop this many low Morton bits (test hook to create long runs of equal keys)
void* sim_build(const float* verts, int64_t nv, const int32_t* faces, int64_t nf, int force_mode,
                int morton_shift) {
    (void)nv;
    SimBvh* b = new SimBvh();
    if (nf <= 0) return b;
    std::vector<float> tribox(6 * nf);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t f = 0; f < nf; f++) {
        const float* a = verts + 3 * (int64_t)faces[3 * f];
        const float* bb = verts + 3 * (int64_t)faces[3 * f + 1];
        const float* c = verts + 3 * (int64_t)faces[3 * f + 2];
        tr_tri_box(a[0], a[1], a[2], bb[0], bb[1], bb[2], c[0], c[1], c[2], &tribox[6 * f], &tribox[6 * f + 3]);
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(mn[k], tribox[6 * f + k]);
            mx[k] = fmaxf(mx[k], tribox[6 * f + 3 + k]);
        }
    }
    tr_qframe_make(mn, mx, &b->frame);   // as the GPU builder: the grid of the 32-byte nodes from the mesh bounds
    std::vector<uint64_t> keys(nf);
    for (int64_t f = 0; f < nf; f++) keys[f] = (tr_morton63(&tribox[6 * f], mn, mx) >> morton_shift) << morton_shift;
    std::vector<uint32_t> order(nf);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return keys[x] < keys[y]; });
    std::vector<uint64_t> skeys(nf);
    std::vector<float> sbox(6 * nf);
    b->tris.resize(nf);
    for (int64_t k = 0; k < nf; k++) {
        uint32_t f = order[k];
        skeys[k] = keys[f];
        const float* a = verts + 3 * (int64_t)faces[3 * f];
        const float* bb = verts + 3 * (int64_t)faces[3 * f + 1];
        const float* c = verts + 3 * (int64_t)faces[3 * f + 2];
        tr_tri t;
        t.ax = a[0]; t.ay = a[1]; t.az = a[2];
        t.bx = bb[0]; t.by = bb[1]; t.bz = bb[2];
        t.cx = c[0]; t.cy = c[1]; t.cz = c[2];
        t.face = (int32_t)f; t.pad1 = 0;
        t.esum = tr_tri_scale(t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz);
        b->tris[k] = t;
        memcpy(&sbox[6 * k], &tribox[6 * f], 24);
    }
    if (nf >= 2) {
        int h = -1;
        if (force_mode == 2 || force_mode == 3 || force_mode == 4) { h = build_hierarchy_split(*b, skeys, sbox, force_mode); b->key_mode = force_mode; }
        else {
        if (force_mode != 1) { h = build_hierarchy<0>(*b, skeys, sbox); b->key_mode = 0; }
        if (force_mode == 1 || (force_mode < 0 && h > 64)) { h = build_hierarchy<1>(*b, skeys, sbox); b->key_mode = 1; }
        }
        b->depth = h;
    }
    return b;
}

void sim_destroy(void* h) { delete (SimBvh*)h; }
int sim_depth(void* h) { return ((SimBvh*)h)->depth; }
int sim_key_mode(void* h) { return ((SimBvh*)h)->key_mode; }
int64_t sim_num_nodes(void* h) { return (int64_t)((SimBvh*)h)->nodes.size(); }
void sim_get_qnodes(void* h, void* qnodes, float* frame6) {
    SimBvh* b = (SimBvh*)h;
    if (qnodes && !b->qnodes.empty()) memcpy(qnodes, b->qnodes.data(), b->qnodes.size() * sizeof(tr_qnode));
    for (int k = 0; k < 3; k++) { frame6[k] = b->frame.base[k]; frame6[3 + k] = b->frame.scale[k]; }
}
void sim_get(void* h, void* nodes, void* links, void* tris) {
    SimBvh* b = (SimBvh*)h;
    // (a single-triangle mesh has no nodes: memcpy from a null pointer is undefined even for zero bytes)
    if (nodes && !b->nodes.empty()) memcpy(nodes, b->nodes.data(), b->nodes.size() * sizeof(tr_node));
    if (links && !b->links.empty()) memcpy(links, b->links.data(), b->links.size() * sizeof(tr_link));
    if (tris && !b->tris.empty()) memcpy(tris, b->tris.data(), b->tris.size() * sizeof(tr_tri));
}
}  // extern "C"

// traverse with externally supplied arrays (e.g. downloaded from the GPU builder)
static const tr_qnode* g_qnodes = nullptr;            // grid nodes + frame of the arrays handed to sim_query
static tr_qframe g_frame = {{0, 0, 0}, {1, 1, 1}};    // (sim_set_qnodes; needed by the unordered schedule only)
static tr_bvh_view view_of(const tr_node* nodes, const tr_link* links, const tr_tri* tris, int64_t nf) {
    tr_bvh_view v; v.nodes = nodes; v.links = links; v.tris = tris; v.num_tris = nf;
    v.qnodes = g_qnodes; v.frame = g_frame; v.frame_dev = nullptr;
    return v;
}

static int g_use_ring = 1;
static int g_fused = 0;   // 0 generic node/leaf schedule, 1 fused trip (64-bit state), 2 fused compact (32-bit)
static int g_unordered = 0;   // 1: any / count / location run the unordered two-phase schedule (tr_unord_step)

template <int Q>
static void run_query(const tr_bvh_view& v, const float* o, const float* d, int64_t n, uint8_t* hit,
                      uint8_t* front, int32_t* tri, float* loc, float* uv, int32_t* count, uint64_t* stats) {
    tr_counters cnt = {0, 0, 0};
    uint64_t tn = 0, tt = 0, tc = 0;
    for (int64_t i = 0; i < n; i++) {
        tr_ray r;
        bool valid = tr_ray_setup_q(r, v.frame, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        tr_result res;
        tr_topk<1> top;
        cnt.nodes = cnt.tris = cnt.climbs = 0;
        int32_t ring_mem[TR_RING];
        tr_ring ring = {g_use_ring ? ring_mem : nullptr, 1};
        int32_t leafq_mem[TR_LEAFQ];
        tr_leafq lq = {leafq_mem, 1};
        if (v.num_tris >= 2 && g_unordered && (Q == TR_Q_ANY || Q == TR_Q_COUNT)) {
            tr_traverse_unordered<Q, 1, true>(v, r, valid, res, top, &cnt, ring, lq);
        } else if (v.num_tris >= 2 && (g_fused & 3)) {
            tr_result_init(res);
            if (valid && (g_fused & 4)) {
                // the streaming launch's flavour: 32-byte grid nodes, no intervals in the leaf FIFO
                if ((g_fused & 3) == 1) {
                    tr_state_t<uint64_t> fs; tr_state_init(fs);
                    while (!tr_done(fs)) {
                        tr_fused_step<Q, 1, true, false, uint64_t, false, true, true>(v, r, fs, res, top, &cnt, ring);
                        if (!tr_done(fs)) tr_fused_step<Q, 1, true, false, uint64_t, false, false, true>(v, r, fs, res, top, &cnt, ring);
                    }
                } else {
                    tr_state_t<uint32_t> fs; tr_state_init(fs);
                    while (!tr_done(fs)) {
                        tr_fused_step<Q, 1, true, true, uint32_t, false, true, true>(v, r, fs, res, top, &cnt, ring);
                        if (!tr_done(fs)) tr_fused_step<Q, 1, true, true, uint32_t, false, false, true>(v, r, fs, res, top, &cnt, ring);
                    }
                }
            } else if (valid) {
                if ((g_fused & 3) == 1) {
                    tr_state_t<uint64_t> fs; tr_state_init(fs);
                    // as the kernels run it: trips with and without the leaf block alternate
                    while (!tr_done(fs)) {
                        tr_fused_step<Q, 1, true, false, uint64_t, false, true>(v, r, fs, res, top, &cnt, ring);
                        if (!tr_done(fs)) tr_fused_step<Q, 1, true, false, uint64_t, false, false>(v, r, fs, res, top, &cnt, ring);
                    }
                } else {
                    tr_state_t<uint32_t> fs; tr_state_init(fs);
                    while (!tr_done(fs)) {
                        tr_fused_step<Q, 1, true, true, uint32_t, false, true>(v, r, fs, res, top, &cnt, ring);
                        if (!tr_done(fs)) tr_fused_step<Q, 1, true, true, uint32_t, false, false>(v, r, fs, res, top, &cnt, ring);
                    }
                }
            }
        } else if (v.num_tris >= 2) tr_traverse<Q, 1, true>(v, r, valid, res, top, &cnt, ring);
        else {
            res.best_face = -1; res.count = 0; res.best_t = TR_TMAX;
            if (valid && v.num_tris == 1) {
                const tr_tri& t = v.tris[0]; tr_hit h;
                if (tr_tri_test(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, h)) {
                    res.best_t = h.t; res.best_face = t.face; res.best_slot = 0; res.count = 1;
                }
            }
        }
        tn += cnt.nodes; tt += cnt.tris; tc += cnt.climbs;
        if (Q == TR_Q_ANY) hit[i] = res.best_face >= 0;
        if (Q == TR_Q_FIRST) tri[i] = res.best_face;
        if (Q == TR_Q_COUNT) count[i] = res.count;
        if (Q == TR_Q_CLOSEST) {
            float l3[3] = {0, 0, 0}, u2[2] = {0, 0};
            hit[i] = res.best_face >= 0; front[i] = 0; tri[i] = res.best_face;
            if (res.best_face >= 0) {
                const tr_tri& t = v.tris[res.best_slot];
                front[i] = tr_hit_outputs(r, t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, l3, u2);
            }
            memcpy(loc + 3 * i, l3, 12); memcpy(uv + 2 * i, u2, 8);
        }
    }
    if (stats) { stats[0] = (uint64_t)n; stats[1] = tn; stats[2] = tt; stats[3] = tc; }
}

extern "C" {
void sim_set_qnodes(const void* qnodes, const float* f6) {
    g_qnodes = (const tr_qnode*)qnodes;
    for (int k = 0; k < 3; k++) { g_frame.base[k] = f6[k]; g_frame.scale[k] = f6[3 + k]; }
}
void sim_use_ring(int on) { g_use_ring = on; }
void sim_use_fused(int mode) { g_fused = mode; }
void sim_use_unordered(int on) { g_unordered = on; }
void sim_query(int q, const void* nodes, const void* links, const void* tris, int64_t nf, const float* o,
               const float* d, int64_t n, uint8_t* hit, uint8_t* front, int32_t* tri, float* loc, float* uv,
               int32_t* count, uint64_t* stats) {
    tr_bvh_view v = view_of((const tr_node*)nodes, (const tr_link*)links, (const tr_tri*)tris, nf);
    switch (q) {
        case TR_Q_ANY: run_query<TR_Q_ANY>(v, o, d, n, hit, front, tri, loc, uv, count, stats); break;
        case TR_Q_FIRST: run_query<TR_Q_FIRST>(v, o, d, n, hit, front, tri, loc, uv, count, stats); break;
        case TR_Q_CLOSEST: run_query<TR_Q_CLOSEST>(v, o, d, n, hit, front, tri, loc, uv, count, stats); break;
        case TR_Q_COUNT: run_query<TR_Q_COUNT>(v, o, d, n, hit, front, tri, loc, uv, count, stats); break;
    }
}

// per-ray node-visit / leaf-test counts of the closest-hit traversal (divergence studies)
void sim_steps(const void* nodes, const void* links, const void* tris, int64_t nf, const float* o,
               const float* d, int64_t n, int32_t* node_visits, int32_t* tri_tests) {
    tr_bvh_view v = view_of((const tr_node*)nodes, (const tr_link*)links, (const tr_tri*)tris, nf);
    for (int64_t i = 0; i < n; i++) {
        tr_ray r;
        bool valid = tr_ray_setup(r, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        tr_result res; tr_topk<1> top; tr_counters cnt = {0, 0, 0};
        int32_t ring_mem[TR_RING];
        tr_ring ring = {ring_mem, 1};
        tr_traverse<TR_Q_CLOSEST, 1, true>(v, r, valid, res, top, &cnt, ring);
        node_visits[i] = (int32_t)cnt.nodes; tri_tests[i] = (int32_t)cnt.tris;
    }
}

#ifdef TR_COUNT_BOTTOM
// experiment build (scripts/exp_pair_leaves.py): additionally the visits of nodes whose two children
// are leaves and the number of their children that passed the box test, per ray
void sim_steps_bottom(const void* nodes, const void* links, const void* tris, int64_t nf, const float* o,
                      const float* d, int64_t n, int32_t* node_visits, int32_t* tri_tests,
                      int32_t* bottom, int32_t* bottom_hits) {
    tr_bvh_view v = view_of((const tr_node*)nodes, (const tr_link*)links, (const tr_tri*)tris, nf);
    for (int64_t i = 0; i < n; i++) {
        tr_ray r;
        bool valid = tr_ray_setup(r, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        tr_result res; tr_topk<1> top; tr_counters cnt = {};
        int32_t ring_mem[TR_RING];
        tr_ring ring = {ring_mem, 1};
        tr_traverse<TR_Q_CLOSEST, 1, true>(v, r, valid, res, top, &cnt, ring);
        node_visits[i] = (int32_t)cnt.nodes; tri_tests[i] = (int32_t)cnt.tris;
        bottom[i] = (int32_t)cnt.bottom; bottom_hits[i] = (int32_t)cnt.bottom_hits;
    }
}
#endif

// Packet statistics of the UNORDERED (count) traversal for groups of `group` consecutive rays (a wave: the caller
// hands the rays in wave order, e.g. 8x8 pixel tiles): per group
//   out[0] = node visits summed over the rays          (what per-lane traversal pays in lane-visits)
//   out[1] = node visits of the slowest ray            (>= the wave-trips of the per-lane kernel's node phase)
//   out[2] = distinct nodes visited by ANY ray         (= the trips of a wave-packet traversal: one node per trip)
//   out[3] = leaf tests summed over the rays, out[4] = leaf tests of the busiest ray, out[5] = distinct leaves
// (scripts/round4/exp_packet_count.py: is a packet traversal for count / location on image tiles worth building?)
void sim_packet_stats(const void* nodes_, int64_t nf, const float* o, const float* d, int64_t n, int group, int64_t* out) {
    const tr_node* nodes = (const tr_node*)nodes_;
    std::vector<tr_ray> rays((size_t)group);
    std::vector<int32_t> per_ray((size_t)group), per_ray_leaf((size_t)group);
    std::vector<int32_t> stack;
    std::vector<uint64_t> masks;       // per stack entry: which rays of the group reach the node
    for (int64_t g = 0; g * group < n; g++) {
        const int m = (int)((g + 1) * group <= n ? group : n - g * group);
        uint64_t all = 0;
        for (int k = 0; k < m; k++) {
            const int64_t i = g * group + k;
            if (tr_ray_setup(rays[k], o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2])) all |= 1ull << k;
            per_ray[k] = 0; per_ray_leaf[k] = 0;
        }
        int64_t sum = 0, uni = 0, lsum = 0, luni = 0;
        stack.clear(); masks.clear();
        if (nf >= 2 && all) { stack.push_back(0); masks.push_back(all); }
        while (!stack.empty()) {
            const int32_t nd = stack.back(); stack.pop_back();
            const uint64_t mk = masks.back(); masks.pop_back();
            uni++;
            const tr_node& N = nodes[nd];
            uint64_t h0 = 0, h1 = 0;
            for (int k = 0; k < m; k++) {
                if (!((mk >> k) & 1ull)) continue;
                per_ray[k]++; sum++;
                float tn0, tf0, tn1, tf1;
                const tr_f4* np = reinterpret_cast<const tr_f4*>(&N);
                tr_node_slabs(rays[k], np[0], np[1], np[2], tn0, tf0, tn1, tf1);
                if (tr_slab_hit(tn0, tf0, TR_TLIM)) h0 |= 1ull << k;
                if (tr_slab_hit(tn1, tf1, TR_TLIM)) h1 |= 1ull << k;
            }
            const int32_t cs[2] = {N.c0, N.c1};
            const uint64_t hs[2] = {h0, h1};
            for (int e = 0; e < 2; e++) {
                if (!hs[e]) continue;
                if (cs[e] >= 0) { stack.push_back(cs[e]); masks.push_back(hs[e]); }
                else {
                    luni++;
                    for (int k = 0; k < m; k++) if ((hs[e] >> k) & 1ull) { per_ray_leaf[k]++; lsum++; }
                }
            }
        }
        int32_t mx = 0, lmx = 0;
        for (int k = 0; k < m; k++) { mx = per_ray[k] > mx ? per_ray[k] : mx; lmx = per_ray_leaf[k] > lmx ? per_ray_leaf[k] : lmx; }
        out[6 * g + 0] = sum; out[6 * g + 1] = mx; out[6 * g + 2] = uni;
        out[6 * g + 3] = lsum; out[6 * g + 4] = lmx; out[6 * g + 5] = luni;
    }
}

// The fused box test of the grid nodes (tr_ray_fuse / tr_qnode_slabs, round 5) against the contract's three-step form
// on the same decoded boxes: for every (ray, grid node, child) the fused interval must CONTAIN what tr_slab_hit can
// see of the contract's --  tn' <= max(tn, 0)  and  tf' >= min(tf, TR_TLIM)  -- so that the fused traversal visits a
// superset of the nodes.  Returns the number of violations; out[0] = pairs checked, out[1] = children the fused test
// accepts, out[2] = children the contract's test accepts (the looseness paid for one fma per plane).
int64_t sim_check_fused(const void* qnodes_, int64_t nnodes, const float* f6, const float* o, const float* d, int64_t n,
                        int64_t node_stride, int64_t* out) {
    const tr_qnode* qn = (const tr_qnode*)qnodes_;
    tr_qframe f;
    for (int k = 0; k < 3; k++) { f.base[k] = f6[k]; f.scale[k] = f6[3 + k]; }
    int64_t bad = 0, pairs = 0, acc_f = 0, acc_c = 0;
    for (int64_t i = 0; i < n; i++) {
        tr_ray r;
        if (!tr_ray_setup_q(r, f, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2])) continue;
        for (int64_t j = i % node_stride; j < nnodes; j += node_stride) {
            const tr_i4* w = reinterpret_cast<const tr_i4*>(qn + j);
            float a0, b0, a1, b1, c0, e0, c1, e1;
            tr_qnode_slabs(r, f, w[0], w[1], a0, b0, a1, b1);
            tr_qnode_slabs_contract(r, f, w[0], w[1], c0, e0, c1, e1);
            const float fn[2] = {a0, a1}, ff[2] = {b0, b1}, cn[2] = {c0, c1}, cf[2] = {e0, e1};
            for (int c = 0; c < 2; c++) {
                pairs++;
                if (!(fn[c] <= fmaxf(cn[c], 0.0f)) || !(ff[c] >= fminf(cf[c], TR_TLIM))) bad++;
                acc_f += tr_slab_hit(fn[c], ff[c], TR_TLIM) ? 1 : 0;
                acc_c += tr_slab_hit(cn[c], cf[c], TR_TLIM) ? 1 : 0;
            }
        }
    }
    out[0] = pairs; out[1] = acc_f; out[2] = acc_c;
    return bad;
}

// The same for the 8-wide nodes (tr_wide.h): every wide node of the hierarchy (one per binary node at a depth that is a
// multiple of three, built with the product's tr_wexits / tr_wnode_make), every child, fused (tr_wfuse_axis + one fma
// per plane, as traverse_wide.inc) against the contract's decode - subtract - multiply on the node's own 8-bit grid.
int64_t sim_check_fused_wide(const void* nodes_, int64_t nnodes, const float* f6, const float* o, const float* d, int64_t n,
                             int64_t node_stride, int64_t* out) {
    const tr_node* nodes = (const tr_node*)nodes_;
    tr_qframe f;
    for (int k = 0; k < 3; k++) { f.base[k] = f6[k]; f.scale[k] = f6[3 + k]; }
    // roots of the wide nodes
    std::vector<int32_t> roots, stack;
    std::vector<int64_t> widx((size_t)nnodes, 0);
    stack.push_back(0);
    while (!stack.empty()) {
        const int32_t r0 = stack.back(); stack.pop_back();
        widx[r0] = (int64_t)roots.size();
        roots.push_back(r0);
        tr_wexit ex[8];
        const int ne = tr_wexits(nodes, r0, ex);
        for (int j = 0; j < ne; j++) if (ex[j].id >= 0) stack.push_back(ex[j].id);
    }
    std::vector<tr_wnode> wn(roots.size());
    for (size_t w = 0; w < roots.size(); w++) {
        tr_wexit ex[8];
        const int ne = tr_wexits(nodes, roots[w], ex);
        tr_wnode_make(ex, ne, widx.data(), &wn[w]);
    }
    int64_t bad = 0, pairs = 0, acc_f = 0, acc_c = 0;
    for (int64_t i = 0; i < n; i++) {
        tr_ray r;
        if (!tr_ray_setup_q(r, f, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2])) continue;
        const float ro[3] = {r.ox, r.oy, r.oz}, ri[3] = {r.ix, r.iy, r.iz}, rk[3] = {r.kx, r.ky, r.kz}, re[3] = {r.ex, r.ey, r.ez};
        for (size_t w = (size_t)(i % node_stride); w < wn.size(); w += (size_t)node_stride) {
            const tr_wnode& N = wn[w];
            float A[3], Bn[3], Bf[3], sc[3];
            for (int k = 0; k < 3; k++) { sc[k] = tr_wscale(N.e[k]); tr_wfuse_axis(N.base[k], sc[k], ro[k], rk[k], re[k], A[k], Bn[k], Bf[k]); }
            for (int c = 0; c < (int)N.n; c++) {
                float tn = -INFINITY, tf = INFINITY, cn = -INFINITY, cf = INFINITY;
                for (int k = 0; k < 3; k++) {
                    const bool ng = ri[k] < 0.f;
                    const float qe = (float)(ng ? N.q[3 + k][c] : N.q[k][c]), qx = (float)(ng ? N.q[k][c] : N.q[3 + k][c]);
                    tn = fmaxf(tn, fmaf(qe, A[k], Bn[k]));
                    tf = fminf(tf, fmaf(qx, A[k], Bf[k]));
                    cn = fmaxf(cn, (tr_wdecode((uint32_t)qe, sc[k], N.base[k]) - ro[k]) * ri[k]);
                    cf = fminf(cf, (tr_wdecode((uint32_t)qx, sc[k], N.base[k]) - ro[k]) * ri[k]);
                }
                cf *= TR_SLAB_PAD;
                pairs++;
                if (!(tn <= fmaxf(cn, 0.0f)) || !(tf >= fminf(cf, TR_TLIM))) bad++;
                acc_f += tr_slab_hit(tn, tf, TR_TLIM) ? 1 : 0;
                acc_c += tr_slab_hit(cn, cf, TR_TLIM) ? 1 : 0;
            }
        }
    }
    out[0] = pairs; out[1] = acc_f; out[2] = acc_c;
    return bad;
}

// multi-hit: counts[i] hits (uncapped), first min(count,cap) nearest written at i*cap
void sim_location(const void* nodes, const void* links, const void* tris, int64_t nf, const float* o,
                  const float* d, int64_t n, int32_t cap, int32_t* count, int32_t* tri_out, float* t_out) {
    tr_bvh_view v = view_of((const tr_node*)nodes, (const tr_link*)links, (const tr_tri*)tris, nf);
    for (int64_t i = 0; i < n; i++) {
        tr_ray r;
        bool valid = tr_ray_setup_q(r, v.frame, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        tr_result res; tr_topk<8> top; tr_counters cnt;
        int32_t ring_mem[TR_RING];
        tr_ring ring = {g_use_ring ? ring_mem : nullptr, 1};
        int32_t leafq_mem[TR_LEAFQ];
        tr_leafq lq = {leafq_mem, 1};
        if (g_unordered) tr_traverse_unordered<TR_Q_LOCATION, 8, false>(v, r, valid, res, top, &cnt, ring, lq);
        else tr_traverse<TR_Q_LOCATION, 8, false>(v, r, valid, res, top, &cnt, ring);
        count[i] = res.count;
        for (int k = 0; k < 8 && k < cap; k++) { tri_out[i * cap + k] = k < res.count ? top.face[k] : -1; t_out[i * cap + k] = top.t[k]; }
    }
}
}

// ---- the float32 part of the hit predicate against the float64 part, pair by pair (tests/test_host_sim.py) -----------
// For (ray i, triangle i): code = tr_tri_fast's answer (0 proven miss / out of range, 1 proven hit, 2 undecided) and its
// distance; the exact part's answer and distance; and an INDEPENDENT classification in long double of the three edge
// functions (Plucker-style triple products of the exact float32 inputs, 64-bit mantissas: signs are right unless a value
// is below 1e-17 of its terms): inside = 1, outside = 0, too close to call = 2 -- the arbiter of "proven".
extern "C" void sim_tri_fast_vs_exact(const float* o, const float* d, const float* tri, int64_t n, int32_t* code_fast,
                                      float* t_fast, int32_t* hit_exact, float* t_exact, int32_t* truth) {
    for (int64_t i = 0; i < n; i++) {
        tr_ray r;
        tr_ray_setup(r, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        const float* t = tri + 9 * i;
        tr_hit h; h.t = 0.f;
        code_fast[i] = tr_tri_fast(r, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8],
                                   tr_tri_scale(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]), h);
        t_fast[i] = h.t;
        tr_hit he; he.t = 0.f;
        hit_exact[i] = tr_tri_exact(r, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], he) ? 1 : 0;
        t_exact[i] = he.t;
        // long double triple products d . ((P - o) x (Q - o)) for the three edges
        long double P[3][3];
        for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) P[k][c] = (long double)t[3 * k + c] - (long double)o[3 * i + c];
        const long double D[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
        int pos = 0, neg = 0, amb = 0;
        for (int e = 0; e < 3; e++) {
            const long double* A = P[(e + 1) % 3];
            const long double* B = P[(e + 2) % 3];
            const long double cx = A[1] * B[2] - A[2] * B[1], cy = A[2] * B[0] - A[0] * B[2], cz = A[0] * B[1] - A[1] * B[0];
            const long double v = D[0] * cx + D[1] * cy + D[2] * cz;
            const long double mag = fabsl(D[0]) * (fabsl(A[1] * B[2]) + fabsl(A[2] * B[1])) + fabsl(D[1]) * (fabsl(A[2] * B[0]) + fabsl(A[0] * B[2])) +
                                    fabsl(D[2]) * (fabsl(A[0] * B[1]) + fabsl(A[1] * B[0]));
            if (fabsl(v) <= 1e-17L * mag) amb++;
            else if (v > 0) pos++;
            else neg++;
        }
        truth[i] = amb ? 2 : ((pos && neg) ? 0 : 1);
    }
}
