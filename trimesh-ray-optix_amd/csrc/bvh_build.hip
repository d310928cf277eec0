// bvh_build.hip -- on-GPU LBVH builder for gfx950 (replaces optixAccelBuild/optixAccelCompact,
// triro/backend/ray.cpp:27-100).
//
// Pipeline (all on the caller's stream, one host sync to read the tree height):
//   1. k_tri_bounds    per-triangle exact box + mesh bounds (ordered-uint atomic min/max)
//   2. k_morton        63-bit Morton code of the box centre (21 bits/axis), value = face id
//   3. radix sort      LSD, 8 passes x 8 bits, keys u64 + values u32.  One wave64 per tile:
//                      k_rs_count (LDS histogram per wave + per-digit totals) -> k_rs_scan
//                      (digit-major exclusive scan, one workgroup per digit) -> k_rs_scatter (stable: ranks from
//                      8 __ballot's per round, running per-digit base in LDS)
//   4. k_gather        Morton-ordered triangle records (48 B) + their boxes
//   5. k_karras        Karras 2012 binary radix tree: one thread per internal node
//   6. k_refit_round   bottom-up boxes, ONE KERNEL LAUNCH PER TREE LEVEL.  A node is
//                      computed in round r only if both children were finished in a round
//                      < r, so every cross-workgroup read is ordered by a kernel boundary
//                      (gfx950 L2s are per-XCD and not coherent; the classic single-launch
//                      "second thread to arrive" refit needs agent-scope release/acquire
//                      per node and is wrong-not-slow without it).  #rounds = tree height.
//   7. k_emit          64-B traversal nodes {box0, box1, c0, c1, parent, sibling} + links
//
// If the height exceeds 64 (long runs of identical Morton codes) the hierarchy is rebuilt
// with depth-bounded keys (top 32 Morton bits << 32 | sorted position), which caps the
// height at 64 -- the traversal trail is a single 64-bit word.
#include <string.h>

#include "tr_internal.h"
#include "tr_lbvh.h"

namespace {

constexpr int RS_KPT = 16;                 // keys per lane per tile
constexpr int RS_TILE = 64 * RS_KPT;       // 1024 keys per wave-tile
constexpr int RS_WAVES = 4;                // waves (tiles) per workgroup

__device__ __forceinline__ uint32_t enc_f32(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
    uint32_t b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return __uint_as_float(b);
}

__global__ void k_init_bounds(uint32_t* bounds) {
    int t = threadIdx.x;
    if (t < 3) bounds[t] = 0xffffffffu;        // min (encoded)
    else if (t < 6) bounds[t] = 0u;            // max (encoded)
    else if (t == 6) bounds[6] = 0xffffffffu;  // smallest face id with a vertex index outside [0, nv)
}

// Vertex indices of face f, range-checked against nv (the reference hands unchecked index
// buffers to optixAccelBuild, ray.cpp:44-58).  A face with an index outside [0, nv) is never
// dereferenced: it becomes a degenerate triangle at the origin and its id is reported through
// `bad` (atomicMin -> the build fails with TR_ERR_INVALID_ARG naming the first such face).
__device__ __forceinline__ bool load_face(const float* __restrict__ verts, int64_t nv,
                                          const int32_t* __restrict__ faces, int64_t f, float* a,
                                          float* b, float* c) {
    const int32_t ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
    const bool ok = (uint64_t)(int64_t)ia < (uint64_t)nv && (uint64_t)(int64_t)ib < (uint64_t)nv &&
                    (uint64_t)(int64_t)ic < (uint64_t)nv;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        a[k] = ok ? verts[3 * (int64_t)ia + k] : 0.f;
        b[k] = ok ? verts[3 * (int64_t)ib + k] : 0.f;
        c[k] = ok ? verts[3 * (int64_t)ic + k] : 0.f;
    }
    return ok;
}

// ---- 1. triangle boxes + mesh bounds ------------------------------------------------
// grid-stride over the triangles; wave shuffle + LDS block reduction, then 6 atomics per
// workgroup (one atomic per WAVE on the same six words took 1.4 ms at 1.3 M triangles)
__global__ __launch_bounds__(256) void k_tri_bounds(const float* __restrict__ verts, int64_t nv,
                                                    const int32_t* __restrict__ faces, int64_t nf,
                                                    float* __restrict__ tribox,
                                                    uint32_t* __restrict__ bounds) {
    __shared__ float red[4][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf;
         i += (int64_t)gridDim.x * blockDim.x) {
        float a[3], b[3], c[3];
        if (!load_face(verts, nv, faces, i, a, b, c)) atomicMin(&bounds[6], (uint32_t)i);
        float l[3], h[3];
        tr_tri_box(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], l, h);
        float* o = tribox + 6 * i;
        o[0] = l[0]; o[1] = l[1]; o[2] = l[2];
        o[3] = h[0]; o[4] = h[1]; o[5] = h[2];
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], l[k]); hi[k] = fmaxf(hi[k], h[k]); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float l = lo[k], h = hi[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            l = fminf(l, __shfl_xor(l, off));
            h = fmaxf(h, __shfl_xor(h, off));
        }
        if (lane == 0) { red[wave][k] = l; red[wave][3 + k] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const float l = fminf(fminf(red[0][k], red[1][k]), fminf(red[2][k], red[3][k]));
        const float h = fmaxf(fmaxf(red[0][3 + k], red[1][3 + k]), fmaxf(red[2][3 + k], red[3][3 + k]));
        atomicMin(&bounds[k], enc_f32(l));
        atomicMax(&bounds[3 + k], enc_f32(h));
    }
}

// ---- 2. Morton codes -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_morton(const float* __restrict__ tribox, int64_t nf,
                                                const uint32_t* __restrict__ bounds,
                                                uint64_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    float mn[3], mx[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { mn[k] = dec_f32(bounds[k]); mx[k] = dec_f32(bounds[3 + k]); }
    keys[i] = tr_morton63(tribox + 6 * i, mn, mx);
    vals[i] = (uint32_t)i;
}

// ---- 3. radix sort -----------------------------------------------------------------------
// histogram of one 8-bit digit per wave-tile; hist layout is digit-major: hist[d*ntiles+tile]
static_assert(64 * RS_WAVES == 256, "k_rs_count folds the digit totals with one thread per digit");
__global__ __launch_bounds__(64 * RS_WAVES) void k_rs_count(const uint64_t* __restrict__ keys,
                                                            int64_t n, int shift, int64_t ntiles,
                                                            uint32_t* __restrict__ hist,
                                                            uint32_t* __restrict__ gtot) {
    __shared__ uint32_t lh[RS_WAVES][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * RS_WAVES + wave;
#pragma unroll
    for (int k = 0; k < 4; k++) lh[wave][lane + 64 * k] = 0;
    __syncthreads();
    if (tile < ntiles) {
        const int64_t base = tile * RS_TILE;
#pragma unroll 4
        for (int r = 0; r < RS_KPT; r++) {
            int64_t idx = base + (int64_t)r * 64 + lane;
            if (idx < n) atomicAdd(&lh[wave][(keys[idx] >> shift) & 0xff], 1u);
        }
    }
    __syncthreads();
    if (tile < ntiles) {
#pragma unroll
        for (int k = 0; k < 4; k++) hist[(int64_t)(lane + 64 * k) * ntiles + tile] = lh[wave][lane + 64 * k];
    }
    // per-digit totals of the whole pass (waves past the last tile hold zeros)
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) t += lh[w][threadIdx.x];
    if (t) atomicAdd(&gtot[threadIdx.x], t);
}

// exclusive scan of the digit-major histogram in place.  One workgroup per digit: its row
// starts at the sum of the totals of all smaller digits (gtot, accumulated by k_rs_count), so
// the 256 rows scan independently (a single-workgroup scan of all 256*ntiles entries took
// 89 us per pass at 1.3 M triangles, more than count + scatter together).
__global__ __launch_bounds__(256) void k_rs_scan(uint32_t* __restrict__ hist, int64_t ntiles,
                                                 const uint32_t* __restrict__ gtot) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.x;
    uint32_t g = tid < d ? gtot[tid] : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) g += __shfl_xor(g, off);
    if (lane == 0) wsum[wave] = g;
    __syncthreads();
    if (tid == 0) carry_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    uint32_t* row = hist + (int64_t)d * ntiles;
    for (int64_t base = 0; base < ntiles; base += 1024) {
        int64_t i0 = base + (int64_t)tid * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < ntiles) ? row[i0 + k] : 0u;
        uint32_t s = v[0] + v[1] + v[2] + v[3];
        uint32_t inc = s;   // inclusive wave scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t wpre = 0;
        for (int w = 0; w < wave; w++) wpre += wsum[w];
        const uint32_t carry = carry_s;
        uint32_t ex = carry + wpre + inc - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < ntiles) row[i0 + k] = ex;
            ex += v[k];
        }
        __syncthreads();
        if (tid == 255) carry_s = carry + wpre + inc;
        __syncthreads();
    }
}

// stable scatter: one wave per tile
__global__ __launch_bounds__(64 * RS_WAVES) void k_rs_scatter(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int64_t n,
    int shift, int64_t ntiles, const uint32_t* __restrict__ hist_scanned,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t base[RS_WAVES][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * RS_WAVES + wave;
    if (tile >= ntiles) return;   // whole wave exits together; no block-level barrier below
#pragma unroll
    for (int k = 0; k < 4; k++)
        base[wave][lane + 64 * k] = hist_scanned[(int64_t)(lane + 64 * k) * ntiles + tile];
    __builtin_amdgcn_wave_barrier();
    const uint64_t lt = (1ull << lane) - 1ull;
    const int64_t tbase = tile * RS_TILE;
    for (int r = 0; r < RS_KPT; r++) {
        int64_t idx = tbase + (int64_t)r * 64 + lane;
        bool valid = idx < n;
        uint64_t key = valid ? keys_in[idx] : 0ull;
        uint32_t val = valid ? vals_in[idx] : 0u;
        uint32_t digit = (uint32_t)(key >> shift) & 0xffu;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (digit >> b) & 1u;
            uint64_t bal = __ballot(valid && bit);
            peers &= bit ? bal : ~bal;
        }
        uint32_t rank = (uint32_t)__popcll(peers & lt);
        uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (valid) pos = base[wave][digit] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) base[wave][digit] += cnt;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
    }
}

// ---- 4. gather Morton-ordered triangle records -------------------------------------------
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ verts, int64_t nv,
                                                const int32_t* __restrict__ faces,
                                                const uint32_t* __restrict__ order, int64_t nf,
                                                tr_tri* __restrict__ tris,
                                                float* __restrict__ sbox) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nf) return;
    int64_t f = order[k];
    float a[3], b[3], c[3];
    load_face(verts, nv, faces, f, a, b, c);   // bad faces were reported by k_tri_bounds
    tr_tri t;
    t.ax = a[0]; t.ay = a[1]; t.az = a[2];
    t.bx = b[0]; t.by = b[1]; t.bz = b[2];
    t.cx = c[0]; t.cy = c[1]; t.cz = c[2];
    t.face = (int32_t)f; t.pad1 = 0;
    t.esum = tr_tri_scale(t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz);
    tris[k] = t;
    float lo[3], hi[3];
    tr_tri_box(t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, lo, hi);
    float* o = sbox + 6 * k;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
    o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
}

// ---- 5. Karras hierarchy ------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_karras(const uint64_t* __restrict__ keys, int64_t n,
                                                int32_t* __restrict__ childL,
                                                int32_t* __restrict__ childR,
                                                int32_t* __restrict__ parent,
                                                int32_t* __restrict__ span) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int32_t cl, cr, sp;
    tr_karras_node<MODE>(keys, n, i, &cl, &cr, &sp);
    childL[i] = cl;
    childR[i] = cr;
    span[i] = sp;
    if (cl >= 0) parent[cl] = (int32_t)i;
    if (cr >= 0) parent[cr] = (int32_t)i;
    if (i == 0) parent[0] = -1;
}

// ---- 5b. node layout: positions of the nodes in treelet order, one launch per treelet level (top-down)
__global__ void k_layout_init(int32_t* __restrict__ bases, int32_t* __restrict__ flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { bases[0] = 0; flag[0] = 1; }
}
__global__ __launch_bounds__(256) void k_layout_round(const int32_t* __restrict__ childL, const int32_t* __restrict__ childR,
                                                      const int32_t* __restrict__ span, int32_t* __restrict__ pos,
                                                      int32_t* __restrict__ bases, int32_t* __restrict__ flag,
                                                      int64_t ninternal, int32_t round) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ninternal || flag[i] != round) return;
    tr_treelet_assign(childL, childR, span, (int32_t)i, bases[i], round + 1, pos, bases, flag);
}

// ---- 6. refit, one launch per level ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_refit_round(const int32_t* __restrict__ childL,
                                                     const int32_t* __restrict__ childR,
                                                     const float* __restrict__ sbox,
                                                     float* __restrict__ ibox,
                                                     int32_t* __restrict__ ready, int64_t ninternal,
                                                     int32_t round) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ninternal) return;
    if (ready[i] != 0) return;
    int32_t cl = childL[i], cr = childR[i];
    if (cl >= 0) { int32_t r = ready[cl]; if (r == 0 || r >= round) return; }
    if (cr >= 0) { int32_t r = ready[cr]; if (r == 0 || r >= round) return; }
    const float* a = cl < 0 ? sbox + 6 * (int64_t)(~cl) : ibox + 6 * (int64_t)cl;
    const float* b = cr < 0 ? sbox + 6 * (int64_t)(~cr) : ibox + 6 * (int64_t)cr;
    float* o = ibox + 6 * i;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o[k] = fminf(a[k], b[k]);
        o[3 + k] = fmaxf(a[3 + k], b[3 + k]);
    }
    ready[i] = round;
}

// ---- 7. emit traversal nodes -------------------------------------------------------------------
// quantisation frame of the 32-byte node array from the encoded mesh bounds of step 1 (one
// thread; the host derives the same frame from the same six floats with the same function)
__global__ void k_qframe(const uint32_t* __restrict__ bounds, tr_qframe* __restrict__ frame) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) { mn[k] = dec_f32(bounds[k]); mx[k] = dec_f32(bounds[3 + k]); }
    tr_qframe f;
    tr_qframe_make(mn, mx, &f);
    *frame = f;
}
// the same from a float box lo[3], hi[3] (refit: the new root box)
__global__ void k_qframe_box(const float* __restrict__ box, tr_qframe* __restrict__ frame) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    tr_qframe f;
    tr_qframe_make(box, box + 3, &f);
    *frame = f;
}

__global__ __launch_bounds__(256) void k_emit(const int32_t* __restrict__ childL,
                                              const int32_t* __restrict__ childR,
                                              const int32_t* __restrict__ parent,
                                              const float* __restrict__ sbox,
                                              const float* __restrict__ ibox, int64_t ninternal,
                                              const tr_qframe* __restrict__ frame,
                                              const int32_t* __restrict__ pos,
                                              tr_node* __restrict__ nodes,
                                              tr_link* __restrict__ links,
                                              tr_qnode* __restrict__ qnodes) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ninternal) return;
    int32_t cl = childL[i], cr = childR[i];
    const float* a = cl < 0 ? sbox + 6 * (int64_t)(~cl) : ibox + 6 * (int64_t)cl;
    const float* b = cr < 0 ? sbox + 6 * (int64_t)(~cr) : ibox + 6 * (int64_t)cr;
    tr_node nd;
    tr_node_set_box(nd.box0, a, a + 3);
    tr_node_set_box(nd.box1, b, b + 3);
    int32_t p = parent[i];
    int32_t sib = 0;
    if (p >= 0) sib = (childL[p] == (int32_t)i) ? childR[p] : childL[p];
    // the arrays are written in layout order (tr_lbvh.h, treelets): node i lives at pos[i], ids follow
    // (a node the layout has not reached yet -- a speculative emit of a tree that is higher than guessed
    // -- is skipped: the emit is repeated once the tree is complete)
    if (pos && pos[i] < 0) return;
    if (pos) {
        if (cl >= 0) cl = pos[cl];
        if (cr >= 0) cr = pos[cr];
        if (p >= 0) { if (sib >= 0) sib = pos[sib]; p = pos[p]; }
        i = pos[i];
    }
    nd.c0 = cl; nd.c1 = cr;
    nd.parent = p; nd.sibling = sib;
    nodes[i] = nd;
    tr_link l; l.parent = p; l.sibling = sib;
    links[i] = l;
    // the same two boxes on the 16-bit grid (supersets), for the unordered schedule
    const tr_qframe f = *frame;
    tr_qnode qn;
    tr_qnode_set_box(qn.q, a, a + 3, f);
    tr_qnode_set_box(qn.q + 3, b, b + 3, f);
    qn.c0 = cl; qn.c1 = cr;
    qnodes[i] = qn;
}

// ---- refit (same topology, new vertex positions) ------------------------------------------------
__global__ __launch_bounds__(256) void k_regather(const float* __restrict__ verts, int64_t nv,
                                                  const int32_t* __restrict__ faces, int64_t nf,
                                                  tr_tri* __restrict__ tris, float* __restrict__ sbox,
                                                  uint32_t* __restrict__ bad) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nf) return;
    const int64_t f = tris[k].face;
    float a[3], b[3], c[3];
    if (!load_face(verts, nv, faces, f, a, b, c)) atomicMin(bad, (uint32_t)f);
    tr_tri t;
    t.ax = a[0]; t.ay = a[1]; t.az = a[2];
    t.bx = b[0]; t.by = b[1]; t.bz = b[2];
    t.cx = c[0]; t.cy = c[1]; t.cz = c[2];
    t.face = (int32_t)f; t.pad1 = 0;
    t.esum = tr_tri_scale(t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz);
    tris[k] = t;
    float lo[3], hi[3];
    tr_tri_box(t.ax, t.ay, t.az, t.bx, t.by, t.bz, t.cx, t.cy, t.cz, lo, hi);
    float* o = sbox + 6 * k;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
    o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
}

// one tree level per launch, children taken from the traversal nodes themselves
__global__ __launch_bounds__(256) void k_refit_nodes_round(const tr_node* __restrict__ nodes,
                                                           const float* __restrict__ sbox,
                                                           float* __restrict__ ibox,
                                                           int32_t* __restrict__ ready,
                                                           int64_t ninternal, int32_t round) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ninternal) return;
    if (ready[i] != 0) return;
    const int32_t cl = nodes[i].c0, cr = nodes[i].c1;
    if (cl >= 0) { int32_t r = ready[cl]; if (r == 0 || r >= round) return; }
    if (cr >= 0) { int32_t r = ready[cr]; if (r == 0 || r >= round) return; }
    const float* a = cl < 0 ? sbox + 6 * (int64_t)(~cl) : ibox + 6 * (int64_t)cl;
    const float* b = cr < 0 ? sbox + 6 * (int64_t)(~cr) : ibox + 6 * (int64_t)cr;
    float* o = ibox + 6 * i;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o[k] = fminf(a[k], b[k]);
        o[3 + k] = fmaxf(a[3 + k], b[3 + k]);
    }
    ready[i] = round;
}

__global__ __launch_bounds__(256) void k_update_boxes(tr_node* __restrict__ nodes,
                                                      const float* __restrict__ sbox,
                                                      const float* __restrict__ ibox,
                                                      int64_t ninternal,
                                                      const tr_qframe* __restrict__ frame,
                                                      tr_qnode* __restrict__ qnodes) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ninternal) return;
    const int32_t cl = nodes[i].c0, cr = nodes[i].c1;
    const float* a = cl < 0 ? sbox + 6 * (int64_t)(~cl) : ibox + 6 * (int64_t)cl;
    const float* b = cr < 0 ? sbox + 6 * (int64_t)(~cr) : ibox + 6 * (int64_t)cr;
    tr_node_set_box(nodes[i].box0, a, a + 3);
    tr_node_set_box(nodes[i].box1, b, b + 3);
    const tr_qframe f = *frame;          // the grid follows the new bounds
    tr_qnode_set_box(qnodes[i].q, a, a + 3, f);
    tr_qnode_set_box(qnodes[i].q + 3, b, b + 3, f);
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carver {
    char* base; size_t off = 0;
    template <class T> T* take(size_t count) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

}  // namespace

// Arena layout for `nf` triangles; returns total bytes and sets the pointers when base != 0
static size_t carve_arena(tr_bvh* bvh, char* base, int64_t nf) {
    Carver c{base};
    int64_t nn = nf >= 2 ? nf - 1 : 0;
    tr_node* nodes = c.take<tr_node>((size_t)(nn > 0 ? nn : 1));
    tr_link* links = c.take<tr_link>((size_t)(nn > 0 ? nn : 1));
    tr_tri* tris = c.take<tr_tri>((size_t)(nf > 0 ? nf : 1));
    tr_qnode* qnodes = c.take<tr_qnode>((size_t)(nn > 0 ? nn : 1));
    if (base) { bvh->nodes = nodes; bvh->links = links; bvh->tris = tris; bvh->qnodes = qnodes; }
    return align_up(c.off, 256);
}

// (re)allocate the arena for `nf` triangles and point nodes/links/tris into it
int tr_arena_alloc(tr_bvh* bvh, int64_t nf) {
    if (!bvh->arena || bvh->capacity_tris < nf) {
        // the new arena first: if it cannot be had, the handle keeps a consistent (old) state;
        // the caller resets the hierarchy on any build failure
        size_t bytes = carve_arena(bvh, nullptr, nf);
        void* fresh = nullptr;
        TR_HIP_TRY(hipMalloc(&fresh, bytes));
        if (bvh->arena) (void)hipFree(bvh->arena);   // hipFree synchronises: no launch still reads it
        bvh->arena = fresh;
        bvh->arena_bytes = (int64_t)bytes;
        bvh->capacity_tris = nf;
    }
    carve_arena(bvh, (char*)bvh->arena, nf);
    return TR_OK;
}

int64_t tr_arena_used_bytes(int64_t nf) { return (int64_t)carve_arena(nullptr, nullptr, nf); }

int tr_bvh_sync_frame(tr_bvh* bvh) {
    if (!bvh->frame_dev) TR_HIP_TRY(hipMalloc((void**)&bvh->frame_dev, sizeof(tr_qframe)));
    TR_HIP_TRY(hipMemcpy(bvh->frame_dev, &bvh->frame, sizeof(tr_qframe), hipMemcpyHostToDevice));
    return TR_OK;
}

void tr_bvh_reset(tr_bvh* bvh) {
    bvh->wide_valid = false; bvh->wide_unavailable = false;
    bvh->frame = tr_qframe{{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
    bvh->num_tris = 0; bvh->num_nodes = 0; bvh->depth = 0; bvh->key_mode = 0;
    for (int k = 0; k < 3; k++) { bvh->aabb_min[k] = 0.f; bvh->aabb_max[k] = 0.f; }
    for (int k = 0; k < TR_SCHED_SLOTS; k++) { bvh->sched[k].nblocks = 0; }
}

int tr_build_impl(tr_bvh* bvh, const float* d_vertices, int64_t nv, const int32_t* d_faces,
                  int64_t nf, hipStream_t stream) {
    if (nf < 0 || nv < 0) return tr_fail(TR_ERR_INVALID_ARG, "negative mesh size");
    if (nf >= (int64_t)1 << 31) return tr_fail(TR_ERR_INVALID_ARG, "more than 2^31-1 triangles");
    bvh->wide_valid = false; bvh->wide_unavailable = false;      // the wide nodes are derived data: rebuilt by the next streaming query
    // an empty vertex array may be NULL: every face is then out of range and reported as such
    if (nf > 0 && (!d_faces || (!d_vertices && nv > 0))) return tr_fail(TR_ERR_INVALID_ARG, "null mesh pointer");

    {
        const int as = tr_arena_alloc(bvh, nf);
        if (as != TR_OK) {
            const std::string msg = tr_last_error();
            tr_bvh_reset(bvh);
            tr_set_error(msg);
            return as;
        }
    }
    bvh->num_tris = nf;
    bvh->num_nodes = nf >= 2 ? nf - 1 : 0;
    bvh->depth = 0;
    bvh->key_mode = 0;
    // A learned launch order describes the rays, not the mesh: after a rebuild (an animation step,
    // `update_raw`) it is one frame stale, which is a far better hint than none -- keep it and
    // measure again on the next launches (a deferred sort of the last measurement, if one is pending, runs before them:
    // sched_acquire)
    for (int k = 0; k < TR_SCHED_SLOTS; k++) { bvh->sched[k].launches = 0; }
    for (int k = 0; k < 3; k++) { bvh->aabb_min[k] = 0.f; bvh->aabb_max[k] = 0.f; }
    if (nf == 0) return TR_OK;

    // temporaries: one allocation
    const int64_t ntiles = cdiv(nf, RS_TILE);
    Carver tc{nullptr};
    auto plan = [&](Carver& c, float*& tribox, float*& sbox, float*& ibox, uint64_t*& k0,
                    uint64_t*& k1, uint32_t*& v0, uint32_t*& v1, uint32_t*& hist,
                    uint32_t*& gtot, uint32_t*& bounds, int32_t*& cl, int32_t*& cr, int32_t*& par, int32_t*& ready,
                    int32_t*& lay) {
        tribox = c.take<float>(6 * (size_t)nf);
        sbox = c.take<float>(6 * (size_t)nf);
        ibox = c.take<float>(6 * (size_t)nf);
        k0 = c.take<uint64_t>((size_t)nf);
        k1 = c.take<uint64_t>((size_t)nf);
        v0 = c.take<uint32_t>((size_t)nf);
        v1 = c.take<uint32_t>((size_t)nf);
        hist = c.take<uint32_t>(256 * (size_t)ntiles);
        gtot = c.take<uint32_t>(8 * 256);
        bounds = c.take<uint32_t>(8 + 8);   // 8 words of bounds / flags + the quantisation frame (6 floats)
        cl = c.take<int32_t>((size_t)nf);
        cr = c.take<int32_t>((size_t)nf);
        par = c.take<int32_t>((size_t)nf);
        ready = c.take<int32_t>((size_t)nf);
        lay = c.take<int32_t>(4 * (size_t)nf);      // node layout: span | pos | bases | flag
    };
    float *tribox, *sbox, *ibox; uint64_t *k0, *k1; uint32_t *v0, *v1, *hist, *gtot, *bounds;
    int32_t *cl, *cr, *par, *ready, *lay;
    plan(tc, tribox, sbox, ibox, k0, k1, v0, v1, hist, gtot, bounds, cl, cr, par, ready, lay);
    tr_device_state* st;
    TR_TRY(tr_get_device_state(bvh->device, &st));
    void* temp = nullptr;
    TR_TRY(tr_build_temp_acquire(st, align_up(tc.off, 256), &temp));   // holds st->build_mutex
    Carver tc2{(char*)temp};
    plan(tc2, tribox, sbox, ibox, k0, k1, v0, v1, hist, gtot, bounds, cl, cr, par, ready, lay);
    int32_t* span = lay; int32_t* lpos = lay + nf; int32_t* lbase = lay + 2 * nf; int32_t* lflag = lay + 3 * nf;
    const bool treelets = tr_opts().node_layout != 0;

    int status = TR_OK;
    auto check = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && status == TR_OK)
            status = tr_fail(TR_ERR_HIP, std::string(what) + ": " + hipGetErrorName(e));
    };
    const int TB = 256;
    const unsigned gF = (unsigned)cdiv(nf, TB);

    tr_qframe* d_frame = reinterpret_cast<tr_qframe*>(bounds + 8);
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, stream, bounds);
    hipLaunchKernelGGL(k_tri_bounds, dim3(gF < 1024u ? gF : 1024u), dim3(TB), 0, stream, d_vertices, nv, d_faces, nf, tribox, bounds);
    hipLaunchKernelGGL(k_qframe, dim3(1), dim3(64), 0, stream, bounds, d_frame);
    check(hipGetLastError(), "k_tri_bounds");
    // mesh bounds back to the host; completes with the first synchronisation below
    uint32_t hb[8] = {0};
    hb[6] = 0xffffffffu;
    check(hipMemcpyAsync(hb, bounds, sizeof(uint32_t) * 7, hipMemcpyDeviceToHost, stream), "memcpy bounds");

    if (nf == 1) {
        // single triangle: no hierarchy; queries use the brute-force kernel
        hipLaunchKernelGGL(k_morton, dim3(gF), dim3(TB), 0, stream, tribox, nf, bounds, k0, v0);
        hipLaunchKernelGGL(k_gather, dim3(gF), dim3(TB), 0, stream, d_vertices, nv, d_faces, v0, nf, bvh->tris, sbox);
        check(hipGetLastError(), "k_gather");
    } else {
        hipLaunchKernelGGL(k_morton, dim3(gF), dim3(TB), 0, stream, tribox, nf, bounds, k0, v0);
        check(hipGetLastError(), "k_morton");
        uint64_t* kin = k0; uint64_t* kout = k1; uint32_t* vin = v0; uint32_t* vout = v1;
        const unsigned gT = (unsigned)cdiv(ntiles, RS_WAVES);
        check(hipMemsetAsync(gtot, 0, sizeof(uint32_t) * 8 * 256, stream), "memset digit totals");
        for (int pass = 0; pass < 8; pass++) {
            int shift = 8 * pass;
            hipLaunchKernelGGL(k_rs_count, dim3(gT), dim3(64 * RS_WAVES), 0, stream, kin, nf, shift, ntiles, hist, gtot + 256 * pass);
            hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, stream, hist, ntiles, gtot + 256 * pass);
            hipLaunchKernelGGL(k_rs_scatter, dim3(gT), dim3(64 * RS_WAVES), 0, stream, kin, vin, nf, shift,
                               ntiles, hist, kout, vout);
            uint64_t* tk = kin; kin = kout; kout = tk;
            uint32_t* tv = vin; vin = vout; vout = tv;
        }
        check(hipGetLastError(), "radix sort");
        // after 8 passes the sorted data is back in (k0, v0) == (kin, vin)
        hipLaunchKernelGGL(k_gather, dim3(gF), dim3(TB), 0, stream, d_vertices, nv, d_faces, vin, nf, bvh->tris, sbox);
        check(hipGetLastError(), "k_gather");

        const int64_t ni = nf - 1;
        const unsigned gI = (unsigned)cdiv(ni, TB);
        // Speculative schedule: enqueue the number of refit rounds the tree is expected to need
        // (the previous build's height for a rebuild, log2(n)+10 otherwise) and k_emit, then
        // read the root's round back ONCE.  More rounds follow only if the root was not reached.
        int32_t guess = bvh->depth > 0 ? bvh->depth + 1 : 10;
        if (bvh->depth <= 0) for (int64_t m = 1; m < nf; m <<= 1) ++guess;
        if (guess > 64) guess = 64;
        bvh->depth = 0;
        for (int mode = 0; mode < 2 && status == TR_OK; mode++) {
            if (mode == 0)
                hipLaunchKernelGGL(k_karras<0>, dim3(gI), dim3(TB), 0, stream, kin, nf, cl, cr, par, span);
            else
                hipLaunchKernelGGL(k_karras<1>, dim3(gI), dim3(TB), 0, stream, kin, nf, cl, cr, par, span);
            check(hipGetLastError(), "k_karras");
            check(hipMemsetAsync(ready, 0, sizeof(int32_t) * (size_t)ni, stream), "memset ready");
            // Node layout (top-down, one launch per treelet level) on the builder's side stream, beside
            // the refit rounds (bottom-up) on the caller's: both need only the Karras hierarchy; the first
            // emit waits for both.  As many levels as the guessed height needs; a higher tree gets the
            // rest on the caller's stream further down.
            int32_t lround = 0;              // treelet levels laid out so far
            bool lay_joined = true;
            if (treelets) {
                check(hipEventRecord(st->build_fork, stream), "record fork");
                check(hipStreamWaitEvent(st->build_side, st->build_fork, 0), "side waits");
                check(hipMemsetAsync(lflag, 0, sizeof(int32_t) * (size_t)ni, st->build_side), "memset layout flags");
                check(hipMemsetAsync(lpos, 0xff, sizeof(int32_t) * (size_t)ni, st->build_side), "memset layout positions");
                hipLaunchKernelGGL(k_layout_init, dim3(1), dim3(64), 0, st->build_side, lbase, lflag);
                while (lround * TR_TREELET_LEVELS < guess + TR_TREELET_LEVELS) {
                    ++lround;
                    hipLaunchKernelGGL(k_layout_round, dim3(gI), dim3(TB), 0, st->build_side, cl, cr, span, lpos, lbase, lflag, ni, lround);
                }
                check(hipGetLastError(), "k_layout_round");
                check(hipEventRecord(st->build_join, st->build_side), "record join");
                lay_joined = false;
            }
            int32_t root_ready = 0;
            int32_t round = 0;
            const int32_t max_rounds = 160;   // > 64 + 32 + slack
            int32_t batch = guess;
            while (root_ready == 0 && round < max_rounds && status == TR_OK) {
                for (int k = 0; k < batch; k++) {
                    ++round;
                    hipLaunchKernelGGL(k_refit_round, dim3(gI), dim3(TB), 0, stream, cl, cr, sbox, ibox, ready, ni, round);
                }
                check(hipGetLastError(), "k_refit_round");
                // the layout needs one round per TR_TREELET_LEVELS levels; the tree is at most `round` high
                // if its root has been reached (if not, both loops continue below)
                if (!lay_joined) { check(hipStreamWaitEvent(stream, st->build_join, 0), "join layout"); lay_joined = true; }
                while (treelets && lround * TR_TREELET_LEVELS < round + TR_TREELET_LEVELS) {
                    ++lround;
                    hipLaunchKernelGGL(k_layout_round, dim3(gI), dim3(TB), 0, stream, cl, cr, span, lpos, lbase, lflag, ni, lround);
                }
                // harmless if the root is not final yet: it is launched again below
                hipLaunchKernelGGL(k_emit, dim3(gI), dim3(TB), 0, stream, cl, cr, par, sbox, ibox, ni, d_frame,
                                   treelets ? lpos : nullptr, bvh->nodes, bvh->links, bvh->qnodes);
                check(hipGetLastError(), "k_emit");
                check(hipMemcpyAsync(&root_ready, ready, sizeof(int32_t), hipMemcpyDeviceToHost, stream), "memcpy root");
                check(hipStreamSynchronize(stream), "sync refit");
                if (status == TR_OK && hb[6] != 0xffffffffu) break;   // malformed mesh: reported below
                batch = 8;
            }
            if (hb[6] != 0xffffffffu) break;
            if (status != TR_OK) break;
            if (root_ready == 0) { status = tr_fail(TR_ERR_INTERNAL, "refit did not reach the root"); break; }
            bvh->depth = root_ready;
            bvh->key_mode = mode;
            if (root_ready <= 64) break;
            if (mode == 1) { status = tr_fail(TR_ERR_INTERNAL, "tree height > 64 with bounded keys"); break; }
        }
    }
    // drain the streams before the temporaries are handed back (the side stream too: an error path may
    // have left its layout rounds unjoined)
    check(hipStreamSynchronize(stream), "sync build");
    if (treelets) check(hipStreamSynchronize(st->build_side), "sync build side stream");
    if (status == TR_OK && hb[6] != 0xffffffffu)
        status = tr_fail(TR_ERR_INVALID_ARG, "face " + std::to_string(hb[6]) + " has a vertex index outside [0, " +
                                                 std::to_string(nv) + ")");
    if (status == TR_OK) {
        for (int k = 0; k < 3; k++) {
            uint32_t e0 = hb[k], e1 = hb[3 + k];
            uint32_t b0 = (e0 & 0x80000000u) ? (e0 & 0x7fffffffu) : ~e0;
            uint32_t b1 = (e1 & 0x80000000u) ? (e1 & 0x7fffffffu) : ~e1;
            memcpy(&bvh->aabb_min[k], &b0, 4);
            memcpy(&bvh->aabb_max[k], &b1, 4);
        }
        tr_qframe_make(bvh->aabb_min, bvh->aabb_max, &bvh->frame);   // == k_qframe's (same function, same bounds)
        if (tr_bvh_sync_frame(bvh) != TR_OK && status == TR_OK) status = TR_ERR_HIP;
    }
    int rs = tr_build_temp_release(st);   // the stream is drained: the next build may reuse the buffer
    if (rs != TR_OK && status == TR_OK) status = rs;
    if (status != TR_OK) {   // never leave a half-built hierarchy reachable (keep the error message)
        const std::string msg = tr_last_error();
        tr_bvh_reset(bvh);
        tr_set_error(msg);
    }
    return status;
}


// Refit: keep the hierarchy (Morton order, Karras topology), recompute every box from new
// vertex positions.  No host synchronisation except the final one that orders the free of
// the temporaries: the number of refit rounds is the known tree height.
int tr_refit_impl(tr_bvh* bvh, const float* d_vertices, int64_t nv, const int32_t* d_faces,
                  int64_t nf, hipStream_t stream) {
    if (nf != bvh->num_tris) return tr_fail(TR_ERR_INVALID_ARG, "refit needs the same number of faces as the build");
    bvh->wide_valid = false; bvh->wide_unavailable = false;
    if (nf == 0) return TR_OK;
    if (!d_vertices || !d_faces) return tr_fail(TR_ERR_INVALID_ARG, "null mesh pointer");
    if (nv < 0) return tr_fail(TR_ERR_INVALID_ARG, "negative mesh size");
    const int64_t ni = bvh->num_nodes;
    Carver tc{nullptr};
    tc.take<float>(6 * (size_t)nf); tc.take<float>(6 * (size_t)(ni > 0 ? ni : 1)); tc.take<int32_t>((size_t)(ni > 0 ? ni : 1));
    tc.take<uint32_t>(4 + 8);
    const size_t need = align_up(tc.off, 256);
    if (bvh->refit_temp_bytes < need) {
        if (bvh->refit_temp) { TR_HIP_TRY(hipFree(bvh->refit_temp)); bvh->refit_temp = nullptr; bvh->refit_temp_bytes = 0; }
        TR_HIP_TRY(hipMalloc(&bvh->refit_temp, need));
        bvh->refit_temp_bytes = need;
    }
    Carver c2{(char*)bvh->refit_temp};
    float* sbox = c2.take<float>(6 * (size_t)nf);
    float* ibox = c2.take<float>(6 * (size_t)(ni > 0 ? ni : 1));
    int32_t* ready = c2.take<int32_t>((size_t)(ni > 0 ? ni : 1));
    uint32_t* bad = c2.take<uint32_t>(4 + 8);   // smallest face id with an out-of-range vertex index | frame
    tr_qframe* d_frame = reinterpret_cast<tr_qframe*>(bad + 4);
    uint32_t hbad = 0xffffffffu;
    int status = TR_OK;
    auto check = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && status == TR_OK)
            status = tr_fail(TR_ERR_HIP, std::string(what) + ": " + hipGetErrorName(e));
    };
    const int TB = 256;
    check(hipMemsetAsync(bad, 0xff, sizeof(uint32_t), stream), "memset bad-face word");
    hipLaunchKernelGGL(k_regather, dim3((unsigned)cdiv(nf, TB)), dim3(TB), 0, stream, d_vertices, nv, d_faces, nf, bvh->tris, sbox, bad);
    check(hipGetLastError(), "k_regather");
    check(hipMemcpyAsync(&hbad, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "memcpy bad-face word");
    if (ni > 0) {
        const unsigned gI = (unsigned)cdiv(ni, TB);
        check(hipMemsetAsync(ready, 0, sizeof(int32_t) * (size_t)ni, stream), "memset ready");
        for (int32_t round = 1; round <= bvh->depth; round++)
            hipLaunchKernelGGL(k_refit_nodes_round, dim3(gI), dim3(TB), 0, stream, bvh->nodes, sbox, ibox, ready, ni, round);
        hipLaunchKernelGGL(k_qframe_box, dim3(1), dim3(64), 0, stream, ibox, d_frame);   // the new root box
        hipLaunchKernelGGL(k_update_boxes, dim3(gI), dim3(TB), 0, stream, bvh->nodes, sbox, ibox, ni, d_frame, bvh->qnodes);
        check(hipGetLastError(), "refit rounds");
        float rootbox[6];
        check(hipMemcpyAsync(rootbox, ibox, sizeof(rootbox), hipMemcpyDeviceToHost, stream), "memcpy root box");
        check(hipStreamSynchronize(stream), "sync refit");
        if (status == TR_OK) {
            for (int k = 0; k < 3; k++) { bvh->aabb_min[k] = rootbox[k]; bvh->aabb_max[k] = rootbox[3 + k]; }
            tr_qframe_make(bvh->aabb_min, bvh->aabb_max, &bvh->frame);
            if (tr_bvh_sync_frame(bvh) != TR_OK) status = TR_ERR_HIP;
        }
    } else {
        float box[6];
        check(hipMemcpyAsync(box, sbox, sizeof(box), hipMemcpyDeviceToHost, stream), "memcpy box");
        check(hipStreamSynchronize(stream), "sync refit");
        if (status == TR_OK) {
            for (int k = 0; k < 3; k++) { bvh->aabb_min[k] = box[k]; bvh->aabb_max[k] = box[3 + k]; }
            tr_qframe_make(bvh->aabb_min, bvh->aabb_max, &bvh->frame);      // (the box rays are anchored to: also for one triangle)
            if (tr_bvh_sync_frame(bvh) != TR_OK) status = TR_ERR_HIP;
        }
    }
    if (status == TR_OK && hbad != 0xffffffffu)
        status = tr_fail(TR_ERR_INVALID_ARG, "face " + std::to_string(hbad) + " has a vertex index outside [0, " +
                                                 std::to_string(nv) + ")");
    if (status != TR_OK) {   // the triangle records are partly rewritten: drop the hierarchy
        const std::string msg = tr_last_error();
        tr_bvh_reset(bvh);
        tr_set_error(msg);
    }
    return status;
}
