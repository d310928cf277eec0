/*
 * triro_oracle.c -- CPU ORACLE for the ray/triangle-mesh intersection path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * smoke() entry point and bench.py's cpu_baseline leg may load it.  The
 * shipped library (libtriro_hip.so) never links, loads or calls it.
 *
 * PARITY STATUS: *parity unpinned* against the real reference at the bit level; pinned at 8-bit IMAGE
 * precision on the one output artefact the reference publishes (assets/location.png, its OptiX path's
 * location map of the README quick-start: tests/golden/reference_readme_location_axes.npz,
 * tests/test_oracle.py::test_oracle_reproduces_the_reference_s_published_readme_image -- silhouette to
 * a fraction of a pixel, location values to one 8-bit level on average).  The reference
 * (lcp29/trimesh-ray-optix, "triro" 1.3.1) performs BVH build, traversal and
 * the ray/triangle test inside NVIDIA OptiX (closed source, pinned only as
 * ">= 7.7", README.md:8) on RTX hardware; it cannot be compiled, imported or
 * run here, and its own tests (test/test.py) hold no assertions or golden
 * vectors.  What this file restates is everything the repository itself
 * defines around optixTrace:
 *
 *   - strided ray fetch          triro/backend/shaders.cu:27-63  (getRay/getIndices)
 *   - ray interval tmin=0,tmax=1e7, no culling   shaders.cu:86,112,163,191,238
 *   - intersects_any   -> bool                    shaders.cu:67-89
 *   - intersects_first -> tri index or -1         shaders.cu:93-116
 *   - intersects_closest -> hit, front, tri, loc, uv; miss values
 *                                                  shaders.cu:120-172
 *       loc = u*V1 + v*V2 + (1-u-v)*V0            shaders.cu:143-146
 *       uv  = (1-u-v, u)                          shaders.cu:149
 *   - intersects_count -> every triangle once      shaders.cu:176-194, ray.cpp:60-62
 *   - intersects_location: clamp counts to 8, exclusive scan, fill
 *                                                  ray.cpp:324-378, shaders.cu:196-246,
 *                                                  LaunchParams.h:8
 *
 * and anchors it on the hand-derived known answers for the inputs of
 * test/test.py and README.md (tests/golden/known_answers.json).
 *
 * The ray/triangle arithmetic itself (closed source in the reference; OptiX documents its built-in triangle test as
 * WATERTIGHT) is fixed here as an ARITHMETIC CONTRACT (version 3, round 6) that the HIP kernels follow operation by
 * operation so that every output is bit-exact.  The contract (also in DESIGN.md, "Arithmetic contract"):
 *
 *   ray:   valid iff all six of o,d are finite (else the ray misses everything)
 *   MT:    e1=b-a, e2=c-a, p=cross(d,e2), det=dot(e1,p), s=o-a, U=dot(s,p), q=cross(s,e1), V=dot(d,q), T=dot(e2,q)
 *          cross(x,y).x = fma(x.y, y.z, -(x.z*y.y)) (cyclic);  dot(x,y) = fma(x.z, y.z, fma(x.y, y.y, x.x*y.x))
 *          (Uf,Vf) = (U,V) with the sign bit of det flipped in; Wf = (|det| - Uf) - Vf; m3 = min(min(Uf,Vf),Wf)
 *   bound: E = (|e1x|+|e1y|+|e1z|) + (|e2x|+|e2y|+|e2z|), Ls = max|s_i|, kd = (|dx|+|dy|+|dz|) * 10*2^-24,
 *          mm = fma(kd*E, Ls+E, 2^-100)  -- exceeds the rounding error of Uf, Vf, Wf and det (proof: tr_math.h);
 *          Ls+E > 2^40 or |d|_1 > 2^40 (kd = inf) or NaN -> the float32 part does not answer (overflow)
 *   inside: m3 < -mm -> outside (proven);  m3 > mm -> inside (proven), t = T/det in float32 provided
 *          |det| >= (kd*E*E)*1024 and |T| >= (Ls*2^-10)*(E*E) (relative error of t < 2^-11);
 *          anything else (also NaN) -> the EXACT part: Woop / Benthin / Wald 2013 edge functions in float64 from the
 *          float32 inputs, projective form (woop64 below): inside iff U,V,W all >= 0 or all <= 0 and U+V+W != 0 -- where one of
 *          them is exactly ZERO (the ray passes through an edge or a vertex) only for the triangle that owns that edge (tie_own);
 *          t = (float)(fma(W,Cz, fma(V,Bz, U*Az)) / ((U+V+W) * d[kz]))
 *   accepted iff 0 <= t <= 1e7;  closest = lexicographic minimum of (t, tri_idx) over accepted triangles
 *   outputs of the winning triangle from the float64 edge functions: w0 = (float)(U/det64), w1 = (float)(V/det64),
 *          w2=(1-w0)-w1, loc_i = fma(w0,a_i, fma(w2,c_i, w1*b_i)), uv=(w0,w1); front = (det64 < 0) == (d[kz] < 0)
 *   multi-hit: the (up to) 8 accepted hits with smallest (t, tri_idx), ascending; the reference keeps "the first 8
 *          in traversal order", which is unspecified (shaders.cu:209-212).
 *   boxes (any BVH): pad(x) = |x|*2^-21 + 2^-100 per bound; slab t1 = (lo_i-o_i)*inv_i, t2 = (hi_i-o_i)*inv_i,
 *          inv_i = 1/d_i clamped to +-3e38; tn = max_i min(t1,t2); tf = (min_i max(t1,t2)) * (1+2^-21);
 *          a box is entered iff tn <= tf && tf >= 0 && tn <= limit, limit = best_t * (1+2^-10) for closest hits,
 *          1.001e7 otherwise.
 *
 * The predicate is a pure function of (ray, triangle).  Any conservative BVH returns what the brute-force loop
 * returns because (a) the slab test accepts every box the ray's line truly meets within [0, limit] (the exit pad
 * 1 + 8u covers the three roundings of an entry and of an exit distance), and a triangle that is hit lies in its
 * padded box, and (b) culling leaves 2^-10 of slack where a float32 distance can be off by 2^-11.  Both are here:
 * mode 0 = brute force (ground truth), mode 1 = median-split BVH (fast; used for the CPU baseline and for parity at
 * sizes brute force cannot reach).
 *
 * Build:  see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TR_TMAX 1.0e7f
#define TR_TANCHOR 9.99999e6f /* anchoring only when the whole box is nearer than this: every triangle then lies within tmax of the origin as well */
#define TR_HUGE 3.0e38f
#define TR_SLAB_PAD 1.000000476837158203125f /* 1 + 2^-21 */
#define TR_CULL_SLACK 1.0009765625f         /* 1 + 2^-10 */
#define TR_TLIM 1.001e7f                    /* > TR_TMAX * TR_CULL_SLACK */
#define TR_MAX_HITS_CAP 64
#define TR_PAD_REL 4.76837158203125e-07f  /* 2^-21  */
#define TR_PAD_ABS 7.888609052210118e-31f /* 2^-100 */
#define TR_BAND_K 5.9604644775390625e-07f  /* 10 u, u = 2^-24 */
#define TR_BAND_ABS 7.888609052210118e-31f /* 2^-100 */
#define TR_BAND_MAXLEN 1.099511627776e12f  /* 2^40 */

typedef struct {
    float o[3], d[3], inv[3];
    float kd; /* 2^-20 * |d|_1: the ray's factor of the band margin (contract v2) */
    int valid;
} ray_t;

typedef struct {
    float lo[3], hi[3];
    int32_t left;  /* internal: index of left child; leaf: first prim slot */
    int32_t right; /* internal: index of right child; leaf: -(count)       */
} onode_t;

typedef struct {
    int64_t nv, nf;
    float *verts;   /* nv*3 */
    int32_t *faces; /* nf*3 */
    /* median-split BVH */
    onode_t *nodes;
    int32_t nnodes;
    int32_t *prim; /* permutation of triangle ids */
    /* the box rays are anchored to (contract: ray anchoring): the product's 16-bit grid frame over the padded bounds of
     * the mesh -- base = the minimum, top = fma(65535, scale, base), scale a power of two (tr_qframe_make, tr_bvh.h) */
    float alo[3], ahi[3];
} omesh_t;

static inline float minf_(float a, float b) { return a < b ? a : b; }
static inline float maxf_(float a, float b) { return a > b ? a : b; }

static inline float dot3(const float *x, const float *y) {
    return fmaf(x[2], y[2], fmaf(x[1], y[1], x[0] * y[0]));
}
static inline void cross3(const float *x, const float *y, float *r) {
    r[0] = fmaf(x[1], y[2], -(x[2] * y[1]));
    r[1] = fmaf(x[2], y[0], -(x[0] * y[2]));
    r[2] = fmaf(x[0], y[1], -(x[1] * y[0]));
}

/* ray anchoring (contract 3; csrc/tr_math.h tr_ray_anchor): a ray that starts far outside the mesh's box is moved along
 * itself to just in front of its entry point.  tn, tf = entry / exit of the slab test (reciprocals clamped to +-3e38, no
 * padding), chord = tf - tn; anchored iff 0 < tn <= tf, tf < 1e7 (1 - 1e-6), tn > chord / 2; t0 = tn - max(chord / 16, tn * 2^-18);
 * o' = o + t0 * d with a compensated product (fma) and sum (TwoSum): within an ulp of itself of the exact point. */
static void anchor_ray(const omesh_t *m, const float *o, const float *d, float *oa) {
    float tn = -INFINITY, tf = INFINITY;
    for (int i = 0; i < 3; i++) {
        float inv = 1.0f / d[i];
        if (fabsf(inv) > TR_HUGE) inv = copysignf(TR_HUGE, d[i]);
        const float t1 = (m->alo[i] - o[i]) * inv, t2 = (m->ahi[i] - o[i]) * inv;
        /* (fmaxf / fminf as the product: a NaN operand is ignored) */
        tn = fmaxf(tn, fminf(t1, t2));
        tf = fminf(tf, fmaxf(t1, t2));
    }
    const float chord = tf - tn;
    for (int i = 0; i < 3; i++) oa[i] = o[i];
    if (!(tn > 0.0f && tn <= tf && tf < TR_TANCHOR && tn > 0.5f * chord)) return;
    const float t0 = tn - fmaxf(chord * 0.0625f, tn * 3.814697265625e-06f);
    for (int i = 0; i < 3; i++) {
        const float p = t0 * d[i], e = fmaf(t0, d[i], -p);
        const float s = p + o[i], bb = s - p;
        const float err = (p - (s - bb)) + (o[i] - bb);
        oa[i] = s + (e + err);
    }
}

static void ray_setup(const omesh_t *m, ray_t *r, const float *o_in, const float *d) {
    int ok = 1;
    float o[3];
    anchor_ray(m, o_in, d, o);
    for (int i = 0; i < 3; i++) {
        r->o[i] = o[i];
        r->d[i] = d[i];
        if (!isfinite(o[i]) || !isfinite(d[i])) ok = 0;
        float inv = 1.0f / d[i];
        if (fabsf(inv) > TR_HUGE) inv = copysignf(TR_HUGE, d[i]);
        r->inv[i] = inv;
    }
    {
        const float l1 = (fabsf(d[0]) + fabsf(d[1])) + fabsf(d[2]);
        r->kd = l1 <= TR_BAND_MAXLEN ? l1 * TR_BAND_K : INFINITY; /* too long: the float32 part does not answer */
    }
    r->valid = ok;
}

/* robust slab test (Ize 2013); returns 1 if the (padded) interval is non-empty and overlaps [0, limit] */
static inline int slab(const ray_t *r, const float *lo, const float *hi, float limit, float *tn_out,
                       float *tf_out) {
    float tn = -INFINITY, tf = INFINITY;
    for (int i = 0; i < 3; i++) {
        float t1 = (lo[i] - r->o[i]) * r->inv[i];
        float t2 = (hi[i] - r->o[i]) * r->inv[i];
        tn = maxf_(tn, minf_(t1, t2));
        tf = minf_(tf, maxf_(t1, t2));
    }
    tf = tf * TR_SLAB_PAD;
    *tn_out = tn;
    *tf_out = tf;
    return (tn <= tf) && (tf >= 0.0f) && (tn <= limit);
}

typedef struct {
    float t; /* the key: distance clamped into the triangle's own slab interval */
} hit_t;

/* box of a triangle, padded outward by |x|*2^-21 + 2^-100 per bound (see contract) */
static inline float pad_(float x) { return fabsf(x) * TR_PAD_REL + TR_PAD_ABS; }
static inline void tri_box_padded(const float *a, const float *b, const float *c, float *lo,
                                  float *hi) {
    for (int i = 0; i < 3; i++) {
        float l = minf_(minf_(a[i], b[i]), c[i]);
        float h = maxf_(maxf_(a[i], b[i]), c[i]);
        lo[i] = l - pad_(l);
        hi[i] = h + pad_(h);
    }
}

/* ---- contract v2 (round 6): Moller-Trumbore decides where it provably can, float64 edge functions elsewhere ---- */
/* The EXACT part: Woop / Benthin / Wald 2013 edge functions in float64 from the float32 inputs, in projective form
 * (multiplied through by d[kz]: no division).  kz = the dominant axis of the direction, (kx, ky, kz) cyclic.  A vertex P
 * becomes P' = P - o (float64), Ph = (P'x*dz - dx*P'z, P'y*dz - dy*P'z): a function of (vertex, ray) alone, so the two
 * triangles of a shared edge see the same two points; an edge function is the DIFFERENCE OF TWO ROUNDED PRODUCTS -- its
 * sign is the exact sign of the 2D orientation of the rounded points or zero (rounding is monotone), and swapping the
 * end points negates it exactly.  A ray therefore cannot pass between two triangles that share an edge. */
typedef struct {
    double U, V, W;    /* weights of a, b, c (unnormalised; all >= 0 or all <= 0 inside) */
    double Az, Bz, Cz; /* P'z of the three vertices */
    double dz;
} woop_t;

static inline void woop64(const float *o, const float *d, const float *a, const float *b, const float *c,
                          woop_t *w) {
    int kz = 0;
    if (fabsf(d[1]) > fabsf(d[kz])) kz = 1;
    if (fabsf(d[2]) > fabsf(d[kz])) kz = 2;
    const int kx = (kz + 1) % 3, ky = (kz + 2) % 3;
    const double dx = d[kx], dy = d[ky], dz = d[kz];
    const double ox = o[kx], oy = o[ky], oz = o[kz];
    const double Ax = (double)a[kx] - ox, Ay = (double)a[ky] - oy, Az = (double)a[kz] - oz;
    const double Bx = (double)b[kx] - ox, By = (double)b[ky] - oy, Bz = (double)b[kz] - oz;
    const double Cx = (double)c[kx] - ox, Cy = (double)c[ky] - oy, Cz = (double)c[kz] - oz;
    const double ahx = fma(-dx, Az, Ax * dz), ahy = fma(-dy, Az, Ay * dz);
    const double bhx = fma(-dx, Bz, Bx * dz), bhy = fma(-dy, Bz, By * dz);
    const double chx = fma(-dx, Cz, Cx * dz), chy = fma(-dy, Cz, Cy * dz);
    w->U = chx * bhy - chy * bhx;
    w->V = ahx * chy - ahy * chx;
    w->W = bhx * ahy - bhy * ahx;
    w->Az = Az; w->Bz = Bz; w->Cz = Cz;
    w->dz = dz;
}

/* Exact ties (csrc/tr_math.h "exact ties"): an edge whose function is exactly zero belongs to ONE of its two triangles, as
 * in a rasteriser's fill rule.  The directed edge p -> q seen along the ray (permuted components, float64; q - p is exact):
 * e = ((qx - px) dz - dx (qz - pz), (qy - py) dz - dy (qz - pz)), exactly negated when p and q swap; with the orientation
 * normalised (s = sign(U + V + W)) the owner is the triangle whose s e has ey > 0, or ey = 0 and ex > 0. */
static inline int edge_own(double s, const float *d, int kx, int ky, int kz, const float *p, const float *q) {
    const double ux = (double)q[kx] - (double)p[kx], uy = (double)q[ky] - (double)p[ky], uz = (double)q[kz] - (double)p[kz];
    const double ex = s * fma(-(double)d[kx], uz, ux * (double)d[kz]), ey = s * fma(-(double)d[ky], uz, uy * (double)d[kz]);
    return ey > 0.0 || (ey == 0.0 && ex > 0.0);
}
/* does the triangle own every edge whose function is zero (U: b -> c, V: c -> a, W: a -> b)? */
static inline int tie_own(const woop_t *w, double det, const float *d, const float *a, const float *b, const float *c) {
    int kz = 0;
    if (fabsf(d[1]) > fabsf(d[kz])) kz = 1;
    if (fabsf(d[2]) > fabsf(d[kz])) kz = 2;
    const int kx = (kz + 1) % 3, ky = (kz + 2) % 3;
    const double s = det < 0.0 ? -1.0 : 1.0;
    if (w->U == 0.0 && !edge_own(s, d, kx, ky, kz, b, c)) return 0;
    if (w->V == 0.0 && !edge_own(s, d, kx, ky, kz, c, a)) return 0;
    if (w->W == 0.0 && !edge_own(s, d, kx, ky, kz, a, b)) return 0;
    return 1;
}

/* inside test of the exact part: both windings, no strict sign conflict (zero edge functions: tie_own, at the call sites) */
static inline int woop_inside(const woop_t *w, double *det_out) {
    const double U = w->U, V = w->V, W = w->W;
    if ((U < 0.0 || V < 0.0 || W < 0.0) && (U > 0.0 || V > 0.0 || W > 0.0)) return 0;
    const double det = (U + V) + W;
    *det_out = det;
    return det != 0.0;
}


/* diagnostics (-DORACLE_STATS build, oracle_band_stats): leaf tests that reach the inside decision / that take the float64 part */
static long long g_leaf_tests, g_band_tests;

/* the hit predicate: a pure function of (ray, triangle) -- no box, no clamp */
static inline int tri_hit(const ray_t *r, const float *a, const float *b, const float *c,
                          hit_t *h) {
    float e1[3], e2[3], s[3], p[3], q[3];
    for (int i = 0; i < 3; i++) {
        e1[i] = b[i] - a[i];
        e2[i] = c[i] - a[i];
        s[i] = r->o[i] - a[i];
    }
    cross3(r->d, e2, p);
    float det = dot3(e1, p);
    float U = dot3(s, p);
    /* both orientations at once: flip by the sign BIT of det */
    const int neg = signbit(det) != 0;
    const float Uf = neg ? -U : U;
    /* proven bound on |computed - exact| of Uf, Vf, D1, Wf (+ that of det): 10 u |d|_1 E (|s|_inf + E) + 2^-100,
     * E = |e1|_1 + |e2|_1 */
    const float E = ((fabsf(e1[0]) + fabsf(e1[1])) + fabsf(e1[2])) + ((fabsf(e2[0]) + fabsf(e2[1])) + fabsf(e2[2]));
    const float Ls = maxf_(maxf_(fabsf(s[0]), fabsf(s[1])), fabsf(s[2]));
    const float kE = r->kd * E;
    const float LsE = Ls + E;
    const float mm = fmaf(kE, LsE, TR_BAND_ABS);
#ifdef ORACLE_STATS
    __atomic_fetch_add(&g_leaf_tests, 1, __ATOMIC_RELAXED);
#endif
    /* lengths beyond 2^40 (overflow) and NaN: the float32 part does not answer */
    int exact = !(LsE <= TR_BAND_MAXLEN);
    if (!exact) {
        cross3(s, e1, q);
        const float V = dot3(r->d, q);
        const float Vf = neg ? -V : V;
        const float Wf = (fabsf(det) - Uf) - Vf;
        const float m3 = minf_(minf_(Uf, Vf), Wf);
        if (m3 < -mm) return 0;  /* proven outside */
        exact = !(m3 > mm);      /* not proven inside */
    }
    float t = 0.0f;
    if (!exact) {
        const float T = dot3(e2, q);
        /* the float32 distance only where |T| and |det| exceed 2^12 x their error bounds: relative error < 2^-11 */
        exact = !(fabsf(det) >= (kE * E) * 1024.0f) || !(fabsf(T) >= (Ls * 9.765625e-4f) * (E * E));
        if (!exact) t = T / det;
    }
    if (exact) {
#ifdef ORACLE_STATS
        __atomic_fetch_add(&g_band_tests, 1, __ATOMIC_RELAXED);
#endif
        woop_t w;
        double det64;
        woop64(r->o, r->d, a, b, c, &w);
        if (!woop_inside(&w, &det64)) return 0;
        if ((w.U == 0.0 || w.V == 0.0 || w.W == 0.0) && !tie_own(&w, det64, r->d, a, b, c)) return 0;      /* an exact tie: not this triangle's edge */
        t = (float)(fma(w.W, w.Cz, fma(w.V, w.Bz, w.U * w.Az)) / (det64 * w.dz));
    }
    if (!(t >= 0.0f && t <= TR_TMAX)) return 0;
    h->t = t;
    return 1;
}

static inline void tri_verts(const omesh_t *m, int32_t f, const float **a, const float **b,
                             const float **c) {
    const int32_t *idx = m->faces + 3 * (int64_t)f;
    *a = m->verts + 3 * (int64_t)idx[0];
    *b = m->verts + 3 * (int64_t)idx[1];
    *c = m->verts + 3 * (int64_t)idx[2];
}

/* ------------------------------------------------------------------ */
/* median-split BVH (deliberately NOT the LBVH the GPU library builds) */
/* ------------------------------------------------------------------ */
typedef struct {
    float c[3];
    int32_t id;
} cent_t;

static int g_axis;
static int cmp_cent(const void *x, const void *y) {
    float a = ((const cent_t *)x)->c[g_axis], b = ((const cent_t *)y)->c[g_axis];
    if (a < b) return -1;
    if (a > b) return 1;
    int32_t ia = ((const cent_t *)x)->id, ib = ((const cent_t *)y)->id;
    return (ia > ib) - (ia < ib);
}

static void tri_box(const omesh_t *m, int32_t f, float *lo, float *hi) {
    const float *a, *b, *c;
    tri_verts(m, f, &a, &b, &c);
    tri_box_padded(a, b, c, lo, hi);
}

static int32_t build_rec(omesh_t *m, cent_t *cents, int32_t first, int32_t count) {
    int32_t me = m->nnodes++;
    onode_t *n = &m->nodes[me];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < 3; i++) {
        n->lo[i] = INFINITY;
        n->hi[i] = -INFINITY;
    }
    for (int32_t k = first; k < first + count; k++) {
        float lo[3], hi[3];
        tri_box(m, cents[k].id, lo, hi);
        for (int i = 0; i < 3; i++) {
            n->lo[i] = minf_(n->lo[i], lo[i]);
            n->hi[i] = maxf_(n->hi[i], hi[i]);
            clo[i] = minf_(clo[i], cents[k].c[i]);
            chi[i] = maxf_(chi[i], cents[k].c[i]);
        }
    }
    if (count <= 4) {
        n->left = first;
        n->right = -count;
        return me;
    }
    int axis = 0;
    float ext = chi[0] - clo[0];
    for (int i = 1; i < 3; i++)
        if (chi[i] - clo[i] > ext) {
            ext = chi[i] - clo[i];
            axis = i;
        }
    g_axis = axis;
    qsort(cents + first, (size_t)count, sizeof(cent_t), cmp_cent);
    int32_t half = count / 2;
    int32_t l = build_rec(m, cents, first, half);
    int32_t rr = build_rec(m, cents, first + half, count - half);
    /* m->nodes is preallocated, so n is still valid */
    m->nodes[me].left = l;
    m->nodes[me].right = rr;
    return me;
}

void *oracle_mesh_create(const float *verts, int64_t nv, const int32_t *faces, int64_t nf) {
    omesh_t *m = (omesh_t *)calloc(1, sizeof(omesh_t));
    m->nv = nv;
    m->nf = nf;
    m->verts = (float *)malloc(sizeof(float) * 3 * (size_t)(nv > 0 ? nv : 1));
    m->faces = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(nf > 0 ? nf : 1));
    memcpy(m->verts, verts, sizeof(float) * 3 * (size_t)nv);
    memcpy(m->faces, faces, sizeof(int32_t) * 3 * (size_t)nf);
    if (nf > 0) {
        m->nodes = (onode_t *)malloc(sizeof(onode_t) * (size_t)(2 * nf));
        m->prim = (int32_t *)malloc(sizeof(int32_t) * (size_t)nf);
        cent_t *cents = (cent_t *)malloc(sizeof(cent_t) * (size_t)nf);
        for (int64_t f = 0; f < nf; f++) {
            float lo[3], hi[3];
            tri_box(m, (int32_t)f, lo, hi);
            for (int i = 0; i < 3; i++) cents[f].c[i] = 0.5f * lo[i] + 0.5f * hi[i];
            cents[f].id = (int32_t)f;
        }
        build_rec(m, cents, 0, (int32_t)nf);
        for (int64_t f = 0; f < nf; f++) m->prim[f] = cents[f].id;
        free(cents);
    }
    /* the grid frame over the padded bounds (tr_qframe_make restated): per axis scale = the power of two with
     * fma(65535, scale, base) >= max, found from frexp(ext / 65535) and doubled while the last plane falls short */
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t f = 0; f < nf; f++) {
        float lo[3], hi[3];
        tri_box(m, (int32_t)f, lo, hi);
        for (int i = 0; i < 3; i++) {
            mn[i] = minf_(mn[i], lo[i]);
            mx[i] = maxf_(mx[i], hi[i]);
        }
    }
    for (int i = 0; i < 3; i++) {
        float sc = 1.0f, base = nf > 0 ? mn[i] : 0.0f;
        const float ext = nf > 0 ? mx[i] - mn[i] : 0.0f;
        if (ext > 0.0f && ext <= 3.0e38f) {
            int e = 0;
            (void)frexpf(ext / 65535.0f, &e);
            sc = ldexpf(1.0f, e < -126 ? -126 : e);
        }
        for (int it = 0; it < 300 && nf > 0 && !(fmaf(65535.0f, sc, base) >= mx[i]); it++) sc *= 2.0f;
        m->alo[i] = base;
        m->ahi[i] = fmaf(65535.0f, sc, base);
    }
    return m;
}

void oracle_mesh_destroy(void *h) {
    omesh_t *m = (omesh_t *)h;
    if (!m) return;
    free(m->verts);
    free(m->faces);
    free(m->nodes);
    free(m->prim);
    free(m);
}

/* ------------------------------------------------------------------ */
/* per-ray visitors                                                     */
/* ------------------------------------------------------------------ */
typedef struct {
    int found;
    float t;
    int32_t tri;
} best_t;

static inline void consider(best_t *b, const hit_t *h, int32_t f) {
    if (!b->found || h->t < b->t || (h->t == b->t && f < b->tri)) {
        b->found = 1;
        b->t = h->t;
        b->tri = f;
    }
}

typedef struct {
    int32_t n;   /* entries kept (<= cap)   */
    int32_t cap; /* <= TR_MAX_HITS_CAP       */
    int64_t total;
    float t[TR_MAX_HITS_CAP];
    int32_t tri[TR_MAX_HITS_CAP];
} hitlist_t;

static inline void hl_insert(hitlist_t *l, const hit_t *h, int32_t f) {
    l->total++;
    if (l->cap == 0) return;
    int32_t pos = l->n;
    if (l->n == l->cap) {
        int32_t last = l->n - 1;
        if (!(h->t < l->t[last] || (h->t == l->t[last] && f < l->tri[last]))) return;
        pos = last;
    } else {
        l->n++;
    }
    while (pos > 0 &&
           (h->t < l->t[pos - 1] || (h->t == l->t[pos - 1] && f < l->tri[pos - 1]))) {
        l->t[pos] = l->t[pos - 1];
        l->tri[pos] = l->tri[pos - 1];
        pos--;
    }
    l->t[pos] = h->t;
    l->tri[pos] = f;
}

/* closest hit */
static void closest_ray(const omesh_t *m, const ray_t *r, int mode, best_t *best) {
    best->found = 0;
    best->t = INFINITY;
    best->tri = -1;
    if (!r->valid || m->nf == 0) return;
    hit_t h;
    const float *a, *b, *c;
    if (mode == 0) {
        for (int64_t f = 0; f < m->nf; f++) {
            tri_verts(m, (int32_t)f, &a, &b, &c);
            if (tri_hit(r, a, b, c, &h)) consider(best, &h, (int32_t)f);
        }
        return;
    }
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const onode_t *n = &m->nodes[stack[--sp]];
        float tn, tf;
        /* culled only beyond best * (1 + 2^-10): a float32 distance can be 2^-11 off (contract, TR_CULL_SLACK) */
        if (!slab(r, n->lo, n->hi, best->found ? best->t * TR_CULL_SLACK : TR_TLIM, &tn, &tf)) continue;
        if (n->right < 0) {
            for (int32_t k = 0; k < -n->right; k++) {
                int32_t f = m->prim[n->left + k];
                tri_verts(m, f, &a, &b, &c);
                if (tri_hit(r, a, b, c, &h)) consider(best, &h, f);
            }
        } else {
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
}

/* all hits: count (uncapped) and the cap nearest */
static void allhits_ray(const omesh_t *m, const ray_t *r, int mode, hitlist_t *l) {
    l->n = 0;
    l->total = 0;
    if (!r->valid || m->nf == 0) return;
    hit_t h;
    const float *a, *b, *c;
    if (mode == 0) {
        for (int64_t f = 0; f < m->nf; f++) {
            tri_verts(m, (int32_t)f, &a, &b, &c);
            if (tri_hit(r, a, b, c, &h)) hl_insert(l, &h, (int32_t)f);
        }
        return;
    }
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const onode_t *n = &m->nodes[stack[--sp]];
        float tn, tf;
        if (!slab(r, n->lo, n->hi, TR_TLIM, &tn, &tf)) continue;
        if (n->right < 0) {
            for (int32_t k = 0; k < -n->right; k++) {
                int32_t f = m->prim[n->left + k];
                tri_verts(m, f, &a, &b, &c);
                if (tri_hit(r, a, b, c, &h)) hl_insert(l, &h, f);
            }
        } else {
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
}

/* outputs of a hit: the barycentrics of the WINNING triangle from the float64 edge functions of (ray, triangle) --
 * w0, w1 = the weights of face vertices 0 and 1 = the reference's uv (shaders.cu:149: (1-u-v, u)), each the float32
 * rounding of a float64 quotient --, then float32: w2 = (1 - w0) - w1, loc = fma(w0, a, fma(w2, c, w1 * b)) (:143-146).
 * front (optixIsFrontFaceHit, :151) = counter-clockwise seen from the origin = d . ((b-a) x (c-a)) < 0: the sign of the
 * edge functions' sum times that of d[kz]. */
static inline void hit_outputs(const omesh_t *m, const ray_t *r, int32_t f, float *loc, float *uv,
                               uint8_t *front) {
    const float *a, *b, *c;
    tri_verts(m, f, &a, &b, &c);
    woop_t wq;
    woop64(r->o, r->d, a, b, c, &wq);
    const double det = (wq.U + wq.V) + wq.W;
    const float w0 = (float)(wq.U / det), w1 = (float)(wq.V / det);
    const float w2 = (1.0f - w0) - w1;
    if (loc)
        for (int i = 0; i < 3; i++) loc[i] = fmaf(w0, a[i], fmaf(w2, c[i], w1 * b[i]));
    if (uv) {
        uv[0] = w0;
        uv[1] = w1;
    }
    if (front) *front = (det < 0.0) == (wq.dz < 0.0);
}

static void set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

/* intersects_closest: shaders.cu:120-172; outputs may be NULL.  t is an extra
 * diagnostic output (the reference does not return it). */
int oracle_closest(const void *mesh, const float *o, const float *d, int64_t n, int mode,
                   int nthreads, uint8_t *hit, uint8_t *front, int32_t *tri, float *loc,
                   float *uv, float *t) {
    const omesh_t *m = (const omesh_t *)mesh;
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        ray_t r;
        best_t b;
        ray_setup(m, &r, o + 3 * i, d + 3 * i);
        closest_ray(m, &r, mode, &b);
        float l3[3] = {0, 0, 0}, uv2[2] = {0, 0};
        uint8_t fr = 0;
        if (b.found) hit_outputs(m, &r, b.tri, l3, uv2, &fr);
        if (hit) hit[i] = (uint8_t)b.found;
        if (front) front[i] = fr;
        if (tri) tri[i] = b.found ? b.tri : -1;
        if (loc) memcpy(loc + 3 * i, l3, sizeof l3);
        if (uv) memcpy(uv + 2 * i, uv2, sizeof uv2);
        if (t) t[i] = b.found ? b.t : INFINITY;
    }
    return 0;
}

/* intersects_count (shaders.cu:176-194) and intersects_any (:67-89): counts are
 * uncapped; any = count > 0 (the reference's anyhit does not terminate the ray) */
int oracle_count(const void *mesh, const float *o, const float *d, int64_t n, int mode,
                 int nthreads, int32_t *count) {
    const omesh_t *m = (const omesh_t *)mesh;
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        ray_t r;
        hitlist_t l;
        l.cap = 0;
        ray_setup(m, &r, o + 3 * i, d + 3 * i);
        allhits_ray(m, &r, mode, &l);
        count[i] = (int32_t)l.total;
    }
    return 0;
}

/* intersects_location second pass (ray.cpp:344-378, shaders.cu:207-246): for ray
 * i write min(count,cap) hits at offsets[i]..; hits ordered by (t_key, tri). */
int oracle_location_fill(const void *mesh, const float *o, const float *d, int64_t n,
                         int mode, int nthreads, int32_t cap, const int64_t *offsets,
                         float *loc, int32_t *ray_idx, int32_t *tri_idx, float *t_out) {
    const omesh_t *m = (const omesh_t *)mesh;
    if (cap > TR_MAX_HITS_CAP) return -1;
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        ray_t r;
        hitlist_t l;
        l.cap = cap;
        ray_setup(m, &r, o + 3 * i, d + 3 * i);
        allhits_ray(m, &r, mode, &l);
        for (int32_t k = 0; k < l.n; k++) {
            int64_t g = offsets[i] + k;
            hit_outputs(m, &r, l.tri[k], loc + 3 * g, NULL, NULL);
            ray_idx[g] = (int32_t)i;
            tri_idx[g] = l.tri[k];
            if (t_out) t_out[g] = l.t[k];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* WATERTIGHT float64 reference (NOT the contract): the error bar on    */
/* "bit-exact vs the reference".  The reference's triangle test runs on */
/* RT cores (optixTrace, shaders.cu:86,163), documented as watertight;  */
/* the contract above is Moller-Trumbore in float32, which can lose a   */
/* ray that passes exactly through a shared edge.  This section answers */
/* "how many rays of a batch can that touch": Woop / Benthin / Wald     */
/* 2013 ("Watertight Ray/Triangle Intersection", JCGT 2(1)) evaluated   */
/* in float64 from the float32 inputs -- edge functions of a shared     */
/* edge are exact negations of each other for the two triangles, so a   */
/* ray through the edge hits at least one of them -- behind a double    */
/* precision slab test on slightly inflated boxes.  Nothing of the      */
/* contract (padded boxes, t_key clamp, fma placement) is used.         */
/* ------------------------------------------------------------------ */
typedef struct {
    double o[3], d[3];
    int kx, ky, kz;
    double Sx, Sy, Sz;
    int valid;
} wray_t;

static void wray_setup(wray_t *w, const float *o, const float *d) {
    w->valid = 1;
    for (int i = 0; i < 3; i++) {
        w->o[i] = o[i];
        w->d[i] = d[i];
        if (!isfinite(o[i]) || !isfinite(d[i])) w->valid = 0;
    }
    int kz = 0;
    if (fabs(w->d[1]) > fabs(w->d[kz])) kz = 1;
    if (fabs(w->d[2]) > fabs(w->d[kz])) kz = 2;
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    if (w->d[kz] < 0.0) { int t = kx; kx = ky; ky = t; }
    w->kx = kx; w->ky = ky; w->kz = kz;
    if (w->d[kz] == 0.0) { w->valid = 0; return; }   /* zero direction: hits nothing */
    w->Sx = w->d[kx] / w->d[kz];
    w->Sy = w->d[ky] / w->d[kz];
    w->Sz = 1.0 / w->d[kz];
}

static inline int wt_tri(const wray_t *w, const float *a, const float *b, const float *c, double *t_out, double *bary) {
    const int kx = w->kx, ky = w->ky, kz = w->kz;
    const double A[3] = {a[0] - w->o[0], a[1] - w->o[1], a[2] - w->o[2]};
    const double B[3] = {b[0] - w->o[0], b[1] - w->o[1], b[2] - w->o[2]};
    const double C[3] = {c[0] - w->o[0], c[1] - w->o[1], c[2] - w->o[2]};
    const double Ax = A[kx] - w->Sx * A[kz], Ay = A[ky] - w->Sy * A[kz];
    const double Bx = B[kx] - w->Sx * B[kz], By = B[ky] - w->Sy * B[kz];
    const double Cx = C[kx] - w->Sx * C[kz], Cy = C[ky] - w->Sy * C[kz];
    const double U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if ((U < 0.0 || V < 0.0 || W < 0.0) && (U > 0.0 || V > 0.0 || W > 0.0)) return 0;   /* no culling: both windings */
    const double det = U + V + W;
    if (det == 0.0) return 0;
    const double Az = w->Sz * A[kz], Bz = w->Sz * B[kz], Cz = w->Sz * C[kz];
    const double t = (U * Az + V * Bz + W * Cz) / det;
    if (!(t >= 0.0 && t <= (double)TR_TMAX)) return 0;
    *t_out = t;
    if (bary) { bary[0] = U / det; bary[1] = V / det; bary[2] = W / det; }   /* weights of a, b, c */
    return 1;
}

/* double-precision slab on the node's float box inflated by 1e-7 of its magnitude (+ 1e-30) */
static inline int wt_box(const wray_t *w, const float *lo, const float *hi, double tlimit) {
    double tn = 0.0, tf = tlimit;
    for (int i = 0; i < 3; i++) {
        const double e = 1e-7 * (fabs((double)lo[i]) + fabs((double)hi[i])) + 1e-30;
        const double l = (double)lo[i] - e, h = (double)hi[i] + e;
        if (w->d[i] == 0.0) {
            if (w->o[i] < l || w->o[i] > h) return 0;
            continue;
        }
        double t1 = (l - w->o[i]) / w->d[i], t2 = (h - w->o[i]) / w->d[i];
        if (t1 > t2) { double x = t1; t1 = t2; t2 = x; }
        if (t1 > tn) tn = t1;
        if (t2 < tf) tf = t2;
    }
    return tn <= tf * (1.0 + 1e-12) + 1e-300;
}

/* closest hit and hit count of every ray under the watertight float64 test; tri = -1 / t = inf on a miss.
 * uvw (optional, [n,3]): float64 barycentric weights of the hit on face vertices 0, 1, 2; loc (optional, [n,3]): the
 * float64 location w0 V0 + w1 V1 + w2 V2 -- the yardstick for the uv / loc the contract returns (0 on a miss). */
int oracle_watertight(const void *mesh, const float *o, const float *d, int64_t n, int nthreads,
                      int32_t *tri, double *t, int32_t *count, double *uvw, double *loc) {
    const omesh_t *m = (const omesh_t *)mesh;
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        wray_t w;
        wray_setup(&w, o + 3 * i, d + 3 * i);
        int32_t best = -1, cnt = 0;
        double bt = INFINITY, bb[3] = {0, 0, 0};
        if (w.valid && m->nf > 0) {
            int32_t stack[128];
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0) {
                const onode_t *nd = &m->nodes[stack[--sp]];
                if (!wt_box(&w, nd->lo, nd->hi, (double)TR_TMAX)) continue;
                if (nd->right < 0) {
                    for (int32_t k = 0; k < -nd->right; k++) {
                        const int32_t f = m->prim[nd->left + k];
                        const float *a, *b, *c;
                        tri_verts(m, f, &a, &b, &c);
                        double th, ba[3];
                        if (wt_tri(&w, a, b, c, &th, ba)) {
                            cnt++;
                            if (th < bt || (th == bt && f < best)) { bt = th; best = f; bb[0] = ba[0]; bb[1] = ba[1]; bb[2] = ba[2]; }
                        }
                    }
                } else {
                    stack[sp++] = nd->left;
                    stack[sp++] = nd->right;
                }
            }
        }
        if (tri) tri[i] = best;
        if (t) t[i] = bt;
        if (count) count[i] = cnt;
        if (uvw) { uvw[3 * i] = bb[0]; uvw[3 * i + 1] = bb[1]; uvw[3 * i + 2] = bb[2]; }
        if (loc) {
            double l3[3] = {0, 0, 0};
            if (best >= 0) {
                const float *a, *b, *c;
                tri_verts(m, best, &a, &b, &c);
                for (int k = 0; k < 3; k++) l3[k] = bb[0] * (double)a[k] + bb[1] * (double)b[k] + bb[2] * (double)c[k];
            }
            loc[3 * i] = l3[0]; loc[3 * i + 1] = l3[1]; loc[3 * i + 2] = l3[2];
        }
    }
    return 0;
}

/* Strided ray fetch: shaders.cu:27-63 (getIndices + getRay) with the launch-param
 * marshaling of ray.cpp:151-159,177-179: shape/strides right-aligned in 4 slots,
 * shape padded with INT64_MAX, strides padded with 0, strides in elements.
 * Unlike the reference (int arithmetic, shaders.cu:37) the index math is 64-bit. */
int oracle_fetch_rays(const float *obase, const float *dbase, const int64_t shape[4],
                      const int64_t ostride[4], const int64_t dstride[4], int64_t n,
                      float *o_out, float *d_out) {
    for (int64_t idx = 0; idx < n; idx++) {
        int64_t fi = idx * 3, ind[4];
        for (int i = 3; i >= 0; i--) {
            ind[i] = fi % shape[i];
            fi /= shape[i];
        }
        int64_t oi = 0, di = 0;
        for (int i = 0; i < 4; i++) {
            oi += ind[i] * ostride[i];
            di += ind[i] * dstride[i];
        }
        for (int k = 0; k < 3; k++) {
            o_out[3 * idx + k] = obase[oi + k * ostride[3]];
            d_out[3 * idx + k] = dbase[di + k * dstride[3]];
        }
    }
    return 0;
}

/* the anchored origins of a batch (what every query above traces): for comparisons with references that take plain rays */
int oracle_anchor_rays(const void *mesh, const float *o, const float *d, int64_t n, float *o_out) {
    const omesh_t *m = (const omesh_t *)mesh;
    for (int64_t i = 0; i < n; i++) anchor_ray(m, o + 3 * i, d + 3 * i, o_out + 3 * i);
    return 0;
}

/* out = {inside decisions, of which in float64}; zero unless built with -DORACLE_STATS (make stats) */
void oracle_band_stats(long long out[2], int reset) {
    out[0] = __atomic_load_n(&g_leaf_tests, __ATOMIC_RELAXED);
    out[1] = __atomic_load_n(&g_band_tests, __ATOMIC_RELAXED);
    if (reset) {
        __atomic_store_n(&g_leaf_tests, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&g_band_tests, 0, __ATOMIC_RELAXED);
    }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
