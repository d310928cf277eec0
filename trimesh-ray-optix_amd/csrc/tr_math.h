// tr_math.h -- the arithmetic contract of the ray/triangle path ("Arithmetic
// contract"; version 3, round 6).  Every operation below is fixed, in this order, with explicit fma
// where fused and -ffp-contract=off everywhere else, so that any IEEE-754 implementation (gfx950
// VALU, or the CPU oracle's independent restatement in oracle/triro_oracle.c) produces bit-identical
// hit masks, triangle indices, distances and barycentrics.
//
// What the reference leaves to OptiX (optixTrace, shaders.cu:86,112,163,191,238; built-in WATERTIGHT
// triangle test; optixGetTriangleBarycentrics :139; optixIsFrontFaceHit :151) is replaced by:
//   * inside test: Moller-Trumbore in float32 wherever its answer is PROVEN (a running bound on the
//     rounding error of U, V, det - U - V; the three must clear it), and otherwise -- the ray passes
//     within rounding of an edge, a vertex or the triangle's plane -- Woop / Benthin / Wald 2013 edge
//     functions in float64 (tr_woop64): exact negations for the two triangles of a shared edge, so no ray slips
//     between two triangles of a closed mesh; a function that is exactly ZERO (the ray runs through the edge) counts
//     for the ONE triangle that owns the edge (tr_tie_own: the fill rule of rasterisers), so none is hit twice;
//   * distance: T / det in float32 where its relative error is proven below 2^-11, float64 otherwise;
//   * outputs: the barycentrics of the WINNING triangle from the float64 edge functions.
// The predicate is a pure function of (ray, triangle) -- no box, no clamp (contract 2 clamped the distance
// into the triangle's own slab interval, 45 instructions per leaf test) -- and is independent of the BVH
// that finds the candidates because (a) every box test accepts a box that the ray truly meets within
// [0, limit] (robust slab test, TR_SLAB_PAD; conservative forms tr_bvh.h / tr_wide.h) and (b) boxes are
// culled against the best distance times TR_CULL_SLACK, which exceeds what a float32 distance can be off by.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TR_HD __host__ __device__ __forceinline__
#define TR_HDM __host__ __device__ __forceinline__
// a REAL call on the device: the float64 part of the inside test runs for ~1 % of the leaf tests and needs ~50 registers;
// inlined it raised every traversal kernel's allocation (77 -> 117 VGPRs: four waves per SIMD instead of six)
#define TR_HD_CALL static __host__ __device__ __attribute__((noinline))
#if defined(__HIP_DEVICE_COMPILE__)
// A convergent no-op: marks a point where all control-flow paths of a loop body must merge,
// so that jump threading cannot give the loop several back edges (LLVM then nests the loop
// and lanes on different paths serialise).  Emits no instruction.
#define TR_CONVERGE() __builtin_amdgcn_wave_barrier()
// true when the predicate holds in any active lane of the wave (wave-uniform branch)
#define TR_WAVE_ANY(x) (__ballot(x) != 0ull)
#else
#define TR_CONVERGE() ((void)0)
#define TR_WAVE_ANY(x) (x)
#endif
#else
#define TR_CONVERGE() ((void)0)
#define TR_WAVE_ANY(x) (x)
#define TR_HD static inline
#define TR_HDM inline
#define TR_HD_CALL static inline
#endif

#define TR_TMIN 0.0f
#define TR_TMAX 1.0e7f                          // shaders.cu:86 (tmax of every optixTrace)
#define TR_TANCHOR 9.99999e6f                   // a ray is anchored only if its mesh's whole box lies nearer than this (tr_ray_anchor)
#define TR_HUGE 3.0e38f                         // finite stand-in for 1/0
// Robust slab (Ize 2013): a plane distance (lo - o) * inv is three roundings from the exact one (subtract, reciprocal,
// multiply): within (1 +- 2^-24)^3.  An entry distance can come out 3 u too large and an exit distance 3 u too small, in
// different axes; the exit is therefore padded by 1 + 8 u, which after its own rounding still leaves > (1 + 3 u)^2: a box
// that the ray's line truly meets is accepted, whatever the roundings (contract 2 had 1 + 4 u: enough in practice, not
// in proof).
#define TR_SLAB_PAD 1.000000476837158203125f    // 1 + 2^-21
// Culling of boxes against the best hit so far: a float32 distance accepted without float64 is within 2^-11 relative
// of the exact one (tr_tri_test), a box's entry distance at most 3 u beyond the exact entry -- so a box is only culled
// when its entry lies beyond best * (1 + 2^-10); TR_TLIM bounds the interval for the queries that do not cull.
#define TR_CULL_SLACK 1.0009765625f             // 1 + 2^-10
#define TR_TLIM 1.001e7f                        // > TR_TMAX * TR_CULL_SLACK
// margins of the float32 inside test (u = 2^-24): see tr_tri_test
#define TR_BAND_K 5.9604644775390625e-07f       // 10 u
#define TR_BAND_ABS 7.888609052210118e-31f      // 2^-100
#define TR_BAND_MAXLEN 1.099511627776e12f       // 2^40: lengths beyond it leave the float32 part undecided (overflow)

struct tr_ray {
    float ox, oy, oz;
    float dx, dy, dz;
    float ix, iy, iz;   // clamped reciprocals
    float kd;           // TR_BAND_K * (|dx| + |dy| + |dz|): the ray's factor of the inside test's error bound
    // byte selectors (v_perm_b32) that pick, from a grid node's 16-bit plane pairs, the planes this ray ENTERS a box
    // through (lo where its direction is positive, hi where negative) and the ones it leaves through (tr_qnode_slabs)
    uint32_t sel_n, sel_f, sel_z;
    // Round 5: constants of the FUSED conservative box test of the grid nodes and the 8-wide nodes (tr_ray_fuse,
    // tr_bvh.h): t(plane q) = fma(q, qa, qn | qf) -- one fused multiply-add per plane instead of decode, subtract,
    // multiply.  (kx.., ex..: the per-axis reciprocal clamped so that every product stays finite, and the outward
    // margin that covers the rounding differences to the contract's three-step form: what a node with its own frame
    // -- the 8-wide nodes -- folds into its A / B.)  Only read by the kernels that walk those nodes.
    // (field order = the register pairs of the packed fma: (ax, ay) (nx, ny) (fx, fy) (nz, fz))
    float qax, qay;           // A  = scale * k
    float qnx, qny;           // B of the planes the ray ENTERS a box through:  (base - o) * k - e
    float qfx, qfy;           // B of the planes it leaves through:             (base - o) * k + e
    float qnz, qfz;
    float qaz, qaz2;          // (A of z twice: a register pair of its own)
    float kx, ky, kz;
    float ex, ey, ez;
};

TR_HD uint32_t tr_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
TR_HD float tr_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

TR_HD float tr_inv(float d) {
    float inv = 1.0f / d;
    if (fabsf(inv) > TR_HUGE) inv = copysignf(TR_HUGE, d);   // inf (d = +-0 or denormal)
    return inv;
}

TR_HD bool tr_finite(float x) { return fabsf(x) <= 3.4028234663852886e38f; }  // false for NaN

// ---- ray anchoring (contract 3, round 6) ---------------------------------------------------------------------------
// A ray that starts FAR outside the mesh's bounding box is MOVED ALONG ITSELF to just in front of the point where it
// enters the box, and everything downstream -- box tests, the inside test, the float64 part, the barycentrics -- sees the
// ray (o', d).  Why: the rounding error of the float32 inside test (tr_tri_fast) grows with |o - vertex|; from a camera
// 100 mesh sizes away a tenth of all leaf tests, from 1 000 sizes all of them, had to be decided in float64 (the same
// image 1.2x / 2x slower: profiles/r06_far_camera.txt).  From the anchor that distance is about the box's size,
// wherever the camera stands.
//   tn, tf = entry / exit distance of the slab test on the box [lo, hi] (clamped reciprocals, no padding); chord = tf - tn;
//   anchored iff  0 < tn <= tf,  tf < 1e7 (1 - 1e-6)  and  tn > chord / 2  (an origin closer than half a chord gains nothing;
//   a box that reaches beyond the reference's tmax keeps its ray where it is, see below);
//   t0 = tn - max(chord / 16, tn * 2^-18)          (in front of the entry by more than the slab test can be wrong);
//   o' = o + t0 * d  per component with a COMPENSATED product and sum (the product's rounding error from an fma, the
//   sum's from TwoSum, both added back): o' is within an ulp OF ITSELF of the exact point of the ray, however large o
//   is -- the displaced ray is the float32 noise of any ray transform, the same point for every triangle (watertightness
//   is untouched), and more accurate than round 5's arithmetic from the far origin was.
// Distances are measured from the anchor: the interval [0, 1e7] of the reference (shaders.cu:86) starts there (nothing
// lies in front of the box) -- and it ends where the reference's does for every triangle of the mesh: a ray is only
// anchored when the whole box lies within 1e7 (1 - 1e-6) of its origin (tf, which the slab arithmetic gets wrong by a few
// 2^-24 at most), so every triangle -- all inside the box -- is nearer than 1e7 from the origin AND from the anchor.  A pure function of (ray, box): the oracle restates it, every rank of a sharded run computes
// the same anchor.
TR_HD void tr_ray_anchor(const float* lo, const float* hi, float& ox, float& oy, float& oz, float dx, float dy, float dz) {
    const float ix = tr_inv(dx), iy = tr_inv(dy), iz = tr_inv(dz);
    const float x1 = (lo[0] - ox) * ix, x2 = (hi[0] - ox) * ix;
    const float y1 = (lo[1] - oy) * iy, y2 = (hi[1] - oy) * iy;
    const float z1 = (lo[2] - oz) * iz, z2 = (hi[2] - oz) * iz;
    const float tn = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
    const float tf = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
    const float chord = tf - tn;
    if (!(tn > 0.0f && tn <= tf && tf < TR_TANCHOR && tn > 0.5f * chord)) return;      // (NaN: not anchored)
    const float t0 = tn - fmaxf(chord * 0.0625f, tn * 3.814697265625e-06f);
#define TR_ANCHOR_AXIS(o, d)                                         \
    {                                                                \
        const float p = t0 * (d), e = fmaf(t0, (d), -p);             \
        const float s = p + (o), bb = s - p;                         \
        const float err = (p - (s - bb)) + ((o) - bb);               \
        (o) = s + (e + err);                                         \
    }
    TR_ANCHOR_AXIS(ox, dx)
    TR_ANCHOR_AXIS(oy, dy)
    TR_ANCHOR_AXIS(oz, dz)
#undef TR_ANCHOR_AXIS
}

// returns false when the ray has a non-finite component (such rays miss everything)
TR_HD bool tr_ray_setup(tr_ray& r, float ox, float oy, float oz, float dx, float dy, float dz) {
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = tr_inv(dx); r.iy = tr_inv(dy); r.iz = tr_inv(dz);
    {
        // (a direction longer than 2^40 leaves the float32 part of the inside test undecided: kd = inf, tr_tri_fast)
        const float l1 = (fabsf(dx) + fabsf(dy)) + fabsf(dz);
        r.kd = l1 <= TR_BAND_MAXLEN ? l1 * TR_BAND_K : INFINITY;
    }
    {
        const bool nx = r.ix < 0.f, ny = r.iy < 0.f, nz = r.iz < 0.f;
        // v_perm_b32(hi_pair, lo_pair, sel): selector bytes 0-3 pick bytes of lo_pair, 4-7 of hi_pair
        r.sel_n = ((ny ? 0x0706u : 0x0302u) << 16) | (nx ? 0x0504u : 0x0100u);
        r.sel_f = ((ny ? 0x0302u : 0x0706u) << 16) | (nx ? 0x0100u : 0x0504u);
        r.sel_z = nz ? 0x01000302u : 0x03020100u;
    }
    return tr_finite(ox) && tr_finite(oy) && tr_finite(oz) && tr_finite(dx) && tr_finite(dy) &&
           tr_finite(dz);
}

// Slab interval of a box.  tn = entry, tf = padded exit.  Inputs are NaN-free (finite ray,
// finite boxes, finite reciprocals), so fminf/fmaxf are plain min/max.
TR_HD void tr_slab(const tr_ray& r, float lox, float loy, float loz, float hix, float hiy,
                   float hiz, float& tn, float& tf) {
    float x1 = (lox - r.ox) * r.ix, x2 = (hix - r.ox) * r.ix;
    float y1 = (loy - r.oy) * r.iy, y2 = (hiy - r.oy) * r.iy;
    float z1 = (loz - r.oz) * r.iz, z2 = (hiz - r.oz) * r.iz;
    tn = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
    tf = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2)) * TR_SLAB_PAD;
}

// box accepted for traversal with an upper limit on the entry distance
// (tn <= tf && tf >= 0 && tn <= tlimit) with tlimit >= 0, folded into one compare:
// max(tn, 0) <= tf  <=>  tn <= tf && 0 <= tf;  max(tn, 0) <= tlimit  <=>  tn <= tlimit.
TR_HD bool tr_slab_hit(float tn, float tf, float tlimit) {
    return fmaxf(tn, 0.0f) <= fminf(tf, tlimit);
}

TR_HD float tr_dot(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

struct tr_hit {
    float t;     // distance along the ray (parametric: P = o + t d)
};

// ---- the EXACT part of the contract --------------------------------------------------------------------------------
// Woop / Benthin / Wald 2013 edge functions in float64 from the float32 inputs, in projective form (multiplied through
// by d[kz]: no division).  kz = the dominant axis of the direction (the first of equals), (kx, ky, kz) cyclic.  A
// vertex P becomes P' = P - o and Ph = (P'x dz - dx P'z, P'y dz - dy P'z): a function of (vertex, ray) alone, so both
// triangles of a shared edge see the same two points.  An edge function is the DIFFERENCE OF TWO ROUNDED PRODUCTS: its
// sign is the exact sign of the 2D orientation of the rounded points, or zero (rounding is monotone), and swapping the
// end points negates it exactly -- a ray cannot pass between two triangles that share an edge.
struct tr_woop {
    double U, V, W;      // weights of a, b, c (unnormalised): inside iff all >= 0 or all <= 0 (and not all zero)
    double Az, Bz, Cz;   // P'z of the three vertices
    double dz;           // d[kz]
};
// (x, y, z) here are already the permuted components (kx, ky, kz)
TR_HD void tr_woop64_xyz(float ox, float oy, float oz, float dx, float dy, float dz, float ax, float ay, float az,
                         float bx, float by, float bz, float cx, float cy, float cz, tr_woop& w) {
    const double Dx = dx, Dy = dy, Dz = dz, Ox = ox, Oy = oy, Oz = oz;
    const double Ax = (double)ax - Ox, Ay = (double)ay - Oy, Az = (double)az - Oz;
    const double Bx = (double)bx - Ox, By = (double)by - Oy, Bz = (double)bz - Oz;
    const double Cx = (double)cx - Ox, Cy = (double)cy - Oy, Cz = (double)cz - Oz;
    const double ahx = fma(-Dx, Az, Ax * Dz), ahy = fma(-Dy, Az, Ay * Dz);
    const double bhx = fma(-Dx, Bz, Bx * Dz), bhy = fma(-Dy, Bz, By * Dz);
    const double chx = fma(-Dx, Cz, Cx * Dz), chy = fma(-Dy, Cz, Cy * Dz);
    w.U = chx * bhy - chy * bhx;     // (two rounded products and a subtraction: NOT an fma)
    w.V = ahx * chy - ahy * chx;
    w.W = bhx * ahy - bhy * ahx;
    w.Az = Az; w.Bz = Bz; w.Cz = Cz;
    w.dz = Dz;
}
TR_HD void tr_woop64(float ox, float oy, float oz, float dx, float dy, float dz, float ax, float ay, float az,
                     float bx, float by, float bz, float cx, float cy, float cz, tr_woop& w) {
    int kz = 0;
    float m = fabsf(dx);
    if (fabsf(dy) > m) { kz = 1; m = fabsf(dy); }
    if (fabsf(dz) > m) kz = 2;
    // three copies of the arithmetic behind a branch instead of thirty selects in front of one: the lanes of a wave that
    // get here are few
    if (kz == 2) tr_woop64_xyz(ox, oy, oz, dx, dy, dz, ax, ay, az, bx, by, bz, cx, cy, cz, w);
    else if (kz == 0) tr_woop64_xyz(oy, oz, ox, dy, dz, dx, ay, az, ax, by, bz, bx, cy, cz, cx, w);
    else tr_woop64_xyz(oz, ox, oy, dz, dx, dy, az, ax, ay, bz, bx, by, cz, cx, cy, w);
}
// both windings; no strict sign conflict (a ZERO edge function: see tr_tie_code / tr_tie_own)
TR_HD bool tr_woop_inside(const tr_woop& w, double& det) {
    const bool neg = (w.U < 0.0) | (w.V < 0.0) | (w.W < 0.0), pos = (w.U > 0.0) | (w.V > 0.0) | (w.W > 0.0);
    det = (w.U + w.V) + w.W;
    return !(neg & pos) & (det != 0.0);
}
// ---- exact ties ---------------------------------------------------------------------------------------------------------
// A ray through a shared edge makes that edge's function zero in BOTH neighbours, one through a vertex two functions of
// every triangle of the fan.  "Zero counts as inside" (the first form of contract 3) hits them all: watertight, but an entry
// through an edge counts twice and flips the parity of contains_points -- and for rays and meshes on a common grid (axis-
// parallel rays through the vertices of a height field, a voxel surface) that is the ordinary case.  The rule of rasterisers
// instead: an edge belongs to ONE of its two triangles.  The directed edge p -> q of a triangle, seen along the ray, is
//     e = ((qx - px) dz - dx (qz - pz),  (qy - py) dz - dy (qz - pz))       (permuted components, float64; q - p is exact)
// -- a function of (edge, ray direction) alone, exactly negated when p and q swap; with the triangle's orientation
// normalised (s = the sign of U + V + W) the two neighbours of a shared edge see  s e  and  -s e, and the owner is the one
// whose  s e  lies in the half plane  ey > 0  or  (ey = 0 and ex > 0).  A hit through an edge of a closed surface is ONE
// hit; where the surface folds over in the ray's view (a silhouette edge) both neighbours see the same vector: none or both,
// the parity stays.  tr_tie_code: which functions are zero (bits 0-2: U, V, W) and the orientation (bit 3: negative), 0 = no
// tie; tr_tie_own: does the triangle own every edge named in the code (U belongs to b -> c, V to c -> a, W to a -> b).
TR_HD int tr_tie_code(const tr_woop& w, double det) {
    const int z = ((w.U == 0.0) ? 1 : 0) | ((w.V == 0.0) ? 2 : 0) | ((w.W == 0.0) ? 4 : 0);
    return z ? (z | (det < 0.0 ? 8 : 0)) : 0;
}
TR_HD bool tr_edge_own(double s, double dx, double dy, double dz, float px, float py, float pz, float qx, float qy, float qz) {
    const double ux = (double)qx - (double)px, uy = (double)qy - (double)py, uz = (double)qz - (double)pz;
    const double ex = s * fma(-dx, uz, ux * dz), ey = s * fma(-dy, uz, uy * dz);
    return (ey > 0.0) | ((ey == 0.0) & (ex > 0.0));
}
// (x, y, z) already permuted
TR_HD bool tr_tie_own_xyz(float dx, float dy, float dz, float ax, float ay, float az, float bx, float by, float bz,
                          float cx, float cy, float cz, int code) {
    const double s = (code & 8) ? -1.0 : 1.0;
    bool ok = true;
    if (code & 1) ok &= tr_edge_own(s, dx, dy, dz, bx, by, bz, cx, cy, cz);
    if (code & 2) ok &= tr_edge_own(s, dx, dy, dz, cx, cy, cz, ax, ay, az);
    if (code & 4) ok &= tr_edge_own(s, dx, dy, dz, ax, ay, az, bx, by, bz);
    return ok;
}
// a function of its own (a call on the device): it runs on exact ties only, and inlined into the float64 part it would
// double that function's registers -- which every kernel that calls it must keep free
TR_HD_CALL bool tr_tie_own(float dx, float dy, float dz, float ax, float ay, float az, float bx, float by, float bz,
                           float cx, float cy, float cz, int code) {
    int kz = 0;
    float m = fabsf(dx);
    if (fabsf(dy) > m) { kz = 1; m = fabsf(dy); }
    if (fabsf(dz) > m) kz = 2;
    if (kz == 2) return tr_tie_own_xyz(dx, dy, dz, ax, ay, az, bx, by, bz, cx, cy, cz, code);
    if (kz == 0) return tr_tie_own_xyz(dy, dz, dx, ay, az, ax, by, bz, bx, cy, cz, cx, code);
    return tr_tie_own_xyz(dz, dx, dy, az, ax, ay, bz, bx, by, cz, cx, cy, code);
}
TR_HD float tr_woop_t(const tr_woop& w, double det) {
    return (float)(fma(w.W, w.Cz, fma(w.V, w.Bz, w.U * w.Az)) / (det * w.dz));
}
// The exact part as one function: true iff the ray hits the triangle within [0, 1e7]; t = the distance.
// A REAL CALL on the device (TR_HD_CALL): it runs for < 1 % of the leaf tests and owns ~30 registers.
struct tr_exact_res { float t; int tie; };      // tie != 0: the hit stands only if the triangle owns its zero edges (tr_tie_own)
TR_HD_CALL tr_exact_res tr_tri_exact_t(float ox, float oy, float oz, float dx, float dy, float dz, float ax, float ay, float az,
                                       float bx, float by, float bz, float cx, float cy, float cz) {
    tr_woop w;
    double d64;
    tr_woop64(ox, oy, oz, dx, dy, dz, ax, ay, az, bx, by, bz, cx, cy, cz, w);
    tr_exact_res e;
    e.tie = 0;
    e.t = -1.0f;                                      // (no accepted distance is negative)
    if (tr_woop_inside(w, d64)) {
        e.t = tr_woop_t(w, d64);
        e.tie = tr_tie_code(w, d64);
    }
    return e;
}
TR_HD bool tr_tri_exact(const tr_ray& r, float ax, float ay, float az, float bx, float by, float bz,
                        float cx, float cy, float cz, tr_hit& h) {
    const tr_exact_res e = tr_tri_exact_t(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, ax, ay, az, bx, by, bz, cx, cy, cz);
    float t = e.t;
    if (__builtin_expect(e.tie != 0, 0)) {
        if (!tr_tie_own(r.dx, r.dy, r.dz, ax, ay, az, bx, by, bz, cx, cy, cz, e.tie)) t = -1.0f;
    }
    h.t = t;
    return t >= TR_TMIN && t <= TR_TMAX;
}

// ---- the hit predicate: a pure function of (ray, triangle) ----------------------------------------------------------
// The float32 part: Moller-Trumbore with fixed fma placement, both orientations through a sign flip:
//   (Uf, Vf, Wf) = sign(det) * (U, V, det - U - V);  exact arithmetic: inside  <=>  all three >= 0 (det != 0).
// Rounding: every product term of U = s . (d x e2) passes at most 7 roundings (s, e2, the inner product, the fma of the
// cross product, the outer product, two accumulating fma) and the terms' magnitudes sum to at most |s|_inf |d|_1 |e2|_1;
// the same for V with e1; det's terms to |d|_1 |e1|_inf |e2|_1 <= |d|_1 E^2 / 4 with E = |e1|_1 + |e2|_1.  Hence, with
// Wf = (|det| - Uf) - Vf,
//   |Uf - exact| + |Vf - exact| <= 7 u |s|_inf |d|_1 E,   |det - exact| <= 2 u |d|_1 E^2,
//   |Wf - exact| <= 8 u |s|_inf |d|_1 E + 2 u |d|_1 E^2   (the roundings of the two subtractions included),
// and with mm = 10 u |d|_1 E (|s|_inf + E) + 2^-100  (> every bound above + det's, the quantities of mm itself rounded)
// and m3 = min(Uf, Vf, Wf):
//   m3 < -mm  : one exact value is < 0, and since |det| cannot be wrong by more than mm another one is > 0
//                                                                                            -> outside, proven;
//   m3 >  mm  : all three exact values are > 0 with the true sign of det                    -> inside, proven;
//   otherwise : undecided -- the float64 edge functions decide (NaN from overflowing coordinates lands here, too).
// The bounds assume that no product overflows and that what underflows (an absolute error of 2^-149 per product, times
// one more factor) stays below the 2^-100 in mm: both hold while |s|_inf + E and |d|_1 are at most 2^40 (1.1e12) --
// larger lengths are left UNDECIDED, tiny ones are undecided by themselves (|Uf| < 2^-100).  The distance of a proven hit: t = T / det in float32 when |T| and |det| exceed
// 2^12 x their own error bounds (4 u |s|_inf E^2 and 4 u |d|_1 E^2: relative error of t < 2^-11, what TR_CULL_SLACK
// absorbs), otherwise undecided, too.  Accepted iff 0 <= t <= 1e7 (shaders.cu:86: tmin 0, tmax 1e7; no culling).
// Early exits are kept: in a wave most candidate triangles are proven outside, and the compiler skips the rest when no
// lane is left (s_cbranch_execz).
enum { TR_MISS = 0, TR_HIT = 1, TR_UNDECIDED = 2 };
TR_HD float tr_tri_scale(float ax, float ay, float az, float bx, float by, float bz, float cx, float cy, float cz) {
    const float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    const float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    return ((fabsf(e1x) + fabsf(e1y)) + fabsf(e1z)) + ((fabsf(e2x) + fabsf(e2y)) + fabsf(e2z));
}
// E = tr_tri_scale of the triangle (the kernels read it from the triangle's record, where the builder put it)
TR_HD int tr_tri_fast(const tr_ray& r, float ax, float ay, float az, float bx, float by, float bz,
                      float cx, float cy, float cz, float E, tr_hit& h) {
    float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    // p = d x e2
    float px = fmaf(r.dy, e2z, -(r.dz * e2y));
    float py = fmaf(r.dz, e2x, -(r.dx * e2z));
    float pz = fmaf(r.dx, e2y, -(r.dy * e2x));
    float det = tr_dot(e1x, e1y, e1z, px, py, pz);
    float sx = r.ox - ax, sy = r.oy - ay, sz = r.oz - az;
    float U = tr_dot(sx, sy, sz, px, py, pz);
    const uint32_t flip = tr_f2u(det) & 0x80000000u;
    const float Uf = tr_u2f(tr_f2u(U) ^ flip);
    const float Ls = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
    const float kE = r.kd * E;
    const float LsE = Ls + E;
    const float mm = fmaf(kE, LsE, TR_BAND_ABS);
    // the bounds hold while nothing overflows and what underflows stays below the 2^-100 of mm: lengths up to 2^40
    // (the direction's: kd = inf makes mm inf or NaN, and nothing below is decided)
    if (!(LsE <= TR_BAND_MAXLEN)) return TR_UNDECIDED;      // (NaN, too)
    // q = s x e1
    float qx = fmaf(sy, e1z, -(sz * e1y));
    float qy = fmaf(sz, e1x, -(sx * e1z));
    float qz = fmaf(sx, e1y, -(sy * e1x));
    float V = tr_dot(r.dx, r.dy, r.dz, qx, qy, qz);
    const float Vf = tr_u2f(tr_f2u(V) ^ flip);
    const float Wf = (fabsf(det) - Uf) - Vf;
    // ONE decision from the smallest of the three (round 6 tried exits after U and after V: with five lanes of a wave in
    // a leaf block some lane nearly always survives them, the branches cost more than they skipped; and FEWER levels of
    // nested exits -- two, or one branch around the division: 13 scalar instructions less per trip, no time: commit 650222c
    // has the variants (TR_TRI_FLAT, TR_FOLD_FLAT), profiles/r06_ab_flat_leaf.txt the numbers)
    const float m3 = fminf(fminf(Uf, Vf), Wf);
    if (m3 < -mm) return TR_MISS;
    if (!(m3 > mm)) return TR_UNDECIDED;
    const float T = tr_dot(e2x, e2y, e2z, qx, qy, qz);
    if ((int)!(fabsf(det) >= (kE * E) * 1024.0f) | (int)!(fabsf(T) >= (Ls * 9.765625e-4f) * (E * E))) return TR_UNDECIDED;
    const float t = T / det;
    h.t = t;
    return (t >= TR_TMIN && t <= TR_TMAX) ? TR_HIT : TR_MISS;
}
// the whole predicate in one call (brute force, single-triangle meshes, host code); the traversal kernels run the two
// parts apart (tr_fold_leaf / tr_drain_exact, tr_bvh.h)
TR_HD bool tr_tri_test(const tr_ray& r, float ax, float ay, float az, float bx, float by, float bz,
                       float cx, float cy, float cz, tr_hit& h) {
    const int c = tr_tri_fast(r, ax, ay, az, bx, by, bz, cx, cy, cz, tr_tri_scale(ax, ay, az, bx, by, bz, cx, cy, cz), h);
    if (c != TR_UNDECIDED) return c == TR_HIT;
    return tr_tri_exact(r, ax, ay, az, bx, by, bz, cx, cy, cz, h);
}

// outputs of a hit (shaders.cu:137-153).  The reference returns uv = (1-u-v, u) with (u, v) OptiX' barycentrics = the
// weights of face vertices 1 and 2 (:139,149), i.e. uv = (w0, w1), the weights of vertices 0 and 1 -- here each the
// float32 rounding of a float64 quotient of the edge functions of (ray, WINNING triangle), so both are correctly
// rounded whatever their magnitude; front (optixIsFrontFaceHit, :151) = counter-clockwise seen from the origin =
// d . ((b - a) x (c - a)) < 0  <=>  the edge functions' sum and d[kz] have the same sign.
// (a real call on the device, like the exact part of the predicate: the float64 arithmetic keeps its ~30 registers to
// itself -- inlined into the streaming kernels' refill path it sent 20-28 values of the persistent loop to scratch)
struct tr_bary { float w0, w1; int front; };
TR_HD_CALL tr_bary tr_tri_bary_call(float ox, float oy, float oz, float dx, float dy, float dz, float ax, float ay, float az,
                                    float bx, float by, float bz, float cx, float cy, float cz) {
    tr_woop w;
    tr_woop64(ox, oy, oz, dx, dy, dz, ax, ay, az, bx, by, bz, cx, cy, cz, w);
    const double det = (w.U + w.V) + w.W;
    tr_bary o;
    o.w0 = (float)(w.U / det);
    o.w1 = (float)(w.V / det);
    o.front = ((det < 0.0) == (w.dz < 0.0)) ? 1 : 0;
    return o;
}
TR_HD void tr_tri_bary(const tr_ray& r, float ax, float ay, float az, float bx, float by, float bz,
                       float cx, float cy, float cz, float& w0, float& w1, bool& front) {
    const tr_bary o = tr_tri_bary_call(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, ax, ay, az, bx, by, bz, cx, cy, cz);
    w0 = o.w0; w1 = o.w1; front = o.front != 0;
}

// Box of a triangle, padded outward by |x|*2^-21 + 2^-100 per bound.  The padding makes a
// ray that lies exactly in a bounding plane with a zero direction component (axis-aligned
// rays on axis-aligned geometry) fall inside the slab on both the lower and the upper side:
// (2^-100) * TR_HUGE = 2.4e8 > TR_TMAX.
#define TR_PAD_REL 4.76837158203125e-07f     /* 2^-21  */
#define TR_PAD_ABS 7.888609052210118e-31f    /* 2^-100 */
TR_HD float tr_pad(float x) { return fabsf(x) * TR_PAD_REL + TR_PAD_ABS; }
TR_HD void tr_tri_box(float ax, float ay, float az, float bx, float by, float bz, float cx,
                      float cy, float cz, float* lo, float* hi) {
    float l, h;
    l = fminf(fminf(ax, bx), cx); h = fmaxf(fmaxf(ax, bx), cx); lo[0] = l - tr_pad(l); hi[0] = h + tr_pad(h);
    l = fminf(fminf(ay, by), cy); h = fmaxf(fmaxf(ay, by), cy); lo[1] = l - tr_pad(l); hi[1] = h + tr_pad(h);
    l = fminf(fminf(az, bz), cz); h = fmaxf(fmaxf(az, bz), cz); lo[2] = l - tr_pad(l); hi[2] = h + tr_pad(h);
}

// (t, tri) lexicographic order used for closest hit and multi-hit ordering
TR_HD bool tr_closer(float t, int32_t tri, float bt, int32_t btri) {
    return (t < bt) || (t == bt && tri < btri);
}

// loc = u*V1 + v*V2 + (1-u-v)*V0 (shaders.cu:143-146) in float32 from the float32 (w0, w1): w2 = (1 - w0) - w1,
// loc = fma(w0, V0, fma(w2, V2, w1 * V1)); uv = (w0, w1) (:149).  A result can travel as (triangle, w0, w1) -- 12 bytes
// instead of 26 -- and be expanded elsewhere with the very same operations (tr_closest_expand).
TR_HD void tr_bary_outputs(float w0, float w1, float ax, float ay, float az, float bx, float by,
                           float bz, float cx, float cy, float cz, float* loc, float* uv) {
    const float w2 = (1.0f - w0) - w1;
    loc[0] = fmaf(w0, ax, fmaf(w2, cx, w1 * bx));
    loc[1] = fmaf(w0, ay, fmaf(w2, cy, w1 * by));
    loc[2] = fmaf(w0, az, fmaf(w2, cz, w1 * bz));
    uv[0] = w0;
    uv[1] = w1;
}
// all outputs of the hit of ray r on the triangle; returns front
TR_HD bool tr_hit_outputs(const tr_ray& r, float ax, float ay, float az, float bx, float by,
                          float bz, float cx, float cy, float cz, float* loc, float* uv) {
    float w0, w1;
    bool front;
    tr_tri_bary(r, ax, ay, az, bx, by, bz, cx, cy, cz, w0, w1, front);
    tr_bary_outputs(w0, w1, ax, ay, az, bx, by, bz, cx, cy, cz, loc, uv);
    return front;
}
