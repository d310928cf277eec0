// tr_math.h -- the arithmetic contract of the ray/triangle path ("Arithmetic
// contract").  Every float operation below is fixed, in this order, with explicit fma
// where fused and -ffp-contract=off everywhere else, so that any IEEE-754 binary32
// implementation (gfx950 VALU, or the CPU oracle's independent restatement in
// oracle/triro_oracle.c) produces bit-identical hit masks, triangle indices and keys.
//
// What the reference leaves to OptiX (optixTrace, shaders.cu:86,112,163,191,238; built-in
// triangle test; optixGetTriangleBarycentrics :139; optixIsFrontFaceHit :151) is replaced
// by: robust slab test -> Moller-Trumbore -> distance clamped into the triangle's own slab
// interval.  The hit predicate is a pure function of (ray, triangle), therefore independent
// of the BVH that finds the candidates.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TR_HD __host__ __device__ __forceinline__
#define TR_HDM __host__ __device__ __forceinline__
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
#endif

#define TR_TMIN 0.0f
#define TR_TMAX 1.0e7f                          // shaders.cu:86 (tmax of every optixTrace)
#define TR_HUGE 3.0e38f                         // finite stand-in for 1/0
#define TR_SLAB_PAD 1.00000023841857910156f     // 1 + 2^-22 (Ize 2013 robust slab)

struct tr_ray {
    float ox, oy, oz;
    float dx, dy, dz;
    float ix, iy, iz;   // clamped reciprocals
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

// returns false when the ray has a non-finite component (such rays miss everything)
TR_HD bool tr_ray_setup(tr_ray& r, float ox, float oy, float oz, float dx, float dy, float dz) {
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = tr_inv(dx); r.iy = tr_inv(dy); r.iz = tr_inv(dz);
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
    float t;     // key distance (MT distance clamped into the triangle's slab interval)
    float U, V;  // unnormalised barycentrics of vertices 1 and 2
    float det;   // > 0 : front face (CCW from the ray origin)
};

// det, U, V of Moller-Trumbore for (ray, triangle) -- a pure function of its inputs: the
// closest-hit kernels keep only (t_key, face, slot) per ray while traversing and call this
// again on the winning triangle to get the same bits for the outputs.
TR_HD void tr_tri_duv(const tr_ray& r, float ax, float ay, float az, float bx, float by, float bz,
                      float cx, float cy, float cz, float& det, float& U, float& V) {
    float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    float px = fmaf(r.dy, e2z, -(r.dz * e2y));
    float py = fmaf(r.dz, e2x, -(r.dx * e2z));
    float pz = fmaf(r.dx, e2y, -(r.dy * e2x));
    det = tr_dot(e1x, e1y, e1z, px, py, pz);
    float sx = r.ox - ax, sy = r.oy - ay, sz = r.oz - az;
    U = tr_dot(sx, sy, sz, px, py, pz);
    float qx = fmaf(sy, e1z, -(sz * e1y));
    float qy = fmaf(sz, e1x, -(sx * e1z));
    float qz = fmaf(sx, e1y, -(sy * e1x));
    V = tr_dot(r.dx, r.dy, r.dz, qx, qy, qz);
}

// Moller-Trumbore given the triangle's own slab interval [tn, tf].  Early exits are kept: in
// a wave most candidate triangles fail on det / U / V, and the compiler skips the rest of the
// test when no lane is left (s_cbranch_execz).
TR_HD bool tr_tri_mt(const tr_ray& r, float ax, float ay, float az, float bx, float by, float bz,
                     float cx, float cy, float cz, float tn, float tf, tr_hit& h) {
    float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    // p = d x e2
    float px = fmaf(r.dy, e2z, -(r.dz * e2y));
    float py = fmaf(r.dz, e2x, -(r.dx * e2z));
    float pz = fmaf(r.dx, e2y, -(r.dy * e2x));
    float det = tr_dot(e1x, e1y, e1z, px, py, pz);
    if (det == 0.0f) return false;
    float sx = r.ox - ax, sy = r.oy - ay, sz = r.oz - az;
    float U = tr_dot(sx, sy, sz, px, py, pz);
    // q = s x e1
    float qx = fmaf(sy, e1z, -(sz * e1y));
    float qy = fmaf(sz, e1x, -(sx * e1z));
    float qz = fmaf(sx, e1y, -(sy * e1x));
    float V = tr_dot(r.dx, r.dy, r.dz, qx, qy, qz);
    // inside test, both orientations at once: flipping the sign of U, V and det when det < 0
    // is exact (and -(U+V) == (-U)+(-V)), so  det>0 ? (U>=0 && V>=0 && U+V<=det)
    //                                              : (U<=0 && V<=0 && U+V>=det)
    // becomes three compares on the flipped values -- no branch on the orientation.
    const uint32_t flip = tr_f2u(det) & 0x80000000u;
    const float Uf = tr_u2f(tr_f2u(U) ^ flip), Vf = tr_u2f(tr_f2u(V) ^ flip);
    const bool ok = (Uf >= 0.0f) & (Vf >= 0.0f) & ((Uf + Vf) <= fabsf(det));
    if (!ok) return false;
    float T = tr_dot(e2x, e2y, e2z, qx, qy, qz);
    float t = T / det;
    float tk = fminf(fmaxf(t, tn), tf);
    if (!(tk >= TR_TMIN && tk <= TR_TMAX)) return false;
    h.t = tk; h.U = U; h.V = V; h.det = det;
    return true;
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

// full predicate: slab of the triangle's box, then MT (used where the box is not already
// known from the parent node, e.g. the brute-force kernel for single-triangle meshes)
TR_HD bool tr_tri_hit(const tr_ray& r, float ax, float ay, float az, float bx, float by, float bz,
                      float cx, float cy, float cz, tr_hit& h) {
    float lo[3], hi[3], tn, tf;
    tr_tri_box(ax, ay, az, bx, by, bz, cx, cy, cz, lo, hi);
    tr_slab(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tn, tf);
    if (!tr_slab_hit(tn, tf, TR_TMAX)) return false;
    return tr_tri_mt(r, ax, ay, az, bx, by, bz, cx, cy, cz, tn, tf, h);
}

// (t, tri) lexicographic order used for closest hit and multi-hit ordering
TR_HD bool tr_closer(float t, int32_t tri, float bt, int32_t btri) {
    return (t < bt) || (t == bt && tri < btri);
}

// outputs of a hit: shaders.cu:137-153.  loc = u*V1 + v*V2 + (1-u-v)*V0 (:143-146),
// uv = (1-u-v, u) (:149).  In two steps so that a result can travel as (triangle, u, v) -- 12 bytes
// instead of 26 -- and be expanded elsewhere with the very same operations (tr_closest_expand).
TR_HD void tr_hit_bary(const tr_hit& h, float& u, float& v) {
    u = h.U / h.det;
    v = h.V / h.det;
}
TR_HD void tr_bary_outputs(float u, float v, float ax, float ay, float az, float bx, float by,
                           float bz, float cx, float cy, float cz, float* loc, float* uv) {
    float w = (1.0f - u) - v;
    loc[0] = fmaf(w, ax, fmaf(v, cx, u * bx));
    loc[1] = fmaf(w, ay, fmaf(v, cy, u * by));
    loc[2] = fmaf(w, az, fmaf(v, cz, u * bz));
    uv[0] = w;
    uv[1] = u;
}
TR_HD void tr_hit_outputs(const tr_hit& h, float ax, float ay, float az, float bx, float by,
                          float bz, float cx, float cy, float cz, float* loc, float* uv) {
    float u, v;
    tr_hit_bary(h, u, v);
    tr_bary_outputs(u, v, ax, ay, az, bx, by, bz, cx, cy, cz, loc, uv);
}
