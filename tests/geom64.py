"""Independent float64 ray/triangle geometry (numpy, brute force) used to cross-check the oracle
and the GPU path "from outside" the arithmetic contract: textbook Moller-Trumbore in double
precision, no boxes, no clamps.  It is NOT bit-comparable (different precision); the tests below
compare only rays whose nearest hit is robust (well inside a triangle, not near-tangent, not a
near-tie between two triangles)."""
import numpy as np


def closest_f64(vertices, faces, origins, directions, tmax=1.0e7, chunk=2048):
    v = np.asarray(vertices, np.float64)
    f = np.asarray(faces)
    o = np.asarray(origins, np.float64).reshape(-1, 3)
    d = np.asarray(directions, np.float64).reshape(-1, 3)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    e1, e2 = b - a, c - a
    n = len(o)
    t_best = np.full(n, np.inf)
    tri = np.full(n, -1, np.int64)
    margin = np.zeros(n)            # barycentric distance of the best hit from the triangle border
    second = np.full(n, np.inf)     # distance of the runner-up hit
    graze = np.zeros(n, bool)       # the ray passes within 1e-4 (barycentric) of some triangle border
    for s in range(0, n, chunk):
        oo, dd = o[s:s + chunk, None, :], d[s:s + chunk, None, :]
        p = np.cross(dd, e2[None])
        det = np.einsum("rfk,fk->rf", p, e1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            sv = oo - a[None]
            u = np.einsum("rfk,rfk->rf", sv, p) * inv
            q = np.cross(sv, e1[None])
            vv = np.einsum("rfk,rfk->rf", np.broadcast_to(dd, q.shape), q) * inv
            t = np.einsum("rfk,fk->rf", q, e2) * inv
            border = np.minimum(np.minimum(u, vv), 1 - u - vv)
        ok = (det != 0) & (border >= 0) & (t >= 0) & (t <= tmax)
        tt = np.where(ok, t, np.inf)
        k = np.argmin(tt, axis=1)
        r = np.arange(len(k))
        t_best[s:s + chunk] = tt[r, k]
        tri[s:s + chunk] = np.where(np.isfinite(tt[r, k]), k, -1)
        margin[s:s + chunk] = border[r, k]
        tt[r, k] = np.inf
        second[s:s + chunk] = tt.min(axis=1)
        graze[s:s + chunk] = ((det != 0) & (t >= -1e-4) & (np.abs(border) < 1e-4)).any(axis=1)
    hit = tri >= 0
    loc = o + d * np.where(hit, t_best, 0.0)[:, None]
    with np.errstate(invalid="ignore"):
        gap = np.where(np.isfinite(second), second - t_best, np.inf)
    robust = ~graze & np.where(hit, (margin > 1e-4) & (gap > 1e-4 * np.maximum(1.0, np.where(hit, t_best, 1.0))), True)
    return hit, tri, t_best, loc, robust
