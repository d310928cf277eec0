"""Synthetic meshes and ray batches for tests and bench.py (NOT product code).

Everything is a deterministic function of its arguments (no network, no files):
  * icosphere(subdivisions)           -- trimesh.creation.icosphere's construction
    (icosahedron of Andreas Kahler's ordering, midpoint subdivision, renormalise);
    80 tris at 1 subdivision (BASELINE.json config 1), 1 310 720 at 8 (headline)
  * displaced(...)                    -- smooth radial displacement ("noise", seed)
  * nested_shells(...)                -- BASELINE.md config C4 (<= 8 hits per ray)
  * ortho_grid / readme_perspective   -- README.md:35-39, test/test.py:17-24 inputs
  * pinhole_grid                      -- BASELINE.md C2/C5(i): 40 deg vFOV, raster order
  * hash_rays                         -- BASELINE.md C3/C5(ii): pure function of the
    global ray index; numpy and torch versions produce identical bits
  * interior_room / ref_shape_rays    -- an interior scene with the camera INSIDE and the ray shape
    of the reference's own published timing (test/performance_test.py:10-20, 39-44)
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------
# meshes
# ----------------------------------------------------------------------------


def icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([-1, t, 0, 1, t, 0, -1, -t, 0, 1, -t, 0,
                  0, -1, t, 0, 1, t, 0, -1, -t, 0, 1, -t,
                  t, 0, -1, t, 0, 1, -t, 0, -1, -t, 0, 1], dtype=np.float64).reshape(-1, 3)
    f = np.array([0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11,
                  1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8,
                  3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9,
                  4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1], dtype=np.int64).reshape(-1, 3)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v, f


def _subdivide(v, f):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    e.sort(axis=1)
    key = e[:, 0] * np.int64(len(v)) + e[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    a, b = uniq // len(v), uniq % len(v)
    mid = 0.5 * (v[a] + v[b])
    nf = len(f)
    m01 = inv[:nf] + len(v)
    m12 = inv[nf:2 * nf] + len(v)
    m20 = inv[2 * nf:] + len(v)
    v2 = np.concatenate([v, mid], 0)
    f2 = np.concatenate([
        np.stack([f[:, 0], m01, m20], 1),
        np.stack([f[:, 1], m12, m01], 1),
        np.stack([f[:, 2], m20, m12], 1),
        np.stack([m01, m12, m20], 1)], 0)
    return v2, f2


def icosphere(subdivisions: int = 3, radius: float = 1.0):
    """Unit icosphere, outward-wound (CCW seen from outside): 20*4^s faces."""
    v, f = icosahedron()
    for _ in range(subdivisions):
        v, f = _subdivide(v, f)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    return (v * radius).astype(np.float32), f.astype(np.int32)


def displaced(v, seed: int = 0, amplitude: float = 0.08, octaves: int = 3):
    """Smooth deterministic radial displacement of a star-shaped mesh about 0."""
    rng = np.random.default_rng(seed)
    v64 = v.astype(np.float64)
    r = np.linalg.norm(v64, axis=1, keepdims=True)
    n = v64 / r
    disp = np.zeros(len(v64))
    for k in range(octaves):
        for _ in range(4):
            axis = rng.normal(size=3)
            axis /= np.linalg.norm(axis)
            freq = (3.0 * 2 ** k) * (0.75 + 0.5 * rng.random())
            phase = rng.random() * 2 * np.pi
            disp += (0.5 ** k) * np.sin(freq * (n @ axis) * np.pi + phase) / 4.0
    return (v64 * (1.0 + amplitude * disp[:, None])).astype(np.float32)


def bunny_standin():
    """BASELINE.md C2 stand-in (no Stanford bunny file in this image): icosphere(6)
    = 81 920 tris with 3-octave displacement, seed 0.  Always labelled 'stand-in'."""
    v, f = icosphere(6)
    return displaced(v, seed=0, amplitude=0.12), f


def load_mesh_file(path: str):
    """Triangle mesh from a Stanford PLY (ascii or binary_little_endian; e.g. bun_zipper.ply) or a Wavefront
    OBJ file -> (float32 [nv,3], int32 [nf,3]); polygons are fanned into triangles.  No trimesh needed."""
    import os
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        vs, fs = [], []
        for ln in open(path, errors="replace"):
            t = ln.split()
            if not t:
                continue
            if t[0] == "v":
                vs.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(x.split("/")[0]) for x in t[1:]]
                idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
                fs += [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]
        return np.asarray(vs, np.float32), np.asarray(fs, np.int32)
    if ext != ".ply":
        raise ValueError(f"unsupported mesh file type: {path}")
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, elems, cur = None, [], None
        while True:
            t = fh.readline().decode("ascii", "replace").split()
            if not t:
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                cur = {"name": t[1], "count": int(t[2]), "props": []}
                elems.append(cur)
            elif t[0] == "property":
                cur["props"].append(t[1:])
            elif t[0] == "end_header":
                break
        types = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
                 "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}
        v = f = None
        if fmt == "ascii":
            toks = fh.read().split()
            pos = 0
            for e in elems:
                if e["name"] == "vertex":
                    k = len(e["props"])
                    names = [p[-1] for p in e["props"]]
                    arr = np.array(toks[pos:pos + k * e["count"]], dtype=np.float64).reshape(e["count"], k)
                    v = arr[:, [names.index("x"), names.index("y"), names.index("z")]]
                    pos += k * e["count"]
                elif e["name"] == "face":
                    tris = []
                    for _ in range(e["count"]):
                        m = int(toks[pos])
                        idx = [int(x) for x in toks[pos + 1:pos + 1 + m]]
                        tris += [[idx[0], idx[j], idx[j + 1]] for j in range(1, m - 1)]
                        pos += 1 + m
                    f = np.asarray(tris)
                else:
                    raise ValueError("ascii PLY with extra elements is not supported")
        elif fmt == "binary_little_endian":
            for e in elems:
                if e["name"] == "vertex":
                    dt = np.dtype([(p[-1], "<" + types[p[0]]) for p in e["props"]])
                    arr = np.frombuffer(fh.read(dt.itemsize * e["count"]), dt)
                    v = np.stack([arr["x"], arr["y"], arr["z"]], 1)
                elif e["name"] == "face":
                    p = e["props"][0]            # list <count type> <index type> vertex_indices
                    ct, it = np.dtype("<" + types[p[1]]), np.dtype("<" + types[p[2]])
                    tris = []
                    for _ in range(e["count"]):
                        m = int(np.frombuffer(fh.read(ct.itemsize), ct)[0])
                        idx = np.frombuffer(fh.read(it.itemsize * m), it)
                        tris += [[idx[0], idx[j], idx[j + 1]] for j in range(1, m - 1)]
                    f = np.asarray(tris)
                else:
                    raise ValueError("binary PLY with extra elements is not supported")
        else:
            raise ValueError(f"unsupported PLY format {fmt}")
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def bunny_mesh():
    """BASELINE.md C2/C3 mesh: the file $TRIRO_BUNNY points to (canonical: bun_zipper.ply, 69 451 tris) if
    set, else the labelled procedural stand-in.  Returns (vertices, faces, label)."""
    import os
    path = os.environ.get("TRIRO_BUNNY")
    if path:
        v, f = load_mesh_file(path)
        return v, f, f"{os.path.basename(path)} ({len(f)} tris)"
    v, f = bunny_standin()
    return v, f, f"stand-in ({len(f)} tris; no Stanford bunny file in this image)"


def headline_mesh(subdivisions: int = 8):
    """icosphere(8) = 1 310 720 tris + displacement, seed 0."""
    v, f = icosphere(subdivisions)
    return displaced(v, seed=0, amplitude=0.08), f


def nested_shells(subdivisions: int = 7, radii=(1.0, 0.8, 0.6, 0.4)):
    """concentric icospheres; central rays have 2*len(radii) hits."""
    v0, f0 = icosphere(subdivisions)
    vs, fs = [], []
    for i, r in enumerate(radii):
        vs.append(v0 * np.float32(r))
        fs.append(f0 + np.int32(i * len(v0)))
    return np.concatenate(vs, 0), np.concatenate(fs, 0)


def two_triangles():
    """test/test.py:47-58 geometry."""
    v = np.array([[0.5, -0.5, 0], [0, 0.5, 0], [-0.5, -0.5, 0],
                  [0.5, -0.5, -1], [0, 0.5, -1], [-0.5, -0.5, -1]], np.float32)
    f = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    return v, f


def random_soup(n_tris: int, seed: int = 0, extent: float = 1.0, size: float = 0.2):
    """Unstructured triangle soup (overlapping, arbitrary winding)."""
    rng = np.random.default_rng(seed)
    c = (rng.random((n_tris, 1, 3)) * 2 - 1) * extent
    v = (c + (rng.random((n_tris, 3, 3)) * 2 - 1) * size).reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * n_tris, dtype=np.int32).reshape(-1, 3)
    return v, f


def deep_tree_mesh(reps: int = 4000):
    """Adversarial input for the builder: 63 tiny triangles whose Morton codes are the single
    bits 1<<j plus a run of `reps` identical triangles at the origin -> a Karras tree of
    height 63 + log2(reps) > 64 with plain Morton keys (forces the depth-bounded key mode)."""
    tri = np.array([[0, 0, 0], [1e-9, 0, 0], [0, 1e-9, 0]], np.float32)
    cs = [np.ones(3)]
    for j in range(63):
        c = np.zeros(3)
        c[2 - (j % 3)] = 2.0 ** (j // 3 - 21)
        cs.append(c)
    vs = [tri + c.astype(np.float32) for c in cs] + [tri.copy() for _ in range(reps)]
    v = np.concatenate(vs).astype(np.float32)
    return v, np.arange(len(v), dtype=np.int32).reshape(-1, 3)


def _grid_patch(p0, du, dv, nu: int, nv: int, height=None):
    """(nu+1) x (nv+1) vertices p0 + i/nu*du + j/nv*dv (+ height(i/nu, j/nv) along du x dv), 2*nu*nv
    triangles wound counter-clockwise seen from the side the normal du x dv points to."""
    p0, du, dv = (np.asarray(a, np.float64) for a in (p0, du, dv))
    iu, iv = np.meshgrid(np.arange(nu + 1), np.arange(nv + 1), indexing="ij")
    s, t = iu / nu, iv / nv
    v = p0 + s[..., None] * du + t[..., None] * dv
    if height is not None:
        n = np.cross(du, dv)
        v = v + height(s, t)[..., None] * (n / np.linalg.norm(n))
    idx = (iu * (nv + 1) + iv)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    f = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)], 0)
    return v.reshape(-1, 3), f


def _box(center, half, n: int):
    """axis-aligned box, every face an n x n grid, outward-wound"""
    c, h = np.asarray(center, np.float64), np.asarray(half, np.float64)
    vs, fs, base = [], [], 0
    ex, ey, ez = np.eye(3) * 2 * h
    for p0, du, dv in ((c - h, ey, ex), (c - h + ez, ex, ey),          # z- (normal -z), z+
                       (c - h, ez, ey), (c - h + ex, ey, ez),          # x-, x+
                       (c - h, ex, ez), (c - h + ey, ez, ex)):         # y-, y+
        v, f = _grid_patch(p0, du, dv, n, n)
        vs.append(v)
        fs.append(f + base)
        base += len(v)
    return np.concatenate(vs), np.concatenate(fs)


def interior_room(seed: int = 0, detail: float = 1.0):
    """An INTERIOR scene (the reference's only published timing is a camera inside a bedroom model,
    test/performance_test.py:39-52; that asset is not obtainable here): a closed 8 x 3 x 6 room whose
    walls face inwards, a finely tessellated floor with a small height field, 24 boxes of "furniture"
    standing on it and 8 spheres.  detail = 1 -> about 0.9 M triangles; every ray cast from inside hits
    something, most cross several objects (depth complexity >= 4 from the default camera)."""
    rng = np.random.default_rng(seed)
    k = lambda n: max(2, int(round(n * np.sqrt(detail))))          # noqa: E731
    X, Y, Z = 4.0, 3.0, 3.0
    parts = []
    ph = rng.random(4) * 2 * np.pi

    def floor_h(s, t):
        return 0.02 * (np.sin(9 * np.pi * s + ph[0]) * np.sin(7 * np.pi * t + ph[1]) +
                       0.5 * np.sin(23 * np.pi * s + ph[2]) * np.sin(19 * np.pi * t + ph[3]))
    parts.append(_grid_patch((-X, 0, -Z), (0, 0, 2 * Z), (2 * X, 0, 0), k(360), k(480), floor_h))   # floor, normal +y
    parts.append(_grid_patch((-X, Y, -Z), (2 * X, 0, 0), (0, 0, 2 * Z), k(80), k(60)))              # ceiling, normal -y
    parts.append(_grid_patch((-X, 0, -Z), (2 * X, 0, 0), (0, Y, 0), k(80), k(30)))                  # wall z = -Z, normal +z
    parts.append(_grid_patch((-X, 0, Z), (0, Y, 0), (2 * X, 0, 0), k(30), k(80)))                   # wall z = +Z, normal -z
    parts.append(_grid_patch((-X, 0, -Z), (0, Y, 0), (0, 0, 2 * Z), k(30), k(60)))                  # wall x = -X, normal +x
    parts.append(_grid_patch((X, 0, -Z), (0, 0, 2 * Z), (0, Y, 0), k(60), k(30)))                   # wall x = +X, normal -x
    tops = []
    for i in range(6):
        for j in range(4):
            cx = -X + (i + 0.5 + 0.3 * (rng.random() - 0.5)) * (2 * X / 6)
            cz = -Z + (j + 0.5 + 0.3 * (rng.random() - 0.5)) * (2 * Z / 4)
            hx, hz = 0.22 + 0.25 * rng.random(), 0.22 + 0.25 * rng.random()
            hy = 0.2 + 0.7 * rng.random()
            parts.append(_box((cx, 0.05 + hy, cz), (hx, hy, hz), k(36)))
            tops.append((cx, 0.05 + 2 * hy, cz))
    sv, sf = icosphere(max(2, int(round(5 + np.log2(max(detail, 1e-3)) / 2))))
    for t in rng.choice(len(tops), 8, replace=False):
        r = 0.18 + 0.1 * rng.random()
        parts.append((sv.astype(np.float64) * r + np.array([tops[t][0], tops[t][1] + r + 0.02, tops[t][2]]), sf.astype(np.int64)))
    vs, fs, base = [], [], 0
    for v, f in parts:
        vs.append(v)
        fs.append(f + base)
        base += len(v)
    return np.concatenate(vs).astype(np.float32), np.concatenate(fs).astype(np.int32)


# a camera in one corner, about a metre above the floor, looking across the room: 5.7 surfaces per
# ray on average, up to 13 (more than the multi-hit cap of 8)
INTERIOR_EYE, INTERIOR_TARGET = (-3.6, 1.0, -2.6), (3.2, 0.5, 2.3)


# ----------------------------------------------------------------------------
# rays (numpy; float32)
# ----------------------------------------------------------------------------


def ortho_grid(n: int = 800, z: float = 3.0):
    """BASELINE config 1: origins (x, y, z), x in linspace(-1,1,n), y in linspace(1,-1,n),
    direction (0,0,-1).  Shapes [n,n,3]."""
    y, x = np.meshgrid(np.linspace(1, -1, n, dtype=np.float32),
                       np.linspace(-1, 1, n, dtype=np.float32), indexing="ij")
    o = np.stack([x, y, np.full_like(x, z)], -1)
    d = np.broadcast_to(np.array([0, 0, -1], np.float32), o.shape)
    return o, d


def readme_perspective(n: int = 800):
    """README.md:35-39: stride-0 origin (0,0,3), un-normalised directions (x,y,-1)."""
    y, x = np.meshgrid(np.linspace(1, -1, n, dtype=np.float32),
                       np.linspace(-1, 1, n, dtype=np.float32), indexing="ij")
    d = np.stack([x, y, -np.ones_like(x)], -1)
    o = np.broadcast_to(np.array([0, 0, 3], np.float32), d.shape)
    return o, d


def pinhole_grid(width: int = 1024, height: int = 1024, vfov_deg: float = 40.0,
                 distance: float = 2.5, center=(0.0, 0.0, 0.0)):
    """Camera on +z at `distance` from `center`, looking down -z; unit directions,
    raster order (row 0 = top).  Origins are a stride-0 broadcast like the README."""
    f = 0.5 * height / np.tan(np.radians(vfov_deg) / 2)
    ys, xs = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64),
                         indexing="ij")
    x = xs - (width - 1) / 2
    y = (height - 1) / 2 - ys
    d = np.stack([x, y, -np.full_like(x, f)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    c = np.asarray(center, np.float64)
    o = np.broadcast_to((c + np.array([0, 0, distance])).astype(np.float32), d.shape)
    return o, d.astype(np.float32)


def terrain(n: int = 724, seed: int = 0, size: float = 40.0, relief: float = 2.5):
    """An open height field seen from just above the ground: n x n cells (2 n^2 triangles; n = 724 ->
    1 048 352) over a size x size square in the xz plane, heights = five octaves of sines (ridges up to
    `relief`).  With TERRAIN_EYE / TERRAIN_TARGET most rays GRAZE the surface for a long way before they hit
    (long traversals near the horizon), and the upper part of the image misses everything (sky) -- the
    opposite of the closed blobs and of the interior scene: skewed block costs, rays that leave the mesh."""
    rng = np.random.default_rng(seed)
    ph = rng.random((5, 4)) * 2 * np.pi
    fr = np.array([[1.0, 1.3], [2.1, 1.7], [4.3, 3.9], [8.9, 7.7], [17.0, 19.0]])

    def h(s, t):
        out = np.zeros_like(s)
        for k in range(5):
            out += (0.5 ** k) * (np.sin(2 * np.pi * fr[k, 0] * s + ph[k, 0]) * np.sin(2 * np.pi * fr[k, 1] * t + ph[k, 1]) +
                                 0.5 * np.sin(2 * np.pi * (fr[k, 0] * s + fr[k, 1] * t) + ph[k, 2]))
        return relief * out / 3.0
    v, f = _grid_patch((-size / 2, 0.0, -size / 2), (0, 0, size), (size, 0, 0), n, n, h)      # normal +y
    return v.astype(np.float32), f.astype(np.int32)


TERRAIN_EYE, TERRAIN_TARGET = (-17.0, 3.2, -15.0), (12.0, -1.0, 11.0)


def ref_shape_rays(eye, target, w: int = 640, h: int = 360, f: float = 444.0, up=(0.0, 1.0, 0.0)):
    """The reference's published benchmark shape (test/performance_test.py:10-20, 29-31, 39-44): w x h
    pinhole rays, focal length f pixels, unit directions x = col - (w-1)/2, y = row - (h-1)/2, z = -f
    rotated by the camera matrix; the origin is ONE point broadcast with stride 0."""
    eye, target, up = (np.asarray(a, np.float64) for a in (eye, target, up))
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    d = np.stack([xs - (w - 1) / 2, ys - (h - 1) / 2, -np.full_like(xs, f)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cam = np.stack([right, down, -fwd], 1)            # columns: image x, image y (down), backwards
    d = d @ cam.T
    o = np.broadcast_to(eye.astype(np.float32), d.shape)
    return o, d.astype(np.float32)


_M32 = 0xFFFFFFFF


def _hash32_np(x):
    x = x & _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def hash_rays(n: int, seed: int, lo, hi, start: int = 0):
    """n rays from an integer hash of (global index, seed, channel).
    origins uniform in [lo,hi]^3 (per-axis arrays), directions uniform in [-1,1)^3,
    NOT normalised.  Value = (h >> 8) * 2^-24 mapped affinely, all in float32."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    lo = np.asarray(lo, np.float32)
    hi = np.asarray(hi, np.float32)
    out = []
    for ch in range(6):
        h = _hash32_np(idx * np.uint64(6) + np.uint64(ch) + (np.uint64(seed) << np.uint64(20)))
        out.append(((h >> np.uint64(8)).astype(np.float32)) * np.float32(2.0 ** -24))
    o = np.stack([lo[i] + out[i] * (hi[i] - lo[i]) for i in range(3)], -1).astype(np.float32)
    d = np.stack([out[3 + i] * np.float32(2.0) - np.float32(1.0) for i in range(3)], -1)
    return o, d.astype(np.float32)


def hash_rays_torch(n: int, seed: int, lo, hi, start: int = 0, device="cuda"):
    """Same bits as hash_rays, generated on `device` with torch (int64 arithmetic)."""
    import torch
    idx = torch.arange(start, start + n, dtype=torch.int64, device=device)

    def h32(x):
        x = x & _M32
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & _M32
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & _M32
        x = x ^ (x >> 16)
        return x

    lo_t = torch.as_tensor(np.asarray(lo, np.float32), device=device)
    hi_t = torch.as_tensor(np.asarray(hi, np.float32), device=device)
    u = [((h32(idx * 6 + ch + (seed << 20)) >> 8).to(torch.float32)) * (2.0 ** -24)
         for ch in range(6)]
    o = torch.stack([lo_t[i] + u[i] * (hi_t[i] - lo_t[i]) for i in range(3)], -1)
    d = torch.stack([u[3 + i] * 2.0 - 1.0 for i in range(3)], -1)
    return o.contiguous(), d.contiguous()
