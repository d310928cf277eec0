#!/usr/bin/env python3
"""Occupancy timeline of one query launch (experiment; needs the TR_TIMELINE build):

  make -C trimesh-ray-optix_amd/csrc OUTDIR=../lib_var/timeline EXTRA=-DTR_TIMELINE=131072
  TRIRO_HIP_LIBRARY=trimesh-ray-optix_amd/lib_var/timeline/libtriro_hip.so \
      python scripts/exp_timeline.py [--res 1024] [--query closest] [--opt k=v ...]

Every wave of k_query_direct records (start, end) in 100 MHz wall-clock ticks, its HW_ID / XCC_ID and
its logical block.  Prints one JSON object: launch span, the distribution of wave durations, the number
of resident waves per 10 us bin (chip-wide and for the busiest / idlest CU), when the last wave
started, and how much of the wave-slot time (7 waves x 4 SIMDs x 256 CUs x span) was occupied."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from triro.backend import ops as hops  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--query", default="closest")
ap.add_argument("--warmup", type=int, default=8)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--waves-per-simd", type=int, default=7)
ap.add_argument("--mesh", default="c5i", help="c5i (headline mesh), c4 (4 nested shells), c2 (bunny stand-in)")
ap.add_argument("--hash-rays", type=int, default=0, help="N incoherent hash rays instead of the pinhole image (streaming launch)")
a = ap.parse_args()
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
for kv in a.opt:
    k, v_ = kv.split("=", 1)
    hops.set_option(k, int(v_))
v, f = {"c5i": lambda: W.headline_mesh(8), "c4": lambda: W.nested_shells(7), "c2": W.bunny_standin}[a.mesh]()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
rad = float(np.linalg.norm(v, axis=1).max())
on, dn = W.pinhole_grid(a.res, a.res, distance=2.5 if a.mesh == "c4" else 2.5 * rad)
o, d = T(on), T(dn)
if a.hash_rays:
    o, d = W.hash_rays_torch(a.hash_rays, 1234, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
fn = {"closest": lambda: r.intersects_closest(o, d), "any": lambda: r.intersects_any(o, d),
      "count": lambda: r.intersects_count(o, d), "location": lambda: r.intersects_location(o, d)}[a.query]
for _ in range(a.warmup):
    fn()
torch.cuda.synchronize()
lib = ctypes.CDLL(hops.library_path())
susp = hasattr(lib, "tr_debug_susp")        # the suspend experiment build (-DTR_SUSPEND_EXP=G): reset its flag and counters
if susp:
    lib.tr_debug_susp(1, None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
susp_counts = None
if susp:
    sc = (ctypes.c_uint * 4)()
    lib.tr_debug_susp(0, sc)
    susp_counts = {"flag": int(sc[0]), "entries": int(sc[1]), "lanes": int(sc[2]), "waves": int(sc[3])}
nw = min(a.res * a.res // 64 + 4096, 131072)          # + the extra launch slots of split blocks
if a.hash_rays:
    nw = min((a.hash_rays + 63) // 64, 131072)
buf = np.zeros((nw, 4), np.uint64)
rc = lib.tr_debug_timeline(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(nw))
assert rc == 0, rc
st, en = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
ok = en > 0
st, en = st[ok], en[ok]
blk = (buf[ok, 3] & np.uint64(0x0fffffff)).astype(np.int64)
trips = ((buf[ok, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
hand = ((buf[ok, 3] >> np.uint64(48)) & np.uint64(0xffff)).astype(np.int64)
t0 = st.min()
st = (st - t0) / 100.0          # us
en = (en - t0) / 100.0
span = float(en.max())
dur = en - st
hw = (buf[ok, 2] >> np.uint64(32)).astype(np.int64)
xcc = (buf[ok, 2] & np.uint64(0xf)).astype(np.int64)
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cu_id = ((xcc * 8 + se) * 2 + sh) * 16 + cu
bins = np.arange(0, span + 10, 10.0)
mid = bins[:-1] + 5
resident = [(int(((st <= m) & (en > m)).sum())) for m in mid]
slots = a.waves_per_simd * 4 * 256
ucu = np.unique(cu_id)
per_cu_time = np.array([dur[cu_id == c].sum() for c in ucu])
last_end_cu = np.array([en[cu_id == c].max() for c in ucu])
q = lambda x, p: float(np.percentile(x, p))  # noqa: E731
out = {"mesh": a.mesh, "res": a.res, "query": a.query, "opts": a.opt, "event_ms": round(e0.elapsed_time(e1), 4),
       "waves": int(ok.sum()), "span_us": round(span, 1),
       "wave_us": {"mean": round(float(dur.mean()), 1), "p50": round(q(dur, 50), 1), "p90": round(q(dur, 90), 1),
                   "p99": round(q(dur, 99), 1), "max": round(float(dur.max()), 1)},
       "last_start_us": round(float(st.max()), 1),
       "start_pcts_us": {p: round(q(st, p), 1) for p in (50, 90, 99)},
       "occupied_frac_of_slots": round(float(dur.sum()) / (slots * span), 3),
       "resident_waves_per_10us": resident, "slots": slots,
       "cus_seen": int(len(ucu)),
       "cu_busy_us": {"min": round(float(per_cu_time.min()) / (a.waves_per_simd * 4), 1),
                      "mean": round(float(per_cu_time.mean()) / (a.waves_per_simd * 4), 1),
                      "max": round(float(per_cu_time.max()) / (a.waves_per_simd * 4), 1)},
       "cu_last_end_us": {"min": round(float(last_end_cu.min()), 1), "p50": round(q(last_end_cu, 50), 1),
                          "max": round(float(last_end_cu.max()), 1)},
       # the waves that end in the last 20 % of the span: how long were they, when did they start
       "tail_waves": {"n": int((en > 0.8 * span).sum()),
                      "dur_mean_us": round(float(dur[en > 0.8 * span].mean()), 1),
                      "start_mean_us": round(float(st[en > 0.8 * span].mean()), 1)}}
top = np.argsort(-dur)[:12]
out["top_waves"] = [{"us": round(float(dur[k]), 1), "start": round(float(st[k]), 1), "trips": int(trips[k]),
                     "handovers": int(hand[k]), "first_ray": int(blk[k]) * 128} for k in top]
if trips.max() > 0:
    out["us_per_trip"] = {"all": round(float(dur.sum() / max(1, trips.sum())), 3),
                          "top12": round(float(dur[top].sum() / max(1, trips[top].sum())), 3)}
    out["trips"] = {"mean": round(float(trips.mean()), 1), "p99": round(q(trips, 99), 1), "max": int(trips.max())}
if susp_counts is not None:
    out["suspended"] = susp_counts
print(json.dumps(out))
