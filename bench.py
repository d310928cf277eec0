#!/usr/bin/env python3
"""Headline benchmark: Mrays/s closest-hit, 1M-tri mesh, 1024^2 ray batch (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one `RayMeshIntersector.intersects_closest` call (the reference's timed call,
test/performance_test.py:54-57) over one 1024x1024 pinhole ray batch against the 1 310 720-
triangle headline mesh; rays and BVH are resident in HBM before the
timed region.  With N > 1 every rank owns a BVH replica and its own 1024^2-ray shard (weak
scaling; the path has no exchange step -- `--gather` adds the RCCL gather of the results to
rank 0 to the timed region).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "trimesh-ray-optix_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s spec (6.29 TB/s achievable)
BYTES_PER_RAY_CLOSEST = 50      # 24 B in + 26 B out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--subdiv", type=int, default=8, help="icosphere subdivisions (8 = 1 310 720 tris)")
    ap.add_argument("--res", type=int, default=1024, help="ray grid is res x res")
    ap.add_argument("--rays", choices=["pinhole", "hash"], default="pinhole")
    ap.add_argument("--gather", action="store_true", help="gather results to rank 0 inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--persistent", type=int, default=None)
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (tr_set_option)")
    ap.add_argument("--stats", action="store_true", help="also print traversal counters (diagnostic kernel)")
    return ap.parse_args()


def cpu_baseline(v, f, o, d, budget_s=20.0):
    """The oracle's BVH mode ("port": our CPU restatement, NOT Embree -- trimesh/pyembree are
    not installed in this image) on the host cores, same mesh, same rays, bounded time."""
    from oracle.oracle import OracleIntersector, num_threads
    try:   # trimesh + Embree if the GPU box happens to have them
        import trimesh  # noqa: F401
        import embreex  # noqa: F401
        have_embree = True
    except Exception:
        have_embree = False
    if have_embree:
        try:
            import trimesh
            m = trimesh.Trimesh(vertices=v, faces=f, process=False)
            oo = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
            dd = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
            t0 = time.perf_counter()
            m.ray.intersects_location(oo, dd, multiple_hits=False)   # test/performance_test.py:75
            el = time.perf_counter() - t0
            return {"value": round(len(oo) / el / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "reference",
                    "sample": f"trimesh+embree mesh.ray.intersects_location(multiple_hits=False), {len(oo)} rays, 1 pass"}
        except Exception:
            pass
    R = OracleIntersector(v, f, mode=1)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    R.intersects_first(o[:4096], d[:4096])   # page in
    passes, t0 = 0, time.perf_counter()
    while True:
        R.closest_raw(o, d)
        passes += 1
        el = time.perf_counter() - t0
        if el > budget_s or passes >= 20:
            break
    return {"value": round(len(o) * passes / el / 1e6, 3), "unit": "Mrays/s", "cores": num_threads(),
            "kind": "port",
            "sample": f"full {len(o)}-ray batch x {passes} passes, oracle median-split BVH + contract "
                      f"arithmetic, OpenMP over {num_threads()} host threads"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # launched by torch.distributed.run
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import triro.backend.ops as hops
    from triro.ray.ray_optix import RayMeshIntersector
    if args.persistent is not None:
        hops.set_option("persistent", args.persistent)
    if args.blocks_per_cu is not None:
        hops.set_option("blocks_per_cu", args.blocks_per_cu)
    for kv in args.opt:
        k, v_ = kv.split("=", 1)
        hops.set_option(k, int(v_))

    # ---- workload (synthetic, deterministic) -------------------------------------------------
    v, f = W.headline_mesh(args.subdiv)
    rad = float(np.linalg.norm(v, axis=1).max())
    n = args.res * args.res
    if args.rays == "pinhole":
        o_np, d_np = W.pinhole_grid(args.res, args.res, distance=2.5 * rad)
        # every rank traces its own shard: same camera, rolled by `rank` rows so shards differ
        o_np = np.ascontiguousarray(o_np)
        d_np = np.roll(d_np, rank * 7, axis=0)
        origins = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev)
        dirs = torch.from_numpy(np.ascontiguousarray(d_np)).to(dev)
    else:
        lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
        origins, dirs = W.hash_rays_torch(n, 99, lo, hi, start=rank * n, device=dev)
        o_np, d_np = None, None
    vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = RayMeshIntersector(vertices=vt, faces=ft)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    info = r.bvh_info()

    gather_bufs = None
    if dist_on and args.gather:
        import torch.distributed as dist
        shapes = [((n,), torch.uint8), ((n,), torch.uint8), ((n,), torch.int32), ((n, 3), torch.float32),
                  ((n, 2), torch.float32)]
        if rank == 0:
            gather_bufs = [[torch.empty(s, dtype=t, device=dev) for _ in range(world)] for s, t in shapes]

    lead = origins.dim() - 1

    def flat(x):
        x = x.contiguous()
        if x.dtype == torch.bool:
            x = x.view(torch.uint8)
        return x.reshape(n, *x.shape[lead:])

    def step():
        out = r.intersects_closest(origins, dirs)
        if dist_on and args.gather:
            import torch.distributed as dist
            for k, x in enumerate(out):
                dist.gather(flat(x), gather_bufs[k] if rank == 0 else None, dst=0)
        return out

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    t_first = time.perf_counter()
    out = step()                      # very first call: no learned launch order yet
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t_first) * 1e3
    for _ in range(args.warmup):
        out = step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    import gc
    gc.collect()
    gc.disable()            # a step is ~0.3 ms: keep collector pauses out of the timed region
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        out = step()
        ev[k][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    if os.environ.get("TRIRO_BENCH_TRACE") and rank == 0:   # per-step durations, launch order
        print("per-step ms:", " ".join(f"{x:.3f}" for x in kernel_ms), file=sys.stderr)
    kernel_ms.sort()
    kernel_avg_ms = float(np.mean(kernel_ms))

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist_on:
        import torch.distributed as dist
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        total_rays = n * world * args.steps
        value = total_rays / elapsed / 1e6
        bvh_bytes = info["node_bytes"] + info["tri_bytes"]
        algo_bytes = n * BYTES_PER_RAY_CLOSEST + bvh_bytes
        achieved = algo_bytes / (kernel_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("closest_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "Mrays/s closest-hit, 1M-tri mesh, 1024^2 ray batch",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"C5(i): icosphere({args.subdiv})+displacement seed 0, {len(f)} tris; "
                                   f"{args.res}x{args.res} {args.rays} rays per GPU; intersects_closest "
                                   f"(stream_compaction=False)",
                       "rays_per_gpu": n, "triangles": int(len(f)),
                       "parallelism": f"ray-sharded x{world}, BVH replicated" + (", results gathered to rank 0" if (dist_on and args.gather) else ""),
                       "bvh_depth": info["depth"], "bvh_bytes": int(bvh_bytes), "bvh_build_ms": round(build_ms, 2),
                       "hit_fraction": round(float(out[0].float().mean().item()), 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "kernel": "k_query_direct<CLOSEST>", "kernel_avg_ms": round(kernel_avg_ms, 4),
                         "kernel_min_ms": round(kernel_ms[0], 4), "first_call_ms": round(first_call_ms, 4),
                         "algorithmic_bytes": int(algo_bytes),
                         "note": "50 B/ray compulsory I/O + one read of the BVH arena per launch; the path "
                                 "is cache-latency/divergence bound, not HBM-bandwidth bound (DESIGN.md)"},
        }
        if args.stats:
            st = hops.trace_stats_closest(r.as_wrapper, origins, dirs)
            res["trace_stats"] = {k: (v_ / st["rays"] if k != "rays" else v_) for k, v_ in st.items()}
        if world == 1 and not args.no_cpu_baseline and o_np is not None:
            res["cpu_baseline"] = cpu_baseline(v, f, o_np, d_np)
        print(json.dumps(res), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
