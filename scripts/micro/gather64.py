#!/usr/bin/env python3
"""Random 64-byte gather ceiling over a table the size of the headline BVH.
Builds scripts/micro/gather64.hip with hipcc if needed.  usage (GPU box): python scripts/micro/gather64.py"""
import ctypes as C, json, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libgather64.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC",
                           os.path.join(HERE, "gather64.hip"), "-o", so])
lib = C.CDLL(so)
dev = torch.device("cuda:0")
nrows = 146_800_576 // 64                       # the headline arena: 147 MB of 64-B records
table = torch.empty(nrows, 16, dtype=torch.float32, device=dev).uniform_()
g = torch.Generator(device=dev); g.manual_seed(1)
nxt = torch.randint(0, nrows, (nrows,), device=dev, dtype=torch.int32, generator=g)
table.view(torch.int32)[:, 15] = nxt            # record.w of the 4th quarter = next index (dependent walk)
res = []
for waves_per_simd in (0, 1, 2, 4, 8):             # 0 = ONE workgroup on the whole GPU (unloaded latency)
    nthreads = 256 * 4 * 64 * waves_per_simd if waves_per_simd else 256
    idx = torch.randint(0, nrows, (nthreads,), device=dev, dtype=torch.int32, generator=g)
    out = torch.empty(nthreads, device=dev)
    for dep, steps in ((1, 256), (0, 256)):
        s = torch.cuda.current_stream().cuda_stream
        def run():
            rc = lib.gather64_run(C.c_void_p(table.data_ptr()), C.c_uint32(nrows), C.c_void_p(idx.data_ptr()),
                                  C.c_int64(nthreads), C.c_int(steps), C.c_int(dep), C.c_void_p(out.data_ptr()), C.c_void_p(s))
            assert rc == 0
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        nbytes = nthreads * steps * 64
        res.append({"waves_per_simd": waves_per_simd, "dependent": bool(dep), "ms": round(ms, 4),
                    "GBps": round(nbytes / ms / 1e6, 1), "ns_per_step": round(ms * 1e6 / steps, 1)})
        print(json.dumps(res[-1]), flush=True)
