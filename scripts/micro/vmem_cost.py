#!/usr/bin/env python3
"""Cost of one gather wave-instruction in the CU texture path vs bytes per lane and active lanes
(scripts/micro/vmem_cost.hip).  usage (GPU box): python scripts/micro/vmem_cost.py > gpurun_out/vmem_cost.jsonl"""
import ctypes as C, json, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libvmem_cost.so")
src = os.path.join(HERE, "vmem_cost.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", src, "-o", so])
lib = C.CDLL(so)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
CLK = 2.4e9


def run(table_bytes, stride, nq, nd, ns, waves_per_simd, lane_mask=(1 << 64) - 1, steps=512, share=1):
    """share = number of neighbouring lanes that walk the SAME chain (same address every step):
    64 // share distinct cache lines per wave-instruction"""
    nrec = table_bytes // stride
    table = torch.empty(nrec * stride // 4, dtype=torch.float32, device=dev).uniform_()
    nxt = torch.randint(0, nrec, (nrec,), device=dev, dtype=torch.int32, generator=g)
    table.view(torch.int32)[:: stride // 4] = nxt           # first dword of every record = next index
    nthreads = 256 * 4 * 64 * waves_per_simd
    idx = torch.randint(0, nrec, (nthreads,), device=dev, dtype=torch.int32, generator=g)
    if share > 1:
        idx = idx.reshape(-1, share)[:, :1].expand(-1, share).reshape(-1).contiguous()
    out = torch.empty(nthreads, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def go():
        rc = lib.vmem_cost_run(C.c_void_p(table.data_ptr()), C.c_uint32(nrec), C.c_uint32(stride), C.c_void_p(idx.data_ptr()),
                               C.c_int64(nthreads), C.c_int(steps), C.c_int(nq), C.c_int(nd), C.c_int(ns),
                               C.c_uint64(lane_mask), C.c_void_p(out.data_ptr()), C.c_void_p(s))
        assert rc == 0, rc
    go(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    waves_per_cu = 4 * waves_per_simd
    # CU-cycles per wave-step (all waves of a CU share its TA/TD): time * clk / (steps * waves on the CU)
    cyc = best * 1e-3 * CLK / (steps * waves_per_cu)
    active = bin(lane_mask).count("1")
    rec = {"distinct_lines_per_instr": 64 // share, "table_KiB": table_bytes // 1024, "stride": stride, "loads": f"{nq}x16+{nd}x8+{ns}x4", "waves_per_simd": waves_per_simd,
           "active_lanes": active, "ms": round(best, 4), "cu_cycles_per_wave_step@2.4GHz": round(cyc, 1),
           "G_lane_steps_per_s": round(nthreads * (active / 64) * steps / best / 1e6, 1)}
    print(json.dumps(rec), flush=True)


FULL = (1 << 64) - 1
# per-instruction floor by width: ONE active lane per wave (and 4 instructions per step so that the
# chain latency does not dominate): dwordx4 vs dwordx2
if len(sys.argv) > 1 and sys.argv[1] == "share":
    # coherence: 64 active lanes, 64 / share distinct lines per wave-instruction (the traversal's case:
    # neighbouring rays visit the same node)
    for tb in (16 << 10, 2 << 20):
        for share in (64, 32, 16, 8, 4, 2, 1):
            run(tb, 64, 4, 0, 0, 7, share=share)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "floor":
    for tb in (16 << 10,):
        for mask in (0x1, 0x0001000100010001, FULL):
            run(tb, 64, 4, 0, 0, 7, mask)      # 4 x dwordx4
            run(tb, 64, 0, 4, 0, 7, mask)      # 4 x dwordx2
            run(tb, 64, 2, 0, 0, 7, mask)      # 2 x dwordx4
            run(tb, 64, 0, 2, 0, 7, mask)      # 2 x dwordx2
            run(tb, 64, 1, 0, 0, 7, mask)
            run(tb, 64, 0, 1, 0, 7, mask)
    sys.exit(0)
for tb in (16 << 10, 2 << 20, 64 << 20):                      # L1-resident, L2-resident, Infinity-Cache-resident
    for (nq, nd, ns, stride) in ((4, 0, 0, 64), (3, 0, 0, 64), (2, 0, 0, 64), (2, 0, 0, 32), (1, 0, 0, 64), (1, 0, 0, 16), (0, 1, 0, 64), (0, 0, 1, 64),
                                 (0, 2, 0, 64), (0, 4, 0, 64), (2, 0, 1, 64), (7, 0, 0, 128), (4, 1, 1, 128)):
        run(tb, stride, nq, nd, ns, 7)
# active-lane dependence (64-B records, L1-resident and L2-resident)
for tb in (16 << 10, 2 << 20):
    for mask in (FULL, 0xFFFFFFFF, 0x5555555555555555, 0x1111111111111111, 0x000000000000FFFF, 0x0001000100010001, 0xF, 0x1):
        run(tb, 64, 4, 0, 0, 7, mask)
# occupancy dependence
for w in (1, 2, 4, 7, 8):
    run(2 << 20, 64, 4, 0, 0, w)
    run(2 << 20, 32, 2, 0, 0, w)
