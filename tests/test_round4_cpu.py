"""Round 4 CPU tests: resources of the shipped kernels, the CPU baseline's parallel efficiency."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_no_query_kernel_of_the_shipped_library_spills_inside_its_trips():
    """VERDICT r03 #7a, restated in round 6.  The traversal kernels ask for a wave per SIMD more than the compiler's own
    budget gives them -- the streaming launches five (96 registers: at six they spill inside the trip once the float64
    call is in the kernel, profiles/r06_ab_stream_waves.txt), the direct launches six (80), where the float64 part of
    the hit predicate is a real call at the end of a trip (tr_drain_exact: the callee owns v0-v31).  What that costs is
    pinned here, read off the shipped ISA: the budgets; a bound on the spilled registers; NO scratch instruction inside
    the traversal trips of any launch family except the save / restore around that call (a cold branch: 0.6 % of the
    leaf tests); and, for the kernels of hierarchies deeper than 32 levels, no scratch instruction anywhere else at all
    (the pattern that faulted: profiles/r06_deep_fault_probe.txt)."""
    import code_object_notes as con
    sys.path.insert(0, os.path.join(ROOT, "scripts", "round5"))
    import isa_loops
    so = os.path.join(ROOT, "trimesh-ray-optix_amd", "lib", "libtriro_hip.so")
    if not os.path.exists(so) or not os.path.exists(con.READELF):
        pytest.skip("library not built / llvm-readelf not available")
    ks = [k for k in con.kernels(so) if "k_query" in k["name"]]
    assert len(ks) > 50
    stream = [k for k in ks if k["name"].startswith("void k_query_stream<")]
    wide = [k for k in ks if k["name"].startswith("void k_query_wide<")]
    direct = [k for k in ks if k not in stream and k not in wide and "_stats" not in k["name"] and "_wide" not in k["name"]]
    # (<Q, COMPACT, BS, DEEP>: five waves per SIMD since round 6 -- 96 registers; four for the instantiations with 64-bit addressing)
    assert stream and all(k["vgpr"] <= (96 if ", true, 128," in k["name"] else 128) and k["vgpr_spill"] <= 8 for k in stream), [(k["name"], k["vgpr"], k["vgpr_spill"]) for k in stream]
    assert wide and all(k["vgpr"] <= 96 and k["vgpr_spill"] <= 16 for k in wide), [(k["name"], k["vgpr"], k["vgpr_spill"]) for k in wide]
    bad = [(k["name"], k["vgpr"], k["vgpr_spill"]) for k in direct if k["vgpr_spill"] > (12 if "count_steal" in k["name"] else 8)]
    assert not bad, bad
    # <Q, STATS, COMPACT, MODE, DEEP, QN>: the instantiations for hierarchies of more than 32 levels (64-bit trail words) are
    # the ones that died with a memory access fault when KERNEL-LIFETIME values were forced into scratch (stored in the
    # prologue, reloaded in the stealing loop's hand-over code: profiles/r06_deep_spill_fault.txt).  They may
    # save registers around the float64 call (the cold drain branch) and nothing else: every scratch instruction of a DEEP
    # kernel sits within a few dozen instructions of an s_swappc.
    import re
    deep = [k for k in direct if re.search(r"k_query_direct<\d, \w+, \w+, \d, true, \w+>", k["name"]) or re.search(r"k_query_direct_sort<\d, true", k["name"])
            or re.search(r"k_query_count_steal<\w+, true>", k["name"]) or re.search(r"k_query_count_steal_sort<true", k["name"])]      # (<COMPACT, DEEP> / _sort<DEEP, COMPACT>)
    assert len(deep) >= 10, [k["name"] for k in deep]
    checked = 0
    for want in ("k_query_direct<", "k_query_direct_sort<", "k_query_count_steal"):
        for name, lines in isa_loops.disassemble(so, want):
            # (64-bit trail words: the DEEP instantiations and the ones with 64-bit addressing -- <.., COMPACT = false, ..>;
            # the 32-bit ones have run with kernel-lifetime spills since round 5 and pass everything: bounded, not forbidden)
            wide_word = any(name.startswith(k["name"].split("(")[0]) or k["name"].startswith(name) for k in deep) or re.search(r"k_query_direct<\d, \w+, false,", name) \
                or re.search(r"k_query_count_steal<false", name)
            checked += 1
            ins = isa_loops.instructions(lines)
            calls = [i for i, (_, t) in enumerate(ins) if t.startswith("s_swappc_b64")]
            far = [t for i, (_, t) in enumerate(ins) if t.startswith("scratch_") and not any(abs(i - c) <= 64 for c in calls)]
            # (the 32-bit count launches run at seven waves per SIMD -- 72 registers, 8-10 kernel-lifetime spills: a dozen scratch instructions)
            assert len(far) <= (0 if wide_word else (16 if "count_steal" in name else 8)), (name, far)
    assert checked >= 40, checked
    # (round 6, tr_drain_exact<COLD>: what the float64 call clobbers is saved AROUND the call, on the 0.6 % of leaf tests
    # that reach it -- those stores / loads sit within a few dozen instructions of the s_swappc and are not "in the trips")
    for want, kw in (("k_query_stream<", dict(near_call=64)), ("k_query_direct<", dict(near_call=64)), ("k_query_direct_sort<", dict(near_call=64)),
                     ("k_query_count_steal", dict(near_call=64)),
                     # (the 8-bit planes' decode marks the trips of the 8-wide walk; a visit -- the root's -- also sits in the
                     # refill path: the SMALLEST loop around a decode is the trip loop)
                     ("k_query_wide<", dict(marker="v_cvt_f32_ubyte", smallest_only=True, near_call=64))):
        inside = isa_loops.scratch_in_trip_loops(so, want, **kw)
        assert inside and not any(inside.values()), {k: v for k, v in inside.items() if v}
    # the stealing closest launch of the headline: six waves per SIMD (80 registers)
    headline = [k for k in ks if k["name"].startswith("void k_query_direct<2, false, true, 1, false, true>")]
    assert len(headline) == 1 and headline[0]["vgpr"] <= 80 and headline[0]["vgpr_spill"] <= 8, headline


def test_cpu_baseline_scales_with_threads():
    """the cpu_baseline leg times only the C entry point; N threads must buy at least 0.5 x N
    over one thread on this container's cores (round 3's figure was serial-dominated: 1.3x from 128 threads)"""
    import workloads as W
    from oracle.oracle import OracleIntersector, usable_cpus, num_threads
    ncpu = min(usable_cpus(), num_threads())
    if ncpu < 2:
        pytest.skip("one usable CPU")
    v, f = W.headline_mesh(6)
    o, d = W.pinhole_grid(1024, 512, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    o, d = np.ascontiguousarray(o).reshape(-1, 3), d.reshape(-1, 3)
    R = OracleIntersector(v, f, mode=1)
    R.closest_timed(o, d, ncpu, passes=2)          # wake the thread team, warm the caches
    best = 0.0
    for _ in range(3):      # best of three: the container's cores are shared
        t1 = R.closest_timed(o[:1 << 17], d[:1 << 17], 1) * (len(o) / (1 << 17))
        R.closest_timed(o, d, ncpu)
        tn = R.closest_timed(o, d, ncpu, passes=3) / 3
        best = max(best, t1 / tn / ncpu)
        if best >= 0.5:
            break
    assert best >= 0.5, f"parallel efficiency {best:.2f} on {ncpu} threads"
