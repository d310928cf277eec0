// traverse.hip -- query kernels and their C-ABI launchers (replace the five OptiX pipelines
// of triro/backend/shaders.cu:67-246 and the launch wrappers of ray.cpp:161-378).
//
// One ray per lane, wave64; the per-ray state machine is tr_fused_step (tr_bvh.h): stackless
// trail + LDS far-child ring, one node visit and one queued leaf test per trip.  Launch shapes:
//   direct     : grid = ceil(n/BS) workgroups of BS = 64/128/256 rays (default, k_query_direct).
//                Which ray block a workgroup takes is a scheduling choice: the measured
//                per-XCD cost order of the previous launch (k_sched_sort), else the
//                XCD-chunked, scrambled static map.
//   persistent : grid = CUs x blocks_per_cu; each wave pulls 64-ray batches from a global
//                work counter (option, not faster at the measured sizes).  The per-lane refill
//                variant of round 1 ("active-ray repacking": 0.8 vs 3.6 Grays/s) was removed in
//                round 2 after the randomised sweep found a mismatch in it.
// Kernel parameters travel by value (no per-call malloc/memcpy/free as in ray.cpp:279-287).
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "tr_internal.h"

namespace {

#include "kernels_common.inc"
#include "kernels_direct.inc"
#include "kernels_stream.inc"
#include "kernels_lists.inc"
#include "kernels_expand.inc"
#include "launch_policy.inc"

}  // namespace

void tr_wide_rebuild(tr_bvh* bvh, hipStream_t stream) {
    if (bvh && bvh->wnodes && !bvh->wide_valid && bvh->num_tris >= 2) (void)ensure_wide(bvh, stream);
}

#include "abi.inc"


#ifdef TR_USTEAL_DEBUG
extern "C" int tr_debug_usteal(unsigned* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_usteal_debug), 16);
}
#endif
#ifdef TR_TIMELINE
extern "C" int tr_debug_timeline(unsigned long long* host_out, long long n_waves) {
    if (n_waves > TR_TIMELINE) n_waves = TR_TIMELINE;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline), (size_t)n_waves * 32);
}
#endif
#ifdef TR_DEBUG_SCHED
// experiment builds only (not part of the ABI): the scheduling buffer of (handle, stream, class) -> host
extern "C" int tr_debug_sched(tr_bvh* bvh, void* stream, int cls, uint32_t* host_out, long long words) {
    for (int k = 0; k < TR_SCHED_SLOTS; k++)
        if (bvh->sched[k].used && bvh->sched[k].stream == (hipStream_t)stream && bvh->sched[k].cls == cls) {
            (void)hipStreamSynchronize((hipStream_t)stream);
            return (int)hipMemcpy(host_out, bvh->sched[k].buf, sizeof(uint32_t) * (size_t)words, hipMemcpyDeviceToHost);
        }
    return -1;
}
#endif

