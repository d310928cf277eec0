// traverse.hip -- the query kernels and their C-ABI launchers: ONE translation unit assembled from the .inc files below
// (replaces the five OptiX pipelines of triro/backend/shaders.cu:67-246 and the launch wrappers of ray.cpp:161-378).
//
//   kernels_common.inc   RayFetch / QueryOut (the per-call ABI of the kernels, LaunchParams.h:11-47), the strided ray
//                        fetch (shaders.cu:27-63), write_result (shaders.cu:120-172)
//   kernels_direct.inc   the DIRECT launch: one workgroup of 128 rays (two waves), one ray per lane; plain, with
//                        intra-wave work stealing, the unordered two-phase schedule (count / location) with and without
//                        stealing; the block -> ray map (XCD chunks, 8x8 tiles, learned order, split slots);
//                        k_query_direct, k_query_direct_sort (carries the deferred sort of the measured block costs),
//                        k_query_count_steal[_sort], k_query_direct_wide, k_sched_sort / k_sched_rescale
//   traverse_wide.inc    the walk over the 8-wide compressed nodes (tr_wide.h): k_query_wide (persistent waves, refill)
//   kernels_stream.inc   the STREAMING launch for incoherent batches: persistent waves, ranges from a work counter,
//                        __ballot / mbcnt refill of idle lanes; k_probe_coherence picks between the shapes on the device
//   kernels_lists.inc    multi-hit lists (k_location, k_fill_list), scans, stream compaction
//   kernels_expand.inc   12-byte / 4-byte closest-hit records -> the five dense outputs (the sharded gather's far end)
//   launch_policy.inc    which launch a batch gets: every threshold in one table (tr_policy) with the profile that set it
//   abi.inc              the extern "C" entry points of include/triro_hip.h
//
// The per-lane state machine (stackless trail + LDS far-child ring, one node visit and one queued leaf test per trip, the
// deferred float64 leaf decision) is tr_bvh.h; the arithmetic contract is tr_math.h.  Kernel parameters travel by value
// (no per-call malloc / memcpy / free as in ray.cpp:279-287); the mesh's grid frame is re-read from device memory
// (TR_VIEW_LIVE) so that a launch captured in a HIP graph follows a later refit.
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

