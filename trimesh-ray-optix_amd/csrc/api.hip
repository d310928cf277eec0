// api.hip -- runtime bring-up, error reporting and acceleration-structure entry points of the
// C ABI (include/triro_hip.h).  Replaces triro/backend/base.cpp (global OptiX context, module,
// pipelines, SBTs: base.cpp:15-157) -- none of those concepts survive; what remains is a
// per-device table {CU count, work-counter ring, builder temporaries} created on first use.
#include <atomic>
#include <mutex>
#include <string.h>

#include "tr_internal.h"

namespace {
thread_local std::string g_last_error;
std::mutex g_mutex;
constexpr int TR_MAX_DEVICES = 64;
tr_device_state g_devices[TR_MAX_DEVICES];

// option table: name, field, accepted range (0/1 options are normalised with != 0)
struct opt_desc {
    const char* name;
    int tr_options::*field;
    int64_t lo, hi;
    bool boolean;
};
const opt_desc OPTS[] = {
    {"adaptive", &tr_options::adaptive, 0, 1, true},
    {"compact", &tr_options::compact, 0, 1, true},
    {"xcd_chunk", &tr_options::xcd_chunk, 0, 65536, false},
    {"steal", &tr_options::steal, 0, 4096, false},
    {"tile", &tr_options::tile, 0, 2, false},
    {"tile_small", &tr_options::tile_small, 0, 4, false},
    {"build_cache", &tr_options::build_cache, 0, 1, true},
    {"node_layout", &tr_options::node_layout, 0, 1, true},
    {"stream", &tr_options::stream, 0, 2, false},
    {"stream_rays", &tr_options::stream_rays, 64, 1 << 20, false},
    {"stream_refill", &tr_options::stream_refill, 0, 64, false},
    {"stream_dynamic", &tr_options::stream_dynamic, 0, 1, false},
    {"leaf_vote", &tr_options::leaf_vote, 1, 64, false},
    {"grid_nodes", &tr_options::grid_nodes, 0, 2, false},
    {"split", &tr_options::split, 0, 12, false},
    {"split_steal", &tr_options::split_steal, 0, 4096, false},
    {"split_outlier", &tr_options::split_outlier, 0, 1024, false},
    {"usteal", &tr_options::usteal, 0, 4095, false},
    {"split_floor", &tr_options::split_floor, 0, 100000, false},
    {"expand_cus", &tr_options::expand_cus, 0, 64, false},
    {"expand_tiles", &tr_options::expand_tiles, 0, 1, true},
    {"order_transfer", &tr_options::order_transfer, 0, 1, true},
    {"sort_inline", &tr_options::sort_inline, 0, 1, true},
    {"wide", &tr_options::wide, 0, 2, false},
    {"wide_stack", &tr_options::wide_stack, 1, 12, false},
    {"wide_direct", &tr_options::wide_direct, 0, 3, false},
};
constexpr int NUM_OPTS = (int)(sizeof(OPTS) / sizeof(OPTS[0]));
struct opt_store {
    std::atomic<int> v[NUM_OPTS];
    opt_store() {
        const tr_options d;
        for (int k = 0; k < NUM_OPTS; k++) v[k].store(d.*(OPTS[k].field), std::memory_order_relaxed);
    }
};
opt_store g_opts;

// exact hash of the triangle arena (tr_bvh_replica_hash): per slot a 64-bit mix of its index and the words of its
// record, summed (order-free: any reduction order gives the same value)
__device__ __forceinline__ unsigned long long tr_mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_replica_hash(const uint32_t* __restrict__ words, int64_t num_tris,
                                                      unsigned long long* __restrict__ out) {
    constexpr int W = (int)(sizeof(tr_tri) / 4);
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < num_tris; i += (int64_t)gridDim.x * 256) {
        unsigned long long h = tr_mix64((unsigned long long)i + 0x9e3779b97f4a7c15ull);
        const uint32_t* p = words + i * W;
#pragma unroll
        for (int k = 0; k < W; k++) h = tr_mix64(h ^ ((unsigned long long)p[k] + ((unsigned long long)(k + 1) << 32)));
        acc += h;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

}  // namespace

void tr_set_error(const std::string& msg) { g_last_error = msg; }
int tr_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
tr_options tr_opts() {
    tr_options o;
    for (int k = 0; k < NUM_OPTS; k++) o.*(OPTS[k].field) = g_opts.v[k].load(std::memory_order_relaxed);
    return o;
}

int tr_get_device_state(int device, tr_device_state** out) {
    if (device < 0 || device >= TR_MAX_DEVICES) return tr_fail(TR_ERR_INVALID_ARG, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_mutex);
    tr_device_state& st = g_devices[device];
    if (!st.ready) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return tr_fail(TR_ERR_NO_DEVICE, "no HIP device available (libtriro_hip has no CPU fallback)");
        if (device >= count) return tr_fail(TR_ERR_NO_DEVICE, "device ordinal >= device count");
        tr_device_guard g;
        if (g.enter(device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
        hipDeviceProp_t prop;
        TR_HIP_TRY(hipGetDeviceProperties(&prop, device));
        st.device = device;
        st.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        {
            int nx = 0;
            if (hipDeviceGetAttribute(&nx, hipDeviceAttributeNumberOfXccs, device) != hipSuccess || nx < 1) { (void)hipGetLastError(); nx = 8; }
            st.num_xcd = nx;
            st.l2_bytes = prop.l2CacheSize > 0 ? (int64_t)prop.l2CacheSize : 0;
        }
        TR_HIP_TRY(hipMalloc((void**)&st.counters, sizeof(unsigned long long) * TR_NUM_COUNTERS));
        TR_HIP_TRY(hipMemset(st.counters, 0, sizeof(unsigned long long) * TR_NUM_COUNTERS));
        TR_HIP_TRY(hipStreamCreateWithFlags(&st.build_side, hipStreamNonBlocking));
        TR_HIP_TRY(hipEventCreateWithFlags(&st.build_fork, hipEventDisableTiming));
        TR_HIP_TRY(hipEventCreateWithFlags(&st.build_join, hipEventDisableTiming));
        st.ready = true;
    }
    *out = &st;
    return TR_OK;
}

int tr_build_temp_acquire(tr_device_state* st, size_t bytes, void** out) {
    st->build_mutex.lock();
    if (st->build_temp_bytes < bytes) {
        hipError_t e = hipSuccess;
        if (st->build_temp) { e = hipFree(st->build_temp); st->build_temp = nullptr; st->build_temp_bytes = 0; }
        if (e == hipSuccess) e = hipMalloc(&st->build_temp, bytes);
        if (e != hipSuccess) {
            st->build_temp = nullptr;
            st->build_mutex.unlock();
            return tr_fail(e == hipErrorOutOfMemory ? TR_ERR_OUT_OF_MEMORY : TR_ERR_HIP,
                           std::string("builder temporaries: ") + hipGetErrorName(e));
        }
        st->build_temp_bytes = bytes;
    }
    *out = st->build_temp;
    return TR_OK;
}

int tr_build_temp_release(tr_device_state* st) {
    int status = TR_OK;
    if (!tr_opts().build_cache && st->build_temp) {
        if (hipFree(st->build_temp) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(builder temporaries)");
        st->build_temp = nullptr;
        st->build_temp_bytes = 0;
    }
    st->build_mutex.unlock();
    return status;
}

extern "C" {

int tr_abi_version(void) { return TR_ABI_VERSION; }

const char* tr_last_error(void) { return g_last_error.c_str(); }

int tr_init(int device) {
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess)
            return tr_fail(TR_ERR_NO_DEVICE, "no HIP device available (libtriro_hip has no CPU fallback)");
    }
    tr_device_state* st;
    return tr_get_device_state(device, &st);
}

int tr_bvh_build(const float* d_vertices, int64_t nv, const int32_t* d_faces, int64_t nf,
                 void* stream, tr_bvh** out) {
    if (!out) return tr_fail(TR_ERR_INVALID_ARG, "out == NULL");
    *out = nullptr;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return tr_fail(TR_ERR_NO_DEVICE, "no HIP device available (libtriro_hip has no CPU fallback)");
    if (nf > 0 && d_vertices) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, d_vertices) == hipSuccess && attr.type == hipMemoryTypeDevice)
            device = attr.device;
        else (void)hipGetLastError();
    }
    tr_device_state* st;
    TR_TRY(tr_get_device_state(device, &st));
    tr_device_guard g;
    if (g.enter(device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    tr_bvh* bvh = new (std::nothrow) tr_bvh();
    if (!bvh) return tr_fail(TR_ERR_OUT_OF_MEMORY, "host allocation failed");
    bvh->device = device;
    bvh->sched_mutex = new (std::nothrow) std::mutex();
    int s = tr_build_impl(bvh, d_vertices, nv, d_faces, nf, (hipStream_t)stream);
    if (s != TR_OK) {
        if (bvh->arena) (void)hipFree(bvh->arena);
        delete bvh->sched_mutex;
        delete bvh;
        return s;
    }
    *out = bvh;
    return TR_OK;
}

int tr_bvh_update(tr_bvh* bvh, const float* d_vertices, int64_t nv, const int32_t* d_faces,
                  int64_t nf, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    tr_device_guard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    const int s = tr_build_impl(bvh, d_vertices, nv, d_faces, nf, (hipStream_t)stream);
    // a handle that has walked 8-wide nodes gets them rebuilt at once, in place where they fit: a HIP graph that
    // captured a wide launch keeps reading current geometry (as it does with the arena's nodes)
    if (s == TR_OK) tr_wide_rebuild(bvh, (hipStream_t)stream);
    return s;
}

int tr_bvh_refit(tr_bvh* bvh, const float* d_vertices, int64_t nv, const int32_t* d_faces,
                 int64_t nf, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    tr_device_guard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    const int s = tr_refit_impl(bvh, d_vertices, nv, d_faces, nf, (hipStream_t)stream);
    if (s == TR_OK) tr_wide_rebuild(bvh, (hipStream_t)stream);       // (same topology: same records, new boxes, same buffer)
    return s;
}

// ---- (de)serialisation: header + the arena, byte for byte -------------------------------------
namespace {
struct tr_blob_header {
    char magic[8];          // "TRBVH\0\0\2"
    int64_t num_tris, num_nodes, arena_bytes;
    int32_t depth, key_mode;
    float aabb_min[3], aabb_max[3];
    uint32_t sizeof_node, sizeof_tri, sizeof_link, pad;
};
const char TR_MAGIC[8] = {'T', 'R', 'B', 'V', 'H', 0, 0, 4};   // 4: arena = nodes | links | tris (with the scale of contract 3's inside test) | 32-byte grid nodes
}  // namespace

int64_t tr_bvh_serialized_size(const tr_bvh* bvh) {
    if (!bvh) return -1;
    // only the bytes the current mesh uses: after update_raw() to a smaller mesh the arena keeps
    // its larger capacity, which is not part of the hierarchy
    return (int64_t)sizeof(tr_blob_header) + (bvh->num_tris > 0 ? tr_arena_used_bytes(bvh->num_tris) : 0);
}

int tr_bvh_serialize(const tr_bvh* bvh, void* h_buffer, int64_t size, void* stream) {
    if (!bvh || !h_buffer) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    if (size < tr_bvh_serialized_size(bvh)) return tr_fail(TR_ERR_INVALID_ARG, "buffer too small");
    tr_device_guard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    tr_blob_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, TR_MAGIC, 8);
    h.num_tris = bvh->num_tris; h.num_nodes = bvh->num_nodes; h.arena_bytes = bvh->num_tris > 0 ? tr_arena_used_bytes(bvh->num_tris) : 0;
    h.depth = bvh->depth; h.key_mode = bvh->key_mode;
    for (int k = 0; k < 3; k++) { h.aabb_min[k] = bvh->aabb_min[k]; h.aabb_max[k] = bvh->aabb_max[k]; }
    h.sizeof_node = sizeof(tr_node); h.sizeof_tri = sizeof(tr_tri); h.sizeof_link = sizeof(tr_link);
    memcpy(h_buffer, &h, sizeof h);
    if (h.arena_bytes > 0) {
        TR_HIP_TRY(hipMemcpyAsync((char*)h_buffer + sizeof h, bvh->arena, (size_t)h.arena_bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        TR_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    return TR_OK;
}

int tr_bvh_deserialize(const void* h_buffer, int64_t size, void* stream, tr_bvh** out) {
    if (!h_buffer || !out) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (size < (int64_t)sizeof(tr_blob_header)) return tr_fail(TR_ERR_INVALID_ARG, "blob too small");
    tr_blob_header h;
    memcpy(&h, h_buffer, sizeof h);
    if (memcmp(h.magic, TR_MAGIC, 8) != 0) return tr_fail(TR_ERR_INVALID_ARG, "not a triro BVH blob (bad magic/version)");
    if (h.sizeof_node != sizeof(tr_node) || h.sizeof_tri != sizeof(tr_tri) || h.sizeof_link != sizeof(tr_link))
        return tr_fail(TR_ERR_INVALID_ARG, "blob was written with a different record layout");
    if (h.num_tris < 0 || h.num_nodes != (h.num_tris >= 2 ? h.num_tris - 1 : 0) || h.depth < 0 || h.depth > 64 ||
        size < (int64_t)sizeof h + h.arena_bytes)
        return tr_fail(TR_ERR_INVALID_ARG, "inconsistent blob header");
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return tr_fail(TR_ERR_NO_DEVICE, "no HIP device available (libtriro_hip has no CPU fallback)");
    tr_device_state* st;
    TR_TRY(tr_get_device_state(device, &st));
    tr_bvh* bvh = new (std::nothrow) tr_bvh();
    if (!bvh) return tr_fail(TR_ERR_OUT_OF_MEMORY, "host allocation failed");
    bvh->device = device;
    bvh->sched_mutex = new (std::nothrow) std::mutex();
    // an empty build gives the handle a correctly carved arena of the right capacity
    int s = tr_build_impl(bvh, nullptr, 0, nullptr, 0, (hipStream_t)stream);
    if (s == TR_OK && h.num_tris > 0) {
        s = tr_arena_alloc(bvh, h.num_tris);
        if (s == TR_OK && tr_arena_used_bytes(h.num_tris) != h.arena_bytes) s = tr_fail(TR_ERR_INVALID_ARG, "arena size mismatch");
        if (s == TR_OK) {
            if (hipMemcpyAsync(bvh->arena, (const char*)h_buffer + sizeof h, (size_t)h.arena_bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
                hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                s = tr_fail(TR_ERR_HIP, "upload of the BVH arena failed");
        }
        if (s == TR_OK) {
            bvh->num_tris = h.num_tris; bvh->num_nodes = h.num_nodes; bvh->depth = h.depth; bvh->key_mode = h.key_mode;
            for (int k = 0; k < 3; k++) { bvh->aabb_min[k] = h.aabb_min[k]; bvh->aabb_max[k] = h.aabb_max[k]; }
            tr_qframe_make(bvh->aabb_min, bvh->aabb_max, &bvh->frame);   // the grid is a function of the bounds
            s = tr_bvh_sync_frame(bvh);
        }
    }
    if (s != TR_OK) {
        if (bvh->arena) (void)hipFree(bvh->arena);
        delete bvh->sched_mutex;
        delete bvh;
        return s;
    }
    *out = bvh;
    return TR_OK;
}

int tr_bvh_destroy(tr_bvh* bvh) {
    if (!bvh) return TR_OK;
    int status = TR_OK;
    {
        tr_device_guard g;
        if (g.enter(bvh->device) == TR_OK) {
            if (bvh->arena && hipFree(bvh->arena) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(arena)");
            if (bvh->refit_temp && hipFree(bvh->refit_temp) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(refit_temp)");
            if (bvh->frame_dev && hipFree(bvh->frame_dev) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(frame_dev)");
            if (bvh->wnodes && hipFree(bvh->wnodes) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(wnodes)");
            if (bvh->wflag && hipFree(bvh->wflag) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(wflag)");
            if (bvh->widx && hipFree(bvh->widx) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(widx)");
            if (bvh->wide_event) (void)hipEventDestroy(bvh->wide_event);
            for (int k = 0; k < TR_SCHED_SLOTS; k++) {
                if (bvh->sched[k].buf && hipFree(bvh->sched[k].buf) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(sched)");
                if (bvh->sched[k].wspill && hipFree(bvh->sched[k].wspill) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(wspill)");
            }
        }
    }
    delete bvh->sched_mutex;
    delete bvh;
    return status;
}

int tr_bvh_replica_hash(const tr_bvh* bvh, uint64_t* h_hash, void* stream) {
    if (!bvh || !h_hash) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    tr_device_guard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    tr_device_state* st;
    TR_TRY(tr_get_device_state(bvh->device, &st));
    hipStream_t s = (hipStream_t)stream;
    {   // the hash is read back on the host: not something a stream that is being captured into a graph can do
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return tr_fail(TR_ERR_INVALID_ARG, "tr_bvh_replica_hash synchronises: not on a stream that is being captured");
        }
    }
    unsigned long long sum = 0;
    if (bvh->num_tris > 0) {
        unsigned long long* d_sum = nullptr;
        TR_HIP_TRY(hipMalloc((void**)&d_sum, sizeof(unsigned long long)));
        hipError_t e = hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), s);
        if (e == hipSuccess) {
            int64_t blocks = (bvh->num_tris + 255) / 256;
            if (blocks > (int64_t)st->num_cus * 8) blocks = (int64_t)st->num_cus * 8;
            hipLaunchKernelGGL(k_replica_hash, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(bvh->tris),
                               bvh->num_tris, d_sum);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&sum, d_sum, sizeof sum, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(d_sum);
        if (e != hipSuccess) return tr_fail(TR_ERR_HIP, std::string("replica hash: ") + hipGetErrorName(e));
    }
    // (host side of the same mix: the triangle count and the record size are part of the identity)
    unsigned long long x = sum ^ ((unsigned long long)bvh->num_tris * 0x9e3779b97f4a7c15ull) ^ ((unsigned long long)sizeof(tr_tri) << 56);
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    *h_hash = x ^ (x >> 31);
    return TR_OK;
}

int tr_bvh_last_launch(const tr_bvh* bvh, tr_launch_info* info) {
    if (!bvh || !info) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    if (!bvh->sched_mutex) return tr_fail(TR_ERR_INVALID_ARG, "handle without scheduling state");
    std::lock_guard<std::mutex> lock(*bvh->sched_mutex);
    if (!bvh->have_last_launch) return tr_fail(TR_ERR_INVALID_ARG, "no direct launch recorded for this handle yet");
    *info = bvh->last_launch;
    return TR_OK;
}

int tr_bvh_get_info(const tr_bvh* bvh, tr_bvh_info* info) {
    if (!bvh || !info) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    info->device = bvh->device;
    info->num_tris = bvh->num_tris;
    info->num_nodes = bvh->num_nodes;
    info->depth = bvh->depth;
    info->key_mode = bvh->key_mode;
    info->arena_bytes = bvh->arena_bytes;
    info->node_bytes = bvh->num_nodes * (int64_t)sizeof(tr_node);   // the exact nodes; the 32-byte grid copy is in arena_bytes only
    info->tri_bytes = bvh->num_tris * (int64_t)sizeof(tr_tri);
    for (int k = 0; k < 3; k++) { info->aabb_min[k] = bvh->aabb_min[k]; info->aabb_max[k] = bvh->aabb_max[k]; }
    return TR_OK;
}

int tr_bvh_download_qnodes(const tr_bvh* bvh, void* h_qnodes, float* h_frame6, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    hipStream_t s = (hipStream_t)stream;
    tr_device_guard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    if (h_qnodes && bvh->num_nodes)
        TR_HIP_TRY(hipMemcpyAsync(h_qnodes, bvh->qnodes, sizeof(tr_qnode) * (size_t)bvh->num_nodes, hipMemcpyDeviceToHost, s));
    TR_HIP_TRY(hipStreamSynchronize(s));
    if (h_frame6)
        for (int k = 0; k < 3; k++) { h_frame6[k] = bvh->frame.base[k]; h_frame6[3 + k] = bvh->frame.scale[k]; }
    return TR_OK;
}

int tr_bvh_download(const tr_bvh* bvh, void* h_nodes, void* h_links, void* h_tris, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    hipStream_t s = (hipStream_t)stream;
    tr_device_guard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    if (h_nodes && bvh->num_nodes)
        TR_HIP_TRY(hipMemcpyAsync(h_nodes, bvh->nodes, sizeof(tr_node) * (size_t)bvh->num_nodes, hipMemcpyDeviceToHost, s));
    if (h_links && bvh->num_nodes)
        TR_HIP_TRY(hipMemcpyAsync(h_links, bvh->links, sizeof(tr_link) * (size_t)bvh->num_nodes, hipMemcpyDeviceToHost, s));
    if (h_tris && bvh->num_tris)
        TR_HIP_TRY(hipMemcpyAsync(h_tris, bvh->tris, sizeof(tr_tri) * (size_t)bvh->num_tris, hipMemcpyDeviceToHost, s));
    TR_HIP_TRY(hipStreamSynchronize(s));
    return TR_OK;
}

int tr_set_option(const char* name, int64_t value) {
    if (!name) return tr_fail(TR_ERR_INVALID_ARG, "name == NULL");
    for (int k = 0; k < NUM_OPTS; k++) {
        if (strcmp(name, OPTS[k].name)) continue;
        if (OPTS[k].boolean) value = value != 0;
        if (value < OPTS[k].lo || value > OPTS[k].hi)
            return tr_fail(TR_ERR_INVALID_ARG, std::string(name) + " out of range [" + std::to_string(OPTS[k].lo) +
                                                   ", " + std::to_string(OPTS[k].hi) + "]");
        g_opts.v[k].store((int)value, std::memory_order_relaxed);
        return TR_OK;
    }
    // options of launch shapes that no longer exist: accepted and ignored, so that callers written against earlier ABIs
    // keep working -- no option ever changed results.  Round 1-2: the per-lane refill kernel, the early XCD map.  Round 5
    // (VERDICT r04 "next" #6: closed A/Bs kept alive): the persistent launch (persistent, blocks_per_cu), workgroups of 64 /
    // 256 rays (block_size), the unscrambled static order (scramble), the ordered schedule for count / location and the
    // unordered one for any-hit (unordered), the eighth wave per SIMD (occ8), the LDS-staged table of the top levels
    // (lds_top), the other expansion kernels (expand4), the static tail split (tail_split: experiments/static_tail_split.patch)
    for (const char* retired : {"refill", "refill_min", "xcd_segments", "leaf_min", "tail_split", "persistent", "blocks_per_cu",
                                "block_size", "scramble", "unordered", "occ8", "lds_top", "expand4"})
        if (!strcmp(name, retired)) return TR_OK;
    return tr_fail(TR_ERR_INVALID_ARG, std::string("unknown option: ") + name);
}

}  // extern "C"
