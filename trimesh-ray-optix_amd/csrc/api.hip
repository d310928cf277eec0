// api.hip -- runtime bring-up, error reporting and acceleration-structure entry points of the
// C ABI (include/triro_hip.h).  Replaces triro/backend/base.cpp (global OptiX context, module,
// pipelines, SBTs: base.cpp:15-157) -- none of those concepts survive; what remains is a
// per-device table {CU count, work-counter ring, builder temporaries} created on first use.
#include <mutex>
#include <string.h>

#include "tr_internal.h"

namespace {
thread_local std::string g_last_error;
std::mutex g_mutex;
constexpr int TR_MAX_DEVICES = 64;
tr_device_state g_devices[TR_MAX_DEVICES];
tr_options g_options;

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    int enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) return TR_ERR_NO_DEVICE;
        if (prev != device) {
            if (hipSetDevice(device) != hipSuccess) return TR_ERR_NO_DEVICE;
            changed = true;
        }
        return TR_OK;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};
}  // namespace

void tr_set_error(const std::string& msg) { g_last_error = msg; }
int tr_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
tr_options& tr_opts() { return g_options; }

int tr_get_device_state(int device, tr_device_state** out) {
    if (device < 0 || device >= TR_MAX_DEVICES) return tr_fail(TR_ERR_INVALID_ARG, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_mutex);
    tr_device_state& st = g_devices[device];
    if (!st.ready) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return tr_fail(TR_ERR_NO_DEVICE, "no HIP device available (libtriro_hip has no CPU fallback)");
        if (device >= count) return tr_fail(TR_ERR_NO_DEVICE, "device ordinal >= device count");
        DeviceGuard g;
        if (g.enter(device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
        hipDeviceProp_t prop;
        TR_HIP_TRY(hipGetDeviceProperties(&prop, device));
        st.device = device;
        st.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        TR_HIP_TRY(hipMalloc((void**)&st.counters, sizeof(unsigned long long) * TR_NUM_COUNTERS));
        TR_HIP_TRY(hipMemset(st.counters, 0, sizeof(unsigned long long) * TR_NUM_COUNTERS));
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
    if (!g_options.build_cache && st->build_temp) {
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
    DeviceGuard g;
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
    DeviceGuard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    return tr_build_impl(bvh, d_vertices, nv, d_faces, nf, (hipStream_t)stream);
}

int tr_bvh_refit(tr_bvh* bvh, const float* d_vertices, int64_t nv, const int32_t* d_faces,
                 int64_t nf, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    DeviceGuard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    return tr_refit_impl(bvh, d_vertices, nv, d_faces, nf, (hipStream_t)stream);
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
const char TR_MAGIC[8] = {'T', 'R', 'B', 'V', 'H', 0, 0, 2};   // 2: child boxes stored lo.xy|lo.z hi.z|hi.xy
}  // namespace

int64_t tr_bvh_serialized_size(const tr_bvh* bvh) {
    if (!bvh) return -1;
    return (int64_t)sizeof(tr_blob_header) + (bvh->num_tris > 0 ? bvh->arena_bytes : 0);
}

int tr_bvh_serialize(const tr_bvh* bvh, void* h_buffer, int64_t size, void* stream) {
    if (!bvh || !h_buffer) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    if (size < tr_bvh_serialized_size(bvh)) return tr_fail(TR_ERR_INVALID_ARG, "buffer too small");
    DeviceGuard g;
    if (g.enter(bvh->device) != TR_OK) return tr_fail(TR_ERR_NO_DEVICE, "hipSetDevice failed");
    tr_blob_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, TR_MAGIC, 8);
    h.num_tris = bvh->num_tris; h.num_nodes = bvh->num_nodes; h.arena_bytes = bvh->num_tris > 0 ? bvh->arena_bytes : 0;
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
        if (s == TR_OK && bvh->arena_bytes != h.arena_bytes) s = tr_fail(TR_ERR_INVALID_ARG, "arena size mismatch");
        if (s == TR_OK) {
            if (hipMemcpyAsync(bvh->arena, (const char*)h_buffer + sizeof h, (size_t)h.arena_bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
                hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                s = tr_fail(TR_ERR_HIP, "upload of the BVH arena failed");
        }
        if (s == TR_OK) {
            bvh->num_tris = h.num_tris; bvh->num_nodes = h.num_nodes; bvh->depth = h.depth; bvh->key_mode = h.key_mode;
            for (int k = 0; k < 3; k++) { bvh->aabb_min[k] = h.aabb_min[k]; bvh->aabb_max[k] = h.aabb_max[k]; }
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
        DeviceGuard g;
        if (g.enter(bvh->device) == TR_OK && bvh->arena) {
            if (hipFree(bvh->arena) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(arena)");
            if (bvh->refit_temp && hipFree(bvh->refit_temp) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(refit_temp)");
            for (int k = 0; k < TR_SCHED_SLOTS; k++)
                if (bvh->sched[k].buf && hipFree(bvh->sched[k].buf) != hipSuccess) status = tr_fail(TR_ERR_HIP, "hipFree(sched)");
        }
    }
    delete bvh->sched_mutex;
    delete bvh;
    return status;
}

int tr_bvh_get_info(const tr_bvh* bvh, tr_bvh_info* info) {
    if (!bvh || !info) return tr_fail(TR_ERR_INVALID_ARG, "null argument");
    info->device = bvh->device;
    info->num_tris = bvh->num_tris;
    info->num_nodes = bvh->num_nodes;
    info->depth = bvh->depth;
    info->key_mode = bvh->key_mode;
    info->arena_bytes = bvh->arena_bytes;
    info->node_bytes = bvh->num_nodes * (int64_t)sizeof(tr_node);
    info->tri_bytes = bvh->num_tris * (int64_t)sizeof(tr_tri);
    for (int k = 0; k < 3; k++) { info->aabb_min[k] = bvh->aabb_min[k]; info->aabb_max[k] = bvh->aabb_max[k]; }
    return TR_OK;
}

int tr_bvh_download(const tr_bvh* bvh, void* h_nodes, void* h_links, void* h_tris, void* stream) {
    if (!bvh) return tr_fail(TR_ERR_INVALID_ARG, "bvh == NULL");
    hipStream_t s = (hipStream_t)stream;
    DeviceGuard g;
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
    if (!strcmp(name, "persistent")) { g_options.persistent = value != 0; return TR_OK; }
    if (!strcmp(name, "blocks_per_cu")) {
        if (value < 1 || value > 32) return tr_fail(TR_ERR_INVALID_ARG, "blocks_per_cu out of range");
        g_options.blocks_per_cu = (int)value;
        return TR_OK;
    }
    if (!strcmp(name, "refill")) { g_options.refill = value != 0; return TR_OK; }
    if (!strcmp(name, "adaptive")) { g_options.adaptive = value != 0; return TR_OK; }
    if (!strcmp(name, "steal")) {   // 0 off, 1 on (threshold 64 trips), > 1: on with this trip threshold
        if (value < 0 || value > 4096) return tr_fail(TR_ERR_INVALID_ARG, "steal out of range");
        g_options.steal = (int)value; return TR_OK;
    }
    if (!strcmp(name, "tile")) {
        if (value < 0 || value > 2) return tr_fail(TR_ERR_INVALID_ARG, "tile must be 0, 1 or 2");
        g_options.tile = (int)value; return TR_OK;
    }
    if (!strcmp(name, "scramble")) { g_options.scramble = value != 0; return TR_OK; }
    if (!strcmp(name, "build_cache")) { g_options.build_cache = value != 0; return TR_OK; }
    if (!strcmp(name, "block_size")) {
        if (value != 64 && value != 128 && value != 256) return tr_fail(TR_ERR_INVALID_ARG, "block_size must be 64, 128 or 256");
        g_options.block_size = (int)value;
        return TR_OK;
    }
    if (!strcmp(name, "compact")) { g_options.compact = value != 0; return TR_OK; }
    if (!strcmp(name, "xcd_chunk")) {
        if (value < 0 || value > 65536) return tr_fail(TR_ERR_INVALID_ARG, "xcd_chunk out of range");
        g_options.xcd_chunk = (int)value;
        return TR_OK;
    }
    if (!strcmp(name, "xcd_segments")) { g_options.xcd_segments = value != 0; return TR_OK; }
    if (!strcmp(name, "leaf_min")) {
        if (value < 0 || value > 64) return tr_fail(TR_ERR_INVALID_ARG, "leaf_min out of range");
        g_options.leaf_min = (int)value;
        return TR_OK;
    }
    if (!strcmp(name, "refill_min")) {
        if (value < 1 || value > 64) return tr_fail(TR_ERR_INVALID_ARG, "refill_min out of range");
        g_options.refill_min = (int)value;
        return TR_OK;
    }
    return tr_fail(TR_ERR_INVALID_ARG, std::string("unknown option: ") + name);
}

}  // extern "C"
