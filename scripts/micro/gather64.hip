// gather64.hip -- diagnostic microbenchmark: how fast can a wave64 kernel
// gather random 64-byte records (one BVH node per lane: four 16-byte loads, like the traversal)
// from a table the size of the headline BVH arena?  This, not the 8 TB/s stream figure, is the
// memory-side ceiling of a node fetch.  Not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct alignas(16) f4 { float x, y, z, w; };

// each lane walks `steps` records: dependent = 1 chases indices through the records themselves
// (latency-bound, like a traversal), dependent = 0 generates them with an LCG (throughput-bound)
template <int DEP>
__global__ __launch_bounds__(256) void k_gather(const f4* __restrict__ table, uint32_t nrows,
                                                const uint32_t* __restrict__ idx, int steps,
                                                float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = idx[t];
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const f4* p = table + (size_t)r * 4;
        const f4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z;
        if (DEP) r = __float_as_uint(d.w) % nrows;          // next index stored in the record
        else r = (r * 1664525u + 1013904223u) % nrows;       // address independent of the data
    }
    out[t] = acc;
}

extern "C" int gather64_run(const void* table, uint32_t nrows, const void* idx, int64_t nthreads,
                            int steps, int dependent, void* out, void* stream) {
    dim3 grid((unsigned)(nthreads / 256)), block(256);
    if (dependent)
        hipLaunchKernelGGL(k_gather<1>, grid, block, 0, (hipStream_t)stream, (const f4*)table, nrows,
                           (const uint32_t*)idx, steps, (float*)out);
    else
        hipLaunchKernelGGL(k_gather<0>, grid, block, 0, (hipStream_t)stream, (const f4*)table, nrows,
                           (const uint32_t*)idx, steps, (float*)out);
    return (int)hipGetLastError();
}
