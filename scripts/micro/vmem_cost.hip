// vmem_cost.hip -- diagnostic microbenchmark (not part of the product library): what does one
// gather wave-instruction cost in the CU's texture path (TA address / TD data return), as a
// function of the bytes per lane and of the number of active lanes?  Each lane walks a dependent
// chain of random records in a table small enough to stay in L1/L2 (so that the fabric is not
// what is measured) -- the regime of the traversal kernel (L1 hit rate 86 %, TD busy > 80 %).
#include <hip/hip_runtime.h>
#include <stdint.h>

struct alignas(16) f4 { float x, y, z, w; };
struct alignas(8) f2 { float x, y; };

// NQ = number of 16-byte loads per record, ND = extra 8-byte loads, NS = extra 4-byte loads;
// record stride in bytes = `stride`; the next index is read from the first dword of the record.
template <int NQ, int ND, int NS>
__global__ __launch_bounds__(256) void k_chain(const char* __restrict__ table, uint32_t nrec,
                                               uint32_t stride, const uint32_t* __restrict__ idx,
                                               int steps, unsigned long long lane_mask,
                                               float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = idx[t];
    float acc = 0.f;
    const bool active = (lane_mask >> (threadIdx.x & 63)) & 1ull;
    if (active) {
        for (int s = 0; s < steps; s++) {
            const char* p = table + (size_t)r * stride;
            uint32_t nxt = 0;
#pragma unroll
            for (int k = 0; k < NQ; k++) {
                const f4 a = reinterpret_cast<const f4*>(p)[k];
                acc += a.y + a.z + a.w;
                if (k == 0) nxt = __float_as_uint(a.x);
            }
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const f2 a = reinterpret_cast<const f2*>(p + 16 * NQ)[k];
                acc += a.y;
                if (NQ == 0 && k == 0) nxt = __float_as_uint(a.x); else acc += a.x;
            }
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const float a = reinterpret_cast<const float*>(p + 16 * NQ + 8 * ND)[k];
                if (NQ == 0 && ND == 0 && k == 0) nxt = __float_as_uint(a); else acc += a;
            }
            r = nxt;
        }
    }
    out[t] = acc;
}

#define CASE(Q, D, S) if (nq == Q && nd == D && ns == S) { \
    hipLaunchKernelGGL((k_chain<Q, D, S>), grid, block, 0, (hipStream_t)stream, (const char*)table, nrec, stride, \
                       (const uint32_t*)idx, steps, lane_mask, (float*)out); return (int)hipGetLastError(); }

extern "C" int vmem_cost_run(const void* table, uint32_t nrec, uint32_t stride, const void* idx,
                             int64_t nthreads, int steps, int nq, int nd, int ns,
                             unsigned long long lane_mask, void* out, void* stream) {
    dim3 grid((unsigned)(nthreads / 256)), block(256);
    CASE(4, 0, 0) CASE(3, 0, 0) CASE(2, 0, 0) CASE(1, 0, 0) CASE(0, 1, 0) CASE(0, 0, 1)
    CASE(2, 1, 0) CASE(2, 0, 1) CASE(3, 0, 1) CASE(7, 0, 0) CASE(4, 1, 1) CASE(0, 2, 0) CASE(0, 4, 0)
    return -1;
}
