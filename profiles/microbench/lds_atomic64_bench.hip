// Would pairing the two z-adjacent corners of a vote into ONE 64-bit LDS atomic (when the cell index is even) pay?  The centre
// vote issues 8 returning 32-bit LDS atomics per candidate (csrc/vote.hip:vote_deposit), four pairs of adjacent cells.  This
// prices 2 x ds_add_rtn_u32 on (a, a + 1) against 1 x ds_add_rtn_u64 on a (a even), random cells of a 113 KB tile, 16 waves
// per workgroup, one workgroup per CU, as in the product kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic64_bench lds_atomic64_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(1024) void bench(unsigned* out, int iters)
{
    extern __shared__ unsigned tile[];
    for (int k = threadIdx.x; k < 28960; k += 1024) tile[k] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned a[4];
        for (int q = 0; q < 4; ++q) { s = s * 1664525u + 1013904223u; a[q] = ((s >> 8) % 14000u) * 2u; }   // even cells
        if (MODE == 0) {
            unsigned o[8];
            for (int q = 0; q < 4; ++q) { o[2 * q] = atomicAdd(&tile[a[q]], 3u + it); o[2 * q + 1] = atomicAdd(&tile[a[q] + 1], 5u + it); }
            for (int q = 0; q < 8; ++q) acc = max(acc, o[q]);
        } else {
            unsigned long long o[4];
            for (int q = 0; q < 4; ++q)
                o[q] = atomicAdd(reinterpret_cast<unsigned long long*>(&tile[a[q]]), ((unsigned long long)(5u + it) << 32) | (3u + it));
            for (int q = 0; q < 4; ++q) acc = max(acc, max((unsigned)o[q], (unsigned)(o[q] >> 32)));
        }
    }
    __syncthreads();
    out[blockIdx.x * 1024 + threadIdx.x] = acc + tile[threadIdx.x];
}

int main()
{
    unsigned* out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipFuncSetAttribute((const void*)bench<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipFuncSetAttribute((const void*)bench<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        bench<0><<<256, 1024, 116 * 1024>>>(out, 50); hipDeviceSynchronize();
        hipEventRecord(e0); bench<0><<<256, 1024, 116 * 1024>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("8 x ds_add_rtn_u32 per candidate: %8.3f ms  (%.1f cycles per candidate and wave at 2.1 GHz, 16 waves per CU)\n", ms,
               ms * 1e-3 * 2.1e9 / iters);
        bench<1><<<256, 1024, 116 * 1024>>>(out, 50); hipDeviceSynchronize();
        hipEventRecord(e0); bench<1><<<256, 1024, 116 * 1024>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("4 x ds_add_rtn_u64 per candidate: %8.3f ms  (%.1f cycles per candidate and wave)\n", ms, ms * 1e-3 * 2.1e9 / iters);
    }
    return 0;
}
