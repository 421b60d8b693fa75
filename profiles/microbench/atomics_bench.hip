// Micro-benchmark: LDS / L2 atomic-add ceilings on gfx950 (the "measured ceiling" the vote kernel's
// atomics/s are compared against).  hipcc --offload-arch=gfx950 -O3 atomics_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// MODE 0: ds_add_f32 random, 1: ds_add_u32 random, 2: ds_add_f32 conflict-free (lane-linear),
// 3: ds_add_f32 same address per wave, 4: ds_add_f32 trilinear-like (8 atomics around a random cell)
template <int MODE>
__global__ __launch_bounds__(1024) void lds_kernel(float* out, int cells, int iters)
{
    extern __shared__ float tile[];
    for (int k = threadIdx.x; k < cells; k += 1024) tile[k] = 0.f;
    __syncthreads();
    uint32_t s = 0x9e3779b9u * (blockIdx.x * 1024 + threadIdx.x + 1);
    uint32_t* ut = reinterpret_cast<uint32_t*>(tile);
    for (int i = 0; i < iters; ++i) {
        uint32_t r = rng(s);
        if (MODE == 0) atomicAdd(&tile[r % cells], 1.0f);
        if (MODE == 1) atomicAdd(&ut[r % cells], 1u);
        if (MODE == 2) atomicAdd(&tile[(threadIdx.x + i * 1024) % cells], 1.0f);
        if (MODE == 3) atomicAdd(&tile[(r % cells) & ~0u * 0 + ((threadIdx.x >> 6) * 17 + i) % cells], 1.0f);
        if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(tile) + (r % (cells / 2)), 1ull);
        if (MODE == 6) atomicAdd(reinterpret_cast<double*>(tile) + (r % (cells / 2)), 1.0);
        if (MODE == 7) { uint32_t old = atomicAdd(&ut[r % cells], 1u); s += old; }
        if (MODE == 4) {
            int gz = 26, gy = 76, base = r % (cells - gy * gz - gz - 2);
            atomicAdd(&tile[base], 0.1f); atomicAdd(&tile[base + 1], 0.1f);
            atomicAdd(&tile[base + gz], 0.1f); atomicAdd(&tile[base + gz + 1], 0.1f);
            atomicAdd(&tile[base + gy * gz], 0.1f); atomicAdd(&tile[base + gy * gz + 1], 0.1f);
            atomicAdd(&tile[base + gy * gz + gz], 0.1f); atomicAdd(&tile[base + gy * gz + gz + 1], 0.1f);
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int k = threadIdx.x; k < cells; k += 1024) acc += tile[k];
    if (acc == 12345.678f) out[0] = acc;
}

// global (L2 / memory-side) fp32 atomics on a grid of `cells` floats
template <int MODE>
__global__ __launch_bounds__(1024) void global_kernel(float* grid, int cells, int iters)
{
    uint32_t s = 0x9e3779b9u * (blockIdx.x * 1024 + threadIdx.x + 1);
    for (int i = 0; i < iters; ++i) {
        uint32_t r = rng(s);
        if (MODE == 0) atomicAdd(&grid[r % cells], 1.0f);
        if (MODE == 1) atomicAdd(&grid[(r % 64) * 1], 1.0f);  // 64 hot cells
    }
}

template <typename F>
static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    float* out; CK(hipMalloc(&out, 1 << 24));
    CK(hipMemset(out, 0, 1 << 24));
    const int cells = 26000, iters = 512, blocks = 256;
    const size_t lds = cells * sizeof(float);
    const double lane_ops = (double)blocks * 1024 * iters;
#define RUN_LDS(M, name, mult)                                                                               \
    {                                                                                                        \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        float ms = time_ms([&] { hipLaunchKernelGGL(lds_kernel<M>, dim3(blocks), dim3(1024), lds, 0, out, cells, iters); }, 5); \
        printf("%-44s %8.3f ms  %8.2f G lane-atomics/s  (%.2f cycles/wave-instr/CU @2.4GHz)\n", name, ms,     \
               lane_ops * mult / ms / 1e6, ms * 1e-3 * 2.4e9 / ((double)1024 / 64 * iters * mult));           \
    }
    RUN_LDS(0, "LDS ds_add_f32 random (26k cells)", 1)
    RUN_LDS(1, "LDS ds_add_u32 random (26k cells)", 1)
    RUN_LDS(2, "LDS ds_add_f32 lane-linear (conflict-free)", 1)
    RUN_LDS(3, "LDS ds_add_f32 one address per wave", 1)
    RUN_LDS(4, "LDS ds_add_f32 8-corner trilinear pattern", 8)
    RUN_LDS(5, "LDS ds_add_u64 random (13k cells)", 1)
    RUN_LDS(6, "LDS ds_add_f64 random (13k cells)", 1)
    RUN_LDS(7, "LDS ds_add_rtn_u32 random (26k cells)", 1)
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(global_kernel<0>, dim3(blocks * 4), dim3(1024), 0, 0, out, 51376, 128); }, 5);
        printf("%-44s %8.3f ms  %8.2f G lane-atomics/s\n", "L2 global_atomic_add_f32 random (51k cells)", ms, (double)blocks * 4 * 1024 * 128 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(global_kernel<1>, dim3(blocks * 4), dim3(1024), 0, 0, out, 51376, 128); }, 5);
        printf("%-44s %8.3f ms  %8.2f G lane-atomics/s\n", "L2 global_atomic_add_f32 64 hot cells", ms, (double)blocks * 4 * 1024 * 128 / ms / 1e6);
    }
    return 0;
}
