// Could the vote kernel's 256 workgroups add their LDS tiles straight into ONE grid with device-scope integer atomics
// (32-bit fixed point: order-independent, hence deterministic) instead of writing 256 partial tiles (29.6 MB) that a second
// kernel sums (csrc/vote.hip: tile-major dump + reduce_tiles_kernel, ~6 + ~10 us at C2)?  This prices the atomics:
// 256 workgroups x 1024 threads, each workgroup adds `cells` consecutive u32 (coalesced, no return value) into the same
// array, every workgroup starting at a different offset.  Compared with the plain 16-byte store of the same tile to a
// private slot (what the product does).
// Build: hipcc --offload-arch=gfx950 -O3 -o flush_atomics_bench flush_atomics_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(1024) void flush_atomic(unsigned* grid, int cells, int copies)
{
    unsigned* g = grid + (size_t)(blockIdx.x % copies) * cells;
    const int start = (int)((blockIdx.x * 997u) % (unsigned)cells) & ~63;
    for (int k = threadIdx.x; k < cells; k += 1024) {
        int c = start + k;
        if (c >= cells) c -= cells;
        __hip_atomic_fetch_add(&g[c], (unsigned)(k + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ __launch_bounds__(1024) void flush_store(uint4* part, int cells4)
{
    uint4* p = part + (size_t)blockIdx.x * cells4;
    for (int k = threadIdx.x; k < cells4; k += 1024) p[k] = make_uint4(k, blockIdx.x, k, 1u);
}
__global__ void touch(unsigned* g, int n) { for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) g[k] = 0; }

int main()
{
    const int cells = 25688, wgs = 256;
    unsigned* grid; uint4* part;
    hipMalloc(&grid, (size_t)64 * cells * 4);
    hipMalloc(&part, (size_t)wgs * 28960 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int copies : {1, 2, 4, 8, 16, 64}) {
        touch<<<256, 256>>>(grid, 64 * cells);
        flush_atomic<<<wgs, 1024>>>(grid, cells, copies);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) flush_atomic<<<wgs, 1024>>>(grid, cells, copies);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("atomic add, %3d grid copies (workgroup b -> copy b %% copies): %7.2f us per launch (%d x %d u32 atomics)\n", copies,
               ms * 1e3 / 20, wgs, cells);
    }
    flush_store<<<wgs, 1024>>>(part, 28960 / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) flush_store<<<wgs, 1024>>>(part, 25688 / 4);
    hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    printf("16-byte stores to private slots (the product's dump):          %7.2f us per launch (26.3 MB)\n", ms * 1e3 / 20);
    // correctness of the atomics across XCDs: every cell of copy 0 must hold the exact sum
    touch<<<256, 256>>>(grid, cells);
    flush_atomic<<<wgs, 1024>>>(grid, cells, 1);
    hipDeviceSynchronize();
    unsigned* h = new unsigned[cells];
    hipMemcpy(h, grid, cells * 4, hipMemcpyDeviceToHost);
    // cell c receives from workgroup b the value k + b with k = (c - start_b) mod cells
    long bad = 0;
    for (int c = 0; c < cells; ++c) {
        unsigned want = 0;
        for (int b = 0; b < wgs; ++b) {
            const int start = (int)((b * 997u) % (unsigned)cells) & ~63;
            int k = c - start; if (k < 0) k += cells;
            want += (unsigned)(k + b);
        }
        bad += h[c] != want;
    }
    printf("cells with a wrong sum after one launch: %ld of %d\n", bad, cells);
    return 0;
}
