// What would the pair-MLP tile cost on the REAL matrix cores?  
//
// csrc/pair_mlp.hip runs the 16-pair tile as 108 v_mfma_f32_16x16x4_f32 (exact fp32, but on gfx950 that instruction runs on
// the VALU datapath: 32 cycles each, no overlap with the ~650 VALU of PPF + relu + decode).  A 3-way bf16 split
// (x = hi + mid + lo, 6 of the 9 cross products) on v_mfma_f32_16x16x32_bf16 would need ~78 matrix instructions per tile
// plus ~225 VALU to split the activations, and the matrix pipe is a separate pipe.  This benchmark issues both
// instruction mixes with the dependence structure of the real chain (8-step accumulator chains, two chains interleaved)
// beside a stream of independent v_fma_f32 standing for the VALU work, 4 waves per SIMD like the product kernel, and
// reports cycles per tile per SIMD.  It does not compute anything meaningful: it prices the option.
// Build: hipcc --offload-arch=gfx950 -O3 -o mlp_chain_bench mlp_chain_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: fp32 chain: NM MFMAs (16x16x4 f32) + NV VALU per tile
// MODE 1: bf16 chain: NM MFMAs (16x16x32 bf16) + NV VALU per tile
template <int MODE, int NM, int NV>
__global__ __launch_bounds__(1024) void tile_bench(float* out, int tiles)
{
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int t = 0; t < tiles; ++t) {
        // matrix instructions in groups of 16 (8-step chains on two accumulators, as a 32 -> 32 layer issues them), the VALU
        // work spread evenly between them
        constexpr int GROUPS = (NM + 15) / 16;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int m_here = (g + 1) * 16 <= NM ? 16 : NM - g * 16;
#pragma unroll
            for (int m = 0; m < m_here; ++m) {
                if (MODE == 0) acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 1], 0, 0, 0);
                else acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[m & 1], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV / NM; ++q) v[(m + q) & 7] = __builtin_fmaf(v[(m + q) & 7], b, a);
            }
            // the layer's outputs feed the VALU (relu) before the next layer's inputs exist
            v[0] += acc[0][0];
            v[1] += acc[1][1];
            a = v[0] * 1e-30f + a;
        }
#pragma unroll
        for (int q = 0; q < NV % NM; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], b, a);
    }
    float s = acc[0][0] + acc[0][1] + acc[1][2] + acc[1][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NM, int NV>
void run(const char* name, float* out)
{
    const int tiles = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    tile_bench<MODE, NM, NV><<<256, 1024>>>(out, 20);           // one 16-wave workgroup per CU = 4 waves per SIMD
    hipDeviceSynchronize();
    hipEventRecord(e0);
    tile_bench<MODE, NM, NV><<<256, 1024>>>(out, tiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // every SIMD ran 4 waves x `tiles` tiles
    const double us_per_tile = ms * 1e3 / (4.0 * tiles);
    printf("%-58s %8.3f ms   %7.1f ns per tile and SIMD   (the C2 launch = 32 tiles per SIMD -> %6.1f us)\n", name, ms,
           us_per_tile * 1e3, us_per_tile * 32);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    printf("one pair-MLP tile (16 pairs), 4 waves per SIMD, 256 CUs; VALU = independent v_fma_f32\n");
    run<0, 108, 0>("fp32 MFMA 16x16x4 x108, no VALU", out);
    run<0, 108, 650>("fp32 MFMA 16x16x4 x108 + 650 VALU   (today's kernel)", out);
    run<0, 0 + 1, 650>("650 VALU alone (+1 MFMA)", out);
    run<1, 78, 0>("bf16 MFMA 16x16x32 x78, no VALU", out);
    run<1, 78, 875>("bf16 MFMA 16x16x32 x78 + 875 VALU    (3-way split)", out);
    run<1, 1, 875>("875 VALU alone (+1 MFMA)", out);
    run<1, 39, 760>("bf16 MFMA 16x16x32 x39 + 760 VALU    (2-way split, ~16-bit)", out);
    return 0;
}
