// Issue-rate microbenchmarks behind DESIGN.md section 3.1: how fast does gfx950 issue
// v_mfma_f32_16x16x4_f32, plain and packed fp32 VALU, and a mix of both from the same SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bench mfma_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: MFMA only, NACC independent accumulators; 1: one dependent accumulator chain;
// 2: MFMA + V plain v_fma per MFMA; 3: VALU v_fma only; 4: v_pk_fma only; 5: MFMA + V v_pk_fma per MFMA
template <int MODE, int V>
__global__ __launch_bounds__(256) void bench(float* out, long long* clk, int iters)
{
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
    float v[8];
    f32x2 pv[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; pv[i] = f32x2{a + i, b + i}; }
    const long long t0 = __builtin_readcyclecounter();   // s_memtime: shader clock
    const long long r0 = wall_clock64();                  // s_memrealtime: 100 MHz
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 0 || MODE == 2 || MODE == 5) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
            if (MODE == 1) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
            if (MODE == 2 || MODE == 3) {
#pragma unroll
                for (int q = 0; q < V; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], b, a);
            }
            if (MODE == 4 || MODE == 5) {
#pragma unroll
                for (int q = 0; q < V; ++q) pv[q & 7] = __builtin_elementwise_fma(pv[q & 7], f32x2{b, b}, f32x2{a, a});
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = wall_clock64();
    float s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + pv[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int V>
void run(const char* name, int waves_per_simd, float* out, long long* clk)
{
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MODE, V><<<blocks, 256>>>(out, clk, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<MODE, V><<<blocks, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double slots = (double)iters * 4 * waves_per_simd;  // MFMA (or VALU group) slots per SIMD
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);
    printf("%-44s w/SIMD=%d  %8.3f ms  shader clk %6.0f MHz  %7.2f cycles/slot (shader clk)  %7.2f cycles/slot @2.4GHz wall\n",
           name, waves_per_simd, ms, mhz, (double)h[0] / slots, ms * 1e-3 * 2.4e9 / slots);
}

int main()
{
    float* out; long long* clk;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    for (int w : {1, 2, 4}) run<0, 0>("mfma_f32_16x16x4 x4 independent acc", w, out, clk);
    for (int w : {1, 4}) run<1, 0>("mfma_f32_16x16x4 dependent chain", w, out, clk);
    for (int w : {1, 4}) run<3, 8>("8 v_fma_f32 per slot", w, out, clk);
    for (int w : {1, 4}) run<4, 8>("8 v_pk_fma_f32 per slot", w, out, clk);
    for (int w : {1, 4}) run<2, 2>("mfma + 2 v_fma_f32 per slot", w, out, clk);
    for (int w : {1, 4}) run<2, 8>("mfma + 8 v_fma_f32 per slot", w, out, clk);
    for (int w : {1, 4}) run<5, 8>("mfma + 8 v_pk_fma_f32 per slot", w, out, clk);
    return 0;
}
