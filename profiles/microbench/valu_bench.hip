// Per-instruction issue cost on gfx950 (shader-clock cycles per wave64 instruction per SIMD), one wave
// per SIMD and four waves per SIMD.  Each kernel is a 64x unrolled inline-asm body over rotating registers.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_bench valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(1024) void name(float* out, long long* clk, int iters)             \
    {                                                                                               \
        extern __shared__ float lds[];                                                              \
        float a = threadIdx.x * 1e-3f, b = 1.5f, c = 0.25f, d = 3.f;                                \
        float e = a + 1, f = a + 2, g = a + 3, h = a + 4;                                           \
        int ia = threadIdx.x * 4 & 252;                                                             \
        lds[threadIdx.x] = a;                                                                       \
        __syncthreads();                                                                            \
        const long long t0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < iters; ++it) {                                                        \
            asm volatile(REP16(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(ia) \
                         : : "vcc", "s20", "s21", "s22", "s23", "memory");                          \
        }                                                                                           \
        const long long t1 = __builtin_readcyclecounter();                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + ia;            \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                            \
    }

// every body = 4 instructions
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n")
KERNEL(k_add, "v_add_f32 %0, %0, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n v_add_f32 %5, %5, %1\n")
KERNEL(k_max, "v_max_f32 %0, %0, %1\n v_max_f32 %3, %3, %1\n v_max_f32 %4, %4, %1\n v_max_f32 %5, %5, %1\n")
KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc\n v_cndmask_b32 %5, %5, %1, vcc\n")
KERNEL(k_cnd_sgpr, "v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %3, %3, %1, s[20:21]\n v_cndmask_b32 %4, %4, %1, s[22:23]\n v_cndmask_b32 %5, %5, %1, s[22:23]\n")
KERNEL(k_cmp_cnd, "v_cmp_gt_f32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %3, %3, %1, s[20:21]\n v_cmp_gt_f32 s[22:23], %4, %1\n s_nop 1\n v_cndmask_b32 %5, %5, %1, s[22:23]\n")
KERNEL(k_cmp_vcc_cnd, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %3, %3, %1, vcc\n v_cmp_gt_f32 vcc, %4, %1\n v_cndmask_b32 %5, %5, %1, vcc\n")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %3, %3, 3, %1\n v_lshl_add_u32 %4, %4, 3, %1\n v_lshl_add_u32 %5, %5, 3, %1\n")
KERNEL(k_rndne, "v_rndne_f32 %0, %0\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n")
KERNEL(k_cvt, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1\n v_mov_b32 %5, %1\n")
KERNEL(k_perm16, "v_permlane16_swap_b32 %0, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %1, %2\n")
KERNEL(k_perm32, "v_permlane32_swap_b32 %0, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %1, %2\n")
KERNEL(k_swizzle, "ds_swizzle_b32 %0, %0 offset:0x401f\n ds_swizzle_b32 %3, %3 offset:0x401f\n ds_swizzle_b32 %4, %4 offset:0x401f\n ds_swizzle_b32 %5, %5 offset:0x401f\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_bperm, "ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_swz_fma, "ds_swizzle_b32 %0, %0 offset:0x401f\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n")
KERNEL(k_ldsread, "ds_read_b32 %0, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %3, %3, %1, %2\n v_med3_f32 %4, %4, %1, %2\n v_med3_f32 %5, %5, %1, %2\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2\n v_max3_f32 %4, %4, %1, %2\n v_max3_f32 %5, %5, %1, %2\n")
KERNEL(k_cmpx, "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %3, %1\n v_cmp_gt_f32 vcc, %4, %1\n v_cmp_gt_f32 vcc, %5, %1\n")
KERNEL(k_cmp_e64, "v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %3, %1\n v_cmp_gt_f32 s[20:21], %4, %1\n v_cmp_gt_f32 s[22:23], %5, %1\n")

typedef void (*kern_t)(float*, long long*, int);
void run(const char* name, kern_t k, int ninstr, float* out, long long* clk)
{
    const int iters = 2000;
    for (int w : {1, 4}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<<<256, 256 * w, 100 * 1024>>>(out, clk, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<<<256, 256 * w, 100 * 1024>>>(out, clk, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[256];
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
        const double n = (double)iters * 16 * ninstr;   // instructions per wave
        printf("%-16s w/SIMD=%d  %7.2f cycles/instr/wave  %6.2f cycles/instr/SIMD   (%.3f ms wall => %.0f MHz)\n", name, w,
               avg / n, avg / (n * w), ms, avg / (ms * 1e3));
    }
}
int main()
{
    float* out; long long* clk;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    (void)hipMalloc(&clk, 256 * 8);
#define R(k, n) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); run(#k, k, n, out, clk);
    R(k_fma, 4) R(k_add, 4) R(k_max, 4) R(k_max3, 4) R(k_med3, 4) R(k_mov, 4) R(k_cnd_vcc, 4) R(k_cnd_sgpr, 4)
    R(k_cmpx, 4) R(k_cmp_e64, 4) R(k_cmp_cnd, 4) R(k_cmp_vcc_cnd, 4) R(k_addc, 4)
    R(k_lshladd, 4) R(k_rndne, 4) R(k_cvt, 4) R(k_exp, 4) R(k_ldexp, 4) R(k_perm16, 4) R(k_perm32, 4) R(k_dpp, 4)
    R(k_swizzle, 4) R(k_bperm, 4) R(k_ldsread, 4) R(k_swz_fma, 4)
    return 0;
}
