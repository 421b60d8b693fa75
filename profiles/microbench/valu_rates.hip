// Issue rates of the VALU forms the vote kernel's deposit could use (gfx950): scalar fp32 FMA, packed fp32 FMA / MUL,
// fp64 MUL and the f32 <-> f64 conversions.  Eight independent dependency chains per lane, 4 waves per SIMD, every CU busy;
// reports wave-instructions per SIMD per cycle-equivalent (ns * nominal 2.4 GHz).  hipcc --offload-arch=gfx950 -O3 valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed, long long* cyc)
{
    float a[8]; f2 p[8]; double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; d[i] = a[i]; }
    const float m = seed * 0.999f; const f2 pm = {m, m + 1e-3f};
    const long long c0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pm));
            if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (MODE == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)m));
            if (MODE == 4) { double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(a[i])); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(t)); }
            if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 7) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(pm));
            if (MODE == 8) asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 9) asm volatile("v_cvt_rpi_i32_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 10) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 11) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
            if (MODE == 13) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(p[i]) : "v"(pm));
            if (MODE == 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 15) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
            if (MODE == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            if (MODE == 22) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(m) : "s20", "s21");
            if (MODE == 23) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 24) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(m), "v"(seed));
            if (MODE == 25) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 26) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 27) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
            if (MODE == 28) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (MODE == 29) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %0, %1\n s_and_b64 s[20:21], s[20:21], s[22:23]" : : "v"(a[i]), "v"(m) : "s20", "s21", "s22", "s23", "scc");
            if (MODE == 30) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 31) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 32) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 33) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(a[i]));
            if (MODE == 34) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 35) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 36) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %0, %1, %0, s[20:21]\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %0, %1, %0, s[20:21]" : "+v"(a[i]) : "v"(m) : "s20", "s21");
            if (MODE == 37) asm volatile("s_mov_b64 vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 38) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 39) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %0, %1, %0, vcc\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 40) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 41) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 42) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 43) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 19) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 20) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));
            if (MODE == 21) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(seed));
        }
    }
    const long long c1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = c1 - c0;
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int per_iter, float* out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    long long* cyc; hipMalloc(&cyc, 8); k<MODE><<<blocks, 256>>>(out, 1.0f, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, 1.0f, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)ITER * 8 * per_iter * 4;          // 4 waves per SIMD
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction (at 2.4 GHz)   clock64 delta of one wave %lld = %.2f per instruction slot (x4 waves), %.1f MHz\n", name, ms, ms * 1e-3 * 2.4e9 / inst_per_simd, hc, (double)hc / inst_per_simd, hc / (ms * 1e3));
}
int main()
{
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0>("v_fma_f32", 1, out); run<6>("v_mul_f32", 1, out); run<1>("v_pk_fma_f32", 1, out); run<2>("v_pk_mul_f32", 1, out);
    run<7>("v_pk_mul_f32 op_sel_hi", 1, out); run<5>("v_pk_add_f32", 1, out); run<13>("v_pk_mov_b32", 1, out);
    run<3>("v_mul_f64", 1, out); run<4>("v_cvt_f64_f32+v_cvt_f32_f64", 2, out);
    run<8>("v_max3_u32", 1, out); run<9>("v_cvt_rpi_i32_f32", 1, out); run<10>("v_fract_f32", 1, out); run<11>("v_mad_i32_i24", 1, out);
    run<12>("v_cmp_lt_f32", 1, out);
    run<14>("v_add_f32", 1, out); run<15>("v_mul_f32_e64", 1, out); run<16>("v_mov_b32", 1, out); run<17>("v_and_b32", 1, out);
    run<18>("v_cndmask_b32", 1, out); run<19>("v_add_u32", 1, out); run<20>("v_mul_f32 const", 1, out); run<21>("v_fma_f32 3 regs", 1, out);
    run<22>("v_cndmask_b32_e64 sgpr", 1, out); run<23>("v_cmp+v_cndmask vcc", 2, out); run<24>("v_cndmask no chain", 1, out);
    run<25>("v_max_f32", 1, out); run<26>("v_med3_f32", 1, out); run<27>("v_bfe_u32", 1, out); run<28>("v_lshlrev_b32", 1, out);
    run<29>("2 v_cmp_e64 + s_and", 3, out); run<30>("v_cvt_i32_f32", 1, out); run<31>("v_add3_u32", 1, out); run<32>("v_lshl_add_u32", 1, out);
    run<33>("v_sub_f32 1.0", 1, out);
    run<34>("v_cmp + 2 cndmask vcc", 3, out); run<35>("v_cmp + 4 cndmask vcc", 5, out); run<36>("v_cmp + 4 cndmask sgpr", 5, out); run<37>("s_mov vcc + cndmask", 1, out);
    run<38>("v_cmp, v_add, cndmask", 3, out);
    run<39>("v_cmp + 4 cndmask_e64 vcc", 5, out); run<40>("cmp,cnd,add,add,cnd(reuse)", 5, out); run<41>("v_cmp + 2 addc vcc", 3, out);
    run<42>("cmp, nop4, cnd, cnd(reuse)", 3, out); run<43>("cmp,cnd,cmp,cnd", 4, out);
    run<6>("v_mul_f32 again", 1, out); run<0>("v_fma_f32 again", 1, out);
    return 0;
}
