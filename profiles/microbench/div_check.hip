// exhaustive check of div_by() (csrc/vote.hip) against the compiler's IEEE fp32 division on the range the vote uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float refined_rcp(float b) { const float y0 = __builtin_amdgcn_rcpf(b); const float e = fmaf(-b, y0, 1.0f); return fmaf(e, y0, y0); }
__device__ __forceinline__ float div_by(float a, float b, float y)
{
    const float q0 = a * y; const float r0 = fmaf(-b, q0, a); const float q1 = fmaf(r0, y, q0); const float r1 = fmaf(-b, q1, a);
    return fmaf(r1, y, q1);
}
// round 3: the tiled vote's deposit divides through one fp64 product, fl32(fl64(a * fl64(1 / res))) (csrc/vote.hip: v3_deposit)
__global__ void check64(float res, uint32_t e_lo, uint32_t e_hi, unsigned long long* bad, unsigned long long* n)
{
    const double y = 1.0 / (double)res;
    unsigned long long mism = 0, cnt = 0;
    const uint64_t total = (uint64_t)(e_hi - e_lo + 1) << 23;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)(((uint64_t)e_lo << 23) + k);
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float a = __uint_as_float(bits | (sgn ? 0x80000000u : 0u));
            const float q = (float)((double)a * y);
            volatile float rr = res;
            const float ref = a / rr;
            mism += __float_as_uint(q) != __float_as_uint(ref);
            ++cnt;
        }
    }
    atomicAdd(bad, mism); atomicAdd(n, cnt);
}
__global__ void check(float res, uint32_t e_lo, uint32_t e_hi, unsigned long long* bad, unsigned long long* n)
{
    const float y = refined_rcp(res);
    unsigned long long mism = 0, cnt = 0;
    const uint64_t total = (uint64_t)(e_hi - e_lo + 1) << 23;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)(((uint64_t)e_lo << 23) + k);
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float a = __uint_as_float(bits | (sgn ? 0x80000000u : 0u));
            const float q = div_by(a, res, y);
            volatile float rr = res;
            const float ref = a / rr;
            mism += __float_as_uint(q) != __float_as_uint(ref);
            ++cnt;
        }
    }
    atomicAdd(bad, mism); atomicAdd(n, cnt);
}
__global__ void check_rpi(unsigned long long* bad, unsigned long long* n)
{
    unsigned long long mism = 0, cnt = 0;
    // every float in [0, 2^24]
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k <= 0x4b800000ull; k += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)k);
        int r;
        asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
        const long long ref = (long long)floor((double)x + 0.5);
        mism += (long long)r != ref;
        ++cnt;
    }
    atomicAdd(bad, mism); atomicAdd(n, cnt);
}
// the PPF's divisions (csrc/pair_mlp.hip:ppf_from): xy / (d + 1e-7f), |xy| <= d, d = a pair distance
__global__ void check_ppf(unsigned long long seed, unsigned long long* bad, unsigned long long* n)
{
    unsigned long long x = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long mism = 0, cnt = 0;
    for (int it = 0; it < 4096; ++it) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        // d: log-uniform over [2^-30, 2]; a fraction of exact zeros and tiny values
        const int e = (int)((x >> 40) % 32u);
        float d = __uint_as_float(((unsigned)(127 - 30 + e) << 23) | (unsigned)(x & 0x7fffffu));
        if ((x >> 60) == 0) d = 0.f;
        const float t = (float)((x >> 8) & 0xffffffu) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
        const float a = d * t;
        const float den = d + 1e-7f;
        const float y = refined_rcp(den);
        const float q = div_by(a, den, y);
        volatile float dd = den;
        const float ref = a / dd;
        // (a = -0 is left out: pa - pb of equal coordinates is +0, and div_by returns +0 where IEEE gives -0)
        mism += (__float_as_uint(q) != __float_as_uint(ref)) && __float_as_uint(a) != 0x80000000u;
        ++cnt;
    }
    atomicAdd(bad, mism); atomicAdd(n, cnt);
}
int main()
{
    {
        unsigned long long *d, h[2];
        hipMalloc(&d, 16); hipMemset(d, 0, 16);
        for (int r = 0; r < 8; ++r) check_ppf<<<8192, 256>>>(0x1234567ull + r * 77777ull, d, d + 1);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("PPF divisions xy / (d + 1e-7), d log-uniform in [2^-30, 2) or 0, |xy| <= d: %llu random operand pairs, %llu differ from a / b\n", h[1], h[0]);
    }
    {
        unsigned long long *d, h[2];
        hipMalloc(&d, 16); hipMemset(d, 0, 16);
        check_rpi<<<4096, 256>>>(d, d + 1);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("v_cvt_rpi_i32_f32: %llu floats in [0, 2^24], %llu differ from floor(x + 0.5) evaluated exactly\n", h[1], h[0]);
    }
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const float rs[] = {4e-3f, 1e-2f, 3e-2f, 2e-3f, 1.2e-3f, 0.0123f, 0.1f, 1.5e-2f, 8e-3f, 1e-3f, 0.25f, 7.77e-3f};
    for (float res : rs) {
        hipMemset(d, 0, 16);
        check<<<4096, 256>>>(res, 127 - 40, 127 + 6, d, d + 1);   // |a| in [2^-40, 2^7)
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("res %.9g: %llu values of a (both signs, |a| in [2^-40, 2^7)), %llu differ from a / res\n", res, h[1], h[0]);
    }
    // the fp64-product form: every normal a (both signs, |a| in [2^-100, 2^100)) for the same divisors and a few awkward ones
    const float rs64[] = {4e-3f, 1e-2f, 3e-2f, 2e-3f, 1.2e-3f, 0.0123f, 0.1f, 1.5e-2f, 8e-3f, 1e-3f, 0.25f, 7.77e-3f, 3.0f, 0.33333334f, 1.1754944e-3f,
                          0.99999994f, 1.0000001f, 5.9604645e-8f, 1.9999999f, 123.456f};
    for (float res : rs64) {
        hipMemset(d, 0, 16);
        check64<<<8192, 256>>>(res, 127 - 100, 127 + 99, d, d + 1);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("fp64-product form, res %.9g: %llu values of a (both signs, |a| in [2^-100, 2^100)), %llu differ from a / res\n", res, h[1], h[0]);
    }
    return 0;
}
