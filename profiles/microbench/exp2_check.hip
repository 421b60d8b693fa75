// Exhaustive check of the three gfx950 instructions the softmax weight of the decode is built from (csrc/cppf_math.h:det_exp2w)
// against their plain-arithmetic definitions, which is what oracle/cppf_oracle.c:orc_exp2w evaluates on the CPU:
//   v_fract_f32(y)        vs  min(y - floorf(y), 0x1.fffffep-1f)      (and vs the unclamped difference)
//   v_cvt_flr_i32_f32(y)  vs  (int)floorf(y)
//   v_ldexp_f32(p, e)     vs  the correctly rounded p * 2^e, subnormal results included (fp64 product, one conversion)
// and of csrc/cppf_math.h:sqrt_rn against sqrtf for x = 0 and every float in [2^-96, 2^40], and inv_sqrt_rn's reciprocal step against 1.0f / s on [2^-10, 2^20],
// over every float y in [-200, 2] (2.2e9 values; p = a mantissa in [1, 2.0000052) derived from y's bits).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-honor-nans exp2_check.hip -o exp2_check && ./exp2_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__global__ __launch_bounds__(256) void check_kernel(uint32_t first, uint32_t count, unsigned long long* bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < count;
    const uint32_t bits = first + (uint32_t)(live ? i : 0);
    auto tally = [&](int slot, bool cond) {   // one atomic per wave
        const unsigned long long m = __ballot(live && cond);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&bad[slot], (unsigned long long)__popcll(m));
    };
    const float y = __uint_as_float(bits);
    const float fa = __builtin_amdgcn_fractf(y);
    const float fl = floorf(y);
    const float fd = y - fl;
    const float fc = fminf(fd, 0x1.fffffep-1f);
    tally(0, fa != fd);
    tally(1, fa != fc);
    int e;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(e) : "v"(y));
    tally(2, e != (int)fl);
    // p in [1, 2.0000052): mantissa from a hash of the bits, every 64th value pushed to the top of the range
    uint32_t h = bits * 2654435761u;
    h ^= h >> 15;
    float p = __uint_as_float(0x3f800000u | (h & 0x7fffffu));
    if ((bits & 63u) == 0u) p = 2.0f + (float)(h & 15u) * 0x1p-22f;
    const float ra = __builtin_amdgcn_ldexpf(p, e);
    const double rd = (double)p * __longlong_as_double((long long)(1023 + (e < -1022 ? -1022 : e)) << 52);   // exact: 24-bit p, |e| <= 200
    const float rc = (float)rd;                                                                                // one rounding (RNE), subnormals included
    tally(3, __float_as_uint(ra) != __float_as_uint(rc));
    tally(4, ra != 0.f && ra < 1.17549435e-38f);   // how many subnormal results were seen
}

// csrc/cppf_math.h:sqrt_rn (the compiler's correctly rounded fp32 square root without its subnormal / inf / NaN handling) against sqrtf
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const unsigned sb = __float_as_uint(s);
    const float dn = __uint_as_float(max(sb, 1u) - 1u), up = __uint_as_float(sb + 1u);
    const float rdn = fmaf(-dn, s, x), rup = fmaf(-up, s, x);
    float r = rdn <= 0.f ? dn : s;
    r = rup > 0.f ? up : r;
    return r;
}
// cppf_math.h:inv_sqrt_rn's reciprocal step (refined_rcp + div_by with numerator 1) against the IEEE division 1.0f / s
__device__ __forceinline__ float refined_rcp_(float b) { const float y0 = __builtin_amdgcn_rcpf(b); const float e = fmaf(-b, y0, 1.0f); return fmaf(e, y0, y0); }
__device__ __forceinline__ float div_by_(float a, float b, float y)
{
    const float q0 = a * y; const float r0 = fmaf(-b, q0, a); const float q1 = fmaf(r0, y, q0); const float r1 = fmaf(-b, q1, a);
    return fmaf(r1, y, q1);
}
__global__ __launch_bounds__(256) void rcp_kernel(uint32_t first, uint32_t count, unsigned long long* bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < count;
    const float x = __uint_as_float(first + (uint32_t)(live ? i : 0));
    const unsigned long long m = __ballot(live && __float_as_uint(div_by_(1.0f, x, refined_rcp_(x))) != __float_as_uint(1.0f / x));
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&bad[7], (unsigned long long)__popcll(m));
}
__global__ __launch_bounds__(256) void sqrt_kernel(uint32_t first, uint32_t count, unsigned long long* bad, int slot)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < count;
    const float x = __uint_as_float(first + (uint32_t)(live ? i : 0));
    const unsigned long long m = __ballot(live && __float_as_uint(sqrt_rn(x)) != __float_as_uint(sqrtf(x)));
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&bad[slot], (unsigned long long)__popcll(m));
}

int main()
{
    unsigned long long* bad;
    hipMalloc(&bad, 8 * sizeof(unsigned long long));
    hipMemset(bad, 0, 8 * sizeof(unsigned long long));
    struct { uint32_t lo, hi; const char* what; } ranges[2] = {
        {0x80000000u, 0xC3480000u, "[-200, -0]"},
        {0x00000000u, 0x40000000u, "[+0, 2]"},
    };
    unsigned long long total = 0;
    for (auto& r : ranges) {
        uint64_t n = (uint64_t)r.hi - r.lo + 1;
        for (uint64_t off = 0; off < n; off += (1u << 30)) {
            const uint32_t cnt = (uint32_t)((n - off) < (1u << 30) ? (n - off) : (1u << 30));
            hipLaunchKernelGGL(check_kernel, dim3((cnt + 255) / 256), dim3(256), 0, 0, (uint32_t)(r.lo + off), cnt, bad);
        }
        total += n;
    }
    {   // x = +0 and every float in [2^-96, 2^40] (slot 5); the range the kernel does not claim, [2^-126, 2^-96) (slot 6)
        hipLaunchKernelGGL(sqrt_kernel, dim3(1), dim3(256), 0, 0, 0u, 1u, bad, 5);
        struct { uint32_t lo, hi; int slot; } sr[2] = {{0x0f800000u, 0x53800000u, 5}, {0x00800000u, 0x0f7fffffu, 6}};
        for (auto& r : sr) {
            const uint64_t n = (uint64_t)r.hi - r.lo + 1;
            for (uint64_t off = 0; off < n; off += (1u << 30)) {
                const uint32_t cnt = (uint32_t)((n - off) < (1u << 30) ? (n - off) : (1u << 30));
                hipLaunchKernelGGL(sqrt_kernel, dim3((cnt + 255) / 256), dim3(256), 0, 0, (uint32_t)(r.lo + off), cnt, bad, r.slot);
            }
        }
    }
    {   // every float in [2^-10, 2^20]
        const uint32_t lo = 0x3a800000u, hi = 0x49800000u;
        const uint64_t n = (uint64_t)hi - lo + 1;
        for (uint64_t off = 0; off < n; off += (1u << 30)) {
            const uint32_t cnt = (uint32_t)((n - off) < (1u << 30) ? (n - off) : (1u << 30));
            hipLaunchKernelGGL(rcp_kernel, dim3((cnt + 255) / 256), dim3(256), 0, 0, (uint32_t)(lo + off), cnt, bad);
        }
    }
    unsigned long long h[8];
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("floats checked: %llu (y in [-200, -0] and [+0, 2])\n", total);
    printf("v_fract_f32 != y - floorf(y)                      : %llu\n", h[0]);
    printf("v_fract_f32 != min(y - floorf(y), 0x1.fffffep-1f) : %llu\n", h[1]);
    printf("v_cvt_flr_i32_f32 != (int)floorf(y)               : %llu\n", h[2]);
    printf("v_ldexp_f32(p, e) != RNE(p * 2^e)                 : %llu  (subnormal results seen: %llu)\n", h[3], h[4]);
    printf("sqrt_rn(x) != sqrtf(x), x = 0 and [2^-96, 2^40]   : %llu\n", h[5]);
    printf("sqrt_rn(x) != sqrtf(x), [2^-126, 2^-96), not claimed: %llu of %u\n", h[6], 0x0f7fffffu - 0x00800000u + 1u);
    printf("div_by(1, s, refined_rcp(s)) != 1.0f / s, [2^-10, 2^20] : %llu\n", h[7]);
    return (h[1] == 0 || h[0] == 0) && h[2] == 0 && h[3] == 0 && h[5] == 0 && h[7] == 0 ? 0 : 1;
}
