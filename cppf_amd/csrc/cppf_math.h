// Device-side arithmetic shared by the CPPF kernels (gfx950 only).
//
// The library is compiled with -ffp-contract=off: every fused multiply-add below is an explicit
// fmaf()/fma(), every other a*b+c is two roundings.  fp32 divide and sqrt are the correctly
// rounded forms (hipcc default).  cos/sin/tan/exp are fixed polynomial evaluations rather than
// OCML calls so that the discrete outcomes of the vote (trip counts, in/out-of-grid tests, sampled
// bins) do not depend on a vendor math library; their definitions are restated independently in
// oracle/cppf_oracle.c, which is what the parity tests compare against.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CPPF_PI 3.14159265358979323846264338327950288  // reference models/voting.py:6

namespace cppf {

struct f3 { float x, y, z; };

// float3 helper semantics of the reference (models/include/helper_math.cuh:811,994,1245,1288,1417)
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float len3(f3 v) { return sqrtf(dot3(v, v)); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 scl3(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 div3(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ f3 neg3(f3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// a / b for a denominator shared by several divisions (loop-invariant `res` in the vote, the pair distance in the PPF): the compiler's IEEE-exact fp32 division (v_div_scale, v_rcp, two Newton steps on the
// reciprocal, q0, residual, q1, residual, v_div_fmas, v_div_fixup) minus the parts that only act outside the normal
// range -- the reciprocal refinement is hoisted (refined_rcp), operands are never rescaled (|a| <= a few metres over
// res ~ 1e-3..1e-1: no scaling would be applied) and inf/NaN/0 fix-ups are not needed because such coordinates fail
// the bound tests either way.  Same result bit for bit on that range (checked exhaustively on the device against `/`:
// profiles/r2_div_check.txt); 5 instructions per division instead of 9.
__device__ __forceinline__ float refined_rcp(float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, y0, 1.0f);
    return fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_by(float a, float b, float y)
{
    const float q0 = a * y;
    const float r0 = fmaf(-b, q0, a);
    const float q1 = fmaf(r0, y, q0);
    const float r1 = fmaf(-b, q1, a);
    return fmaf(r1, y, q1);
}

__device__ __forceinline__ f3 ld3(const float* __restrict__ p, int i)
{
    return {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
}

// exp(x) for x <= 0 (softmax after the max subtraction; callers guarantee -3e38 < x <= 0):
// 2^(x*log2e) = 2^n * p(f), n = rint(x*log2e), f = x*log2e - n in [-0.5, 0.5], p = degree-4 minimax of 2^f.
// n comes from the float's own rounding: t = fma(x, log2e, 1.5*2^23) has unit spacing, so t - 1.5*2^23 = n exactly and
// the low bits of t's pattern hold n in two's complement -- (bits(t) << 23) IS n << 23, no rint, no float->int conversion;
// f = fma(x, log2e, -n) is rounded once.  Max relative error 3e-6 -- far below what an inverse-CDF draw can resolve -- at
// 8 VALU (5.5 per value on the packed pipe): on gfx950 fp32 MFMA and VALU share one datapath, so every decode
// instruction is paid in full.  oracle/cppf_oracle.c:orc_expf is the same sequence.
#define CPPF_EXP_MAGIC 12582912.0f   // 1.5 * 2^23
__device__ __forceinline__ float det_expf(float x)
{
    x = fmaxf(x, -86.0f);  // keeps 2^n a normal number for the exponent arithmetic below
    const float t = fmaf(x, 1.44269504088896341f, CPPF_EXP_MAGIC);
    const float n = t - CPPF_EXP_MAGIC;
    const float f = fmaf(x, 1.44269504088896341f, -n);
    float p = 9.570102207e-03f;
    p = fmaf(p, f, 5.591785908e-02f);
    p = fmaf(p, f, 2.402474433e-01f);
    p = fmaf(p, f, 6.931217909e-01f);
    p = fmaf(p, f, 9.999992847e-01f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t) << 23));
}

// Two det_expf at once on the packed-fp32 pipe (v_pk_add_f32 / v_pk_fma_f32 are IEEE per component, so each half is
// bit-identical to det_expf).
typedef float cppf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cppf_f32x2 det_expf2(cppf_f32x2 x)
{
    x[0] = fmaxf(x[0], -86.0f); x[1] = fmaxf(x[1], -86.0f);
    const cppf_f32x2 L = {1.44269504088896341f, 1.44269504088896341f}, M = {CPPF_EXP_MAGIC, CPPF_EXP_MAGIC};
    const cppf_f32x2 t = __builtin_elementwise_fma(x, L, M);
    const cppf_f32x2 n = t - M;
    const cppf_f32x2 f = __builtin_elementwise_fma(x, L, -n);
    cppf_f32x2 p = {9.570102207e-03f, 9.570102207e-03f};
    p = __builtin_elementwise_fma(p, f, cppf_f32x2{5.591785908e-02f, 5.591785908e-02f});
    p = __builtin_elementwise_fma(p, f, cppf_f32x2{2.402474433e-01f, 2.402474433e-01f});
    p = __builtin_elementwise_fma(p, f, cppf_f32x2{6.931217909e-01f, 6.931217909e-01f});
    p = __builtin_elementwise_fma(p, f, cppf_f32x2{9.999992847e-01f, 9.999992847e-01f});
    cppf_f32x2 r;
    r[0] = __uint_as_float(__float_as_uint(p[0]) + (__float_as_uint(t[0]) << 23));
    r[1] = __uint_as_float(__float_as_uint(p[1]) + (__float_as_uint(t[1]) << 23));
    return r;
}

// fp64 sin/cos: Cody-Waite by pi/2 + degree-13/14 kernels on [-pi/4, pi/4].
__device__ __forceinline__ void det_sincos(double x, double* s, double* c)
{
    double k = rint(x * 0.63661977236758134308);
    double y = fma(-k, 1.57079632673412561417e+00, x);
    y = fma(-k, 6.07710050650619224932e-11, y);
    double z = y * y;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    double sn = fma(y * z, ps, y);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    int q = ((int)k) & 3;
    double so = (q & 1) ? cs : sn;
    double co = (q & 1) ? sn : cs;
    *s = (q & 2) ? -so : so;
    *c = (q == 1 || q == 2) ? -co : co;
}

// cos/sin of rotation i of n: angle = float(i*2*M_PI/n) evaluated in fp64 then rounded to fp32
// (reference models/voting.py:33), cos/sin of that fp32 angle rounded to fp32.
__device__ __forceinline__ float2 rot_cs(int i, int n)
{
    float angle = (float)((double)(i * 2) * CPPF_PI / (double)n);
    double s, c;
    det_sincos((double)angle, &s, &c);
    return make_float2((float)c, (float)s);
}

__device__ __forceinline__ float det_tanf(float rot)
{
    double s, c;
    det_sincos((double)rot, &s, &c);
    return (float)(s / c);
}

// Front half shared by ppf_voting / backvote / rot_voting (reference models/voting.py:15-29,
// 81-95, 125-136): unit ab with the fp64 "+1e-7", and the unit in-plane direction x.
// Returns false for a degenerate pair (the reference returns early).
__device__ __forceinline__ bool pair_frame(const float* __restrict__ points, int a_idx, int b_idx, f3& a,
                                           f3& ab, f3& xdir)
{
    a = ld3(points, a_idx);
    f3 b = ld3(points, b_idx);
    ab = sub3(a, b);
    float L = len3(ab);
    if ((double)L < 1e-7) return false;
    ab = div3(ab, (float)((double)L + 1e-7));
    f3 co = {0.f, -ab.z, ab.y};
    if ((double)len3(co) < 1e-7) co = {-ab.y, ab.x, 0.f};
    xdir = div3(co, (float)((double)len3(co) + 1e-7));
    return true;
}

// total order on floats as unsigned (monotone): larger float -> larger key
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}

}  // namespace cppf
