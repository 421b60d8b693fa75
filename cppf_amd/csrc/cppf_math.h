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

// sqrtf for x = 0 or x >= 2^-96 (the squared distance of two points of a cloud: coincident, or >= 3.6e-15 m apart): the
// compiler's correctly rounded fp32 square root -- v_sqrt_f32 (1 ulp), then the neighbour below / above if its residual says
// so -- without the parts that only act below 2^-96 (rescaling by 2^32 and back: there the residual leaves the normal
// range) and on infinite or NaN arguments (the class test): 10 instructions instead of 17.  x = 0: the lower neighbour is
// kept at 0 (integer max), both residuals are zeros, the result is 0.  Checked against sqrtf for x = 0 and every float in
// [2^-96, 2^40]: profiles/microbench/exp2_check.hip, profiles/r4_exp2_check.txt (below 2^-96 it differs by an ulp for 1.6 % of
// the arguments; the oracle's sqrtf is exact there too -- such pairs do not occur in metric clouds).
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

// 1 / sqrtf(x) as the two correctly rounded operations it is (the LayerNorm's `1.0f / sqrtf(var + eps)`), for x >= 2^-20: sqrt_rn, then the
// quotient through refined_rcp + div_by -- 18 instructions instead of the 27 of sqrtf + the compiler's IEEE division; the same bits
// (the square root: every float of the range, above; the reciprocal: every float in [2^-10, 2^20], profiles/r4_exp2_check.txt)
__device__ __forceinline__ float inv_sqrt_rn(float x)
{
    const float s = sqrt_rn(x);
    return div_by(1.0f, s, refined_rcp(s));
}

// Softmax weight of a logit l under the shift c = -(max logit * log2e):  w = 2^y,  y = fma(l, log2e, c)  (<= 0 up to the
// rounding of the product), evaluated as ldexp(p(f), floor(y)) with f = y - floor(y) in [0, 1) and p the degree-4 minimax of
// 2^f on [0, 1] (relative error 2.7e-6 in fp32 Horner form -- far below what an inverse-CDF draw can resolve).
// Eight VALU per weight: the fma (max subtraction and log2e scale in one), v_fract_f32, v_cvt_flr_i32_f32, four fma, v_ldexp_f32
// -- no clamp (ldexp underflows through the subnormals to 0 by itself) and no separate subtraction (round 1-3's form -- rint
// through a magic add, exponent bits patched in by hand, x clamped at -86 -- took eleven).  On gfx950 fp32 MFMA and VALU
// share one datapath, so every decode instruction is paid in full.  The three instructions are checked exhaustively against
// their plain-arithmetic definitions (profiles/microbench/exp2_check.hip, profiles/r4_exp2_check.txt), which is what
// oracle/cppf_oracle.c:orc_exp2w evaluates.
#define CPPF_LOG2E 1.44269504088896341f
__device__ __forceinline__ float det_exp2w(float l, float c)
{
    const float y = fmaf(l, CPPF_LOG2E, c);
#ifdef CPPF_PRICE_V_EXP   // pricing aid only (never in the shipped library): the hardware's 2^x (v_exp_f32, ~1 ulp,
    return __builtin_amdgcn_exp2f(y);   // not reproducible on the host) instead of the exact polynomial: what bit-parity of the sampled bins costs
#endif
    const float f = __builtin_amdgcn_fractf(y);          // min(y - floor(y), 0x1.fffffep-1f)
    const int e = (int)floorf(y);                        // v_cvt_flr_i32_f32 (selected under -fno-honor-nans: csrc/Makefile)
    float p = 1.353416778e-02f;
    p = fmaf(p, f, 5.201146007e-02f);
    p = fmaf(p, f, 2.414427549e-01f);
    p = fmaf(p, f, 6.930038333e-01f);
    p = fmaf(p, f, 1.000002623e+00f);
    return __builtin_amdgcn_ldexpf(p, e);
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
    // length() and the divisions of :20-:28 in the forms that are exact on this domain and cost half: sqrt_rn (the radicand is 0 or
    // far above 2^-96 wherever the value is more than compared with 1e-7) and refined_rcp + div_by (numerators of at most a few
    // metres, denominators in [1e-7, ~2]: the PPF's case, checked exhaustively) -- 60 of the ~180 instructions of a pair's frame
    float L = sqrt_rn(dot3(ab, ab));
    if ((double)L < 1e-7) return false;
    {
        const float den = (float)((double)L + 1e-7), r = refined_rcp(den);
        ab = {div_by(ab.x, den, r), div_by(ab.y, den, r), div_by(ab.z, den, r)};
    }
    f3 co = {0.f, -ab.z, ab.y};
    float lc = sqrt_rn(dot3(co, co));
    if ((double)lc < 1e-7) { co = {-ab.y, ab.x, 0.f}; lc = sqrt_rn(dot3(co, co)); }
    {
        const float den = (float)((double)lc + 1e-7), r = refined_rcp(den);
        xdir = {div_by(co.x, den, r), div_by(co.y, den, r), div_by(co.z, den, r)};
    }
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
