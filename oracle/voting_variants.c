/*
 * voting_variants.c -- the three vote kernels of models/voting.py restated with the arithmetic FREEDOM the reference's
 * own toolchain has, to measure what that freedom can change.
 *
 * TEST INFRASTRUCTURE ONLY (see cppf_oracle.c's header): loaded by tests/ only.
 *
 * Why this exists.  The reference compiles its kernels at run time with NVRTC (cupy.RawKernel, models/voting.py:67,113,148:
 * the only option passed is `-I models/include`), so it runs with NVRTC's defaults: --fmad=true (a multiply feeding an add
 * may be contracted into one fused multiply-add, at the compiler's discretion, in NVVM or later in ptxas) and CUDA's device
 * cosf / sinf / tanf (documented maximum error 1-2 ulp for sin/cos, 4 ulp for tan; source not available).  Neither can be
 * executed here (no CUDA), and neither is specified tightly enough to be restated: WHICH products are fused is the compiler's
 * choice.  oracle/cppf_oracle.c therefore fixes one member of the family (no contraction, correctly rounded trigonometry) and
 * the HIP kernels follow it bit for bit.  This file evaluates OTHER members of the family:
 *
 *   ORV_FMAD_LEFT   every `p*q + r` becomes fma(p,q,r); of two products in a sum the LEFT one is fused (`p*q + r*s` ->
 *                   fma(p,q, r*s)), the combine order of LLVM's DAG combiner which NVVM derives from; a 3-term dot product
 *                   becomes fma(az,bz, fma(ax,bx, ay*by)); `x - p*q` -> fma(-p,q,x); `p*q - r*s` -> fma(p,q, -(r*s))
 *   ORV_FMAD_RIGHT  the other choice wherever two products meet: `p*q + r*s` -> fma(r,s, p*q); dot -> fma(az,bz,
 *                   fma(ay,by, ax*bx)); `p*q - r*s` -> fma(-r,s, p*q)
 *   ORV_LIBM        glibc cosf / sinf / tanf on the fp32 angle (another good libm, < 1 ulp)
 *   ORV_ULP_UP / ORV_ULP_DOWN / ORV_ULP_HASH
 *                   every cos / sin / tan result moved by +u / -u / a pseudo-random one of {-u, 0, +u} units in the last
 *                   place, u = bits 8..11 of the variant (0 -> 1): the envelope of a device libm that is "within u ulp"
 *
 * Contraction sites, per kernel (after inlining helper_math.cuh): length() = sqrtf(dot) at voting.py:21,22,27,28 / :87-94 /
 * :131-138; `a - ab * proj_len` (:23 / :89); cross(x, ab) (:29 / :95 / :139); `cos(angle) * x + sin(angle) * y` (:34 / :99 /
 * :141); length(pred_center - gt) (:101); `tan(rot) * offset + (+-ab)` and length(up) (:142-143).  Sites that CANNOT be
 * contracted (no product feeds an add): the `+ 1e-7` sums (fp64), `co / (..) * odist`, `(c + offset - corner) / res` (:35:
 * sums then an IEEE division, --prec-div=true is NVRTC's default), the adaptive trip count `int(odist / res * (2*M_PI))`
 * (:31: a division and a product), fracf, and the trilinear weights (products only).
 *
 * variant == 0 must reproduce cppf_oracle.c bit for bit (tests/test_oracle_variants.py checks it).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORV_FMAD_LEFT 1
#define ORV_FMAD_RIGHT 2
#define ORV_LIBM 4
#define ORV_ULP_UP 8
#define ORV_ULP_DOWN 16
#define ORV_ULP_HASH 32

#define ORV_PI 3.14159265358979323846264338327950288

void orc_sincos(double x, double* s, double* c);   /* cppf_oracle.c */

typedef struct { float x, y, z; } v3;

static inline int contract(int v) { return (v & (ORV_FMAD_LEFT | ORV_FMAD_RIGHT)) != 0; }

/* p*q + r */
static inline float mad(int v, float p, float q, float r) { return contract(v) ? fmaf(p, q, r) : p * q + r; }
/* x - p*q */
static inline float xmsub(int v, float x, float p, float q) { return contract(v) ? fmaf(-p, q, x) : x - p * q; }
/* p*q + r*s */
static inline float mad2(int v, float p, float q, float r, float s)
{
    if (v & ORV_FMAD_RIGHT) return fmaf(r, s, p * q);
    if (v & ORV_FMAD_LEFT) return fmaf(p, q, r * s);
    return p * q + r * s;
}
/* p*q - r*s */
static inline float msub2(int v, float p, float q, float r, float s)
{
    if (v & ORV_FMAD_RIGHT) return fmaf(-r, s, p * q);
    if (v & ORV_FMAD_LEFT) return fmaf(p, q, -(r * s));
    return p * q - r * s;
}
/* helper_math.cuh:1245 dot = a.x*b.x + a.y*b.y + a.z*b.z */
static inline float vdot(int v, v3 a, v3 b)
{
    if (!contract(v)) return (a.x * b.x + a.y * b.y) + a.z * b.z;
    return fmaf(a.z, b.z, mad2(v, a.x, b.x, a.y, b.y));
}
static inline float vlen(int v, v3 a) { return sqrtf(vdot(v, a, a)); }
static inline v3 vsub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 vadd(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 vscl(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline v3 vdiv(v3 a, float s) { v3 r = {a.x / s, a.y / s, a.z / s}; return r; }
static inline v3 vneg(v3 a) { v3 r = {-a.x, -a.y, -a.z}; return r; }
static inline v3 vld(const float* p, int64_t i) { v3 r = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; return r; }
/* helper_math.cuh:1417 */
static inline v3 vcross(int v, v3 a, v3 b)
{
    v3 r = {msub2(v, a.y, b.z, a.z, b.y), msub2(v, a.z, b.x, a.x, b.z), msub2(v, a.x, b.y, a.y, b.x)};
    return r;
}

static inline int sat_int(double v)
{
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

static inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
static inline float nudge(float x, int steps)
{
    for (; steps > 0; --steps) x = nextafterf(x, INFINITY);
    for (; steps < 0; ++steps) x = nextafterf(x, -INFINITY);
    return x;
}
static inline float ulp_variant(int v, float x, uint32_t key0, uint32_t key1, uint32_t which)
{
    int u = (v >> 8) & 15;
    if (u == 0) u = 1;
    if (v & ORV_ULP_UP) return nudge(x, u);
    if (v & ORV_ULP_DOWN) return nudge(x, -u);
    if (v & ORV_ULP_HASH) return nudge(x, ((int)(hash3(key0, key1, which) % 3u) - 1) * u);
    return x;
}

/* cos / sin of the i-th of n rotation angles: angle = float(i*2*M_PI/n) (fp64 expression rounded to fp32, :33) */
static inline void rot_cs_v(int v, int i, int n, float* cs, float* sn)
{
    float angle = (float)((double)(i * 2) * ORV_PI / (double)n);
    if (v & ORV_LIBM) {
        *cs = cosf(angle);
        *sn = sinf(angle);
    } else {
        double s, c;
        orc_sincos((double)angle, &s, &c);
        *cs = (float)c;
        *sn = (float)s;
    }
    *cs = ulp_variant(v, *cs, (uint32_t)i, (uint32_t)n, 0u);
    *sn = ulp_variant(v, *sn, (uint32_t)i, (uint32_t)n, 1u);
}
static inline float tan_v(int v, float rot)
{
    float t;
    if (v & ORV_LIBM) t = tanf(rot);
    else {
        double s, c;
        orc_sincos((double)rot, &s, &c);
        t = (float)(s / c);
    }
    uint32_t b;
    memcpy(&b, &rot, 4);
    return ulp_variant(v, t, b, 77u, 2u);
}

/* shared front half (voting.py:15-29 / :81-95 / :125-136); 0 = degenerate pair (early return) */
static inline int frame_v(int v, const float* points, const int32_t* point_idxs, int64_t idx, v3* a_out, v3* ab_out, v3* xd_out)
{
    v3 a = vld(points, point_idxs[2 * idx]), b = vld(points, point_idxs[2 * idx + 1]);
    v3 ab = vsub(a, b);
    float L = vlen(v, ab);
    if ((double)L < 1e-7) return 0;
    ab = vdiv(ab, (float)((double)L + 1e-7));
    v3 co = {0.f, -ab.z, ab.y};
    if ((double)vlen(v, co) < 1e-7) { co.x = -ab.y; co.y = ab.x; co.z = 0.f; }
    *xd_out = vdiv(co, (float)((double)vlen(v, co) + 1e-7));
    *a_out = a;
    *ab_out = ab;
    return 1;
}

typedef struct { int ok, n; v3 c, x, y; } circle_t;

/* everything of a pair up to the rotation loop of ppf_voting / backvote (:15-31 / :81-97) */
static inline circle_t circle_v(int v, const float* points, const float* outputs, const int32_t* point_idxs, int64_t idx,
                                float res, int n_rots, int adaptive)
{
    circle_t k;
    memset(&k, 0, sizeof k);
    float proj_len = outputs[2 * idx], odist = outputs[2 * idx + 1];
    v3 a, ab, xd;
    k.ok = frame_v(v, points, point_idxs, idx, &a, &ab, &xd);
    k.n = 0;
    if (!k.ok) return k;
    k.c.x = xmsub(v, a.x, ab.x, proj_len);          /* a - ab * proj_len */
    k.c.y = xmsub(v, a.y, ab.y, proj_len);
    k.c.z = xmsub(v, a.z, ab.z, proj_len);
    k.x = vscl(xd, odist);
    k.y = vcross(v, k.x, ab);
    k.n = n_rots;
    if (adaptive) {
        int m = sat_int((double)(odist / res) * (2 * ORV_PI));
        k.n = m < n_rots ? m : n_rots;
    }
    return k;
}
static inline v3 offset_v(int v, const circle_t* k, float cs, float sn)
{
    v3 o = {mad2(v, cs, k->x.x, sn, k->y.x), mad2(v, cs, k->x.y, sn, k->y.y), mad2(v, cs, k->x.z, sn, k->y.z)};
    return o;
}
static inline int in_grid(v3 g, int gx, int gy, int gz)
{
    return !((double)g.x < 0.01 || (double)g.y < 0.01 || (double)g.z < 0.01 || (double)g.x >= (double)gx - 1.01 ||
             (double)g.y >= (double)gy - 1.01 || (double)g.z >= (double)gz - 1.01);
}

/* models/voting.py:8-66 under `variant`, accumulated in fp64 (the exact sum every ordering of the reference's fp32
 * atomicAdd approximates), OpenMP over pair slices with private grids.  grid is ADDED to. */
void orv_ppf_voting_f64(const float* points, const float* outputs, const float* probs, const int32_t* point_idxs,
                        double* grid, const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz,
                        int adaptive, int variant, int threads)
{
    const int v = variant;
    const size_t G = (size_t)gx * gy * gz;
    if (threads < 1) threads = 1;
    double* priv = threads > 1 ? (double*)calloc((size_t)threads * G, sizeof(double)) : grid;
    const v3 cr = {corner[0], corner[1], corner[2]};
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int t = 0, nt = 1;
#endif
        double* g_ = threads > 1 ? priv + (size_t)t * G : grid;
        const int64_t lo = n_ppfs * t / nt, hi = n_ppfs * (t + 1) / nt;
        for (int64_t idx = lo; idx < hi; ++idx) {
            circle_t k = circle_v(v, points, outputs, point_idxs, idx, res, n_rots, adaptive);
            if (!k.ok) continue;
            float pa = probs[point_idxs[2 * idx]], pb = probs[point_idxs[2 * idx + 1]];
            float prob = pa > pb ? pa : pb;
            for (int i = 0; i < k.n; ++i) {
                float cs, sn;
                rot_cs_v(v, i, k.n, &cs, &sn);
                v3 g = vdiv(vsub(vadd(k.c, offset_v(v, &k, cs, sn)), cr), res);
                if (!in_grid(g, gx, gy, gz)) continue;
                int fx = (int)g.x, fy = (int)g.y, fz = (int)g.z;
                float rx = g.x - floorf(g.x), ry = g.y - floorf(g.y), rz = g.z - floorf(g.z);
                float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
                float w[8] = {w0x * w0y * w0z * prob, w0x * w0y * rz * prob, w0x * ry * w0z * prob, w0x * ry * rz * prob,
                              rx * w0y * w0z * prob,  rx * w0y * rz * prob,  rx * ry * w0z * prob,  rx * ry * rz * prob};
                int64_t syz = (int64_t)gy * gz, b = fx * syz + fy * gz + fz;
                int64_t off[8] = {0, 1, gz, gz + 1, syz, syz + 1, syz + gz, syz + gz + 1};
                for (int q = 0; q < 8; ++q) g_[b + off[q]] += (double)w[q];
            }
        }
    }
    if (threads > 1) {
        for (int t = 0; t < threads; ++t)
            for (size_t i = 0; i < G; ++i) grid[i] += priv[(size_t)t * G + i];
        free(priv);
    }
}

/* Sample-by-sample comparison of `variant` against variant 0 on the same inputs.
 *  out[0] pairs whose degenerate test (:21) differs      out[1] pairs whose trip count (:31) differs
 *  out[2] samples compared (pairs alive under both, same trip count)
 *  out[3] samples in the grid under variant 0            out[4] samples whose in-grid test (:36-39) flips
 *  out[5] samples in the grid under both whose floor cell (:40) differs
 *  dmax[0] largest |difference| of a grid coordinate (in cells) over the compared samples */
void orv_vote_flips(const float* points, const float* outputs, const int32_t* point_idxs, const float* corner, float res,
                    int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive, int variant, int64_t* out, double* dmax)
{
    int64_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0;
    double dm = 0.0;
    const v3 cr = {corner[0], corner[1], corner[2]};
#pragma omp parallel for schedule(static) reduction(+ : o0, o1, o2, o3, o4, o5) reduction(max : dm)
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        circle_t k0 = circle_v(0, points, outputs, point_idxs, idx, res, n_rots, adaptive);
        circle_t k1 = circle_v(variant, points, outputs, point_idxs, idx, res, n_rots, adaptive);
        if (k0.ok != k1.ok) { o0 += 1; continue; }
        if (!k0.ok) continue;
        if (k0.n != k1.n) { o1 += 1; continue; }
        for (int i = 0; i < k0.n; ++i) {
            float c0, s0, c1, s1;
            rot_cs_v(0, i, k0.n, &c0, &s0);
            rot_cs_v(variant, i, k0.n, &c1, &s1);
            v3 g0 = vdiv(vsub(vadd(k0.c, offset_v(0, &k0, c0, s0)), cr), res);
            v3 g1 = vdiv(vsub(vadd(k1.c, offset_v(variant, &k1, c1, s1)), cr), res);
            int in0 = in_grid(g0, gx, gy, gz), in1 = in_grid(g1, gx, gy, gz);
            o2 += 1;
            o3 += in0;
            double d = fmax(fmax(fabs((double)g0.x - g1.x), fabs((double)g0.y - g1.y)), fabs((double)g0.z - g1.z));
            if (d > dm) dm = d;
            if (in0 != in1) { o4 += 1; continue; }
            if (in0 && ((int)g0.x != (int)g1.x || (int)g0.y != (int)g1.y || (int)g0.z != (int)g1.z)) o5 += 1;
        }
    }
    out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3; out[4] = o4; out[5] = o5;
    dmax[0] = dm;
}

/* models/voting.py:74-112 under `variant` (+ the mask of nocs/inference.py:230); out_offsets zero-initialised by the caller */
void orv_backvote(const float* points, const float* outputs, float* out_offsets, const int32_t* point_idxs,
                  const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz, const float* gt_center,
                  float tol, uint8_t* mask, int variant)
{
    const int v = variant;
    const v3 cr = {corner[0], corner[1], corner[2]}, gt = {gt_center[0], gt_center[1], gt_center[2]};
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        float* oo = out_offsets + 3 * idx;
        circle_t k = circle_v(v, points, outputs, point_idxs, idx, res, n_rots, 1);   /* :97 always adaptive */
        if (k.ok) {
            oo[0] = oo[1] = oo[2] = 0.f;
            for (int i = 0; i < k.n; ++i) {
                float cs, sn;
                rot_cs_v(v, i, k.n, &cs, &sn);
                v3 offset = offset_v(v, &k, cs, sn);
                v3 pc_ = vadd(k.c, offset);
                if (vlen(v, vsub(pc_, gt)) > tol) continue;
                v3 g = vdiv(vsub(pc_, cr), res);
                if (g.x < 0 || g.y < 0 || g.z < 0 || g.x >= (float)(gx - 1) || g.y >= (float)(gy - 1) || g.z >= (float)(gz - 1))
                    continue;
                oo[0] = -offset.x; oo[1] = -offset.y; oo[2] = -offset.z;
                break;
            }
        }
        if (mask) mask[idx] = (oo[0] != 0.f) || (oo[1] != 0.f) || (oo[2] != 0.f);
    }
}

/* models/voting.py:119-147 under `variant`; outputs_up [n_ppfs, n_rots, 3] zero-initialised by the caller */
void orv_rot_voting(const float* points, const float* preds_rot, float* outputs_up, const int32_t* point_idxs, int64_t n_ppfs,
                    int n_rots, int variant)
{
    const int v = variant;
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        v3 a, ab, x;
        if (!frame_v(v, points, point_idxs, idx, &a, &ab, &x)) continue;
        circle_t k;
        memset(&k, 0, sizeof k);
        k.x = x;
        k.y = vcross(v, x, ab);
        float t = tan_v(v, preds_rot[idx]);
        v3 base = t > 0 ? ab : vneg(ab);
        for (int i = 0; i < n_rots; ++i) {
            float cs, sn;
            rot_cs_v(v, i, n_rots, &cs, &sn);
            v3 offset = offset_v(v, &k, cs, sn);
            v3 up = {mad(v, t, offset.x, base.x), mad(v, t, offset.y, base.y), mad(v, t, offset.z, base.z)};
            up = vdiv(up, (float)((double)vlen(v, up) + 1e-7));
            float* o = outputs_up + ((size_t)idx * n_rots + i) * 3;
            o[0] = up.x; o[1] = up.y; o[2] = up.z;
        }
    }
}
