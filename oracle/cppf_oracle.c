/*
 * cppf_oracle.c -- CPU restatement of the CPPF point-pair-feature + voting hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product (cppf_amd/) never imports it.
 *
 * What it restates (reference = qq456cvb/CPPF, paths relative to the reference root):
 *   orc_pair_mlp         models/model.py:117-137 (PPFEncoder.forward_with_idx) + :27-31 (ResLayer)
 *   orc_decode_*         nocs/inference.py:185-188, :245-256 (softmax + sample + bin->value)
 *   orc_grid_setup       nocs/inference.py:194-196
 *   orc_ppf_voting       models/voting.py:8-66
 *   orc_grid_argmax      nocs/inference.py:207-211
 *   orc_backvote         models/voting.py:74-112 (+ mask of nocs/inference.py:229-231)
 *   orc_rot_voting       models/voting.py:119-147
 *   orc_sphere_count     nocs/inference.py:276-284
 *   orc_axis_sign        nocs/inference.py:287-303
 *   orc_scale            nocs/inference.py:335
 *   float3 helpers       models/include/helper_math.cuh:157,994-1003,811-818,1245,1288,1336,1417
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_pair_mlp is pinned against tests/golden/mlp_*.npz, produced by importing the
 *     reference's own models/model.py on CPU (tests/golden/make_golden.py).
 *   - the closed-form (mu, nu) targets and the Fibonacci sphere are pinned against the
 *     reference's own utils/dataset.py:generate_target / utils/util.py:fibonacci_sphere,
 *     executed from their source text (same script).
 *   - the three vote kernels exist in the reference only as CUDA text inside CuPy RawKernel
 *     strings whose header needs the CUDA toolkit; that is unbuildable in this image without
 *     writing stand-ins, so for orc_ppf_voting / orc_backvote / orc_rot_voting:
 *     PARITY UNPINNED against executed reference code.  They are pinned only through the
 *     closed-form known answer (every vote circle of an exact (mu,nu) passes through the
 *     object centre, utils/dataset.py:27-36) and by line-by-line citation below.
 *
 * Arithmetic conventions (they are what makes the HIP path comparable bit-for-bit):
 *   - compiled with -ffp-contract=off; every fused multiply-add is an explicit fmaf()/fma().
 *   - fp64 sub-expressions of the CUDA text (1e-7, 0.01, 1.01, M_PI literals) are kept in fp64.
 *   - cos/sin/tan/exp are NOT libm: orc_sincos()/orc_exp2w() are fixed polynomial evaluations
 *     (sin/cos: < 1 ulp of fp64; exp: 2.7e-6 relative, it only feeds a softmax sampler) so that CPU and GPU produce the same bits and therefore the same
 *     discrete outcomes (trip counts, in/out-of-grid tests, sampled bins).
 *   - the MLP accumulates each output with an fmaf chain seeded by the bias.  order=0 walks
 *     k = 0..K-1; order=1 walks k in the order the fp32 MFMA tiles of the HIP kernel consume
 *     it, incl. the per-point hoisting of the first layer (documented at orc_k_order).  Both are
 *     within ~1e-6 of torch's GEMM.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846264338327950288 /* models/voting.py:6 */

typedef struct { float x, y, z; } f3;

/* ------------------------------------------------------------------ helpers */
/* helper_math.cuh:1245 dot = a.x*b.x + a.y*b.y + a.z*b.z (left to right, no contraction) */
static inline float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* helper_math.cuh:1288 */
static inline float len3(f3 v) { return sqrtf(dot3(v, v)); }
static inline f3 sub3(f3 a, f3 b) { f3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline f3 add3(f3 a, f3 b) { f3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline f3 scl3(f3 a, float s) { f3 r = {a.x * s, a.y * s, a.z * s}; return r; }   /* :811 */
static inline f3 div3(f3 a, float s) { f3 r = {a.x / s, a.y / s, a.z / s}; return r; }   /* :994 */
static inline f3 neg3(f3 a) { f3 r = {-a.x, -a.y, -a.z}; return r; }
/* helper_math.cuh:1417 */
static inline f3 cross3(f3 a, f3 b)
{
    f3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static inline f3 ld3(const float* p, int64_t i) { f3 r = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; return r; }

/* double -> int like the GPU's conversion (CUDA cvt.rzi.s32.f64 / gfx950 v_cvt_i32_f64): truncate,
 * saturate, NaN -> 0.  Plain C leaves the out-of-range case undefined. */
static inline int sat_int(double v)
{
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}
static inline float bits_f(int32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline int32_t f_bits(float f) { int32_t b; memcpy(&b, &f, 4); return b; }

/* Softmax weight of a logit l under the shift c = -(max logit * log2e):  w = 2^y, y = fmaf(l, log2e, c) (<= 0 up to the
 * rounding of the product), as ldexp(p(f), floor(y)): f = y - floor(y), kept below 1 the way v_fract_f32 keeps it (the
 * difference rounds to 1.0 for tiny negative y), p = degree-4 minimax of 2^f on [0, 1] in Horner form, one fmaf per step
 * (relative error 2.7e-6: far below what an inverse-CDF draw can resolve).  ldexpf rounds a subnormal result once, to
 * nearest even, like v_ldexp_f32; no clamp anywhere.  The device form (csrc/cppf_math.h:det_exp2w) uses v_fract_f32,
 * v_cvt_flr_i32_f32 and v_ldexp_f32 for the three non-fma steps; profiles/microbench/exp2_check.hip checks those three
 * instructions against exactly the expressions below for every float y in [-200, 2]. */
#define ORC_LOG2E 1.44269504088896341f
float orc_exp2w(float l, float c)
{
    const float y = fmaf(l, ORC_LOG2E, c);
    const float fl = floorf(y);
    float f = y - fl;
    if (f >= 1.0f) f = 0x1.fffffep-1f;
    const int e = fl < -100000.0f ? -100000 : (fl > 100000.0f ? 100000 : (int)fl);
    float p = 1.353416778e-02f;
    p = fmaf(p, f, 5.201146007e-02f);
    p = fmaf(p, f, 2.414427549e-01f);
    p = fmaf(p, f, 6.930038333e-01f);
    p = fmaf(p, f, 1.000002623e+00f);
    return ldexpf(p, e);
}
/* exp(x) through the same core (x <= 0): what a softmax weight is for a maximum of 0 */
float orc_expf(float x) { return orc_exp2w(x, 0.0f); }

/* sin and cos of x in fp64: Cody-Waite reduction by pi/2 (two terms; exact for |x| < ~1e5)
 * and the classic degree-13/14 minimax kernels on [-pi/4, pi/4].  < 1 ulp(fp64). */
void orc_sincos(double x, double* s, double* c)
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
    switch (((int)k) & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}

/* cos/sin of the i-th of n rotation angles: angle = float(i*2*M_PI/n) (fp64 expression rounded
 * to fp32, models/voting.py:33), then cos(float)/sin(float) -> fp32 results. */
static inline void rot_cs(int i, int n, float* cs, float* sn)
{
    float angle = (float)((double)(i * 2) * ORC_PI / (double)n);
    double s, c;
    orc_sincos((double)angle, &s, &c);
    *cs = (float)c;
    *sn = (float)s;
}
void orc_rot_cs(int i, int n, float* cs, float* sn) { rot_cs(i, n, cs, sn); }

/* tan(rot) as fp32 (models/voting.py:142) */
static inline float orc_tanf(float rot)
{
    double s, c;
    orc_sincos((double)rot, &s, &c);
    return (float)(s / c);
}
float orc_tan(float rot) { return orc_tanf(rot); }

/* ------------------------------------------------------------------ PPF + MLP */

/* PPF of one pair, models/model.py:118-129: xy = pc[a]-pc[b]; d = ||xy||; u = xy/(d+1e-7) with
 * an fp32 add; ppf = [n_a.u, n_b.u, n_a.n_b, d].  torch.norm / torch.sum reduce 3 elements
 * left to right. */
static inline void ppf4(const float* pc, const float* nrm, int64_t a, int64_t b, float out[4])
{
    f3 pa = ld3(pc, a), pb = ld3(pc, b), na = ld3(nrm, a), nb = ld3(nrm, b);
    f3 xy = sub3(pa, pb);
    float d = sqrtf((xy.x * xy.x + xy.y * xy.y) + xy.z * xy.z);
    float den = d + 1e-7f;
    f3 u = {xy.x / den, xy.y / den, xy.z / den};
    out[0] = (na.x * u.x + na.y * u.y) + na.z * u.z;
    out[1] = (nb.x * u.x + nb.y * u.y) + nb.z * u.z;
    out[2] = (na.x * nb.x + na.y * nb.y) + na.z * nb.z;
    out[3] = d;
}

void orc_ppf_features(const float* pc, const float* nrm, const int64_t* idxs, int64_t P, float* out)
{
    for (int64_t i = 0; i < P; ++i) ppf4(pc, nrm, idxs[2 * i], idxs[2 * i + 1], out + 4 * i);
}

/*
 * Order in which an output's fmaf chain walks its K inputs.
 *  order 0: k = 0..K-1.
 *  order 1 ("mfma"): the HIP kernel feeds v_mfma_f32_16x16x4_f32, which consumes 4 k-values
 *    per instruction (one per 16-lane group g = 0..3, accumulated g = 0,1,2,3) in steps s:
 *      first layer (K = 2F+4): it is linear in cat(feat[a], feat[b], ppf), and the HIP path hoists the
 *        two F-wide blocks out of the pair loop (csrc/pair_mlp.hip): per POINT n and output o
 *           TA[n][o] = fmaf chain over k = 0..F-1 of W[o][k]   * feat[n][k], seeded with the bias
 *           TB[n][o] = fmaf chain over k = 0..F-1 of W[o][F+k] * feat[n][k], seeded with 0
 *        and per pair  y[o] = fmaf chain over k = 0..3 of W[o][2F+k] * ppf[k], seeded with TA[a][o] + TB[b][o]
 *        (handled in orc_pair_mlp, not through a permutation)
 *      hidden layers (K % 16 == 0): step s, group g reads input 16*(s/4) + 4*g + (s%4)
 *        (= the register the previous layer's MFMA left in that lane: no data movement).
 *    Falls back to order 0 when the shape does not fit.
 */
static void orc_k_order(int K, int F, int first, int order, int* perm)
{
    (void)F;
    int ok = 0;
    if (order == 1) {
        if (!first && K % 16 == 0) {
            int n = 0;
            for (int s = 0; s < K / 4; ++s)
                for (int g = 0; g < 4; ++g) perm[n++] = 16 * (s / 4) + 4 * g + (s % 4);
            ok = 1;
        }
    }
    if (!ok)
        for (int k = 0; k < K; ++k) perm[k] = k;
}

/* y[o] = b[o] (+) sum_k W[o][k] x[k] as an fmaf chain in perm order; Wt is [K][Nout] (k-major
 * copy of torch's [Nout][K] weight).  With AVX2+FMA the chain runs 8 outputs per register, 4 registers
 * at a time: _mm256_fmadd_ps is fmaf per lane, so the result is bit-identical to the scalar loop. */
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
static inline void linear_chain(const float* restrict Wt, const float* restrict b, const float* restrict x,
                                const int* restrict perm, int K, int Nout, float* restrict y)
{
    int o = 0;
    for (; o + 32 <= Nout; o += 32) {
        __m256 a0 = _mm256_loadu_ps(b + o), a1 = _mm256_loadu_ps(b + o + 8), a2 = _mm256_loadu_ps(b + o + 16),
               a3 = _mm256_loadu_ps(b + o + 24);
        for (int kk = 0; kk < K; ++kk) {
            const int k = perm[kk];
            const __m256 xv = _mm256_set1_ps(x[k]);
            const float* w = Wt + (size_t)k * Nout + o;
            a0 = _mm256_fmadd_ps(_mm256_loadu_ps(w), xv, a0);
            a1 = _mm256_fmadd_ps(_mm256_loadu_ps(w + 8), xv, a1);
            a2 = _mm256_fmadd_ps(_mm256_loadu_ps(w + 16), xv, a2);
            a3 = _mm256_fmadd_ps(_mm256_loadu_ps(w + 24), xv, a3);
        }
        _mm256_storeu_ps(y + o, a0); _mm256_storeu_ps(y + o + 8, a1);
        _mm256_storeu_ps(y + o + 16, a2); _mm256_storeu_ps(y + o + 24, a3);
    }
    for (; o + 8 <= Nout; o += 8) {
        __m256 a0 = _mm256_loadu_ps(b + o);
        for (int kk = 0; kk < K; ++kk) {
            const int k = perm[kk];
            a0 = _mm256_fmadd_ps(_mm256_loadu_ps(Wt + (size_t)k * Nout + o), _mm256_set1_ps(x[k]), a0);
        }
        _mm256_storeu_ps(y + o, a0);
    }
    for (; o < Nout; ++o) {
        float acc = b[o];
        for (int kk = 0; kk < K; ++kk) acc = fmaf(Wt[(size_t)perm[kk] * Nout + o], x[perm[kk]], acc);
        y[o] = acc;
    }
}
#else
static inline void linear_chain(const float* restrict Wt, const float* restrict b, const float* restrict x,
                                const int* restrict perm, int K, int Nout, float* restrict y)
{
    for (int o = 0; o < Nout; ++o) y[o] = b[o];
    for (int kk = 0; kk < K; ++kk) {
        int k = perm[kk];
        float xv = x[k];
        const float* restrict w = Wt + (size_t)k * Nout;
        for (int o = 0; o < Nout; ++o) y[o] = fmaf(w[o], xv, y[o]);
    }
}
#endif

#define ORC_MAXD 1024

/*
 * PPFEncoder.forward_with_idx.  params: flat fp32 buffer holding torch-layout tensors;
 * offs: 6 entries per res layer {fc1.weight, fc1.bias, fc2.weight, fc2.bias, fc0.weight or -1,
 * fc0.bias or -1} followed by {final.weight, final.bias}.  dims: n_res+1 layer widths
 * (ppffcs, train.py:35), dims[0] must equal 2F+4.  out: [P, out_dim].
 */
int orc_pair_mlp(const float* pc, const float* nrm, const float* feat, const int64_t* idxs, int64_t N,
                 int F, int64_t P, const float* params, const int64_t* offs, const int* dims, int n_res,
                 int out_dim, int order, float* out)
{
    if (dims[0] != 2 * F + 4) return -1;
    for (int i = 0; i <= n_res; ++i)
        if (dims[i] > ORC_MAXD) return -2;
    if (out_dim > ORC_MAXD) return -2;
    /* transposed weight copies + k orders */
    int nl = 3 * n_res + 1;
    float** Wt = (float**)calloc(nl, sizeof(float*));
    int** perm = (int**)calloc(nl, sizeof(int*));
    for (int i = 0; i < n_res; ++i) {
        int K = dims[i], Nn = dims[i + 1];
        const int64_t* o = offs + 6 * i;
        /* slot 3i: fc1 [Nn][K]; 3i+1: fc2 [Nn][Nn]; 3i+2: fc0 [Nn][K] or NULL */
        Wt[3 * i] = (float*)malloc(sizeof(float) * K * Nn);
        for (int r = 0; r < Nn; ++r)
            for (int k = 0; k < K; ++k) Wt[3 * i][k * Nn + r] = params[o[0] + (int64_t)r * K + k];
        Wt[3 * i + 1] = (float*)malloc(sizeof(float) * Nn * Nn);
        for (int r = 0; r < Nn; ++r)
            for (int k = 0; k < Nn; ++k) Wt[3 * i + 1][k * Nn + r] = params[o[2] + (int64_t)r * Nn + k];
        if (o[4] >= 0) {
            Wt[3 * i + 2] = (float*)malloc(sizeof(float) * K * Nn);
            for (int r = 0; r < Nn; ++r)
                for (int k = 0; k < K; ++k) Wt[3 * i + 2][k * Nn + r] = params[o[4] + (int64_t)r * K + k];
        }
        perm[3 * i] = (int*)malloc(sizeof(int) * K);
        orc_k_order(K, F, i == 0, order, perm[3 * i]);
        perm[3 * i + 1] = (int*)malloc(sizeof(int) * Nn);
        orc_k_order(Nn, F, 0, order, perm[3 * i + 1]);
    }
    {
        int K = dims[n_res];
        const int64_t* o = offs + 6 * n_res;
        Wt[3 * n_res] = (float*)malloc(sizeof(float) * K * out_dim);
        for (int r = 0; r < out_dim; ++r)
            for (int k = 0; k < K; ++k) Wt[3 * n_res][k * out_dim + r] = params[o[0] + (int64_t)r * K + k];
        perm[3 * n_res] = (int*)malloc(sizeof(int) * K);
        orc_k_order(K, F, n_res == 0, order, perm[3 * n_res]);
    }

    /* order 1: per-point tables of the first layer (see orc_k_order): tab[w] = {TA | TB} for w = fc1, fc0 */
    float* tab[2] = {NULL, NULL};
    const int hoist = order == 1 && n_res > 0;
    if (hoist) {
        const int K = dims[0], Nn = dims[1];
        int* nat = (int*)malloc(sizeof(int) * F);
        for (int k = 0; k < F; ++k) nat[k] = k;
        float* zero = (float*)calloc(Nn, sizeof(float));
        for (int w = 0; w < 2; ++w) {
            const int64_t ow = offs[w == 0 ? 0 : 4], ob = offs[w == 0 ? 1 : 5];
            if (ow < 0) continue;
            const float* Wk = Wt[w == 0 ? 0 : 2];           /* [K][Nn] k-major */
            tab[w] = (float*)malloc(sizeof(float) * (size_t)N * 2 * Nn);
#pragma omp parallel for schedule(static)
            for (int64_t n = 0; n < N; ++n) {
                linear_chain(Wk, params + ob, feat + n * F, nat, F, Nn, tab[w] + (size_t)n * 2 * Nn);
                linear_chain(Wk + (size_t)F * Nn, zero, feat + n * F, nat, F, Nn, tab[w] + (size_t)n * 2 * Nn + Nn);
            }
        }
        (void)K;
        free(nat); free(zero);
    }

#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        float x[ORC_MAXD], h[ORC_MAXD], y2[ORC_MAXD], y0[ORC_MAXD];
        int64_t a = idxs[2 * i], b = idxs[2 * i + 1];
        /* models/model.py:132 final_feat = cat(feat[a], feat[b], ppf) */
        memcpy(x, feat + a * F, sizeof(float) * F);
        memcpy(x + F, feat + b * F, sizeof(float) * F);
        ppf4(pc, nrm, a, b, x + 2 * F);
        for (int l = 0; l < n_res; ++l) {
            int K = dims[l], Nn = dims[l + 1];
            const int64_t* o = offs + 6 * l;
            /* models/model.py:27-31: x_res = fc0(x) or x; x = relu(fc1(x)); x = fc2(x); x + x_res */
            const int hl = hoist && l == 0;
            if (hl) {   /* TA[a] + TB[b], then the 4 PPF inputs */
                const float *ta = tab[0] + (size_t)a * 2 * Nn, *tb = tab[0] + (size_t)b * 2 * Nn + Nn;
                for (int j = 0; j < Nn; ++j) {
                    float acc = ta[j] + tb[j];
                    for (int k = 0; k < 4; ++k) acc = fmaf(Wt[0][(size_t)(2 * F + k) * Nn + j], x[2 * F + k], acc);
                    h[j] = acc;
                }
            } else {
                linear_chain(Wt[3 * l], params + o[1], x, perm[3 * l], K, Nn, h);
            }
            for (int j = 0; j < Nn; ++j) h[j] = h[j] > 0.0f ? h[j] : 0.0f;
            linear_chain(Wt[3 * l + 1], params + o[3], h, perm[3 * l + 1], Nn, Nn, y2);
            if (o[4] >= 0 && hl) {
                const float *ta = tab[1] + (size_t)a * 2 * Nn, *tb = tab[1] + (size_t)b * 2 * Nn + Nn;
                for (int j = 0; j < Nn; ++j) {
                    float acc = ta[j] + tb[j];
                    for (int k = 0; k < 4; ++k) acc = fmaf(Wt[2][(size_t)(2 * F + k) * Nn + j], x[2 * F + k], acc);
                    y0[j] = acc;
                }
                for (int j = 0; j < Nn; ++j) x[j] = y2[j] + y0[j];
            } else if (o[4] >= 0) {
                linear_chain(Wt[3 * l + 2], params + o[5], x, perm[3 * l], K, Nn, y0);
                for (int j = 0; j < Nn; ++j) x[j] = y2[j] + y0[j];
            } else {
                for (int j = 0; j < Nn; ++j) x[j] = y2[j] + x[j];
            }
        }
        const int64_t* o = offs + 6 * n_res;
        linear_chain(Wt[3 * n_res], params + o[1], x, perm[3 * n_res], dims[n_res], out_dim,
                     out + i * out_dim);
    }
    for (int i = 0; i < nl; ++i) { free(Wt[i]); free(perm[i]); }
    free(Wt); free(perm);
    free(tab[0]); free(tab[1]);
    return 0;
}

/* ------------------------------------------------------------------ decode */

/*
 * Sample one bin from softmax(l[0..nb)) with uniform u in [0,1) by inverse CDF (deterministic
 * stand-in for torch.multinomial, nocs/inference.py:186; the draw itself is supplied by the
 * caller).  u < 0 selects the arg-max bin (first maximum) instead.
 *
 * The CDF is built on the layout the HIP kernel holds the logits in, so that the four lanes that share a pair
 * can build it with two cross-lane exchanges and no serial 32-long dependency: the nb bins are cut into four
 * consecutive segments of NL = ceil(nb/4) bins (one per lane; the final layer's output columns are permuted at
 * pack time so that a lane's registers hold exactly its segment).
 *   e_k   = orc_exp2w(l_k, -(max l * log2e))        = 2^((l_k - max l) log2e), the subtraction inside the fma
 *   T_g   = e summed sequentially over segment g;  total = (T0 + T1) + (T2 + T3);  t = u * total
 *   off_g = 0, T0, T0 + T1, (T0 + T1) + T2
 *   the bin is the first k of the first segment g with (e summed over the segment up to and including k) > t - off_g
 *   (the running sums that end in T_g; the threshold is moved into the segment); none -> the last bin.
 */
int orc_sample_bin(const float* l, int nb, float u, int col0)
{
    (void)col0;   /* (kept in the signature: position of the head in the logit row, unused by this layout) */
    float m = l[0];
    int am = 0;
    for (int k = 1; k < nb; ++k)
        if (l[k] > m) { m = l[k]; am = k; }
    if (u < 0.0f) return am;
    const int NL = (nb + 3) / 4;
    const float c = -(m * ORC_LOG2E);
    float e[ORC_MAXD], T[4];
    for (int g = 0; g < 4; ++g) {
        float acc = 0.0f;
        for (int k = g * NL; k < (g + 1) * NL && k < nb; ++k) {
            e[k] = orc_exp2w(l[k], c);
            acc = k == g * NL ? e[k] : acc + e[k];
        }
        T[g] = acc;
    }
    const float s01 = T[0] + T[1], s23 = T[2] + T[3];
    const float tt = u * (s01 + s23);
    const float off[4] = {0.0f, T[0], s01, s01 + T[2]};
    for (int g = 0; g < 4; ++g) {
        const float t = tt - off[g];
        float b = 0.0f;
        for (int k = g * NL; k < (g + 1) * NL && k < nb; ++k) {
            b = k == g * NL ? e[k] : b + e[k];
            if (b > t) return k;
        }
    }
    return nb - 1;
}

/* nocs/inference.py:185-188: mu = k/(nb-1)*2*vr0 - vr0, nu = k/(nb-1)*vr1, fp32 left to right
 * (true division, as torch does on CPU tensors). logits row stride = ld. */
void orc_decode_center(const float* logits, int64_t P, int ld, int nb, const float* u, float vr0,
                       float vr1, float* outputs, int32_t* bins)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        const float* l = logits + i * ld;
        int k0 = orc_sample_bin(l, nb, u[2 * i], 0);
        int k1 = orc_sample_bin(l + nb, nb, u[2 * i + 1], nb);
        float d = (float)(nb - 1);
        outputs[2 * i] = ((float)k0 / d * 2.0f) * vr0 - vr0;
        outputs[2 * i + 1] = (float)k1 / d * vr1;
        if (bins) { bins[2 * i] = k0; bins[2 * i + 1] = k1; }
    }
}

/* nocs/inference.py:238-256: heads of the second pass.  heads[i] = {theta_up, theta_right,
 * aux_up, aux_right, sx, sy, sz, 0}; theta = k/(rb-1)*pi in fp32.  Column layout train.py:68-75:
 * [0,2tb) centre bins, [2tb,2tb+rb) up, [2tb+rb,2tb+2rb) right, then out_dim-5, -4 aux, -3.. scale. */
void orc_decode_rot(const float* logits, int64_t P, int ld, int out_dim, int tb, int rb, const float* u,
                    float* heads, int32_t* bins)
{
    const float pif = (float)ORC_PI;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        const float* l = logits + i * ld;
        int ku = orc_sample_bin(l + 2 * tb, rb, u[2 * i], 2 * tb);
        int kr = orc_sample_bin(l + 2 * tb + rb, rb, u[2 * i + 1], 2 * tb + rb);
        float d = (float)(rb - 1);
        float* h = heads + 8 * i;
        h[0] = (float)ku / d * pif;
        h[1] = (float)kr / d * pif;
        h[2] = l[out_dim - 5];
        h[3] = l[out_dim - 4];
        h[4] = l[out_dim - 3];
        h[5] = l[out_dim - 2];
        h[6] = l[out_dim - 1];
        h[7] = 0.0f;
        if (bins) { bins[2 * i] = ku; bins[2 * i + 1] = kr; }
    }
}

/* ------------------------------------------------------------------ grid */

/* nocs/inference.py:194-195: corners = [min(pc), max(pc)]; grid_res = int32((max-min)/res) + 1 */
void orc_grid_setup(const float* pc, int64_t N, float res, float* corner, int32_t* dims)
{
    float lo[3] = {pc[0], pc[1], pc[2]}, hi[3] = {pc[0], pc[1], pc[2]};
    for (int64_t i = 1; i < N; ++i)
        for (int j = 0; j < 3; ++j) {
            float v = pc[3 * i + j];
            if (v < lo[j]) lo[j] = v;
            if (v > hi[j]) hi[j] = v;
        }
    for (int j = 0; j < 3; ++j) {
        corner[j] = lo[j];
        dims[j] = (int32_t)((hi[j] - lo[j]) / res) + 1;
    }
}

/* shared front half of the three vote kernels: models/voting.py:15-29 / :81-95 / :125-136.
 * returns 0 for a degenerate pair (early return at :21/:87/:131). */
static inline int pair_frame(const float* points, const int32_t* point_idxs, int64_t idx, f3* a_out,
                             f3* ab_out, f3* xdir_out)
{
    int a_idx = point_idxs[2 * idx], b_idx = point_idxs[2 * idx + 1];
    f3 a = ld3(points, a_idx), b = ld3(points, b_idx);
    f3 ab = sub3(a, b);
    float L = len3(ab);
    if ((double)L < 1e-7) return 0;
    ab = div3(ab, (float)((double)L + 1e-7));              /* ab /= (length(ab) + 1e-7) */
    f3 co = {0.f, -ab.z, ab.y};
    if ((double)len3(co) < 1e-7) { co.x = -ab.y; co.y = ab.x; co.z = 0.f; }
    *xdir_out = div3(co, (float)((double)len3(co) + 1e-7)); /* co / (length(co) + 1e-7) */
    *a_out = a;
    *ab_out = ab;
    return 1;
}

/* models/voting.py:8-66.  Serial: "atomicAdd" order = pair order, rotation order, corner order. */
void orc_ppf_voting(const float* points, const float* outputs, const float* probs,
                    const int32_t* point_idxs, float* grid_obj, const float* corner, float res,
                    int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive, int64_t* n_atomics)
{
    int64_t na = 0;
    f3 cr = {corner[0], corner[1], corner[2]};
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        float proj_len = outputs[2 * idx], odist = outputs[2 * idx + 1];
        f3 a, ab, xd;
        if (!pair_frame(points, point_idxs, idx, &a, &ab, &xd)) continue;
        f3 c = sub3(a, scl3(ab, proj_len));
        float pa = probs[point_idxs[2 * idx]], pb = probs[point_idxs[2 * idx + 1]];
        float prob = pa > pb ? pa : pb;                    /* max(probs[a], probs[b]) :25 */
        f3 x = scl3(xd, odist);
        f3 y = cross3(x, ab);
        int n = n_rots;
        if (adaptive) {                                     /* :31 */
            int m = sat_int((double)(odist / res) * (2 * ORC_PI));
            n = m < n_rots ? m : n_rots;
        }
        for (int i = 0; i < n; ++i) {
            float cs, sn;
            rot_cs(i, n, &cs, &sn);
            f3 offset = add3(scl3(x, cs), scl3(y, sn));     /* cos*x + sin*y :34 */
            f3 g = div3(sub3(add3(c, offset), cr), res);    /* :35 */
            if ((double)g.x < 0.01 || (double)g.y < 0.01 || (double)g.z < 0.01 ||
                (double)g.x >= (double)gx - 1.01 || (double)g.y >= (double)gy - 1.01 ||
                (double)g.z >= (double)gz - 1.01)
                continue;                                   /* :36-39 */
            int fx = (int)g.x, fy = (int)g.y, fz = (int)g.z; /* make_int3 truncation :40 */
            int cx = fx + 1, cy = fy + 1, cz = fz + 1;
            float rx = g.x - floorf(g.x), ry = g.y - floorf(g.y), rz = g.z - floorf(g.z);
            float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
            float lll = w0x * w0y * w0z, llh = w0x * w0y * rz, lhl = w0x * ry * w0z, lhh = w0x * ry * rz;
            float hll = rx * w0y * w0z, hlh = rx * w0y * rz, hhl = rx * ry * w0z, hhh = rx * ry * rz;
            int64_t syz = (int64_t)gy * gz;
            grid_obj[fx * syz + fy * gz + fz] += lll * prob;
            grid_obj[fx * syz + fy * gz + cz] += llh * prob;
            grid_obj[fx * syz + cy * gz + fz] += lhl * prob;
            grid_obj[fx * syz + cy * gz + cz] += lhh * prob;
            grid_obj[cx * syz + fy * gz + fz] += hll * prob;
            grid_obj[cx * syz + fy * gz + cz] += hlh * prob;
            grid_obj[cx * syz + cy * gz + fz] += hhl * prob;
            grid_obj[cx * syz + cy * gz + cz] += hhh * prob;
            na += 8;
        }
    }
    if (n_atomics) *n_atomics = na;
}

/* Same votes accumulated in fp64 (+ optional per-cell deposit counts): the exact sum that every
 * ordering of the reference's fp32 atomicAdd approximates.  The HIP path accumulates in fixed point
 * and is compared against this with a bound of half a quantum per deposit. */
void orc_ppf_voting_f64(const float* points, const float* outputs, const float* probs,
                        const int32_t* point_idxs, double* grid, int32_t* counts, const float* corner, float res,
                        int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive)
{
    f3 cr = {corner[0], corner[1], corner[2]};
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        float proj_len = outputs[2 * idx], odist = outputs[2 * idx + 1];
        f3 a, ab, xd;
        if (!pair_frame(points, point_idxs, idx, &a, &ab, &xd)) continue;
        f3 c = sub3(a, scl3(ab, proj_len));
        float pa = probs[point_idxs[2 * idx]], pb = probs[point_idxs[2 * idx + 1]];
        float prob = pa > pb ? pa : pb;
        f3 x = scl3(xd, odist);
        f3 y = cross3(x, ab);
        int n = n_rots;
        if (adaptive) {
            int m = sat_int((double)(odist / res) * (2 * ORC_PI));
            n = m < n_rots ? m : n_rots;
        }
        for (int i = 0; i < n; ++i) {
            float cs, sn;
            rot_cs(i, n, &cs, &sn);
            f3 offset = add3(scl3(x, cs), scl3(y, sn));
            f3 g = div3(sub3(add3(c, offset), cr), res);
            if ((double)g.x < 0.01 || (double)g.y < 0.01 || (double)g.z < 0.01 ||
                (double)g.x >= (double)gx - 1.01 || (double)g.y >= (double)gy - 1.01 ||
                (double)g.z >= (double)gz - 1.01)
                continue;
            int fx = (int)g.x, fy = (int)g.y, fz = (int)g.z;
            float rx = g.x - floorf(g.x), ry = g.y - floorf(g.y), rz = g.z - floorf(g.z);
            float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
            float w[8] = {w0x * w0y * w0z * prob, w0x * w0y * rz * prob, w0x * ry * w0z * prob, w0x * ry * rz * prob,
                          rx * w0y * w0z * prob,  rx * w0y * rz * prob,  rx * ry * w0z * prob,  rx * ry * rz * prob};
            int64_t syz = (int64_t)gy * gz;
            int64_t b = fx * syz + fy * gz + fz;
            int64_t off[8] = {0, 1, gz, gz + 1, syz, syz + 1, syz + gz, syz + gz + 1};
            for (int k = 0; k < 8; ++k) {
                grid[b + off[k]] += (double)w[k];
                if (counts) counts[b + off[k]] += 1;
            }
        }
    }
}

/* Same votes as EXACT INTEGERS: every deposit rounded to a multiple of the quantum p2 * 2^-bits, q = floor(w * 2^bits / p2 + 1/2)
 * with w the reference's fp32 weight (:47-63, product left to right), summed as int64.  This is the specification of
 * cppf_vote_grid_raw (the image the pair-sharded vote all-reduces, cppf_amd/sharding.py): integer sums are order-independent,
 * so the HIP path must reproduce it bit for bit whichever workgroup / rank took whichever pair.  grid is added to. */
void orc_ppf_voting_fixed(const float* points, const float* outputs, const float* probs, const int32_t* point_idxs,
                          int64_t* grid, const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz,
                          int adaptive, int bits, double p2)
{
    f3 cr = {corner[0], corner[1], corner[2]};
    const double S = ldexp(1.0, bits) / p2;
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        float proj_len = outputs[2 * idx], odist = outputs[2 * idx + 1];
        f3 a, ab, xd;
        if (!pair_frame(points, point_idxs, idx, &a, &ab, &xd)) continue;
        f3 c = sub3(a, scl3(ab, proj_len));
        float pa = probs[point_idxs[2 * idx]], pb = probs[point_idxs[2 * idx + 1]];
        float prob = pa > pb ? pa : pb;
        f3 x = scl3(xd, odist);
        f3 y = cross3(x, ab);
        int n = n_rots;
        if (adaptive) {
            int m = sat_int((double)(odist / res) * (2 * ORC_PI));
            n = m < n_rots ? m : n_rots;
        }
        for (int i = 0; i < n; ++i) {
            float cs, sn;
            rot_cs(i, n, &cs, &sn);
            f3 offset = add3(scl3(x, cs), scl3(y, sn));
            f3 g = div3(sub3(add3(c, offset), cr), res);
            if ((double)g.x < 0.01 || (double)g.y < 0.01 || (double)g.z < 0.01 ||
                (double)g.x >= (double)gx - 1.01 || (double)g.y >= (double)gy - 1.01 ||
                (double)g.z >= (double)gz - 1.01)
                continue;
            int fx = (int)g.x, fy = (int)g.y, fz = (int)g.z;
            float rx = g.x - floorf(g.x), ry = g.y - floorf(g.y), rz = g.z - floorf(g.z);
            float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
            float w[8] = {w0x * w0y * w0z * prob, w0x * w0y * rz * prob, w0x * ry * w0z * prob, w0x * ry * rz * prob,
                          rx * w0y * w0z * prob,  rx * w0y * rz * prob,  rx * ry * w0z * prob,  rx * ry * rz * prob};
            int64_t syz = (int64_t)gy * gz;
            int64_t b = fx * syz + fy * gz + fz;
            int64_t off[8] = {0, 1, gz, gz + 1, syz, syz + 1, syz + gz, syz + gz + 1};
            for (int k = 0; k < 8; ++k) grid[b + off[k]] += (int64_t)floor((double)w[k] * S + 0.5);
        }
    }
}

/* Multi-threaded variant for the timed CPU baseline: per-thread private grids, summed at the end
 * (summation order differs from the serial one; same tolerance class as the GPU's atomics). */
void orc_ppf_voting_mt(const float* points, const float* outputs, const float* probs,
                       const int32_t* point_idxs, float* grid_obj, const float* corner, float res,
                       int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive)
{
#ifdef _OPENMP
    int nt = omp_get_max_threads();
    size_t G = (size_t)gx * gy * gz;
    float* priv = (float*)calloc((size_t)nt * G, sizeof(float));
#pragma omp parallel
    {
        int t = omp_get_thread_num();
        int64_t lo = n_ppfs * t / nt, hi = n_ppfs * (t + 1) / nt;
        orc_ppf_voting(points, outputs + 2 * lo, probs, point_idxs + 2 * lo, priv + (size_t)t * G, corner,
                       res, hi - lo, n_rots, gx, gy, gz, adaptive, NULL);
    }
    for (int t = 0; t < nt; ++t)
        for (size_t i = 0; i < G; ++i) grid_obj[i] += priv[(size_t)t * G + i];
    free(priv);
#else
    orc_ppf_voting(points, outputs, probs, point_idxs, grid_obj, corner, res, n_ppfs, n_rots, gx, gy, gz,
                   adaptive, NULL);
#endif
}

/* np.argmax(grid, axis=None): first maximum in C order (nocs/inference.py:208) */
int64_t orc_grid_argmax(const float* grid, int64_t n, float* val)
{
    int64_t best = 0;
    float bv = grid[0];
    for (int64_t i = 1; i < n; ++i)
        if (grid[i] > bv) { bv = grid[i]; best = i; }
    if (val) *val = bv;
    return best;
}

/* nocs/inference.py:209-210: cand = unravel_index(argmax); T = corners[0] + cand * res  (fp64) */
void orc_center_from_argmax(int64_t flat, int gy, int gz, const float* corner, double res, double* T)
{
    int64_t x = flat / ((int64_t)gy * gz), yz = flat % ((int64_t)gy * gz);
    int64_t y = yz / gz, z = yz % gz;
    T[0] = (double)corner[0] + (double)x * res;
    T[1] = (double)corner[1] + (double)y * res;
    T[2] = (double)corner[2] + (double)z * res;
}

/* models/voting.py:74-112.  out_offsets must be zero-initialised by the caller (degenerate pairs
 * return before the store at :96).  mask[i] = any(out_offsets[i] != 0) (nocs/inference.py:230). */
void orc_backvote(const float* points, const float* outputs, float* out_offsets, const int32_t* point_idxs,
                  const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz,
                  const float* gt_center, float tol, uint8_t* mask)
{
    f3 cr = {corner[0], corner[1], corner[2]};
    f3 gt = {gt_center[0], gt_center[1], gt_center[2]};
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        float proj_len = outputs[2 * idx], odist = outputs[2 * idx + 1];
        f3 a, ab, xd;
        float* oo = out_offsets + 3 * idx;
        if (pair_frame(points, point_idxs, idx, &a, &ab, &xd)) {
            f3 c = sub3(a, scl3(ab, proj_len));
            f3 x = scl3(xd, odist);
            f3 y = cross3(x, ab);
            oo[0] = oo[1] = oo[2] = 0.f;                           /* :96 */
            int m = sat_int((double)(odist / res) * (2 * ORC_PI)); /* :97 always adaptive */
            int n = m < n_rots ? m : n_rots;
            for (int i = 0; i < n; ++i) {
                float cs, sn;
                rot_cs(i, n, &cs, &sn);
                f3 offset = add3(scl3(x, cs), scl3(y, sn));
                f3 pc_ = add3(c, offset);
                if (len3(sub3(pc_, gt)) > tol) continue;          /* :101 */
                f3 g = div3(sub3(pc_, cr), res);
                if (g.x < 0 || g.y < 0 || g.z < 0 || g.x >= (float)(gx - 1) || g.y >= (float)(gy - 1) ||
                    g.z >= (float)(gz - 1))
                    continue;                                      /* :103-107 (int -> float compare) */
                f3 no = neg3(offset);
                oo[0] = no.x; oo[1] = no.y; oo[2] = no.z;          /* :108 */
                break;
            }
        }
        if (mask) mask[idx] = (oo[0] != 0.f) || (oo[1] != 0.f) || (oo[2] != 0.f);
    }
}

/* models/voting.py:119-147.  outputs_up [n_ppfs, n_rots, 3] zero-initialised by the caller. */
void orc_rot_voting(const float* points, const float* preds_rot, float* outputs_up,
                    const int32_t* point_idxs, int64_t n_ppfs, int n_rots)
{
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < n_ppfs; ++idx) {
        f3 a, ab, x;
        if (!pair_frame(points, point_idxs, idx, &a, &ab, &x)) continue;
        f3 y = cross3(x, ab);
        float t = orc_tanf(preds_rot[idx]);
        f3 base = t > 0 ? ab : neg3(ab);                            /* :142 */
        for (int i = 0; i < n_rots; ++i) {
            float cs, sn;
            rot_cs(i, n_rots, &cs, &sn);
            f3 offset = add3(scl3(x, cs), scl3(y, sn));
            f3 up = add3(scl3(offset, t), base);                    /* tan(rot)*offset + (+-ab) */
            up = div3(up, (float)((double)len3(up) + 1e-7));        /* :143 */
            float* o = outputs_up + ((size_t)idx * n_rots + i) * 3;
            o[0] = up.x; o[1] = up.y; o[2] = up.z;
        }
    }
}

/* nocs/inference.py:281-282: cos = candidates.mm(sphere^T) (K=3 fp32 dot, fma chain in k order);
 * counts[j] = #(cos > thr).  cands [M,3], sphere [S,3]. */
void orc_sphere_count(const float* cands, int64_t M, const float* sphere, int S, float thr, int64_t* counts)
{
    for (int j = 0; j < S; ++j) counts[j] = 0;
    for (int64_t m = 0; m < M; ++m) {
        const float* c = cands + 3 * m;
        for (int j = 0; j < S; ++j) {
            const float* s = sphere + 3 * j;
            float d = fmaf(c[2], s[2], fmaf(c[1], s[1], c[0] * s[0]));
            counts[j] += d > thr;
        }
    }
}

/* nocs/inference.py:287-298: ab = pc[a]-pc[b] (numpy fp32), normal n_a flipped to agree with ab
 * (the sign of n.ab_normed equals the sign of n.ab up to rounding; the reference divides first and
 * so do we), target = (n.best_dir > 0) with best_dir in fp64; BCEWithLogits(aux, target) and
 * (aux, 1-target), means accumulated in fp64.  losses[0] = up_loss, losses[1] = down_loss.
 * Returns 1 when the axis must be negated (down_loss < up_loss). */
int orc_axis_sign(const float* pc, const float* nrm, const int32_t* idxs, int64_t P, const float* aux,
                  int aux_stride, const double* best_dir, double* losses)
{
    double up = 0.0, down = 0.0;
    for (int64_t i = 0; i < P; ++i) {
        int a = idxs[2 * i], b = idxs[2 * i + 1];
        f3 ab = sub3(ld3(pc, a), ld3(pc, b));
        float distsq = (ab.x * ab.x + ab.y * ab.y) + ab.z * ab.z;      /* np.sum(ab**2,-1) */
        float den = sqrtf(distsq) + 1e-7f;                             /* fp32 array + python float */
        f3 abn = {ab.x / den, ab.y / den, ab.z / den};
        f3 n = ld3(nrm, a);
        float d = (n.x * abn.x + n.y * abn.y) + n.z * abn.z;
        if (d < 0) n = neg3(n);
        double proj = ((double)n.x * best_dir[0] + (double)n.y * best_dir[1]) + (double)n.z * best_dir[2];
        double t = proj > 0 ? 1.0 : 0.0;
        double x = (double)aux[i * aux_stride];
        /* BCEWithLogits: max(x,0) - x*t + log1p(exp(-|x|)) */
        double sp = (x > 0 ? x : 0) + log1p(exp(-fabs(x)));
        up += sp - x * t;
        down += sp - x * (1.0 - t);
    }
    up /= (double)(P > 0 ? P : 1);
    down /= (double)(P > 0 ? P : 1);
    if (losses) { losses[0] = up; losses[1] = down; }
    return down < up;
}

/* nocs/inference.py:335: exp(mean_P(preds_scale)) * scale_mean * 2; the mean is accumulated in
 * fp64 and rounded to fp32 (torch returns an fp32 mean), np.exp keeps fp32, the products are fp64. */
void orc_scale(const float* scale_logits, int64_t P, int stride, const double* scale_mean, double* out)
{
    double s[3] = {0, 0, 0};
    for (int64_t i = 0; i < P; ++i)
        for (int j = 0; j < 3; ++j) s[j] += (double)scale_logits[i * stride + j];
    for (int j = 0; j < 3; ++j) {
        float m = (float)(s[j] / (double)(P > 0 ? P : 1));
        out[j] = (double)expf(m) * scale_mean[j] * 2.0; /* np.exp(float32) stays fp32 */
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
