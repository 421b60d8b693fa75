/*
 * CPU restatement of the BACKWARD of the reference's PointEncoder (one SparseSO3Conv + GlobalInfoProp stage, the
 * configuration train.py:34 trains: k <= 64, spfcs = {32,64,32,32}, rank 32, 2 neighbour features, out_dim 32 + 8).
 *
 * TEST INFRASTRUCTURE ONLY (see cppf_oracle.c): compiled into liboracle.so, called from tests/ only.
 *
 * The reference has no backward code: train.py:91 calls loss.backward() and torch autograd differentiates
 * models/model.py:46-61 and models/sprin.py:40-107.  Points and normals carry no gradient (train.py:58-60), so the
 * gradients are those of the parameters.  PARITY PINNED on that: tests/golden/make_golden_sprin_bwd.py runs the imported
 * reference module under autograd on CPU for a fixed upstream gradient and stores d/d(parameter); tests/
 * test_oracle_golden.py compares this file against them (2e-5 of each gradient's scale; ATen sums in another order).
 *
 * Deterministic order shared with cppf_amd/csrc/sprin_bwd.hip (HIP vs oracle is bit-exact):
 *   forward      = orc_point_encoder order 1 (sprin_oracle.c), every intermediate recomputed;
 *   pooled part  : dP[c] = sum over chunks of 64 points (ascending) of the chunk's ascending sum of G[n][32 + c];
 *                  share[c] = dP[c] / #{n : lin[n][c] == pooled[c]}  (torch.amax spreads the gradient over ties);
 *   per point    : dlin[c] = share[c] where lin[n][c] == pooled[c], else 0;  dx[o] = G[n][o] + chain_c fmaf(Wa[c][o], dlin[c], .);
 *                  LayerNorm_o: gd = dx * gamma; s1 = seq sum gd; s2 = seq sum gd * xhat;
 *                               dy[o] = ((gd[o] - s1/32) - xhat[o] * (s2/32)) * inv;
 *                  dmixed[c] = chain_o fmaf(Wo[o][c], dy[o], .) from 0;  dkern[j][r] = chain_i fmaf(dmixed[2r + i], nf[j][i], .);
 *   per row      : transposed chains in khid order seeded with 0 (d(a4) = W5^T d(kern) ...), LayerNorm backward
 *                  dz = relu'(z) d(a); gd = dz * gamma; s1, s2 = row sums in the 4-lane order of sum_ord1;
 *                  dy = ((gd - s1/H) - xhat * (s2/H)) * inv;
 *   accumulation : accumulator `part` w (0 <= w < n_parts) owns points w, w + n_parts, ...; rows of a point in ascending
 *                  order.  Weights: ONE fmaf chain per entry over all of w's live rows (per-point ones over its points).
 *                  Biases of the kernel MLP: ((s0 + s1) + s2) + s3 with s_q the sequential sum over w's rows with row % 4 == q.
 *                  LayerNorm gamma / beta of the kernel MLP: sequential sum over the 16 row slots (row % 16) of each slot's
 *                  sequential sum.  Per-point vectors (outnet bias, LayerNorm_o, aggr bias): sequential sums over w's points.
 *                  grad = sum over groups of 32 consecutive parts (ascending) of the group's ascending sum.
 * Gradient layout = the packed parameter layout of cppf_point_encoder_pack (outnet weight transposed [C][n_out]).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

enum { H1 = 32, H2 = 64, H3 = 32, H4 = 32, RK = 32, NIN = 2, NOUT = 32, NGLOB = 8, CMIX = RK * NIN, OUTW = NOUT + NGLOB };
/* packed offsets (floats) */
enum {
    P_W1 = 0, P_B1 = P_W1 + H1 * 6, P_G1 = P_B1 + H1, P_E1 = P_G1 + H1,
    P_W2 = P_E1 + H1, P_B2 = P_W2 + H2 * H1, P_G2 = P_B2 + H2, P_E2 = P_G2 + H2,
    P_W3 = P_E2 + H2, P_B3 = P_W3 + H3 * H2, P_G3 = P_B3 + H3, P_E3 = P_G3 + H3,
    P_W4 = P_E3 + H3, P_B4 = P_W4 + H4 * H3, P_G4 = P_B4 + H4, P_E4 = P_G4 + H4,
    P_W5 = P_E4 + H4, P_B5 = P_W5 + RK * H4,
    P_WO = P_B5 + RK, P_BO = P_WO + CMIX * NOUT, P_GO = P_BO + NOUT, P_EO = P_GO + NOUT,
    P_WA = P_EO + NOUT, P_BA = P_WA + NGLOB * NOUT, P_TOTAL = P_BA + NGLOB
};

static void khid(int K, int* perm)
{
    int n = 0;
    for (int s = 0; s < K / 4; ++s)
        for (int g = 0; g < 4; ++g) perm[n++] = 16 * (s / 4) + 4 * g + (s % 4);
}
static float sum4(const float* v, int n)           /* per-lane (ob, r) partial sums, combined (p0 + p1) + (p2 + p3) */
{
    float p[4];
    for (int g = 0; g < 4; ++g) {
        float acc = 0.f;
        for (int ob = 0; ob < n / 16; ++ob)
            for (int r = 0; r < 4; ++r) acc = acc + v[16 * ob + 4 * g + r];
        p[g] = acc;
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}
typedef struct { float y[64], xh[64], a[64], mean, inv; } ln_state;   /* pre-LN, normalised, relu(affine) */

static void fwd_layer(const float* W, const float* b, const float* x, int n_in, int n_out, const int* perm, float* y)
{
    for (int o = 0; o < n_out; ++o) {
        float acc = b[o];
        for (int q = 0; q < n_in; ++q) acc = fmaf(W[(size_t)o * n_in + perm[q]], x[perm[q]], acc);
        y[o] = acc;
    }
}
static void ln_relu(ln_state* s, int n, const float* gm, const float* bt)
{
    s->mean = sum4(s->y, n) / (float)n;
    float d2[64];
    for (int o = 0; o < n; ++o) { const float d = s->y[o] - s->mean; d2[o] = d * d; }
    s->inv = 1.0f / sqrtf(sum4(d2, n) / (float)n + 1e-5f);
    for (int o = 0; o < n; ++o) {
        s->xh[o] = (s->y[o] - s->mean) * s->inv;
        const float z = s->xh[o] * gm[o] + bt[o];
        s->a[o] = z > 0.f ? z : 0.f;
    }
}
/* d(pre-LN) from d(a); dz (masked) is returned for the gamma / beta sums */
static void ln_relu_bwd(const ln_state* s, int n, const float* gm, const float* da, float* dz, float* dy)
{
    float gd[64], gx[64];
    for (int o = 0; o < n; ++o) {
        dz[o] = s->a[o] > 0.f ? da[o] : 0.f;
        gd[o] = dz[o] * gm[o];
        gx[o] = gd[o] * s->xh[o];
    }
    const float m1 = sum4(gd, n) / (float)n, m2 = sum4(gx, n) / (float)n;
    for (int o = 0; o < n; ++o) dy[o] = ((gd[o] - m1) - s->xh[o] * m2) * s->inv;
}
/* d(in)[i] = chain over the outputs in perm order of fmaf(W[o][i], d[o], .) from 0 */
static void bwd_layer(const float* W, const float* d, int n_in, int n_out, const int* perm, float* dx)
{
    for (int i = 0; i < n_in; ++i) {
        float acc = 0.f;
        for (int q = 0; q < n_out; ++q) acc = fmaf(W[(size_t)perm[q] * n_in + i], d[perm[q]], acc);
        dx[i] = acc;
    }
}
static inline float nrm3(const float* v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
static inline float dt3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

typedef struct {
    float x6[64][6], nf[64][NIN], kern[64][RK];
    ln_state l1[64], l2[64], l3[64], l4[64];
    float mixed[CMIX], y[NOUT], xh[NOUT], x[NOUT], lin[NGLOB], mean, inv;
} point_state;

static void point_forward(const float* pc, const float* nrm, const int32_t* nb, int n, int k, const float* P, const int* kh16,
                          const int* kh32, const int* kh64, point_state* S)
{
    static const int nat6[6] = {0, 1, 2, 3, 4, 5};
    (void)kh16;
    const float* s = pc + 3 * n;
    float rm[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j)
        for (int c = 0; c < 3; ++c) rm[c] = rm[c] + pc[3 * nb[j] + c];
    for (int c = 0; c < 3; ++c) rm[c] = rm[c] / (float)k;
    const float l3[3] = {s[0] - rm[0], s[1] - rm[1], s[2] - rm[2]};
    const float l3n = nrm3(l3);
    for (int j = 0; j < k; ++j) {
        const float* r = pc + 3 * nb[j];
        const float l1[3] = {rm[0] - r[0], rm[1] - r[1], rm[2] - r[2]};
        const float l2[3] = {r[0] - s[0], r[1] - s[1], r[2] - s[2]};
        const float l1n = nrm3(l1), l2n = nrm3(l2);
        float* x = S->x6[j];
        x[0] = l1n; x[1] = l2n; x[2] = l3n;
        x[3] = dt3(l1, l2) / (l1n * l2n + 1e-7f);
        x[4] = dt3(l2, l3) / (l2n * l3n + 1e-7f);
        x[5] = dt3(l3, l1) / (l3n * l1n + 1e-7f);
        S->nf[j][0] = l2n;
        S->nf[j][1] = dt3(nrm + 3 * nb[j], nrm + 3 * n);
        fwd_layer(P + P_W1, P + P_B1, x, 6, H1, nat6, S->l1[j].y); ln_relu(&S->l1[j], H1, P + P_G1, P + P_E1);
        fwd_layer(P + P_W2, P + P_B2, S->l1[j].a, H1, H2, kh32, S->l2[j].y); ln_relu(&S->l2[j], H2, P + P_G2, P + P_E2);
        fwd_layer(P + P_W3, P + P_B3, S->l2[j].a, H2, H3, kh64, S->l3[j].y); ln_relu(&S->l3[j], H3, P + P_G3, P + P_E3);
        fwd_layer(P + P_W4, P + P_B4, S->l3[j].a, H3, H4, kh32, S->l4[j].y); ln_relu(&S->l4[j], H4, P + P_G4, P + P_E4);
        fwd_layer(P + P_W5, P + P_B5, S->l4[j].a, H4, RK, kh32, S->kern[j]);
    }
    for (int r = 0; r < RK; ++r)
        for (int i = 0; i < NIN; ++i) {
            float acc = 0.f;
            for (int j = 0; j < k; ++j) acc = fmaf(S->kern[j][r], S->nf[j][i], acc);
            S->mixed[r * NIN + i] = acc;
        }
    for (int o = 0; o < NOUT; ++o) {
        float acc = P[P_BO + o];
        for (int c = 0; c < CMIX; ++c) acc = fmaf(P[P_WO + c * NOUT + o], S->mixed[c], acc);
        S->y[o] = acc;
    }
    float sm = 0.f;
    for (int o = 0; o < NOUT; ++o) sm = sm + S->y[o];
    S->mean = sm / (float)NOUT;
    float v = 0.f;
    for (int o = 0; o < NOUT; ++o) { const float d = S->y[o] - S->mean; v = v + d * d; }
    S->inv = 1.0f / sqrtf(v / (float)NOUT + 1e-5f);
    for (int o = 0; o < NOUT; ++o) {
        S->xh[o] = (S->y[o] - S->mean) * S->inv;
        S->x[o] = S->xh[o] * P[P_GO + o] + P[P_EO + o];
    }
    for (int c = 0; c < NGLOB; ++c) {
        float acc = P[P_BA + c];
        for (int o = 0; o < NOUT; ++o) acc = fmaf(P[P_WA + c * NOUT + o], S->x[o], acc);
        S->lin[c] = acc;
    }
}

int64_t orc_point_encoder_backward_params(void) { return P_TOTAL; }

/* grad_out: [N][40]; grad_params: [P_TOTAL] in the packed layout.  n_parts: number of accumulators (device: wavefronts). */
int orc_point_encoder_backward(const float* pc, const float* nrm, const int32_t* nbrs, int N, int k, const float* P,
                               const float* grad_out, int n_parts, float* grad_params)
{
    if (k < 1 || k > 64 || N < 1 || n_parts < 1) return -1;
    int kh16[16], kh32[32], kh64[64];
    khid(16, kh16); khid(32, kh32); khid(64, kh64);
    /* pass 0: lin of every point, pooled maxima, tie counts, dP */
    float* lin_all = malloc(sizeof(float) * (size_t)N * NGLOB);
#pragma omp parallel
    {
        point_state* S = malloc(sizeof(point_state));
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            point_forward(pc, nrm, nbrs + (size_t)n * k, n, k, P, kh16, kh32, kh64, S);
            memcpy(lin_all + (size_t)n * NGLOB, S->lin, sizeof(float) * NGLOB);
        }
        free(S);
    }
    float pooled[NGLOB], share[NGLOB];
    for (int c = 0; c < NGLOB; ++c) {
        float mx = -INFINITY;
        for (int n = 0; n < N; ++n) mx = lin_all[(size_t)n * NGLOB + c] > mx ? lin_all[(size_t)n * NGLOB + c] : mx;
        pooled[c] = mx;
        int cnt = 0;
        for (int n = 0; n < N; ++n) cnt += lin_all[(size_t)n * NGLOB + c] == mx;
        float dP = 0.f;
        for (int n0 = 0; n0 < N; n0 += 64) {
            float acc = 0.f;
            for (int n = n0; n < N && n < n0 + 64; ++n) acc = acc + grad_out[(size_t)n * OUTW + NOUT + c];
            dP = dP + acc;
        }
        share[c] = dP / (float)cnt;
    }
    free(lin_all);

    float* parts = calloc((size_t)n_parts * P_TOTAL, sizeof(float));
#pragma omp parallel
    {
        point_state* S = malloc(sizeof(point_state));
        float (*bsub)[5 * 64] = malloc(sizeof(float) * 4 * 5 * 64);       /* [row % 4][layer][feature] */
        float (*gsl)[2 * 4 * 64] = malloc(sizeof(float) * 16 * 2 * 4 * 64); /* [row % 16][gamma | beta][layer][feature] */
#pragma omp for schedule(dynamic, 1)
        for (int w = 0; w < n_parts; ++w) {
            float* part = parts + (size_t)w * P_TOTAL;
            memset(bsub, 0, sizeof(float) * 4 * 5 * 64);
            memset(gsl, 0, sizeof(float) * 16 * 2 * 4 * 64);
            for (int n = w; n < N; n += n_parts) {
                const int32_t* nb = nbrs + (size_t)n * k;
                point_forward(pc, nrm, nb, n, k, P, kh16, kh32, kh64, S);
                const float* G = grad_out + (size_t)n * OUTW;
                float dlin[NGLOB], dx[NOUT], gd[NOUT], dy[NOUT], dmixed[CMIX];
                for (int c = 0; c < NGLOB; ++c) dlin[c] = S->lin[c] == pooled[c] ? share[c] : 0.f;
                for (int o = 0; o < NOUT; ++o) {
                    float t = 0.f;
                    for (int c = 0; c < NGLOB; ++c) t = fmaf(P[P_WA + c * NOUT + o], dlin[c], t);
                    dx[o] = G[o] + t;
                }
                for (int c = 0; c < NGLOB; ++c) {
                    for (int o = 0; o < NOUT; ++o) part[P_WA + c * NOUT + o] = fmaf(dlin[c], S->x[o], part[P_WA + c * NOUT + o]);
                    part[P_BA + c] = part[P_BA + c] + dlin[c];
                }
                float s1 = 0.f, s2 = 0.f;
                for (int o = 0; o < NOUT; ++o) {
                    gd[o] = dx[o] * P[P_GO + o];
                    s1 = s1 + gd[o];
                    s2 = s2 + gd[o] * S->xh[o];
                    part[P_GO + o] = part[P_GO + o] + dx[o] * S->xh[o];
                    part[P_EO + o] = part[P_EO + o] + dx[o];
                }
                const float m1 = s1 / (float)NOUT, m2 = s2 / (float)NOUT;
                for (int o = 0; o < NOUT; ++o) {
                    dy[o] = ((gd[o] - m1) - S->xh[o] * m2) * S->inv;
                    part[P_BO + o] = part[P_BO + o] + dy[o];
                }
                for (int c = 0; c < CMIX; ++c) {
                    float acc = 0.f;
                    for (int o = 0; o < NOUT; ++o) {
                        acc = fmaf(P[P_WO + c * NOUT + o], dy[o], acc);
                        part[P_WO + c * NOUT + o] = fmaf(dy[o], S->mixed[c], part[P_WO + c * NOUT + o]);
                    }
                    dmixed[c] = acc;
                }
                for (int j = 0; j < k; ++j) {
                    float dk[RK], da4[H4], dz4[H4], dy4[H4], da3[H3], dz3[H3], dy3[H3], da2[H2], dz2[H2], dy2[H2], da1[H1],
                        dz1[H1], dy1[H1];
                    for (int r = 0; r < RK; ++r) {
                        float acc = 0.f;
                        for (int i = 0; i < NIN; ++i) acc = fmaf(dmixed[r * NIN + i], S->nf[j][i], acc);
                        dk[r] = acc;
                    }
                    bwd_layer(P + P_W5, dk, H4, RK, kh32, da4);
                    ln_relu_bwd(&S->l4[j], H4, P + P_G4, da4, dz4, dy4);
                    bwd_layer(P + P_W4, dy4, H3, H4, kh32, da3);
                    ln_relu_bwd(&S->l3[j], H3, P + P_G3, da3, dz3, dy3);
                    bwd_layer(P + P_W3, dy3, H2, H3, kh32, da2);
                    ln_relu_bwd(&S->l2[j], H2, P + P_G2, da2, dz2, dy2);
                    bwd_layer(P + P_W2, dy2, H1, H2, kh64, da1);
                    ln_relu_bwd(&S->l1[j], H1, P + P_G1, da1, dz1, dy1);
#define ACC_W(OFF, NI, DELTA, NO, XIN)                                                                \
    for (int o_ = 0; o_ < (NO); ++o_)                                                                 \
        for (int i_ = 0; i_ < (NI); ++i_)                                                             \
            part[(OFF) + o_ * (NI) + i_] = fmaf((DELTA)[o_], (XIN)[i_], part[(OFF) + o_ * (NI) + i_]);
                    ACC_W(P_W5, H4, dk, RK, S->l4[j].a);
                    ACC_W(P_W4, H3, dy4, H4, S->l3[j].a);
                    ACC_W(P_W3, H2, dy3, H3, S->l2[j].a);
                    ACC_W(P_W2, H1, dy2, H2, S->l1[j].a);
                    ACC_W(P_W1, 6, dy1, H1, S->x6[j]);
#undef ACC_W
                    float* bq = bsub[j & 3];
                    for (int o = 0; o < RK; ++o) bq[0 * 64 + o] = bq[0 * 64 + o] + dk[o];
                    for (int o = 0; o < H4; ++o) bq[1 * 64 + o] = bq[1 * 64 + o] + dy4[o];
                    for (int o = 0; o < H3; ++o) bq[2 * 64 + o] = bq[2 * 64 + o] + dy3[o];
                    for (int o = 0; o < H2; ++o) bq[3 * 64 + o] = bq[3 * 64 + o] + dy2[o];
                    for (int o = 0; o < H1; ++o) bq[4 * 64 + o] = bq[4 * 64 + o] + dy1[o];
                    float* gq = gsl[j & 15];
#define ACC_LN(L, NH, DZ, ST)                                                                          \
    for (int o = 0; o < (NH); ++o) {                                                                  \
        gq[(0 * 4 + (L)) * 64 + o] = gq[(0 * 4 + (L)) * 64 + o] + (DZ)[o] * (ST).xh[o];               \
        gq[(1 * 4 + (L)) * 64 + o] = gq[(1 * 4 + (L)) * 64 + o] + (DZ)[o];                            \
    }
                    ACC_LN(3, H4, dz4, S->l4[j]); ACC_LN(2, H3, dz3, S->l3[j]); ACC_LN(1, H2, dz2, S->l2[j]); ACC_LN(0, H1, dz1, S->l1[j]);
#undef ACC_LN
                }
            }
            /* fold the slot sums into the part */
            static const int boff[5] = {P_B5, P_B4, P_B3, P_B2, P_B1}, bn[5] = {RK, H4, H3, H2, H1};
            for (int l = 0; l < 5; ++l)
                for (int o = 0; o < bn[l]; ++o)
                    part[boff[l] + o] = ((bsub[0][l * 64 + o] + bsub[1][l * 64 + o]) + bsub[2][l * 64 + o]) + bsub[3][l * 64 + o];
            static const int goff[4] = {P_G1, P_G2, P_G3, P_G4}, eoff[4] = {P_E1, P_E2, P_E3, P_E4}, hn[4] = {H1, H2, H3, H4};
            for (int l = 0; l < 4; ++l)
                for (int o = 0; o < hn[l]; ++o) {
                    float ag = 0.f, ab = 0.f;
                    for (int s = 0; s < 16; ++s) { ag = ag + gsl[s][(0 * 4 + l) * 64 + o]; ab = ab + gsl[s][(1 * 4 + l) * 64 + o]; }
                    part[goff[l] + o] = ag;
                    part[eoff[l] + o] = ab;
                }
        }
        free(S); free(bsub); free(gsl);
    }
    for (int q = 0; q < P_TOTAL; ++q) {
        float acc = 0.f;
        for (int w0 = 0; w0 < n_parts; w0 += 32) {
            float ga = 0.f;
            for (int w = w0; w < n_parts && w < w0 + 32; ++w) ga = ga + parts[(size_t)w * P_TOTAL + q];
            acc = acc + ga;
        }
        grad_params[q] = acc;
    }
    free(parts);
    return 0;
}
