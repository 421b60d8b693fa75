/*
 * CPU restatement of the BACKWARD of PPFEncoder.forward_with_idx.
 *
 * TEST INFRASTRUCTURE ONLY (see cppf_oracle.c): compiled into liboracle.so, called from tests/ only.
 *
 * The reference has no backward code of its own: train.py:91 calls loss.backward() and torch autograd
 * differentiates models/model.py:117-137 (gather, PPF, cat, ResLayers, final).  PARITY PINNED on exactly
 * that: tests/golden/make_golden_bwd.py runs the imported reference module under autograd on CPU and stores
 * the gradients of every parameter and of `feat` for a fixed upstream gradient; tests/test_oracle_golden.py
 * compares this file against them (2e-5 relative to the gradient's scale; autograd sums in ATen's order).
 *
 * Two summation specs:
 *
 * (1) the standard architecture (F = 40, ppffcs = {84,32,32,16}, train.py:35) follows cppf_amd/csrc/pair_mlp_bwd.hip
 *     exactly (HIP vs oracle is bit-exact for every parameter gradient and for grad_feat).  "khid order" below is the
 *     order the fp32 MFMA walks a K-long contraction: for s < K/4, for g < 4: k = 16*(s/4) + 4*g + s%4.
 *       forward     = orc_pair_mlp order 1 (per-point tables TA | TB of layer 0, khid chains seeded by the bias);
 *       d(x3)[i]    = c0 + c1, c0 / c1 = chains from 0 over the khid order of the 144 padded outputs, first / second
 *                     half (k-steps 0..17 / 18..35), of fmaf(Wf[k][i], g[k], .)            (k >= out_dim: no term)
 *       d(h2)[i]    = relu'(h2[i]) * chain(khid 16) fmaf(fc2[k][i], d(x3)[k], .)
 *       d(x2)[i]    = chain(khid 16) fmaf(fc0[k][i], d(x3)[k], .) continued by chain(khid 16) fmaf(fc1[k][i], d(h2)[k], .)
 *       d(h1)[i]    = relu'(h1[i]) * chain(khid 32) fmaf(fc2[k][i], d(x2)[k], .)
 *       d(x1)[i]    = chain(khid 32) fmaf(fc1[k][i], d(h1)[k], .) seeded with d(x2)[i]     (identity skip)
 *       d(h0)[i]    = relu'(h0[i]) * chain(khid 32) fmaf(fc2[k][i], d(x1)[k], .)
 *       row[p]      = [d(h0) (32) | d(x1) (32)]
 *     Pairs are cut into tiles of 64; accumulator `part` w (0 <= w < n_parts) owns tiles w, w + n_parts, ... and
 *       part[w].W[o][i] = ONE fmaf chain from 0 over all of w's pairs in ascending order of fmaf(delta_o, x_i, .)
 *                         (final: g x x3; layer 2 fc2: d(x3) x h2, fc0: d(x3) x x2, fc1: d(h2) x x2; layer 1 fc2:
 *                         d(x2) x h1, fc1: d(h1) x x1; layer 0 fc2: d(x1) x h0, fc1[:, 80:84]: d(h0) x ppf,
 *                         fc0[:, 80:84]: d(x1) x ppf);
 *       part[w].b[o]    = ((s0 + s1) + s2) + s3, s_q = sequential sum from 0 over w's pairs p with p % 4 == q;
 *       grad            = sum over groups of 32 consecutive parts (ascending) of the group's own ascending sum.
 *     The 2 x 40 feature columns of layer 0 are handled per point:
 *       S[n][0:64]   = sequential sum from 0 of row[p] over the pairs with a == n, ascending; S[n][64:128]: b == n;
 *       grad_feat[n][k] = chain over c = 0..63 of fmaf(Wa[c][k], S[n][c], .) continued over fmaf(Wb[c][k], S[n][64+c], .)
 *                         with Wa[c] = fc1[c][0:40] (c < 32) | fc0[c-32][0:40], Wb the same rows, columns 40:80;
 *       d(fc1 | fc0)[o][40*role + k] = sequential sum over chunks of 64 points (ascending) of the chunk's fmaf chain
 *                         from 0 over its points of fmaf(S[n][64*role + (o | 32 + o)], feat[n][k], .).
 *
 * (2) any other ResLayer stack (the device path uses the torch composite there; this only validates the restatement
 *     against autograd), natural order:
 *     (the formulation of an earlier lane-per-pair kernel)
 *   - pairs are cut into tiles of 64 consecutive pairs; accumulator `part` w (0 <= w < n_parts) owns tiles
 *     w, w + n_parts, ... in ascending order;
 *   - per pair: forward in natural k order (bias-seeded fmaf chains, orc_pair_mlp order 0), then
 *       d(in)_i  = chain over outputs o ascending of fmaf(W[o][i], d(out)_o, .) from 0,
 *       ReLU mask (h > 0), residual: d(x) = d(fc1 path) + d(fc0 path) (or + d(out) for an identity skip);
 *   - per tile and weight W[o][i]:  tile_sum = chain over the tile's pairs j ascending of
 *       fmaf(delta_o(j), x_i(j), .) from 0;  part[w] = part[w] + tile_sum  (biases: plain adds of delta_o(j));
 *   - grad = sum over groups of 32 consecutive parts (ascending) of the group's own ascending sum.
 *   - grad_feat[n] = sequential fp32 sum (from 0) of the per-pair rows d(x0)[0:F] of every pair with a == n in
 *     ascending pair order, then of the rows d(x0)[F:2F] of every pair with b == n in ascending pair order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define BW_MAXD 256
#define BW_MAXL 8
#define BW_TILE 64

void orc_ppf_features(const float* pc, const float* nrm, const int64_t* idxs, int64_t P, float* out);

typedef struct {
    float x[BW_MAXL + 1][BW_MAXD];   /* x[l] = input of res layer l; x[n_res] = input of the final linear */
    float h[BW_MAXL][BW_MAXD];       /* relu(fc1 x) */
    float dy[BW_MAXL + 1][BW_MAXD];  /* dy[l] = gradient wrt x[l] (dy[n_res] = wrt final input) */
    float dh[BW_MAXL][BW_MAXD];      /* gradient wrt h (masked) */
} pair_state;

static void chain_fwd(const float* W, const float* b, const float* x, int K, int Nn, float* y)
{
    for (int o = 0; o < Nn; ++o) {
        float acc = b[o];
        for (int k = 0; k < K; ++k) acc = fmaf(W[(size_t)o * K + k], x[k], acc);
        y[o] = acc;
    }
}
/* d_in[i] = chain over o of fmaf(W[o][i], d_out[o], .) from 0 */
static void chain_bwd(const float* W, const float* d_out, int K, int Nn, float* d_in)
{
    for (int i = 0; i < K; ++i) {
        float acc = 0.f;
        for (int o = 0; o < Nn; ++o) acc = fmaf(W[(size_t)o * K + i], d_out[o], acc);
        d_in[i] = acc;
    }
}

static int backward_natural(const float* pc, const float* nrm, const float* feat, const int64_t* idxs, int64_t N, int F,
                          int64_t P, const float* params, const int64_t* offs, const int* dims, int n_res, int out_dim,
                          const float* grad_out, int n_parts, int64_t n_params, float* grad_params, float* grad_feat)
{
    if (dims[0] != 2 * F + 4 || n_res > BW_MAXL || out_dim > 4 * BW_MAXD) return -1;
    for (int i = 0; i <= n_res; ++i)
        if (dims[i] > BW_MAXD) return -1;
    const int64_t n_tiles = (P + BW_TILE - 1) / BW_TILE;
    if (n_parts < 1) return -1;
    float* parts = calloc((size_t)n_parts * n_params, sizeof(float));
    float* dxall = malloc(sizeof(float) * (size_t)(P > 0 ? P : 1) * 2 * F);
    const int KL = dims[n_res];

#pragma omp parallel
    {
        pair_state* S = malloc(sizeof(pair_state) * BW_TILE);
#pragma omp for schedule(dynamic, 1)
        for (int w = 0; w < n_parts; ++w) {
            float* part = parts + (size_t)w * n_params;
            for (int64_t t = w; t < n_tiles; t += n_parts) {
                const int nv = (int)((P - t * BW_TILE) < BW_TILE ? (P - t * BW_TILE) : BW_TILE);
                for (int j = 0; j < nv; ++j) {
                    const int64_t p = t * BW_TILE + j;
                    const int64_t a = idxs[2 * p], b = idxs[2 * p + 1];
                    pair_state* s = &S[j];
                    memcpy(s->x[0], feat + a * F, sizeof(float) * F);
                    memcpy(s->x[0] + F, feat + b * F, sizeof(float) * F);
                    orc_ppf_features(pc, nrm, idxs + 2 * p, 1, s->x[0] + 2 * F);
                    /* forward (models/model.py:27-31,134-137), natural order */
                    for (int l = 0; l < n_res; ++l) {
                        const int K = dims[l], Nn = dims[l + 1];
                        const int64_t* o = offs + 6 * l;
                        float y2[BW_MAXD], y0[BW_MAXD];
                        chain_fwd(params + o[0], params + o[1], s->x[l], K, Nn, s->h[l]);
                        for (int q = 0; q < Nn; ++q) s->h[l][q] = s->h[l][q] > 0.f ? s->h[l][q] : 0.f;
                        chain_fwd(params + o[2], params + o[3], s->h[l], Nn, Nn, y2);
                        if (o[4] >= 0) {
                            chain_fwd(params + o[4], params + o[5], s->x[l], K, Nn, y0);
                            for (int q = 0; q < Nn; ++q) s->x[l + 1][q] = y2[q] + y0[q];
                        } else {
                            for (int q = 0; q < Nn; ++q) s->x[l + 1][q] = y2[q] + s->x[l][q];
                        }
                    }
                    /* backward */
                    const float* g = grad_out + p * out_dim;
                    const int64_t* of = offs + 6 * n_res;
                    chain_bwd(params + of[0], g, KL, out_dim, s->dy[n_res]);
                    for (int l = n_res - 1; l >= 0; --l) {
                        const int K = dims[l], Nn = dims[l + 1];
                        const int64_t* o = offs + 6 * l;
                        float t1[BW_MAXD], t2[BW_MAXD];
                        chain_bwd(params + o[2], s->dy[l + 1], Nn, Nn, s->dh[l]);
                        for (int q = 0; q < Nn; ++q) s->dh[l][q] = s->h[l][q] > 0.f ? s->dh[l][q] : 0.f;
                        chain_bwd(params + o[0], s->dh[l], K, Nn, t1);
                        if (o[4] >= 0) {
                            chain_bwd(params + o[4], s->dy[l + 1], K, Nn, t2);
                            for (int i = 0; i < K; ++i) s->dy[l][i] = t1[i] + t2[i];
                        } else {
                            for (int i = 0; i < K; ++i) s->dy[l][i] = t1[i] + s->dy[l + 1][i];
                        }
                    }
                    (void)a; (void)b;
                    memcpy(dxall + (size_t)p * 2 * F, s->dy[0], sizeof(float) * 2 * F);
                }
                /* weight gradients of this tile, then part += tile_sum */
#define OUTER(OFFW, OFFB, DELTA, XIN, O_, I_)                                                          \
    do {                                                                                              \
        for (int o_ = 0; o_ < (O_); ++o_) {                                                           \
            for (int i_ = 0; i_ < (I_); ++i_) {                                                       \
                float acc = 0.f;                                                                      \
                for (int j = 0; j < nv; ++j) acc = fmaf(DELTA(j)[o_], XIN(j)[i_], acc);               \
                part[(OFFW) + (int64_t)o_ * (I_) + i_] = part[(OFFW) + (int64_t)o_ * (I_) + i_] + acc; \
            }                                                                                         \
            float accb = 0.f;                                                                         \
            for (int j = 0; j < nv; ++j) accb = accb + DELTA(j)[o_];                                  \
            part[(OFFB) + o_] = part[(OFFB) + o_] + accb;                                             \
        }                                                                                             \
    } while (0)
                {
                    const int64_t* of = offs + 6 * n_res;
#define D_G(j) (grad_out + (t * BW_TILE + (j)) * out_dim)
#define X_L(j) (S[j].x[n_res])
                    OUTER(of[0], of[1], D_G, X_L, out_dim, KL);
#undef D_G
#undef X_L
                }
                for (int l = 0; l < n_res; ++l) {
                    const int K = dims[l], Nn = dims[l + 1];
                    const int64_t* o = offs + 6 * l;
#define D_Y(j) (S[j].dy[l + 1])
#define D_H(j) (S[j].dh[l])
#define X_H(j) (S[j].h[l])
#define X_X(j) (S[j].x[l])
                    OUTER(o[2], o[3], D_Y, X_H, Nn, Nn);
                    OUTER(o[0], o[1], D_H, X_X, Nn, K);
                    if (o[4] >= 0) OUTER(o[4], o[5], D_Y, X_X, Nn, K);
#undef D_Y
#undef D_H
#undef X_H
#undef X_X
                }
            }
        }
        free(S);
    }
    for (int64_t q = 0; q < n_params; ++q) {
        float acc = 0.f;
        for (int w0 = 0; w0 < n_parts; w0 += 32) {
            float ga = 0.f;
            for (int w = w0; w < n_parts && w < w0 + 32; ++w) ga = ga + parts[(size_t)w * n_params + q];
            acc = acc + ga;
        }
        grad_params[q] = acc;
    }
    for (size_t q = 0; q < (size_t)N * F; ++q) grad_feat[q] = 0.f;
    for (int half = 0; half < 2; ++half)
        for (int64_t p = 0; p < P; ++p) {
            float* g = grad_feat + idxs[2 * p + half] * F;
            const float* row = dxall + (size_t)p * 2 * F + (size_t)half * F;
            for (int c = 0; c < F; ++c) g[c] = g[c] + row[c];
        }
    free(parts); free(dxall);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * spec (1): the standard architecture, the order of cppf_amd/csrc/pair_mlp_bwd.hip
 * ------------------------------------------------------------------------------------------------------------------ */
static void khid_order(int K, int* perm)
{
    int n = 0;
    for (int s = 0; s < K / 4; ++s)
        for (int g = 0; g < 4; ++g) perm[n++] = 16 * (s / 4) + 4 * g + (s % 4);
}
/* y[o] = chain in perm order of fmaf(W[o][k], x[k], .) seeded with seed[o];  W is [Nn][K] */
static void fwd_chain(const float* W, const float* seed, const float* x, const int* perm, int K, int Nn, float* y)
{
    for (int o = 0; o < Nn; ++o) {
        float acc = seed[o];
        for (int q = 0; q < K; ++q) acc = fmaf(W[(size_t)o * K + perm[q]], x[perm[q]], acc);
        y[o] = acc;
    }
}
/* acc[i] = chain in perm order (entries q0 <= q < q1) over outputs k of fmaf(W[k][i], d[k], acc[i]);  W is [Nn][K], k < kmax */
static void bwd_chain(const float* W, const float* d, const int* perm, int q0, int q1, int kmax, int K, float* acc)
{
    for (int i = 0; i < K; ++i) {
        float a = acc[i];
        for (int q = q0; q < q1; ++q)
            if (perm[q] < kmax) a = fmaf(W[(size_t)perm[q] * K + i], d[perm[q]], a);
        acc[i] = a;
    }
}

typedef struct {
    float ppf[4], h0[32], x1[32], h1[32], x2[32], h2[16], x3[16];
    float dy3[16], dh2[16], dy2[32], dh1[32], dy1[32], dh0[32];
} std_state;

static int backward_std(const float* pc, const float* nrm, const float* feat, const int64_t* idxs, int64_t N, int64_t P,
                        const float* params, const int64_t* offs, int out_dim, const float* grad_out, int n_parts,
                        int64_t n_params, float* grad_params, float* grad_feat)
{
    enum { F = 40, D0 = 84 };
    if (out_dim > 144 || n_parts < 1) return -1;
    const float *w1_0 = params + offs[0], *b1_0 = params + offs[1], *w2_0 = params + offs[2], *b2_0 = params + offs[3];
    const float *w0_0 = params + offs[4], *b0_0 = params + offs[5];
    const float *w1_1 = params + offs[6], *b1_1 = params + offs[7], *w2_1 = params + offs[8], *b2_1 = params + offs[9];
    const float *w1_2 = params + offs[12], *b1_2 = params + offs[13], *w2_2 = params + offs[14], *b2_2 = params + offs[15];
    const float *w0_2 = params + offs[16], *b0_2 = params + offs[17];
    const float* wf = params + offs[18];
    int kh16[16], kh32[32], kh144[144];
    khid_order(16, kh16); khid_order(32, kh32); khid_order(144, kh144);

    /* layer-0 tables (csrc/pair_mlp.hip:point_proj_kernel): T[n] = {TA1 (32) | TA0 (32) | TB1 (32) | TB0 (32)} */
    float* tab = malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * 128);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int r = 0; r < 128; ++r) {
            const int oc = r & 63;
            const float* w = oc < 32 ? w1_0 + oc * D0 : w0_0 + (oc - 32) * D0;
            float acc = r < 64 ? (oc < 32 ? b1_0[oc] : b0_0[oc - 32]) : 0.f;
            for (int k = 0; k < F; ++k) acc = fmaf(w[(r < 64 ? 0 : F) + k], feat[n * F + k], acc);
            tab[n * 128 + r] = acc;
        }

    const int64_t n_tiles = (P + BW_TILE - 1) / BW_TILE;
    float* parts = calloc((size_t)n_parts * n_params, sizeof(float));
    float* rows = malloc(sizeof(float) * (size_t)(P > 0 ? P : 1) * 64);
    static const float zero32[32] = {0};

#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < n_parts; ++w) {
        float* part = parts + (size_t)w * n_params;
        float bsub[4][19 * 16];          /* bias sub-sums per p % 4: g (9 blocks) | dy3 | dh2 | dy2 (2) | dh1 (2) | dy1 (2) | dh0 (2) */
        memset(bsub, 0, sizeof(bsub));
        for (int64_t t = w; t < n_tiles; t += n_parts) {
            const int nv = (int)((P - t * BW_TILE) < BW_TILE ? (P - t * BW_TILE) : BW_TILE);
            for (int jj = 0; jj < nv; ++jj) {
                const int64_t p = t * BW_TILE + jj;
                const int64_t a = idxs[2 * p], b = idxs[2 * p + 1];
                std_state s;
                float pre1[32], y0[32], a2[32], a1[32], c0[16], c1[16];
                orc_ppf_features(pc, nrm, idxs + 2 * p, 1, s.ppf);
                const float *ta = tab + a * 128, *tb = tab + b * 128 + 64;
                for (int o = 0; o < 32; ++o) {
                    float acc = ta[o] + tb[o];
                    for (int k = 0; k < 4; ++k) acc = fmaf(w1_0[o * D0 + 2 * F + k], s.ppf[k], acc);
                    pre1[o] = acc;
                    acc = ta[32 + o] + tb[32 + o];
                    for (int k = 0; k < 4; ++k) acc = fmaf(w0_0[o * D0 + 2 * F + k], s.ppf[k], acc);
                    y0[o] = acc;
                    s.h0[o] = pre1[o] > 0.f ? pre1[o] : 0.f;
                }
                fwd_chain(w2_0, b2_0, s.h0, kh32, 32, 32, a2);
                for (int o = 0; o < 32; ++o) s.x1[o] = a2[o] + y0[o];
                fwd_chain(w1_1, b1_1, s.x1, kh32, 32, 32, a1);
                for (int o = 0; o < 32; ++o) s.h1[o] = a1[o] > 0.f ? a1[o] : 0.f;
                fwd_chain(w2_1, b2_1, s.h1, kh32, 32, 32, a2);
                for (int o = 0; o < 32; ++o) s.x2[o] = a2[o] + s.x1[o];
                fwd_chain(w1_2, b1_2, s.x2, kh32, 32, 16, a1);
                fwd_chain(w0_2, b0_2, s.x2, kh32, 32, 16, y0);
                for (int o = 0; o < 16; ++o) s.h2[o] = a1[o] > 0.f ? a1[o] : 0.f;
                fwd_chain(w2_2, b2_2, s.h2, kh16, 16, 16, a2);
                for (int o = 0; o < 16; ++o) s.x3[o] = a2[o] + y0[o];
                /* backward-data */
                const float* g = grad_out + p * out_dim;
                memset(c0, 0, sizeof(c0)); memset(c1, 0, sizeof(c1));
                bwd_chain(wf, g, kh144, 0, 72, out_dim, 16, c0);
                bwd_chain(wf, g, kh144, 72, 144, out_dim, 16, c1);
                for (int i = 0; i < 16; ++i) s.dy3[i] = c0[i] + c1[i];
                memset(s.dh2, 0, sizeof(s.dh2));
                bwd_chain(w2_2, s.dy3, kh16, 0, 16, 16, 16, s.dh2);
                for (int i = 0; i < 16; ++i) s.dh2[i] = s.h2[i] > 0.f ? s.dh2[i] : 0.f;
                memset(s.dy2, 0, sizeof(s.dy2));
                bwd_chain(w0_2, s.dy3, kh16, 0, 16, 16, 32, s.dy2);
                bwd_chain(w1_2, s.dh2, kh16, 0, 16, 16, 32, s.dy2);
                memset(s.dh1, 0, sizeof(s.dh1));
                bwd_chain(w2_1, s.dy2, kh32, 0, 32, 32, 32, s.dh1);
                for (int i = 0; i < 32; ++i) s.dh1[i] = s.h1[i] > 0.f ? s.dh1[i] : 0.f;
                memcpy(s.dy1, s.dy2, sizeof(s.dy1));
                bwd_chain(w1_1, s.dh1, kh32, 0, 32, 32, 32, s.dy1);
                memset(s.dh0, 0, sizeof(s.dh0));
                bwd_chain(w2_0, s.dy1, kh32, 0, 32, 32, 32, s.dh0);
                for (int i = 0; i < 32; ++i) s.dh0[i] = s.h0[i] > 0.f ? s.dh0[i] : 0.f;
                memcpy(rows + (size_t)p * 64, s.dh0, sizeof(float) * 32);
                memcpy(rows + (size_t)p * 64 + 32, s.dy1, sizeof(float) * 32);
                /* weight gradients: one running chain per entry */
#define ACC_W(OFF, LD, COL0, DELTA, NO, XIN, NI)                                                        \
    for (int o_ = 0; o_ < (NO); ++o_)                                                                   \
        for (int i_ = 0; i_ < (NI); ++i_) {                                                             \
            float* q_ = part + (OFF) + (int64_t)o_ * (LD) + (COL0) + i_;                                \
            *q_ = fmaf((DELTA)[o_], (XIN)[i_], *q_);                                                    \
        }
                ACC_W(offs[18], 16, 0, g, out_dim, s.x3, 16);
                ACC_W(offs[14], 16, 0, s.dy3, 16, s.h2, 16);
                ACC_W(offs[16], 32, 0, s.dy3, 16, s.x2, 32);
                ACC_W(offs[12], 32, 0, s.dh2, 16, s.x2, 32);
                ACC_W(offs[8], 32, 0, s.dy2, 32, s.h1, 32);
                ACC_W(offs[6], 32, 0, s.dh1, 32, s.x1, 32);
                ACC_W(offs[2], 32, 0, s.dy1, 32, s.h0, 32);
                ACC_W(offs[0], D0, 2 * F, s.dh0, 32, s.ppf, 4);
                ACC_W(offs[4], D0, 2 * F, s.dy1, 32, s.ppf, 4);
#undef ACC_W
                float* bs = bsub[p & 3];
                for (int o = 0; o < out_dim; ++o) bs[o] = bs[o] + g[o];
                for (int o = 0; o < 16; ++o) { bs[144 + o] = bs[144 + o] + s.dy3[o]; bs[160 + o] = bs[160 + o] + s.dh2[o]; }
                for (int o = 0; o < 32; ++o) {
                    bs[176 + o] = bs[176 + o] + s.dy2[o]; bs[208 + o] = bs[208 + o] + s.dh1[o];
                    bs[240 + o] = bs[240 + o] + s.dy1[o]; bs[272 + o] = bs[272 + o] + s.dh0[o];
                }
            }
        }
#define BIAS_OUT(OFF, NO, SLOT)                                                                            \
    for (int o_ = 0; o_ < (NO); ++o_)                                                                      \
        part[(OFF) + o_] = ((bsub[0][(SLOT) + o_] + bsub[1][(SLOT) + o_]) + bsub[2][(SLOT) + o_]) + bsub[3][(SLOT) + o_];
        BIAS_OUT(offs[19], out_dim, 0);
        BIAS_OUT(offs[15], 16, 144); BIAS_OUT(offs[17], 16, 144);   /* layer 2 fc2.bias, fc0.bias: d(x3) */
        BIAS_OUT(offs[13], 16, 160);                                 /* layer 2 fc1.bias: d(h2) */
        BIAS_OUT(offs[9], 32, 176);                                  /* layer 1 fc2.bias: d(x2) */
        BIAS_OUT(offs[7], 32, 208);                                  /* layer 1 fc1.bias: d(h1) */
        BIAS_OUT(offs[3], 32, 240); BIAS_OUT(offs[5], 32, 240);     /* layer 0 fc2.bias, fc0.bias: d(x1) */
        BIAS_OUT(offs[1], 32, 272);                                  /* layer 0 fc1.bias: d(h0) */
#undef BIAS_OUT
    }
    (void)zero32;
    for (int64_t q = 0; q < n_params; ++q) {
        float acc = 0.f;
        for (int w0 = 0; w0 < n_parts; w0 += 32) {
            float ga = 0.f;
            for (int w = w0; w < n_parts && w < w0 + 32; ++w) ga = ga + parts[(size_t)w * n_params + q];
            acc = acc + ga;
        }
        grad_params[q] = acc;
    }
    /* per-point sums S[n] = {role a (64) | role b (64)}, pairs ascending */
    float* S = calloc((size_t)(N > 0 ? N : 1) * 128, sizeof(float));
    for (int half = 0; half < 2; ++half)
        for (int64_t p = 0; p < P; ++p) {
            float* sp = S + idxs[2 * p + half] * 128 + 64 * half;
            const float* row = rows + (size_t)p * 64;
            for (int c = 0; c < 64; ++c) sp[c] = sp[c] + row[c];
        }
    for (int64_t n = 0; n < N; ++n)
        for (int k = 0; k < F; ++k) {
            float acc = 0.f;
            for (int c = 0; c < 64; ++c) acc = fmaf((c < 32 ? w1_0 + c * D0 : w0_0 + (c - 32) * D0)[k], S[n * 128 + c], acc);
            for (int c = 0; c < 64; ++c) acc = fmaf((c < 32 ? w1_0 + c * D0 : w0_0 + (c - 32) * D0)[F + k], S[n * 128 + 64 + c], acc);
            grad_feat[n * F + k] = 0.f + acc;
        }
    /* feature columns of d(fc1.weight), d(fc0.weight): chunks of 64 points */
    for (int r = 0; r < 128; ++r)
        for (int k = 0; k < F; ++k) {
            float tot = 0.f;
            for (int64_t n0 = 0; n0 < N; n0 += 64) {
                float acc = 0.f;
                for (int64_t n = n0; n < N && n < n0 + 64; ++n) acc = fmaf(S[n * 128 + r], feat[n * F + k], acc);
                tot = tot + acc;
            }
            const int role = r >> 6, o = r & 63;
            grad_params[(o < 32 ? offs[0] + (int64_t)o * D0 : offs[4] + (int64_t)(o - 32) * D0) + F * role + k] = tot;
        }
    free(tab); free(parts); free(rows); free(S);
    return 0;
}

int orc_pair_mlp_backward(const float* pc, const float* nrm, const float* feat, const int64_t* idxs, int64_t N, int F,
                          int64_t P, const float* params, const int64_t* offs, const int* dims, int n_res, int out_dim,
                          const float* grad_out, int n_parts, int64_t n_params, float* grad_params, float* grad_feat)
{
    if (F == 40 && n_res == 3 && dims[0] == 84 && dims[1] == 32 && dims[2] == 32 && dims[3] == 16 && out_dim <= 144 &&
        offs[4] >= 0 && offs[10] < 0 && offs[16] >= 0)
        return backward_std(pc, nrm, feat, idxs, N, P, params, offs, out_dim, grad_out, n_parts, n_params, grad_params,
                            grad_feat);
    return backward_natural(pc, nrm, feat, idxs, N, F, P, params, offs, dims, n_res, out_dim, grad_out, n_parts, n_params,
                            grad_params, grad_feat);
}
