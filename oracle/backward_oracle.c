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
 * Deterministic summation spec shared with cppf_amd/csrc/pair_mlp_bwd.hip (so HIP vs oracle parameter
 * gradients are bit-exact):
 *   - pairs are cut into tiles of 64 consecutive pairs; accumulator `part` w (0 <= w < n_parts) owns tiles
 *     w, w + n_parts, ... in ascending order;
 *   - per pair: forward in natural k order (bias-seeded fmaf chains, orc_pair_mlp order 0), then
 *       d(in)_i  = chain over outputs o ascending of fmaf(W[o][i], d(out)_o, .) from 0,
 *       ReLU mask (h > 0), residual: d(x) = d(fc1 path) + d(fc0 path) (or + d(out) for an identity skip);
 *   - per tile and weight W[o][i]:  tile_sum = chain over the tile's pairs j ascending of
 *       fmaf(delta_o(j), x_i(j), .) from 0;  part[w] = part[w] + tile_sum  (biases: plain adds of delta_o(j));
 *   - grad = sum over groups of 32 consecutive parts (ascending) of the group's own ascending sum.
 *   - grad_feat[n] = sequential fp32 sum (from 0) of the per-pair rows d(x0)[0:F] of every pair with a == n in
 *     ascending pair order, then of the rows d(x0)[F:2F] of every pair with b == n in ascending pair order (the
 *     device sorts the 2P (point, entry) keys stably and adds in that order).
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

int orc_pair_mlp_backward(const float* pc, const float* nrm, const float* feat, const int64_t* idxs, int64_t N, int F,
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
