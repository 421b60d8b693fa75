/*
 * CPU restatement of the SPRIN point encoder that feeds the CPPF pair path.
 *
 * TEST INFRASTRUCTURE ONLY.  Like cppf_oracle.c, this file is the checker: it is compiled into
 * liboracle.so and may be called from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
 * never from cppf_amd/.
 *
 * PARITY PINNED: the reference module (models/model.py:PointEncoder, models/sprin.py) is importable on
 * CPU; tests/golden/make_golden.py runs it with seeded weights and stores inputs, neighbour sets and the
 * [N,40] output (tests/golden/sprin_*.npz).  tests/test_oracle_golden.py checks this file against those
 * vectors to 2e-5 (the reference sums in ATen's order, this file sums sequentially).
 *
 * What is restated (paths relative to qq456cvb/CPPF):
 *   PointEncoder.forward / forward_nbrs        models/model.py:46-78
 *   rifeat                                     models/sprin.py:40-61
 *   conv_kernel (Linear, LayerNorm, ReLU)*     models/sprin.py:64-72
 *   SparseSO3Conv (rank contraction, outnet)   models/sprin.py:87-107
 *   GlobalInfoProp                             models/sprin.py:75-84
 *
 * Arithmetic conventions (the HIP kernels in cppf_amd/csrc/sprin.hip follow the same ones -- with `order` = 1,
 * the summation order of the MFMA kernel-MLP, see linear_ord / layer_norm_ord -- so HIP vs oracle is bit-exact): compiled with -ffp-contract=off; every Linear is a bias-seeded fmaf chain over
 * ascending input index; every other sum is sequential over ascending index with separate multiply and
 * add; LayerNorm is mean = sum/n, var = sum((y-mean)^2)/n, z = ((y-mean) * (1/sqrt(var+1e-5))) * g + b
 * with correctly rounded sqrt and divide; neighbour order is ascending point index.
 *
 * Packed parameter layout (float32), produced by cppf_amd.models.sprin.pack_point_encoder and by
 * oracle.py:pack_point_encoder from a state_dict -- per SparseSO3Conv+GlobalInfoProp layer:
 *   for each hidden width h_i (input in_i = 6 or h_{i-1}):  W[h_i][in_i], b[h_i], ln_g[h_i], ln_b[h_i]
 *   Wk[rank][h_last], bk[rank]
 *   Wo_t[rank*n_in][n_out] (outnet weight TRANSPOSED), bo[n_out], ln_g[n_out], ln_b[n_out]
 *   Wa[n_glob][n_out], ba[n_glob]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define SPRIN_MAX_W 256  /* widest hidden layer the oracle accepts */

/* ---- k nearest neighbours (PointEncoder.forward: torch.topk(dist, k, largest=False), models/model.py:47) ----
 * key = dist[i][j] when `dist` is given, else the squared distance ((dx*dx + dy*dy) + dz*dz) (same order).
 * The k smallest keys, ties to the lower index; output sorted by ascending index. */
typedef struct { float key; int32_t idx; } knn_ent;
static int knn_cmp(const void* a, const void* b)
{
    const knn_ent *x = a, *y = b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static int idx_cmp(const void* a, const void* b) { return (*(const int32_t*)a > *(const int32_t*)b) - (*(const int32_t*)a < *(const int32_t*)b); }

int orc_knn(const float* pc, const float* dist, int N, int k, int32_t* out)
{
    if (k > N || k <= 0) return -1;
#pragma omp parallel
    {
        knn_ent* e = malloc(sizeof(knn_ent) * (size_t)N);
#pragma omp for schedule(static)
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j) {
                float key;
                if (dist) key = dist[(size_t)i * N + j];
                else {
                    const float dx = pc[3 * j] - pc[3 * i], dy = pc[3 * j + 1] - pc[3 * i + 1], dz = pc[3 * j + 2] - pc[3 * i + 2];
                    key = (dx * dx + dy * dy) + dz * dz;
                }
                e[j].key = key; e[j].idx = j;
            }
            qsort(e, (size_t)N, sizeof(knn_ent), knn_cmp);
            int32_t* o = out + (size_t)i * k;
            for (int j = 0; j < k; ++j) o[j] = e[j].idx;
            qsort(o, (size_t)k, sizeof(int32_t), idx_cmp);
        }
        free(e);
    }
    return 0;
}

/* ---- pieces ---- */
static inline float norm3(const float* v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
static inline float dot3p(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

static void linear(const float* W, const float* b, const float* x, int n_in, int n_out, float* y)
{
    for (int o = 0; o < n_out; ++o) {
        float acc = b[o];
        for (int k = 0; k < n_in; ++k) acc = fmaf(W[(size_t)o * n_in + k], x[k], acc);
        y[o] = acc;
    }
}
/* order 1 (the MFMA kernel's order, cppf_amd/csrc/sprin.hip): a layer whose input width is a multiple of 16
 * walks its inputs as k(s, g) = 16*(s/4) + 4*g + s%4 for s = 0.., g = 0..3 -- the D layout of one
 * v_mfma_f32_16x16x4_f32 layer is the B layout of the next (same rule as orc_k_order in cppf_oracle.c). */
static void linear_ord(const float* W, const float* b, const float* x, int n_in, int n_out, float* y, int order)
{
    if (order != 1 || n_in % 16 != 0) { linear(W, b, x, n_in, n_out, y); return; }
    for (int o = 0; o < n_out; ++o) {
        float acc = b[o];
        for (int s_ = 0; s_ < n_in / 4; ++s_)
            for (int g = 0; g < 4; ++g) {
                const int k = 16 * (s_ / 4) + 4 * g + (s_ % 4);
                acc = fmaf(W[(size_t)o * n_in + k], x[k], acc);
            }
        y[o] = acc;
    }
}
/* order 1: the n features of a row live in 4 lanes, lane g holding features 16*ob + 4*g + r; each lane sums its
 * own values (ob, then r ascending) and the four partial sums combine as (p0 + p1) + (p2 + p3). */
static float sum_ord1(const float* v, int n, int squared, float mean)
{
    float p[4];
    for (int g = 0; g < 4; ++g) {
        float acc = 0.f;
        for (int ob = 0; ob < n / 16; ++ob)
            for (int r = 0; r < 4; ++r) {
                const float x = v[16 * ob + 4 * g + r];
                if (squared) { const float d = x - mean; acc = acc + d * d; }
                else acc = acc + x;
            }
        p[g] = acc;
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}
static void layer_norm(float* y, int n, const float* g, const float* b);
static void layer_norm_ord(float* y, int n, const float* g, const float* b, int order)
{
    if (order != 1 || n % 16 != 0) { layer_norm(y, n, g, b); return; }
    const float mean = sum_ord1(y, n, 0, 0.f) / (float)n;
    const float inv = 1.0f / sqrtf(sum_ord1(y, n, 1, mean) / (float)n + 1e-5f);
    for (int o = 0; o < n; ++o) y[o] = ((y[o] - mean) * inv) * g[o] + b[o];
}
/* nn.LayerNorm(n), eps 1e-5, affine (models/sprin.py:68,93) */
static void layer_norm(float* y, int n, const float* g, const float* b)
{
    float s = 0.f;
    for (int o = 0; o < n; ++o) s = s + y[o];
    const float mean = s / (float)n;
    float v = 0.f;
    for (int o = 0; o < n; ++o) { const float d = y[o] - mean; v = v + d * d; }
    const float inv = 1.0f / sqrtf(v / (float)n + 1e-5f);
    for (int o = 0; o < n; ++o) y[o] = ((y[o] - mean) * inv) * g[o] + b[o];
}

/* One SparseSO3Conv layer (models/sprin.py:87-107 via models/model.py:57,60) for all points.
 * feat_in: NULL for the first layer (neighbour features are [|p_j - p_i|, n_j . n_i], models/model.py:50-55)
 * or [N][n_in] features gathered by neighbour index (models/model.py:59).  Returns a pointer past the
 * parameters consumed (the aggr parameters follow). */
int orc_sprin_conv(const float* pc, const float* nrm, const float* feat_in, int n_in, const int32_t* nbrs, int N, int k,
                   const float* params, const int32_t* hidden, int n_hidden, int rank, int n_out, int order, float* out)
{
    for (int i = 0; i < n_hidden; ++i)
        if (hidden[i] > SPRIN_MAX_W) return -1;
    if (rank > SPRIN_MAX_W || n_out > SPRIN_MAX_W) return -1;
#pragma omp parallel
    {
        float* kern = malloc(sizeof(float) * (size_t)k * rank);
        float* nf = malloc(sizeof(float) * (size_t)k * n_in);
        float* contracted = malloc(sizeof(float) * (size_t)rank * n_in);
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            const int32_t* nb = nbrs + (size_t)n * k;
            const float* s = pc + 3 * n;
            /* rifeat (models/sprin.py:40-61): r_mean over the k neighbours */
            float rm[3] = {0.f, 0.f, 0.f};
            for (int j = 0; j < k; ++j)
                for (int c = 0; c < 3; ++c) rm[c] = rm[c] + pc[3 * nb[j] + c];
            for (int c = 0; c < 3; ++c) rm[c] = rm[c] / (float)k;
            float l3[3] = {s[0] - rm[0], s[1] - rm[1], s[2] - rm[2]};
            const float l3n = norm3(l3);
            for (int j = 0; j < k; ++j) {
                const float* r = pc + 3 * nb[j];
                float l1[3] = {rm[0] - r[0], rm[1] - r[1], rm[2] - r[2]};
                float l2[3] = {r[0] - s[0], r[1] - s[1], r[2] - s[2]};
                const float l1n = norm3(l1), l2n = norm3(l2);
                float x[SPRIN_MAX_W], y[SPRIN_MAX_W];
                x[0] = l1n; x[1] = l2n; x[2] = l3n;
                x[3] = dot3p(l1, l2) / (l1n * l2n + 1e-7f);
                x[4] = dot3p(l2, l3) / (l2n * l3n + 1e-7f);
                x[5] = dot3p(l3, l1) / (l3n * l1n + 1e-7f);
                /* neighbour features */
                if (feat_in) {
                    for (int i = 0; i < n_in; ++i) nf[(size_t)j * n_in + i] = feat_in[(size_t)nb[j] * n_in + i];
                } else {
                    nf[(size_t)j * n_in + 0] = l2n;                                 /* torch.norm(pc_nbrs - pc), models/model.py:51 */
                    nf[(size_t)j * n_in + 1] = dot3p(nrm + 3 * nb[j], nrm + 3 * n); /* models/model.py:54 */
                }
                /* conv_kernel(6, rank, *hidden) (models/sprin.py:64-72) */
                const float* p = params;
                int in = 6;
                for (int i = 0; i < n_hidden; ++i) {
                    const int h = hidden[i];
                    linear_ord(p, p + (size_t)h * in, x, in, h, y, order);
                    p += (size_t)h * in + h;
                    layer_norm_ord(y, h, p, p + h, order);
                    p += 2 * h;
                    for (int o = 0; o < h; ++o) x[o] = y[o] > 0.f ? y[o] : 0.f;
                    in = h;
                }
                linear_ord(p, p + (size_t)rank * in, x, in, rank, kern + (size_t)j * rank, order);
            }
            const float* p = params;
            {
                int in = 6;
                for (int i = 0; i < n_hidden; ++i) { p += (size_t)hidden[i] * in + 3 * hidden[i]; in = hidden[i]; }
                p += (size_t)rank * in + rank;
            }
            /* einsum("bnkr,bnki->bnri").flatten(-2) (models/sprin.py:99) */
            for (int r = 0; r < rank; ++r)
                for (int i = 0; i < n_in; ++i) {
                    float acc = 0.f;
                    for (int j = 0; j < k; ++j) acc = fmaf(kern[(size_t)j * rank + r], nf[(size_t)j * n_in + i], acc);
                    contracted[r * n_in + i] = acc;
                }
            /* outnet + LayerNorm (models/sprin.py:100,105) */
            const int C = rank * n_in;
            float y[SPRIN_MAX_W];
            const float* bo = p + (size_t)C * n_out;
            for (int o = 0; o < n_out; ++o) {
                float acc = bo[o];
                for (int c = 0; c < C; ++c) acc = fmaf(p[(size_t)c * n_out + o], contracted[c], acc);
                y[o] = acc;
            }
            layer_norm(y, n_out, bo + n_out, bo + 2 * n_out);
            for (int o = 0; o < n_out; ++o) out[(size_t)n * n_out + o] = y[o];
        }
        free(kern); free(nf); free(contracted);
    }
    return 0;
}

/* floats consumed by one conv layer (without the aggr block) */
int64_t orc_sprin_conv_params(const int32_t* hidden, int n_hidden, int rank, int n_in, int n_out)
{
    int64_t n = 0;
    int in = 6;
    for (int i = 0; i < n_hidden; ++i) { n += (int64_t)hidden[i] * in + 3 * hidden[i]; in = hidden[i]; }
    n += (int64_t)rank * in + rank;
    n += (int64_t)rank * n_in * n_out + 3 * n_out;
    return n;
}

/* GlobalInfoProp (models/sprin.py:75-84): out[n] = [conv[n] (n_out), max_n(linear(conv))[n_glob]] */
void orc_sprin_global(const float* conv, int N, int n_out, int n_glob, const float* Wa, const float* ba, float* out)
{
    float glob[SPRIN_MAX_W];
    for (int g = 0; g < n_glob; ++g) glob[g] = -INFINITY;
    for (int n = 0; n < N; ++n) {
        float t[SPRIN_MAX_W];
        linear(Wa, ba, conv + (size_t)n * n_out, n_out, n_glob, t);
        for (int g = 0; g < n_glob; ++g) glob[g] = t[g] > glob[g] ? t[g] : glob[g];
    }
    const int W = n_out + n_glob;
    for (int n = 0; n < N; ++n) {
        for (int o = 0; o < n_out; ++o) out[(size_t)n * W + o] = conv[(size_t)n * n_out + o];
        for (int g = 0; g < n_glob; ++g) out[(size_t)n * W + n_out + g] = glob[g];
    }
}

/* PointEncoder.forward_nbrs (models/model.py:63-78): num_layers conv+aggr stages.  out: [N][n_out+n_glob]. */
int orc_point_encoder(const float* pc, const float* nrm, const int32_t* nbrs, int N, int k, const float* params,
                      const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob,
                      int num_layers, int order, float* out)
{
    const int W = n_out + n_glob;
    float* conv = malloc(sizeof(float) * (size_t)N * n_out);
    float* prev = num_layers > 1 ? malloc(sizeof(float) * (size_t)N * W) : NULL;
    const float* p = params;
    int rc = 0;
    for (int l = 0; l < num_layers && rc == 0; ++l) {
        const int n_in = l == 0 ? n_nbr_feats : W;
        if (l > 0) memcpy(prev, out, sizeof(float) * (size_t)N * W);
        rc = orc_sprin_conv(pc, nrm, l == 0 ? NULL : prev, n_in, nbrs, N, k, p, hidden, n_hidden, rank, n_out, order, conv);
        p += orc_sprin_conv_params(hidden, n_hidden, rank, n_in, n_out);
        orc_sprin_global(conv, N, n_out, n_glob, p, p + (size_t)n_glob * n_out, out);
        p += (size_t)n_glob * n_out + n_glob;
    }
    free(conv); free(prev);
    return rc;
}
