// SPRIN point encoder for gfx950 (C ABI in include/cppf.h).
//
// Replaces models/model.py:36-78 (PointEncoder) + models/sprin.py:40-107 of the reference:
//   knn_kernel           torch.topk(dist, k, largest=False)            models/model.py:47
//   sprin_conv_kernel    gather, rifeat, conv_kernel MLP, rank contraction, outnet, LayerNorm
//                                                                      models/model.py:48-57, models/sprin.py:40-107
//                        + GlobalInfoProp's linear and the workgroup's channel maxima (models/sprin.py:75-84)
//   sprin_fill_kernel    GlobalInfoProp: maximum over the workgroups, concat           models/sprin.py:75-84
//
// Arithmetic follows oracle/sprin_oracle.c to the bit (-ffp-contract=off, bias-seeded fmaf chains over
// ascending input index, sequential sums over ascending neighbour index, correctly rounded sqrt/divide),
// which is what lets the parity tests demand exact equality.  The work is ~0.4 MMAC per point in
// 6->32->64->32->32->32 per-neighbour MLPs with a LayerNorm after every hidden layer; one wavefront owns
// one point, one lane one neighbour, so every LayerNorm is lane-local and the weights (27 KB) arrive
// through the scalar cache as SGPR operands of v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/cppf.h"
#include "cppf_math.h"
#include "sprin_layout.h"
#include "compact.h"

using namespace cppf;
using namespace sprin;

namespace {

constexpr int KNN_WAVES = 4;    // queries per workgroup
constexpr int KNN_CAP = 512;    // candidate slots per query in LDS

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Selection key of point j for the query: order-preserving bits of dist[q][j], or of the exact squared
// distance ((dx*dx + dy*dy) + dz*dz), which is never negative so its raw bits already order correctly.
template <bool FROM_DIST>
__device__ __forceinline__ uint32_t knn_key(const float* __restrict__ pc, const float* __restrict__ row, float qx, float qy,
                                            float qz, int j)
{
    if (FROM_DIST) return f2ord(row[j]);
    const float dx = pc[3 * j] - qx, dy = pc[3 * j + 1] - qy, dz = pc[3 * j + 2] - qz;
    return __float_as_uint((dx * dx + dy * dy) + dz * dz) | 0x80000000u;  // == f2ord() for non-negative floats
}

// Sum of one int per lane over the wavefront (DPP butterflies inside each row of 16, then four readlanes).
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

// One wavefront per query, keys streamed twice, nothing kept per point:
//   1. every lane keeps the minimum of its keys (j = lane mod 64).  The k-th smallest of those 64 minima, T0,
//      bounds the k-th smallest key from above (the k minima below it are k distinct keys), and for points in
//      arbitrary order only ~3k keys are <= T0;
//   2. keys <= T0 are compacted, in ascending index order, into LDS (ballot prefix);
//   3. a 32-step bitwise bisection over the candidates finds the exact k-th smallest key (per-lane VALU counts
//      + one DPP wave sum per step: a ballot + scalar popcount per key costs a VALU->SGPR->SALU round trip);
//   4. candidates below it, and the first ties, are written out in slot (= index) order.
// More than KNN_CAP candidates (duplicate-heavy inputs) run the bisection over re-streamed keys instead.
template <bool FROM_DIST>
__device__ __forceinline__ void knn_body(const float* __restrict__ pc, const float* __restrict__ dist, int N, int k,
                                         int32_t* __restrict__ out, const int32_t* __restrict__ n_dev)
{
    if (n_dev) N = min(*n_dev, N);   // *_dyn: the launch is sized for a capacity, the point count is in memory
    __shared__ uint32_t cand_key[KNN_WAVES][KNN_CAP];
    __shared__ int cand_idx[KNN_WAVES][KNN_CAP];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int q = blockIdx.x * KNN_WAVES + w;
    if (q >= N) return;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (!FROM_DIST) { qx = pc[3 * q]; qy = pc[3 * q + 1]; qz = pc[3 * q + 2]; }
    const float* row = FROM_DIST ? dist + (size_t)q * N : nullptr;
    // 1. lane minima and their k-th smallest
    // (both streaming loops are unrolled 8x: a query is one wavefront and there are only N of them, so the
    //  loads of several chunks have to be in flight at once to cover L2 latency)
    uint32_t lmin = 0xffffffffu;
    {
        int j = lane;
        for (; j + 7 * 64 < N; j += 8 * 64) {
            uint32_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = knn_key<FROM_DIST>(pc, row, qx, qy, qz, j + u * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) lmin = kk[u] < lmin ? kk[u] : lmin;
        }
        for (; j < N; j += 64) {
            const uint32_t key = knn_key<FROM_DIST>(pc, row, qx, qy, qz, j);
            lmin = key < lmin ? key : lmin;
        }
    }
    uint32_t T0 = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t c = T0 | (1u << bit);
        if (__popcll(__ballot(lmin < c)) < k) T0 = c;
    }
    // 2. candidates
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int n_cand = 0;
    for (int j0 = 0; j0 < N; j0 += 8 * 64) {
        uint32_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * 64 + lane;
            kk[u] = j < N ? knn_key<FROM_DIST>(pc, row, qx, qy, qz, j) : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * 64 + lane;
            const bool c = j < N && kk[u] <= T0;
            const unsigned long long m = __ballot(c);
            const int pos = n_cand + __popcll(m & lt_mask);
            if (c && pos < KNN_CAP) { cand_key[w][pos] = kk[u]; cand_idx[w][pos] = j; }
            n_cand += __popcll(m);
        }
    }
    int32_t* o = out + (size_t)q * k;
    if (n_cand <= KNN_CAP) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before other lanes read them
        const int R = (n_cand + 63) >> 6;
        // 3. exact k-th smallest among the candidates (held in registers): two key bits per step, the three
        //    counts packed into one int (10 bits each, n_cand <= 512) so that a step costs one wave sum
        constexpr int RMAX = KNN_CAP / 64;
        uint32_t ck[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int s_ = r * 64 + lane;
            ck[r] = s_ < n_cand ? cand_key[w][s_] : 0xffffffffu;
        }
        uint32_t T = 0;
        for (int bit = 30; bit >= 0; bit -= 2) {
            const uint32_t c1 = T | (1u << bit), c2 = T | (2u << bit), c3 = T | (3u << bit);
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R) cnt += (ck[r] < c1 ? 1 : 0) + (ck[r] < c2 ? 1 << 10 : 0) + (ck[r] < c3 ? 1 << 20 : 0);
            cnt = wave_sum(cnt);
            const int n1 = cnt & 1023, n2 = (cnt >> 10) & 1023, n3 = cnt >> 20;
            T = n3 < k ? c3 : (n2 < k ? c2 : (n1 < k ? c1 : T));   // largest threshold with fewer than k keys below it
        }
        int below = 0;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) below += ck[r] < T ? 1 : 0;
        below = wave_sum(below);
        // 4. emit
        const int remaining = k - below;
        int base = 0, eq_seen = 0;
        for (int r = 0; r < R; ++r) {
            const int s_ = r * 64 + lane;
            const bool in = s_ < n_cand;
            const uint32_t key = in ? cand_key[w][s_] : 0xffffffffu;
            const bool eq = in && key == T;
            const unsigned long long eqm = __ballot(eq);
            const bool take = in && (key < T || (eq && eq_seen + __popcll(eqm & lt_mask) < remaining));
            const unsigned long long tm = __ballot(take);
            if (take) o[base + __popcll(tm & lt_mask)] = cand_idx[w][s_];
            base += __popcll(tm);
            eq_seen += __popcll(eqm);
        }
        return;
    }
    // rare: bisection over re-streamed keys
    uint32_t T = 0;
    int below = 0;
    for (int bit = 31; bit >= -1; --bit) {
        const uint32_t c = bit >= 0 ? (T | (1u << bit)) : T;
        int cnt = 0;
        for (int j = lane; j < N; j += 64) cnt += knn_key<FROM_DIST>(pc, row, qx, qy, qz, j) < c ? 1 : 0;
        cnt = wave_sum(cnt);
        if (bit < 0) below = cnt;
        else if (cnt < k) T = c;
    }
    const int remaining = k - below;
    int base = 0, eq_seen = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j < N;
        const uint32_t key = in ? knn_key<FROM_DIST>(pc, row, qx, qy, qz, j) : 0xffffffffu;
        const bool eq = in && key == T;
        const unsigned long long eqm = __ballot(eq);
        const bool take = in && (key < T || (eq && eq_seen + __popcll(eqm & lt_mask) < remaining));
        const unsigned long long tm = __ballot(take);
        if (take) o[base + __popcll(tm & lt_mask)] = j;
        base += __popcll(tm);
        eq_seen += __popcll(eqm);
    }
}

template <bool FROM_DIST>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_kernel(const float* __restrict__ pc, const float* __restrict__ dist,
                                                             int N, int k, int32_t* __restrict__ out,
                                                             const int32_t* __restrict__ n_dev)
{
    knn_body<FROM_DIST>(pc, dist, N, k, out, n_dev);
}

// The searches of a chain's members in ONE launch (cppf_point_encoder_forward_batch): blockIdx.y = member, blockIdx.x its query
// blocks (the grid is sized for the largest member; the others' surplus blocks leave at once).  A cloud of 700-2000 points is 700-2000
// wavefronts: one member fills a quarter of the chip's wave slots, four or eight of them fill it.
constexpr int SP_BATCH_MAX = 8;
struct KnnBatch { const float* pc[SP_BATCH_MAX]; int32_t* out[SP_BATCH_MAX]; const int32_t* n_dev[SP_BATCH_MAX]; int N[SP_BATCH_MAX]; int k[SP_BATCH_MAX]; };
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_batch_kernel(KnnBatch B)
{
    const int i = blockIdx.y;
    if (!B.out[i]) return;               // (a member whose neighbour sets are already there)
    knn_body<false>(B.pc[i], nullptr, B.N[i], B.k[i], B.out[i], B.n_dev[i]);
}

// --------------------------------------------------------------------------------------------- conv
__host__ __device__ constexpr int sp_per_wave(int n_in) { return 16 * SP_KSTRIDE + 64 * n_in + SP_RANK * n_in + 64 * 3 + SP_NOUT + 64 * 8; }
__host__ __device__ constexpr int sp_waves(int n_in) { return n_in <= 4 ? SP_WAVES_MAX : 4; }

struct ConvArgs {
    const float* pc;
    const float* nrm;
    const float* feat_in;   // null for the first layer, else [N][n_in]
    const int32_t* nbrs;    // [N][k]
    const float* params;    // this layer's parameters, natural layout (outnet part is used from here)
    const float* wimg;      // this layer's MFMA image (SPW_FLOATS), cppf_point_encoder_pack
    float* out;             // [N][out_stride], columns 0..31 written
    int N, k, n_in, out_stride;
    float* mixed_out;       // optional [N][32 * n_in]: the contraction, kept for the backward (training)
    const int32_t* n_dev;   // *_dyn: point count in memory (N is then the capacity the launch is sized for)
    // GlobalInfoProp (models/sprin.py:75-84) in the epilogue: linear(n_out -> n_glob) per point, maximum over the workgroup's
    // points to wgmax[blockIdx][32] (plain stores: no atomics, nothing to zero); sprin_fill_kernel finishes the maximum
    const float* glob_w;    // Wa[n_glob][n_out], ba[n_glob]
    uint32_t* wgmax;
    int n_glob;
};

// hidden = {32, 64, 32, 32}, rank 32, n_out 32 (train.py:34).  Dynamic LDS: the workgroup's weight image
// (SPW_FLOATS), then per wave  kern[16][33] | nf[64][n_in] | contracted[32*n_in] | r[64][3] | y[32] | x6[64][8]
__device__ __forceinline__ void sprin_conv_body(const ConvArgs& A)
{
    extern __shared__ __attribute__((aligned(16))) float sp_lds[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    __shared__ uint32_t gmax[32];
    __shared__ float gw[32 * SP_NOUT + 32];   // GlobalInfoProp's weights and bias, read in the epilogue
    const int N = A.n_dev ? min(*A.n_dev, A.N) : A.N;
    if (blockIdx.x * (blockDim.x >> 6) >= N) return;   // whole workgroup past the cloud (capacity launch; a shorter member of a batch)
    if (threadIdx.x < 32) gmax[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < A.n_glob * (SP_NOUT + 1); i += blockDim.x) gw[i] = A.glob_w[i];
    const int n_in = A.n_in, k = A.k;
    const int per_wave = sp_per_wave(n_in);
    float* Wl = sp_lds;                                   // 16-byte aligned image
    float* kern = sp_lds + SPW_FLOATS + (size_t)w * per_wave;
    float* nf = kern + 16 * SP_KSTRIDE;
    float* contracted = nf + 64 * n_in;
    float* rr = contracted + SP_RANK * n_in;
    float* yv = rr + 64 * 3;
    float* x6l = yv + SP_NOUT;
    for (int i = threadIdx.x; i < SPW_FLOATS / 4; i += blockDim.x)
        reinterpret_cast<f32x4*>(Wl)[i] = reinterpret_cast<const f32x4*>(A.wimg)[i];
    const int n = blockIdx.x * (blockDim.x >> 6) + w;
    const bool live = n < N;
    const int nc = live ? n : N - 1;
    const int jc = lane < k ? lane : k - 1;
    const int nb = A.nbrs[(size_t)nc * k + jc];
    const float rx = A.pc[3 * nb], ry = A.pc[3 * nb + 1], rz = A.pc[3 * nb + 2];
    const float sx = A.pc[3 * nc], sy = A.pc[3 * nc + 1], sz = A.pc[3 * nc + 2];
    rr[3 * lane] = rx; rr[3 * lane + 1] = ry; rr[3 * lane + 2] = rz;
    __syncthreads();
    // r_mean: sequential over neighbours (every lane redundantly; LDS broadcast reads)
    float mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll 8
    for (int j = 0; j < k; ++j) { mx = mx + rr[3 * j]; my = my + rr[3 * j + 1]; mz = mz + rr[3 * j + 2]; }
    mx = mx / (float)k; my = my / (float)k; mz = mz / (float)k;
    // rifeat (models/sprin.py:40-61)
    const float l1x = mx - rx, l1y = my - ry, l1z = mz - rz;
    const float l2x = rx - sx, l2y = ry - sy, l2z = rz - sz;
    const float l3x = sx - mx, l3y = sy - my, l3z = sz - mz;
    const float l1n = norm3(l1x, l1y, l1z), l2n = norm3(l2x, l2y, l2z), l3n = norm3(l3x, l3y, l3z);
    float x6[6];
    x6[0] = l1n; x6[1] = l2n; x6[2] = l3n;
    x6[3] = ((l1x * l2x + l1y * l2y) + l1z * l2z) / (l1n * l2n + 1e-7f);
    x6[4] = ((l2x * l3x + l2y * l3y) + l2z * l3z) / (l2n * l3n + 1e-7f);
    x6[5] = ((l3x * l1x + l3y * l1y) + l3z * l1z) / (l3n * l1n + 1e-7f);
    // neighbour features
    if (A.feat_in) {
        for (int i = 0; i < n_in; ++i) nf[lane * n_in + i] = A.feat_in[(size_t)nb * n_in + i];
    } else {
        const float nax = A.nrm[3 * nb], nay = A.nrm[3 * nb + 1], naz = A.nrm[3 * nb + 2];
        const float nsx = A.nrm[3 * nc], nsy = A.nrm[3 * nc + 1], nsz = A.nrm[3 * nc + 2];
        nf[lane * 2] = l2n;                                         // |p_j - p_i|     (models/model.py:50-51)
        nf[lane * 2 + 1] = (nax * nsx + nay * nsy) + naz * nsz;     // n_j . n_i       (models/model.py:53-54)
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) x6l[lane * 8 + c] = x6[c];
    x6l[lane * 8 + 6] = 0.f; x6l[lane * 8 + 7] = 0.f;
    __syncthreads();
    // conv_kernel(6, 32, 32, 64, 32, 32) (models/sprin.py:64-72) on MFMA, 16 neighbour rows at a time -- TWO row blocks in flight
    // (sp_mfma_layer2: shared weight reads, two independent MFMA chains per output block; one block's LayerNorm under the other's
    // MFMAs), their results contracted one after the other, in neighbour order
    {
        const int j = lane & 15, g = lane >> 4;
#pragma unroll 1
        for (int rb = 0; rb < 4; rb += 2) {
            f32x4 a1[2], a2[4], a3[2], a4[2], kr[2];
            f32x4 b1[2], b2[4], b3[2], b4[2], ks[2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) { a1[ob] = *reinterpret_cast<const f32x4*>(Wl + SPW_B1 + 16 * ob + 4 * g); b1[ob] = a1[ob]; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float bx = x6l[(16 * rb + j) * 8 + 4 * s + g], by = x6l[(16 * (rb + 1) + j) * 8 + 4 * s + g];
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const float w = Wl[SPW_L1 + (ob * 2 + s) * 64 + lane];
                    a1[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, bx, a1[ob], 0, 0, 0);
                    b1[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, by, b1[ob], 0, 0, 0);
                }
            }
            sp_ln_relu4<2>(a1, Wl + SPW_B1 + 32, Wl + SPW_B1 + 64, lane, g);
            sp_ln_relu4<2>(b1, Wl + SPW_B1 + 32, Wl + SPW_B1 + 64, lane, g);
            sp_mfma_layer2<2, 4>(Wl + SPW_L2, Wl + SPW_B2, a1, b1, a2, b2, lane, g);
            sp_ln_relu4<4>(a2, Wl + SPW_B2 + 64, Wl + SPW_B2 + 128, lane, g);
            sp_ln_relu4<4>(b2, Wl + SPW_B2 + 64, Wl + SPW_B2 + 128, lane, g);
            sp_mfma_layer2<4, 2>(Wl + SPW_L3, Wl + SPW_B3, a2, b2, a3, b3, lane, g);
            sp_ln_relu4<2>(a3, Wl + SPW_B3 + 32, Wl + SPW_B3 + 64, lane, g);
            sp_ln_relu4<2>(b3, Wl + SPW_B3 + 32, Wl + SPW_B3 + 64, lane, g);
            sp_mfma_layer2<2, 2>(Wl + SPW_L4, Wl + SPW_B4, a3, b3, a4, b4, lane, g);
            sp_ln_relu4<2>(a4, Wl + SPW_B4 + 32, Wl + SPW_B4 + 64, lane, g);
            sp_ln_relu4<2>(b4, Wl + SPW_B4 + 32, Wl + SPW_B4 + 64, lane, g);
            sp_mfma_layer2<2, 2>(Wl + SPW_L5, Wl + SPW_B5, a4, b4, kr, ks, lane, g);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int rbh = rb + half;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r) kern[j * SP_KSTRIDE + 16 * ob + 4 * g + r] = half ? ks[ob][r] : kr[ob][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes, before other lanes read them
                // einsum("bnkr,bnki->bnri") (models/sprin.py:99): contracted[r*n_in + i] accumulates these 16 neighbours,
                // sequentially and in neighbour order across the row blocks
                const int jn = min(16, k - 16 * rbh);
                for (int t = lane; t < SP_RANK * n_in; t += 64) {
                    const int r = t / n_in, i = t - r * n_in;
                    float acc = rbh == 0 ? 0.f : contracted[t];
#pragma unroll 8
                    for (int jj = 0; jj < jn; ++jj) acc = fmaf(kern[jj * SP_KSTRIDE + r], nf[(16 * rbh + jj) * n_in + i], acc);
                    contracted[t] = acc;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (kern is rewritten by the other half / the next trip)
            }
        }
    }
    const float* p = A.params + SP_NAT_KERNEL;   // outnet parameters follow the kernel-MLP in the natural layout
    const int C = SP_RANK * n_in;
    if (A.mixed_out && live)
        for (int t = lane; t < C; t += 64) A.mixed_out[(size_t)n * C + t] = contracted[t];
    __syncthreads();
    // outnet (transposed weights: lane o reads Wo_t[c][o], coalesced) + LayerNorm (models/sprin.py:100,105)
    const float* Wo = p;
    const float* bo = Wo + (size_t)C * SP_NOUT;
    const int o = lane & (SP_NOUT - 1);
    float acc = bo[o];
#pragma unroll 8
    for (int c = 0; c < C; ++c) acc = fmaf(Wo[(size_t)c * SP_NOUT + o], contracted[c], acc);
    if (lane < SP_NOUT) yv[lane] = acc;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < SP_NOUT; ++q) s = s + yv[q];
    const float mean = s / (float)SP_NOUT;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < SP_NOUT; ++q) { const float dd = yv[q] - mean; v = v + dd * dd; }
    const float inv = inv_sqrt_rn(v / (float)SP_NOUT + 1e-5f);   // = 1.0f / sqrtf(.), bit for bit
    const float z = ((acc - mean) * inv) * bo[SP_NOUT + o] + bo[2 * SP_NOUT + o];
    if (live && lane < SP_NOUT) A.out[(size_t)n * A.out_stride + lane] = z;
    // GlobalInfoProp's linear on the row just written: lane g owns channel g, a bias-seeded fmaf chain over ascending input
    // index (what sprin_glob_kernel did with one thread per point and 32 strided reads of the row), then the maximum -- exact
    // in any order -- over the workgroup's points
    if (lane < SP_NOUT) yv[lane] = z;     // (this wave's reads of yv above are done: LDS operations of a wave stay in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int gch = lane < A.n_glob ? lane : 0;
    const float* Wa = gw + gch * SP_NOUT;
    float ga = gw[A.n_glob * SP_NOUT + gch];
#pragma unroll 8
    for (int q = 0; q < SP_NOUT; ++q) ga = fmaf(Wa[q], yv[q], ga);
    if (live && lane < A.n_glob) atomicMax(&gmax[lane], f2ord(ga));
    __syncthreads();
    if (threadIdx.x < 32) A.wgmax[(size_t)blockIdx.x * 32 + threadIdx.x] = gmax[threadIdx.x];
}

__global__ __launch_bounds__(SP_WAVES_MAX * 64) void sprin_conv_kernel(ConvArgs A) { sprin_conv_body(A); }

// ... and the members' convolutions in one launch (blockIdx.y = member: its own cloud, neighbour sets, weight image -- members of
// different categories carry different encoders -- and output)
struct ConvBatch { ConvArgs item[SP_BATCH_MAX]; };
__global__ __launch_bounds__(SP_WAVES_MAX * 64) void sprin_conv_batch_kernel(ConvBatch B) { sprin_conv_body(B.item[blockIdx.y]); }

// GlobalInfoProp (models/sprin.py:75-84), second half: the maximum over the workgroups' maxima (every block recomputes it from
// L2: <= N/4 x 32 words) and the broadcast into columns n_out.. of every point's row.
__device__ __forceinline__ void sprin_fill_body(float* __restrict__ out, int N, int stride, int n_glob,
                                                const uint32_t* __restrict__ wgmax, int waves, const int32_t* __restrict__ n_dev)
{
    if (n_dev) N = min(*n_dev, N);
    if ((int64_t)blockIdx.x * 256 >= (int64_t)N * n_glob) return;
    __shared__ uint32_t part[256];
    __shared__ uint32_t gl[32];
    const int n_wgs = (N + waves - 1) / waves;        // the conv launch's workgroups that held points
    const int pad = n_glob <= 8 ? 8 : (n_glob <= 16 ? 16 : 32), rows = 256 / pad;   // thread = (row r, channel g)
    const int g = threadIdx.x & (pad - 1), r = threadIdx.x / pad;
    uint32_t m = 0u;
    if (g < n_glob) {
        int b = r;
        for (; b + 3 * rows < n_wgs; b += 4 * rows) {   // four independent loads in flight
            const uint32_t v0 = wgmax[(size_t)b * 32 + g], v1 = wgmax[(size_t)(b + rows) * 32 + g],
                           v2 = wgmax[(size_t)(b + 2 * rows) * 32 + g], v3 = wgmax[(size_t)(b + 3 * rows) * 32 + g];
            const uint32_t a = v0 > v1 ? v0 : v1, c = v2 > v3 ? v2 : v3;
            m = a > m ? a : m;
            m = c > m ? c : m;
        }
        for (; b < n_wgs; b += rows) { const uint32_t v = wgmax[(size_t)b * 32 + g]; m = v > m ? v : m; }
    }
    part[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x < n_glob) {
        uint32_t v = part[threadIdx.x];
        for (int k = 1; k < rows; ++k) { const uint32_t o = part[k * pad + threadIdx.x]; v = o > v ? o : v; }
        gl[threadIdx.x] = v;
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * n_glob) return;
    const int n = t / n_glob, c = t - n * n_glob;
    out[(size_t)n * stride + SP_NOUT + c] = ord2f(gl[c]);
}

__global__ __launch_bounds__(256) void sprin_fill_kernel(float* __restrict__ out, int N, int stride, int n_glob,
                                                         const uint32_t* __restrict__ wgmax, int waves,
                                                         const int32_t* __restrict__ n_dev)
{
    sprin_fill_body(out, N, stride, n_glob, wgmax, waves, n_dev);
}

struct FillBatch { float* out[SP_BATCH_MAX]; const uint32_t* wgmax[SP_BATCH_MAX]; const int32_t* n_dev[SP_BATCH_MAX]; int N[SP_BATCH_MAX]; int stride, n_glob, waves; };
__global__ __launch_bounds__(256) void sprin_fill_batch_kernel(FillBatch B)
{
    const int i = blockIdx.y;
    sprin_fill_body(B.out[i], B.N[i], B.stride, B.n_glob, B.wgmax[i], B.waves, B.n_dev[i]);
}

int64_t conv_params(const int32_t* hidden, int n_hidden, int rank, int n_in, int n_out)
{
    int64_t n = 0;
    int in = 6;
    for (int i = 0; i < n_hidden; ++i) { n += (int64_t)hidden[i] * in + 3 * hidden[i]; in = hidden[i]; }
    n += (int64_t)rank * in + rank;
    n += (int64_t)rank * n_in * n_out + 3 * n_out;
    return n;
}

}  // namespace

// (compact.h) the k-neighbour searches of up to 8 clouds in one launch, each with its own k, capacity and device point count
int cppf_internal_knn_batch(int n_items, const CppfKnnBatchItem* items, void* stream)
{
    if (n_items < 1 || n_items > SP_BATCH_MAX || !items) return CPPF_EINVAL;
    KnnBatch KB;
    memset(&KB, 0, sizeof(KB));
    int n_max = 0;
    for (int i = 0; i < n_items; ++i) {
        const CppfKnnBatchItem& it = items[i];
        if (!it.pc || !it.nbrs || !it.n_dev || it.n_cap < 1 || it.k < 1 || it.k > it.n_cap) return CPPF_EINVAL;
        KB.pc[i] = it.pc; KB.out[i] = it.nbrs; KB.n_dev[i] = it.n_dev; KB.N[i] = it.n_cap; KB.k[i] = it.k;
        n_max = it.n_cap > n_max ? it.n_cap : n_max;
    }
    knn_batch_kernel<<<dim3((n_max + KNN_WAVES - 1) / KNN_WAVES, n_items), KNN_WAVES * 64, 0, (hipStream_t)stream>>>(KB);
    return (int)hipGetLastError();
}

extern "C" {

int cppf_knn(const float* pc, const float* dist, int n_points, int k, int32_t* nbrs, void* stream)
{
    if (n_points < 0 || k <= 0) return CPPF_EINVAL;
    if (n_points == 0) return 0;
    if ((!pc && !dist) || !nbrs || k > n_points) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (n_points + KNN_WAVES - 1) / KNN_WAVES;
    if (dist) knn_kernel<true><<<blocks, KNN_WAVES * 64, 0, st>>>(pc, dist, n_points, k, nbrs, nullptr);
    else knn_kernel<false><<<blocks, KNN_WAVES * 64, 0, st>>>(pc, dist, n_points, k, nbrs, nullptr);
    return (int)hipGetLastError();
}

int cppf_knn_dyn(const float* pc, int n_cap, const int32_t* n_dev, int k, int32_t* nbrs, void* stream)
{
    if (n_cap < 1 || k <= 0 || !pc || !nbrs || !n_dev || k > n_cap) return CPPF_EINVAL;
    const int blocks = (n_cap + KNN_WAVES - 1) / KNN_WAVES;
    knn_kernel<false><<<blocks, KNN_WAVES * 64, 0, (hipStream_t)stream>>>(pc, nullptr, n_cap, k, nbrs, n_dev);
    return (int)hipGetLastError();
}

static bool sp_std_shape(const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob)
{
    return n_hidden == 4 && hidden[0] == 32 && hidden[1] == 64 && hidden[2] == 32 && hidden[3] == 32 && rank == SP_RANK &&
           n_out == SP_NOUT && n_glob >= 1 && n_glob <= 32 && n_nbr_feats == 2;
}
static size_t sp_natural_floats(const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob,
                                int num_layers)
{
    size_t n = 0;
    for (int l = 0; l < num_layers; ++l)
        n += (size_t)conv_params(hidden, n_hidden, rank, l == 0 ? n_nbr_feats : n_out + n_glob, n_out) +
             (size_t)n_glob * n_out + n_glob;
    return n;
}

size_t cppf_point_encoder_packed_floats(const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out,
                                        int n_glob, int num_layers)
{
    if (!hidden || n_hidden <= 0 || num_layers <= 0) return 0;
    size_t n = sp_natural_floats(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers);
    if (sp_std_shape(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob)) n += (size_t)num_layers * SPW_FLOATS;
    return n;
}

// natural parameters (host) -> device image (host buffer): the natural block verbatim, then one lane-ordered MFMA
// image of the kernel-MLP per layer
int cppf_point_encoder_pack(const float* natural, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                            int n_out, int n_glob, int num_layers, float* out)
{
    if (!natural || !hidden || !out || n_hidden <= 0 || num_layers <= 0) return CPPF_EINVAL;
    const size_t nat = sp_natural_floats(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers);
    memcpy(out, natural, nat * sizeof(float));
    if (!sp_std_shape(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob)) return 0;
    const float* p = natural;
    for (int l = 0; l < num_layers; ++l) {
        const int n_in = l == 0 ? n_nbr_feats : n_out + n_glob;
        float* img = out + nat + (size_t)l * SPW_FLOATS;
        for (int i = 0; i < SPW_FLOATS; ++i) img[i] = sp_image_elem(i, p);
        p += conv_params(hidden, n_hidden, rank, n_in, n_out) + (size_t)n_glob * n_out + n_glob;
    }
    return 0;
}

size_t cppf_point_encoder_workspace_bytes(int n_points, int n_out, int n_glob, int num_layers)
{
    // 256 B (spare) + (multi-layer only) one [N, n_out+n_glob] ping buffer + the conv workgroups' channel maxima
    size_t b = 256;
    if (num_layers > 1) b += ((size_t)n_points * (n_out + n_glob) * sizeof(float) + 255) / 256 * 256;
    b += (size_t)((n_points + 3) / 4) * 32 * sizeof(uint32_t);
    return b;
}

static int sp_forward(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k, const float* packed,
                      const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob, int num_layers,
                      float* out, float* mixed_out, void* workspace, size_t workspace_bytes, void* stream,
                      const int32_t* n_dev = nullptr);

int cppf_point_encoder_forward_dyn(const float* pc, const float* nrm, const int32_t* nbrs, int n_cap, const int32_t* n_dev, int k,
                                   const float* packed, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                                   int n_out, int n_glob, int num_layers, float* out, void* workspace,
                                   size_t workspace_bytes, void* stream)
{
    if (!n_dev) return CPPF_EINVAL;
    return sp_forward(pc, nrm, nbrs, n_cap, k, packed, hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers, out,
                      nullptr, workspace, workspace_bytes, stream, n_dev);
}

int cppf_point_encoder_forward(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k,
                               const float* packed, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                               int n_out, int n_glob, int num_layers, float* out, void* workspace,
                               size_t workspace_bytes, void* stream)
{
    return sp_forward(pc, nrm, nbrs, n_points, k, packed, hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers, out,
                      nullptr, workspace, workspace_bytes, stream);
}

int cppf_point_encoder_forward_train(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k,
                                     const float* packed, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                                     int n_out, int n_glob, int num_layers, float* out, float* contraction_out, void* workspace,
                                     size_t workspace_bytes, void* stream)
{
    if (num_layers != 1 || !contraction_out) return num_layers != 1 ? CPPF_EUNSUPPORTED : CPPF_EINVAL;
    return sp_forward(pc, nrm, nbrs, n_points, k, packed, hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers, out,
                      contraction_out, workspace, workspace_bytes, stream);
}

// kNN + convolution + GlobalInfoProp of up to 8 clouds in three launches (the one-layer standard encoder, train.py:34): what
// cppf_knn_dyn + cppf_point_encoder_forward_dyn do per cloud, bit for bit (the same device functions; a member only picks its
// arguments by blockIdx.y).
int cppf_point_encoder_forward_batch(int n_items, const CppfPointEncItem* items, int k, const int32_t* hidden, int n_hidden, int rank,
                                     int n_nbr_feats, int n_out, int n_glob, int num_layers, void* stream)
{
    if (n_items < 1 || n_items > SP_BATCH_MAX || !items || k <= 0 || !hidden) return CPPF_EINVAL;
    if (num_layers != 1 || !sp_std_shape(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob) || k > 64) return CPPF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int W = n_out + n_glob, waves = sp_waves(n_nbr_feats);
    const size_t nat = sp_natural_floats(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, 1);
    const int64_t cp = conv_params(hidden, n_hidden, rank, n_nbr_feats, n_out);
    KnnBatch KB;
    ConvBatch CB;
    FillBatch FB;
    memset(&KB, 0, sizeof(KB)); memset(&CB, 0, sizeof(CB)); memset(&FB, 0, sizeof(FB));
    FB.stride = W; FB.n_glob = n_glob; FB.waves = waves;
    int n_max = 0, any_search = 0;
    for (int i = 0; i < n_items; ++i) {
        const CppfPointEncItem& it = items[i];
        if (it.n_cap < 1 || k > it.n_cap) return CPPF_EINVAL;
        if (!it.pc || !it.nrm || !it.nbrs || !it.packed || !it.out) return CPPF_EINVAL;
        if (!it.workspace || it.workspace_bytes < cppf_point_encoder_workspace_bytes(it.n_cap, n_out, n_glob, 1)) return CPPF_EWORKSPACE;
        uint32_t* wgmax = (uint32_t*)((char*)it.workspace + 256);
        KB.pc[i] = it.pc; KB.out[i] = it.nbrs_ready ? nullptr : it.nbrs; KB.n_dev[i] = it.n_dev; KB.N[i] = it.n_cap; KB.k[i] = k;
        any_search |= !it.nbrs_ready;
        CB.item[i] = ConvArgs{it.pc, it.nrm, nullptr, it.nbrs, it.packed, it.packed + nat, it.out, it.n_cap, k, n_nbr_feats, W, nullptr,
                              it.n_dev, it.packed + cp, wgmax, n_glob};
        FB.out[i] = it.out; FB.wgmax[i] = wgmax; FB.n_dev[i] = it.n_dev; FB.N[i] = it.n_cap;
        n_max = it.n_cap > n_max ? it.n_cap : n_max;
    }
    if (any_search)
        knn_batch_kernel<<<dim3((n_max + KNN_WAVES - 1) / KNN_WAVES, n_items), KNN_WAVES * 64, 0, st>>>(KB);
    const size_t lds = ((size_t)SPW_FLOATS + (size_t)waves * sp_per_wave(n_nbr_feats)) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)sprin_conv_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    sprin_conv_batch_kernel<<<dim3((n_max + waves - 1) / waves, n_items), waves * 64, lds, st>>>(CB);
    sprin_fill_batch_kernel<<<dim3((n_max * n_glob + 255) / 256, n_items), 256, 0, st>>>(FB);
    return (int)hipGetLastError();
}

static int sp_forward(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k, const float* packed,
                      const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob, int num_layers,
                      float* out, float* mixed_out, void* workspace, size_t workspace_bytes, void* stream,
                      const int32_t* n_dev)
{
    if (n_points < 0 || k <= 0 || num_layers <= 0 || !hidden) return CPPF_EINVAL;
    if (n_points == 0) return 0;
    if (!pc || !nrm || !nbrs || !packed || !out) return CPPF_EINVAL;
    if (!sp_std_shape(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob) || k > 64 || k > n_points) return CPPF_EUNSUPPORTED;
    if (!workspace || workspace_bytes < cppf_point_encoder_workspace_bytes(n_points, n_out, n_glob, num_layers))
        return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int W = n_out + n_glob;
    float* ping = (float*)((char*)workspace + 256);
    uint32_t* wgmax = (uint32_t*)((char*)workspace + 256 +
                                  (num_layers > 1 ? ((size_t)n_points * W * sizeof(float) + 255) / 256 * 256 : 0));
    const float* p = packed;
    const float* images = packed + sp_natural_floats(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers);
    // layer l writes `dst`; the last layer must land in `out`
    for (int l = 0; l < num_layers; ++l) {
        const int n_in = l == 0 ? n_nbr_feats : W;
        float* dst = ((num_layers - 1 - l) & 1) ? ping : out;
        const float* src = l == 0 ? nullptr : (dst == out ? ping : out);
        const float* glob_w = p + conv_params(hidden, n_hidden, rank, n_in, n_out);
        ConvArgs A{pc, nrm, src, nbrs, p, images + (size_t)l * SPW_FLOATS, dst, n_points, k, n_in, W, l == 0 ? mixed_out : nullptr, n_dev,
                   glob_w, wgmax, n_glob};
        const int waves = sp_waves(n_in);
        const size_t lds = ((size_t)SPW_FLOATS + (size_t)waves * sp_per_wave(n_in)) * sizeof(float);
        hipError_t e = hipFuncSetAttribute((const void*)sprin_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        sprin_conv_kernel<<<(n_points + waves - 1) / waves, waves * 64, lds, st>>>(A);
        p += conv_params(hidden, n_hidden, rank, n_in, n_out);
        sprin_fill_kernel<<<(n_points * n_glob + 255) / 256, 256, 0, st>>>(dst, n_points, W, n_glob, wgmax, waves, n_dev);
        p += (size_t)n_glob * n_out + n_glob;
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // extern "C"
