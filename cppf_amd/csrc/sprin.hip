// SPRIN point encoder for gfx950 (C ABI in include/cppf.h).
//
// Replaces models/model.py:36-78 (PointEncoder) + models/sprin.py:40-107 of the reference:
//   knn_kernel           torch.topk(dist, k, largest=False)            models/model.py:47
//   sprin_conv_kernel    gather, rifeat, conv_kernel MLP, rank contraction, outnet, LayerNorm
//                                                                      models/model.py:48-57, models/sprin.py:40-107
//   sprin_glob_kernel /  GlobalInfoProp: linear, max over points, concat models/sprin.py:75-84
//   sprin_fill_kernel
//
// Arithmetic follows oracle/sprin_oracle.c to the bit (-ffp-contract=off, bias-seeded fmaf chains over
// ascending input index, sequential sums over ascending neighbour index, correctly rounded sqrt/divide),
// which is what lets the parity tests demand exact equality.  The work is ~0.4 MMAC per point in
// 6->32->64->32->32->32 per-neighbour MLPs with a LayerNorm after every hidden layer; one wavefront owns
// one point, one lane one neighbour, so every LayerNorm is lane-local and the weights (27 KB) arrive
// through the scalar cache as SGPR operands of v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cppf.h"
#include "cppf_math.h"

using namespace cppf;

namespace {

constexpr int KNN_WAVES = 4;        // queries per workgroup
constexpr int KNN_LDS_MAX_N = 8192; // keys staged in LDS up to this N (4 x 32 KB + histograms)

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint32_t knn_key(const float* __restrict__ pc, const float* __restrict__ dist, int N, int q,
                                            float qx, float qy, float qz, int j)
{
    float d;
    if (dist) d = dist[(size_t)q * N + j];
    else {
        const float dx = pc[3 * j] - qx, dy = pc[3 * j + 1] - qy, dz = pc[3 * j + 2] - qz;
        d = (dx * dx + dy * dy) + dz * dz;
    }
    return f2ord(d);
}

// One wavefront per query point.  Four 8-bit radix passes find the exact key of the k-th smallest
// entry (per-wave 256-bin LDS histogram, wave scan), then one ordered pass emits the indices.
template <bool KEYS_IN_LDS>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_kernel(const float* __restrict__ pc, const float* __restrict__ dist,
                                                             int N, int k, int32_t* __restrict__ out)
{
    extern __shared__ uint32_t knn_lds[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    uint32_t* hist = knn_lds + w * 256;
    uint32_t* keys = knn_lds + KNN_WAVES * 256 + (size_t)w * (KEYS_IN_LDS ? N : 0);
    const int q = blockIdx.x * KNN_WAVES + w;
    const bool live = q < N;
    const int qc = live ? q : N - 1;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (!dist) { qx = pc[3 * qc]; qy = pc[3 * qc + 1]; qz = pc[3 * qc + 2]; }
    if (KEYS_IN_LDS)
        for (int j = lane; j < N; j += 64) keys[j] = knn_key(pc, dist, N, qc, qx, qy, qz, j);
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int b = lane; b < 256; b += 64) hist[b] = 0;
        __syncthreads();
        for (int j = lane; j < N; j += 64) {
            const uint32_t key = KEYS_IN_LDS ? keys[j] : knn_key(pc, dist, N, qc, qx, qy, qz, j);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // lane owns bins 4*lane .. 4*lane+3
        const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const int lsum = (int)(c0 + c1 + c2 + c3);
        int incl = lsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        const int excl = incl - lsum;
        const bool hit = excl < remaining && remaining <= incl;
        int digit = 0, below = 0;
        if (hit) {
            int need = remaining - excl;  // 1-based rank inside this lane's 4 bins
            int b = 0;
            int acc = 0;
            if (need > (int)c0) { acc += c0; b = 1; if (need > (int)(c0 + c1)) { acc += c1; b = 2; if (need > (int)(c0 + c1 + c2)) { acc += c2; b = 3; } } }
            digit = 4 * lane + b;
            below = excl + acc;
        }
        const unsigned long long hm = __ballot(hit);
        const int src = __ffsll((long long)hm) - 1;  // exactly one lane hits (k <= N)
        digit = __shfl(digit, src);
        below = __shfl(below, src);
        remaining -= below;
        prefix |= (uint32_t)digit << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    // prefix = key of the k-th smallest; take every key below it and the first `remaining` equal to it
    int base = 0, eq_seen = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j < N;
        const uint32_t key = in ? (KEYS_IN_LDS ? keys[j] : knn_key(pc, dist, N, qc, qx, qy, qz, j)) : 0xffffffffu;
        const bool eq = in && key == prefix;
        const unsigned long long eqm = __ballot(eq);
        const bool take = in && (key < prefix || (eq && eq_seen + __popcll(eqm & lt_mask) < remaining));
        const unsigned long long tm = __ballot(take);
        if (take && live) out[(size_t)q * k + base + __popcll(tm & lt_mask)] = j;
        base += __popcll(tm);
        eq_seen += __popcll(eqm);
    }
}

// --------------------------------------------------------------------------------------------- conv
constexpr int SP_WAVES = 4;   // points per workgroup
constexpr int SP_RANK = 32, SP_NOUT = 32;
constexpr int SP_KSTRIDE = SP_RANK + 1;  // kern[j][r] row stride in LDS (odd: conflict-free column walks)

template <int IN, int OUT>
__device__ __forceinline__ void sp_linear(const float* __restrict__ W, const float* __restrict__ b, const float (&x)[IN],
                                          float (&y)[OUT])
{
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        float acc = b[o];
#pragma unroll
        for (int k = 0; k < IN; ++k) acc = fmaf(W[o * IN + k], x[k], acc);
        y[o] = acc;
    }
}
// nn.LayerNorm (eps 1e-5, affine) + ReLU, lane-local (oracle/sprin_oracle.c:layer_norm)
template <int H>
__device__ __forceinline__ void sp_ln_relu(const float (&y)[H], const float* __restrict__ g, const float* __restrict__ b,
                                           float (&x)[H])
{
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < H; ++o) s = s + y[o];
    const float mean = s / (float)H;
    float v = 0.f;
#pragma unroll
    for (int o = 0; o < H; ++o) { const float d = y[o] - mean; v = v + d * d; }
    const float inv = 1.0f / sqrtf(v / (float)H + 1e-5f);
#pragma unroll
    for (int o = 0; o < H; ++o) {
        const float z = ((y[o] - mean) * inv) * g[o] + b[o];
        x[o] = z > 0.f ? z : 0.f;
    }
}
template <int IN, int H>
__device__ __forceinline__ const float* sp_hidden(const float* __restrict__ p, const float (&x)[IN], float (&xo)[H])
{
    float y[H];
    sp_linear<IN, H>(p, p + H * IN, x, y);
    p += H * IN + H;
    sp_ln_relu<H>(y, p, p + H, xo);
    return p + 2 * H;
}

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

struct ConvArgs {
    const float* pc;
    const float* nrm;
    const float* feat_in;   // null for the first layer, else [N][n_in]
    const int32_t* nbrs;    // [N][k]
    const float* params;    // this layer's packed parameters
    float* out;             // [N][out_stride], columns 0..31 written
    int N, k, n_in, out_stride;
};

// hidden = {32, 64, 32, 32}, rank 32, n_out 32 (train.py:34).  Dynamic LDS per wave:
//   kern[64][33] | nf[64][n_in] | contracted[32*n_in] | r[64][3] | y[32]
__global__ __launch_bounds__(SP_WAVES * 64) void sprin_conv_kernel(ConvArgs A)
{
    extern __shared__ float sp_lds[];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int n_in = A.n_in, k = A.k;
    const int per_wave = 64 * SP_KSTRIDE + 64 * n_in + SP_RANK * n_in + 64 * 3 + SP_NOUT;
    float* kern = sp_lds + (size_t)w * per_wave;
    float* nf = kern + 64 * SP_KSTRIDE;
    float* contracted = nf + 64 * n_in;
    float* rr = contracted + SP_RANK * n_in;
    float* yv = rr + 64 * 3;
    const int n = blockIdx.x * SP_WAVES + w;
    const bool live = n < A.N;
    const int nc = live ? n : A.N - 1;
    const int jc = lane < k ? lane : k - 1;
    const int nb = A.nbrs[(size_t)nc * k + jc];
    const float rx = A.pc[3 * nb], ry = A.pc[3 * nb + 1], rz = A.pc[3 * nb + 2];
    const float sx = A.pc[3 * nc], sy = A.pc[3 * nc + 1], sz = A.pc[3 * nc + 2];
    rr[3 * lane] = rx; rr[3 * lane + 1] = ry; rr[3 * lane + 2] = rz;
    __syncthreads();
    // r_mean: sequential over neighbours (every lane redundantly; LDS broadcast reads)
    float mx = 0.f, my = 0.f, mz = 0.f;
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
    // conv_kernel(6, 32, 32, 64, 32, 32) (models/sprin.py:64-72): lane-local, weights via scalar loads
    const float* p = A.params;
    {
        float a[32], b[64], c[32], d[32], kr[SP_RANK];
        p = sp_hidden<6, 32>(p, x6, a);
        p = sp_hidden<32, 64>(p, a, b);
        p = sp_hidden<64, 32>(p, b, c);
        p = sp_hidden<32, 32>(p, c, d);
        sp_linear<32, SP_RANK>(p, p + SP_RANK * 32, d, kr);
        p += SP_RANK * 32 + SP_RANK;
#pragma unroll
        for (int r = 0; r < SP_RANK; ++r) kern[lane * SP_KSTRIDE + r] = kr[r];
    }
    __syncthreads();
    // einsum("bnkr,bnki->bnri") (models/sprin.py:99): contracted[r*n_in + i], sequential over neighbours
    const int C = SP_RANK * n_in;
    for (int t = lane; t < C; t += 64) {
        const int r = t / n_in, i = t - r * n_in;
        float acc = 0.f;
        for (int j = 0; j < k; ++j) acc = fmaf(kern[j * SP_KSTRIDE + r], nf[j * n_in + i], acc);
        contracted[t] = acc;
    }
    __syncthreads();
    // outnet (transposed weights: lane o reads Wo_t[c][o], coalesced) + LayerNorm (models/sprin.py:100,105)
    const float* Wo = p;
    const float* bo = Wo + (size_t)C * SP_NOUT;
    const int o = lane & (SP_NOUT - 1);
    float acc = bo[o];
    for (int c = 0; c < C; ++c) acc = fmaf(Wo[(size_t)c * SP_NOUT + o], contracted[c], acc);
    if (lane < SP_NOUT) yv[lane] = acc;
    __syncthreads();
    float s = 0.f;
    for (int q = 0; q < SP_NOUT; ++q) s = s + yv[q];
    const float mean = s / (float)SP_NOUT;
    float v = 0.f;
    for (int q = 0; q < SP_NOUT; ++q) { const float dd = yv[q] - mean; v = v + dd * dd; }
    const float inv = 1.0f / sqrtf(v / (float)SP_NOUT + 1e-5f);
    const float z = ((acc - mean) * inv) * bo[SP_NOUT + o] + bo[2 * SP_NOUT + o];
    if (live && lane < SP_NOUT) A.out[(size_t)n * A.out_stride + lane] = z;
}

// GlobalInfoProp (models/sprin.py:75-84): one thread per point computes linear(n_out -> n_glob), the
// per-channel maximum goes to glob[] as an order-preserving uint (max is exact in any order).
__global__ __launch_bounds__(256) void sprin_glob_kernel(const float* __restrict__ feat, int N, int stride, int n_glob,
                                                         const float* __restrict__ Wa, uint32_t* __restrict__ glob)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int nc = n < N ? n : N - 1;
    float x[SP_NOUT];
#pragma unroll
    for (int o = 0; o < SP_NOUT; ++o) x[o] = feat[(size_t)nc * stride + o];
    const float* ba = Wa + n_glob * SP_NOUT;
    for (int g = 0; g < n_glob; ++g) {
        float acc = ba[g];
#pragma unroll
        for (int o = 0; o < SP_NOUT; ++o) acc = fmaf(Wa[g * SP_NOUT + o], x[o], acc);
        uint32_t key = f2ord(acc);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)key, d);
            key = other > key ? other : key;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&glob[g], key);
    }
}
__global__ __launch_bounds__(256) void sprin_fill_kernel(float* __restrict__ out, int N, int stride, int n_glob,
                                                         const uint32_t* __restrict__ glob)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * n_glob) return;
    const int n = t / n_glob, g = t - n * n_glob;
    out[(size_t)n * stride + SP_NOUT + g] = ord2f(glob[g]);
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

extern "C" {

int cppf_knn(const float* pc, const float* dist, int n_points, int k, int32_t* nbrs, void* stream)
{
    if (n_points < 0 || k <= 0) return CPPF_EINVAL;
    if (n_points == 0) return 0;
    if ((!pc && !dist) || !nbrs || k > n_points) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (n_points + KNN_WAVES - 1) / KNN_WAVES;
    const bool in_lds = n_points <= KNN_LDS_MAX_N;
    const size_t lds = (size_t)KNN_WAVES * (256 + (in_lds ? n_points : 0)) * sizeof(uint32_t);
    if (in_lds) {
        hipError_t e = hipFuncSetAttribute((const void*)knn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        knn_kernel<true><<<blocks, KNN_WAVES * 64, lds, st>>>(pc, dist, n_points, k, nbrs);
    } else {
        knn_kernel<false><<<blocks, KNN_WAVES * 64, lds, st>>>(pc, dist, n_points, k, nbrs);
    }
    return (int)hipGetLastError();
}

size_t cppf_point_encoder_packed_floats(const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out,
                                        int n_glob, int num_layers)
{
    if (!hidden || n_hidden <= 0 || num_layers <= 0) return 0;
    size_t n = 0;
    for (int l = 0; l < num_layers; ++l)
        n += (size_t)conv_params(hidden, n_hidden, rank, l == 0 ? n_nbr_feats : n_out + n_glob, n_out) +
             (size_t)n_glob * n_out + n_glob;
    return n;
}

size_t cppf_point_encoder_workspace_bytes(int n_points, int n_out, int n_glob, int num_layers)
{
    // 256 B of per-channel maxima + (multi-layer only) one [N, n_out+n_glob] ping buffer
    size_t b = 256;
    if (num_layers > 1) b += (size_t)n_points * (n_out + n_glob) * sizeof(float);
    return b;
}

int cppf_point_encoder_forward(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k,
                               const float* packed, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                               int n_out, int n_glob, int num_layers, float* out, void* workspace,
                               size_t workspace_bytes, void* stream)
{
    if (n_points < 0 || k <= 0 || num_layers <= 0 || !hidden) return CPPF_EINVAL;
    if (n_points == 0) return 0;
    if (!pc || !nrm || !nbrs || !packed || !out) return CPPF_EINVAL;
    if (n_hidden != 4 || hidden[0] != 32 || hidden[1] != 64 || hidden[2] != 32 || hidden[3] != 32 || rank != SP_RANK ||
        n_out != SP_NOUT || n_glob < 1 || n_glob > 32 || n_nbr_feats != 2 || k > 64 || k > n_points)
        return CPPF_EUNSUPPORTED;
    if (!workspace || workspace_bytes < cppf_point_encoder_workspace_bytes(n_points, n_out, n_glob, num_layers))
        return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int W = n_out + n_glob;
    uint32_t* glob = (uint32_t*)workspace;
    float* ping = (float*)((char*)workspace + 256);
    const float* p = packed;
    // layer l writes `dst`; the last layer must land in `out`
    for (int l = 0; l < num_layers; ++l) {
        const int n_in = l == 0 ? n_nbr_feats : W;
        float* dst = ((num_layers - 1 - l) & 1) ? ping : out;
        const float* src = l == 0 ? nullptr : (dst == out ? ping : out);
        ConvArgs A{pc, nrm, src, nbrs, p, dst, n_points, k, n_in, W};
        const size_t lds = (size_t)SP_WAVES * (64 * SP_KSTRIDE + 64 * n_in + SP_RANK * n_in + 64 * 3 + SP_NOUT) * sizeof(float);
        hipError_t e = hipFuncSetAttribute((const void*)sprin_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipMemsetAsync(glob, 0, 256, st);
        if (e != hipSuccess) return (int)e;
        sprin_conv_kernel<<<(n_points + SP_WAVES - 1) / SP_WAVES, SP_WAVES * 64, lds, st>>>(A);
        p += conv_params(hidden, n_hidden, rank, n_in, n_out);
        sprin_glob_kernel<<<(n_points + 255) / 256, 256, 0, st>>>(dst, n_points, W, n_glob, p, glob);
        sprin_fill_kernel<<<(n_points * n_glob + 255) / 256, 256, 0, st>>>(dst, n_points, W, n_glob, glob);
        p += (size_t)n_glob * n_out + n_glob;
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // extern "C"
