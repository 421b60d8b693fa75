// Backward of PPFEncoder.forward_with_idx for gfx950 (C ABI in include/cppf.h).
//
// The reference has no backward code: train.py:91 calls loss.backward() and autograd differentiates
// models/model.py:117-137.  This kernel computes the same gradients -- every parameter of the three ResLayers
// and the final linear, and d/d(feat) (scatter-add over the pair indices) -- for the standard shape
// ppffcs = [84, 32, 32, 16] (train.py:35) and any out_dim, in one pass that recomputes the forward.
//
// One wavefront owns tiles of 64 consecutive pairs, one pair per lane:
//   1. forward and backward-data are lane-local (activations and deltas in registers, weights as SGPR operands
//      of v_fmac_f32 through the scalar cache);
//   2. each weight gradient is an outer-product sum over the tile's 64 pairs: delta and input are staged in
//      LDS as [pair][feature], lanes own input columns and walk the pairs in ascending order;
//   3. the tile's sums are added to the wavefront's own partial-gradient slice in the workspace (no atomics);
//      a second kernel adds the slices in a fixed two-level order (groups of 32, ascending).
// The summation order is therefore fixed and is restated in oracle/backward_oracle.c: parameter gradients
// are bit-identical to the oracle.  d/d(feat) is a scatter-add over the pair indices: the per-pair rows d(x0)[0:80]
// go to the workspace, the 2P (point, entry) keys are radix-sorted (stable: a-entries in pair order, then
// b-entries), and one wavefront per point adds its rows in that order -- no atomics, deterministic, also bit-exact.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "../../include/cppf.h"
#include "cppf_math.h"

using namespace cppf;

namespace {

constexpr int BW_F = 40, BW_D0 = 84, BW_D1 = 32, BW_D2 = 32, BW_D3 = 16;
constexpr int BW_DSTR = 36;  // delta rows in LDS (16-byte aligned rows: broadcast ds_read_b128)
constexpr int BW_XSTR = 43;  // input rows in LDS (odd stride: conflict-free row writes); x0 is staged in two halves of 42

struct BwdArgs {
    const float* pc;
    const float* nrm;
    const float* feat;
    const void* idxs;
    const float* params;
    const float* grad_out;
    float* parts;      // [n_parts][n_params]
    float* dx;         // [P][2F] per-pair d(x0) feature columns (workspace)
    int64_t P;
    int64_t n_params;
    int64_t offs[20];  // 6 per res layer {fc1.w, fc1.b, fc2.w, fc2.b, fc0.w | -1, fc0.b | -1}, final.w, final.b
    int out_dim, idx64, n_parts;
};

// y = b + W x  (W[o][k] torch layout).  Inputs are register-resident (static k), the output loop is ROLLED
// four rows at a time -- a fully unrolled 84x32 block makes the scheduler hoist every scalar load and spill
// thousands of registers -- and the results pass through the lane's LDS row T to reach static registers.
template <int K, int NN>
__device__ __forceinline__ void fwd_lin(const float* __restrict__ W, const float* __restrict__ b, const float (&x)[K],
                                        float* __restrict__ T, float (&y)[NN])
{
#pragma unroll 1
    for (int o = 0; o < NN; o += 4) {
        const float* __restrict__ w = W + o * K;
        float a0 = b[o], a1 = b[o + 1], a2 = b[o + 2], a3 = b[o + 3];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            a0 = fmaf(w[k], x[k], a0);
            a1 = fmaf(w[K + k], x[k], a1);
            a2 = fmaf(w[2 * K + k], x[k], a2);
            a3 = fmaf(w[3 * K + k], x[k], a3);
        }
        T[o] = a0; T[o + 1] = a1; T[o + 2] = a2; T[o + 3] = a3;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int o = 0; o < NN; ++o) y[o] = T[o];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// dx[i] = chain over o ascending of fmaf(W[o][i], d[o], .) from 0, i < KO (row stride K).  The deltas come from
// the lane's LDS row D (dynamic o), the KO accumulators are static registers, the o loop is rolled.
template <int K, int KO, int NN>
__device__ __forceinline__ void bwd_lin(const float* __restrict__ W, const float* __restrict__ D, float (&dx)[KO])
{
#pragma unroll
    for (int i = 0; i < KO; ++i) dx[i] = 0.f;
#pragma unroll 2
    for (int o = 0; o < NN; ++o) {
        const float dv = D[o];
        const float* __restrict__ w = W + o * K;
#pragma unroll
        for (int i = 0; i < KO; ++i) dx[i] = fmaf(w[i], dv, dx[i]);
    }
}
template <int NN>
__device__ __forceinline__ void relu(float (&h)[NN])
{
#pragma unroll
    for (int o = 0; o < NN; ++o) h[o] = h[o] > 0.f ? h[o] : 0.f;
}
template <int NN>
__device__ __forceinline__ void mask(float (&d)[NN], const float (&h)[NN])
{
#pragma unroll
    for (int o = 0; o < NN; ++o) d[o] = h[o] > 0.f ? d[o] : 0.f;
}
template <int NN>
__device__ __forceinline__ void stage(float* __restrict__ L, int stride, int lane, const float (&v)[NN])
{
#pragma unroll
    for (int c = 0; c < NN; ++c) L[lane * stride + c] = v[c];
}
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// part[offW + o*I + i] += sum_j delta[j][o] * x[j][i];  part[offB + o] += sum_j delta[j][o]   (j ascending)
// I columns of a weight whose rows are ISTR long, starting at column col0 (XL holds just those columns); BIAS: also
// the bias gradient
template <int O, int I, int ISTR = I, bool BIAS = true>
__device__ __forceinline__ void outer(const float* __restrict__ DL, const float* __restrict__ XL, float* __restrict__ part,
                                      int64_t offW, int64_t offB, int lane, int col0 = 0)
{
    lds_fence();
    constexpr int PASSES = (I + 63) / 64;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int i = lane + 64 * ps;
        const bool act = i < I;
        const int ic = act ? i : I - 1;
        float acc[O];
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = 0.f;
#pragma unroll 2
        for (int j = 0; j < 64; ++j) {
            const float xv = XL[j * BW_XSTR + ic];
#pragma unroll
            for (int o4 = 0; o4 < O; o4 += 4) {
                const float4 d = *reinterpret_cast<const float4*>(DL + j * BW_DSTR + o4);
                acc[o4] = fmaf(d.x, xv, acc[o4]);
                acc[o4 + 1] = fmaf(d.y, xv, acc[o4 + 1]);
                acc[o4 + 2] = fmaf(d.z, xv, acc[o4 + 2]);
                acc[o4 + 3] = fmaf(d.w, xv, acc[o4 + 3]);
            }
        }
        if (act) {
#pragma unroll
            for (int o = 0; o < O; ++o) part[offW + (int64_t)o * ISTR + col0 + i] = part[offW + (int64_t)o * ISTR + col0 + i] + acc[o];
        }
    }
    if (BIAS && lane < O) {
        float accb = 0.f;
#pragma unroll 4
        for (int j = 0; j < 64; ++j) accb = accb + DL[j * BW_DSTR + lane];
        part[offB + lane] = part[offB + lane] + accb;
    }
    lds_fence();
}

__device__ __forceinline__ void ppf4(const float* __restrict__ pc, const float* __restrict__ nrm, int a, int b, float* out)
{   // models/model.py:118-129 (fp32 `+ 1e-7`, divisions)
    const f3 pa = ld3(pc, a), pb = ld3(pc, b), na = ld3(nrm, a), nb = ld3(nrm, b);
    const f3 xy = sub3(pa, pb);
    const float d = sqrtf((xy.x * xy.x + xy.y * xy.y) + xy.z * xy.z);
    const float den = d + 1e-7f;
    const f3 u = {xy.x / den, xy.y / den, xy.z / den};
    out[0] = (na.x * u.x + na.y * u.y) + na.z * u.z;
    out[1] = (nb.x * u.x + nb.y * u.y) + nb.z * u.z;
    out[2] = (na.x * nb.x + na.y * nb.y) + na.z * nb.z;
    out[3] = d;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_mlp_bwd_kernel(BwdArgs A)
{
    __shared__ __attribute__((aligned(16))) float DL[64 * BW_DSTR];
    __shared__ float XL[64 * BW_XSTR];
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    float* part = A.parts + (size_t)w * A.n_params;
    const float* Pm = A.params;
    const int64_t n_tiles = (A.P + 63) / 64;
    const int OD = A.out_dim;
    for (int64_t t = w; t < n_tiles; t += A.n_parts) {
        const int64_t p = t * 64 + lane;
        const bool live = p < A.P;
        const int64_t pcl = live ? p : A.P - 1;
        int ia, ib;
        if (A.idx64) {
            const longlong2 v = reinterpret_cast<const longlong2*>(A.idxs)[pcl];
            ia = (int)v.x; ib = (int)v.y;
        } else {
            const int2 v = reinterpret_cast<const int2*>(A.idxs)[pcl];
            ia = v.x; ib = v.y;
        }
        // ---- forward (natural order); x0 is gathered again when it is staged for the layer-0 outer products
        float h0[BW_D1], x1[BW_D1], h1[BW_D2], x2[BW_D2], h2[BW_D3], x3[BW_D3], ppf[4];
        ppf4(A.pc, A.nrm, ia, ib, ppf);
        float* const T = DL + lane * BW_DSTR;     // the lane's own LDS row: matvec outputs, then its deltas
        {
            float x0[BW_D0];
#pragma unroll
            for (int c = 0; c < BW_F; ++c) { x0[c] = A.feat[(size_t)ia * BW_F + c]; x0[BW_F + c] = A.feat[(size_t)ib * BW_F + c]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) x0[2 * BW_F + c] = ppf[c];
            float y0[BW_D1];
            fwd_lin<BW_D0, BW_D1>(Pm + A.offs[0], Pm + A.offs[1], x0, T, h0); relu(h0);
            fwd_lin<BW_D0, BW_D1>(Pm + A.offs[4], Pm + A.offs[5], x0, T, y0);
            fwd_lin<BW_D1, BW_D1>(Pm + A.offs[2], Pm + A.offs[3], h0, T, x1);
#pragma unroll
            for (int q = 0; q < BW_D1; ++q) x1[q] = x1[q] + y0[q];
        }
        fwd_lin<BW_D1, BW_D2>(Pm + A.offs[6], Pm + A.offs[7], x1, T, h1); relu(h1);
        fwd_lin<BW_D2, BW_D2>(Pm + A.offs[8], Pm + A.offs[9], h1, T, x2);
#pragma unroll
        for (int q = 0; q < BW_D2; ++q) x2[q] = x2[q] + x1[q];
        {
            float y0[BW_D3];
            fwd_lin<BW_D2, BW_D3>(Pm + A.offs[12], Pm + A.offs[13], x2, T, h2); relu(h2);
            fwd_lin<BW_D2, BW_D3>(Pm + A.offs[16], Pm + A.offs[17], x2, T, y0);
            fwd_lin<BW_D3, BW_D3>(Pm + A.offs[14], Pm + A.offs[15], h2, T, x3);
#pragma unroll
            for (int q = 0; q < BW_D3; ++q) x3[q] = x3[q] + y0[q];
        }
        // ---- final linear: dy3 = Wf^T g;  dWf, dbf
        const float* g = A.grad_out + (size_t)pcl * OD;
        float dy3[BW_D3];
#pragma unroll
        for (int i = 0; i < BW_D3; ++i) dy3[i] = 0.f;
        {
            const float* Wf = Pm + A.offs[18];
            int o = 0;
            for (; o + 4 <= OD; o += 4) {
                float go[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) go[u] = live ? g[o + u] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int i = 0; i < BW_D3; ++i) dy3[i] = fmaf(Wf[(o + u) * BW_D3 + i], go[u], dy3[i]);
                }
            }
            for (; o < OD; ++o) {
                const float go = live ? g[o] : 0.f;
#pragma unroll
                for (int i = 0; i < BW_D3; ++i) dy3[i] = fmaf(Wf[o * BW_D3 + i], go, dy3[i]);
            }
        }
        stage(XL, BW_XSTR, lane, x3);
        lds_fence();
        for (int o = lane; o < (OD + 63) / 64 * 64; o += 64) {   // lanes own outputs here
            const bool act = o < OD;
            float acc[BW_D3], accb = 0.f;
#pragma unroll
            for (int i = 0; i < BW_D3; ++i) acc[i] = 0.f;
            for (int j0 = 0; j0 < 64; j0 += 8) {   // eight independent loads in flight per step
                float gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t pj = t * 64 + j0 + u;
                    gv[u] = (act && pj < A.P) ? A.grad_out[(size_t)pj * OD + o] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int i = 0; i < BW_D3; ++i) acc[i] = fmaf(gv[u], XL[(j0 + u) * BW_XSTR + i], acc[i]);
                    accb = accb + gv[u];
                }
            }
            if (act) {
#pragma unroll
                for (int i = 0; i < BW_D3; ++i) {
                    const int64_t q = A.offs[18] + (int64_t)o * BW_D3 + i;
                    part[q] = part[q] + acc[i];
                }
                part[A.offs[19] + o] = part[A.offs[19] + o] + accb;
            }
        }
        lds_fence();
        // ---- res layer 2 (32 -> 16, fc0): deltas dy3 (wrt x3), dh2
        float dh2[BW_D3], dy2[BW_D2];
        stage(DL, BW_DSTR, lane, dy3); stage(XL, BW_XSTR, lane, h2);
        lds_fence();
        bwd_lin<BW_D3, BW_D3, BW_D3>(Pm + A.offs[14], T, dh2); mask(dh2, h2);
        bwd_lin<BW_D2, BW_D2, BW_D3>(Pm + A.offs[16], T, dy2);                      // fc0 path of d(x2)
        outer<BW_D3, BW_D3>(DL, XL, part, A.offs[14], A.offs[15], lane);            // fc2: dy3 x h2
        stage(XL, BW_XSTR, lane, x2);
        outer<BW_D3, BW_D2>(DL, XL, part, A.offs[16], A.offs[17], lane);            // fc0: dy3 x x2
        stage(DL, BW_DSTR, lane, dh2);
        lds_fence();
        {
            float t1[BW_D2];
            bwd_lin<BW_D2, BW_D2, BW_D3>(Pm + A.offs[12], T, t1);
#pragma unroll
            for (int i = 0; i < BW_D2; ++i) dy2[i] = t1[i] + dy2[i];
        }
        outer<BW_D3, BW_D2>(DL, XL, part, A.offs[12], A.offs[13], lane);            // fc1: dh2 x x2
        // ---- res layer 1 (32 -> 32, identity skip): deltas dy2 (wrt x2), dh1
        float dh1[BW_D2], dy1[BW_D1];
        stage(DL, BW_DSTR, lane, dy2); stage(XL, BW_XSTR, lane, h1);
        lds_fence();
        bwd_lin<BW_D2, BW_D2, BW_D2>(Pm + A.offs[8], T, dh1); mask(dh1, h1);
        outer<BW_D2, BW_D2>(DL, XL, part, A.offs[8], A.offs[9], lane);              // fc2: dy2 x h1
        stage(DL, BW_DSTR, lane, dh1); stage(XL, BW_XSTR, lane, x1);
        lds_fence();
        bwd_lin<BW_D1, BW_D1, BW_D2>(Pm + A.offs[6], T, dy1);
#pragma unroll
        for (int i = 0; i < BW_D1; ++i) dy1[i] = dy1[i] + dy2[i];
        outer<BW_D2, BW_D1>(DL, XL, part, A.offs[6], A.offs[7], lane);              // fc1: dh1 x x1
        // ---- res layer 0 (84 -> 32, fc0): deltas dy1 (wrt x1), dh0
        float dh0[BW_D1];
        stage(DL, BW_DSTR, lane, dy1); stage(XL, BW_XSTR, lane, h0);
        lds_fence();
        bwd_lin<BW_D1, BW_D1, BW_D1>(Pm + A.offs[2], T, dh0); mask(dh0, h0);
        outer<BW_D1, BW_D1>(DL, XL, part, A.offs[2], A.offs[3], lane);              // fc2: dy1 x h0
        // d/d(feat) = the 80 feature columns of d(x0) = W1^T dh0 + W0^T dy1: the W0 term now (T holds dy1), 20 columns at
        // a time, parked in the pair's workspace row; the W1 term is added to it once T holds dh0
        float* const dxrow = A.dx + (size_t)pcl * (2 * BW_F);
        for (int cc = 0; cc < 4; ++cc) {
            float t2[20];
            bwd_lin<BW_D0, 20, BW_D1>(Pm + A.offs[4] + 20 * cc, T, t2);
            if (live) {
#pragma unroll
                for (int c = 0; c < 20; ++c) dxrow[20 * cc + c] = t2[c];
            }
        }
        // x0 = [feat[a] (40), feat[b] (40), ppf (4)] is staged 42 columns at a time
        auto stage_x0 = [&](int half) {
            lds_fence();
            for (int c = 0; c < 42; ++c) {
                const int col = 42 * half + c;
                XL[lane * BW_XSTR + c] = col < BW_F ? A.feat[(size_t)ia * BW_F + col]
                                         : (col < 2 * BW_F ? A.feat[(size_t)ib * BW_F + col - BW_F]
                                                           : (col == 80 ? ppf[0] : (col == 81 ? ppf[1] : (col == 82 ? ppf[2] : ppf[3]))));
            }
        };
        stage_x0(0);
        outer<BW_D1, 42, BW_D0, true>(DL, XL, part, A.offs[4], A.offs[5], lane, 0);   // fc0: dy1 x x0[0:42], bias
        stage_x0(1);
        outer<BW_D1, 42, BW_D0, false>(DL, XL, part, A.offs[4], A.offs[5], lane, 42); // fc0: dy1 x x0[42:84]
        stage(DL, BW_DSTR, lane, dh0);                                               // T now holds dh0
        outer<BW_D1, 42, BW_D0, true>(DL, XL, part, A.offs[0], A.offs[1], lane, 42);  // fc1: dh0 x x0[42:84], bias
        stage_x0(0);
        outer<BW_D1, 42, BW_D0, false>(DL, XL, part, A.offs[0], A.offs[1], lane, 0);  // fc1: dh0 x x0[0:42]
        for (int cc = 0; cc < 4; ++cc) {
            float t1[20];
            bwd_lin<BW_D0, 20, BW_D1>(Pm + A.offs[0] + 20 * cc, T, t1);
            if (live) {
#pragma unroll
                for (int c = 0; c < 20; ++c) dxrow[20 * cc + c] = t1[c] + dxrow[20 * cc + c];
            }
        }
        lds_fence();
    }
}

// grad[q] = sum over groups of 32 consecutive partials (ascending) of the group's sum (ascending): a fixed
// two-level order (oracle/backward_oracle.c), 32 + 64 dependent adds instead of 2 048.
constexpr int BW_GROUP = 32;
__global__ __launch_bounds__(256) void bwd_reduce_kernel(const float* __restrict__ parts, int n_parts, int64_t n_params,
                                                         float* __restrict__ grad)
{
    __shared__ float gs[(CPPF_BWD_MAX_PARTS + BW_GROUP - 1) / BW_GROUP][64];
    const int qi = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 64 + qi;
    const int n_groups = (n_parts + BW_GROUP - 1) / BW_GROUP;
    if (q < n_params) {
        for (int gidx = slot; gidx < n_groups; gidx += 4) {
            const int w0 = gidx * BW_GROUP, w1 = min(w0 + BW_GROUP, n_parts);
            float acc = 0.f;
#pragma unroll 8
            for (int w = w0; w < w1; ++w) acc = acc + parts[(size_t)w * n_params + q];
            gs[gidx][qi] = acc;
        }
    }
    __syncthreads();
    if (slot == 0 && q < n_params) {
        float acc = 0.f;
        for (int gidx = 0; gidx < n_groups; ++gidx) acc = acc + gs[gidx][qi];
        grad[q] = acc;
    }
}

// ---- deterministic scatter-add of the per-pair rows into grad_feat ------------------------------------------------
// entry e in [0, 2P): e < P is pair e's a-half (key idx[e][0]), e >= P is pair (e - P)'s b-half (key idx[e-P][1])
__global__ __launch_bounds__(256) void bwd_keys_kernel(const void* __restrict__ idxs, int idx64, int64_t P, int32_t* __restrict__ keys,
                                                       int32_t* __restrict__ vals)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * P) return;
    const int64_t p = e < P ? e : e - P;
    const int half = e < P ? 0 : 1;
    keys[e] = idx64 ? (int32_t)reinterpret_cast<const int64_t*>(idxs)[2 * p + half]
                    : reinterpret_cast<const int32_t*>(idxs)[2 * p + half];
    vals[e] = (int32_t)e;
}
// seg[n] = first sorted position with key >= n (seg[N] = 2P)
__global__ __launch_bounds__(256) void bwd_seg_kernel(const int32_t* __restrict__ skeys, int64_t M, int64_t N, int32_t* __restrict__ seg)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > M) return;
    const int64_t lo = i == 0 ? 0 : (int64_t)skeys[i - 1] + 1;
    const int64_t hi = i == M ? N : (int64_t)skeys[i];
    for (int64_t n = lo; n <= hi && n <= N; ++n) seg[n] = (int32_t)i;   // (every n in (key[i-1], key[i]] starts at i)
}
// one wavefront per point: grad_feat[n][c] += rows of its entries in sorted (= entry) order, lanes = columns
__global__ __launch_bounds__(256) void bwd_gather_kernel(const float* __restrict__ dx, const int32_t* __restrict__ svals,
                                                         const int32_t* __restrict__ seg, int64_t P, int64_t N,
                                                         float* __restrict__ grad_feat)
{
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (n >= N || c >= BW_F) return;
    const int b = seg[n], e_ = seg[n + 1];
    auto row_of = [&](int i) {
        const int64_t e = svals[i];
        return e < P ? dx + (size_t)e * (2 * BW_F) : dx + (size_t)(e - P) * (2 * BW_F) + BW_F;
    };
    float acc = 0.f;
    int i = b;
    for (; i + 8 <= e_; i += 8) {   // eight independent row loads in flight, added in order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = row_of(i + u)[c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; i < e_; ++i) acc = acc + row_of(i)[c];
    grad_feat[(size_t)n * BW_F + c] = grad_feat[(size_t)n * BW_F + c] + acc;
}

bool std_shape(int F, const int* dims, int n_res)
{
    return F == BW_F && n_res == 3 && dims[0] == BW_D0 && dims[1] == BW_D1 && dims[2] == BW_D2 && dims[3] == BW_D3;
}
int64_t count_params(const int* dims, int n_res, int out_dim)
{
    int64_t n = 0;
    for (int l = 0; l < n_res; ++l) {
        const int K = dims[l], M = dims[l + 1];
        n += (int64_t)M * K + M + (int64_t)M * M + M + (K != M ? (int64_t)M * K + M : 0);
    }
    return n + (int64_t)out_dim * dims[n_res] + out_dim;
}
// number of partial accumulators = wavefronts: every wavefront gets the same number of tiles (+-1), at most
// CPPF_BWD_MAX_PARTS of them
int n_parts_for(int64_t P)
{
    const int64_t t = (P + 63) / 64;
    if (t <= 1) return 1;
    const int64_t per = (t + CPPF_BWD_MAX_PARTS - 1) / CPPF_BWD_MAX_PARTS;
    return (int)((t + per - 1) / per);
}

}  // namespace

extern "C" {

// workspace: [partial gradients][dx P*2F f32][keys, vals, sorted keys, sorted vals: 4 x 2P i32][seg N+1 i32][sort temp]
struct BwdLayout { size_t parts, dx, keys, seg, temp, temp_bytes, total; };
static BwdLayout bwd_layout(int64_t n_pairs, int64_t n_points, const int* dims, int n_res, int out_dim)
{
    BwdLayout L;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L.parts = 0;
    L.dx = up((size_t)n_parts_for(n_pairs) * (size_t)count_params(dims, n_res, out_dim) * sizeof(float));
    L.keys = L.dx + up((size_t)n_pairs * 2 * BW_F * sizeof(float));
    L.seg = L.keys + up((size_t)8 * n_pairs * sizeof(int32_t));
    L.temp = L.seg + up((size_t)(n_points + 1) * sizeof(int32_t));
    L.temp_bytes = up((size_t)64 * n_pairs + (1u << 20));   // generous bound for the radix sort's scratch (checked at run time)
    L.total = L.temp + L.temp_bytes;
    return L;
}

size_t cppf_pair_mlp_backward_workspace_bytes(int64_t n_pairs, int64_t n_points, int F, const int* dims, int n_res, int out_dim)
{
    if (!dims || n_pairs < 0 || n_points < 0 || !std_shape(F, dims, n_res) || out_dim < 1) return 0;
    return bwd_layout(n_pairs, n_points, dims, n_res, out_dim).total;
}

int cppf_pair_mlp_backward(const float* pc, const float* nrm, const float* feat, const void* idxs, int idx_is_i64,
                           const float* params, const int64_t* offs, int64_t n_points, int F, const int* dims, int n_res,
                           int64_t n_pairs, int out_dim, const float* grad_out, float* grad_params, float* grad_feat,
                           void* workspace, size_t workspace_bytes, void* stream)
{
    if (n_pairs < 0 || n_points < 0 || !dims || !offs || out_dim < 1 || n_pairs > 0x3fffffffll) return CPPF_EINVAL;
    if (!std_shape(F, dims, n_res)) return CPPF_EUNSUPPORTED;
    if (!grad_params) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_params = count_params(dims, n_res, out_dim);
    if (n_pairs == 0) return (int)hipMemsetAsync(grad_params, 0, n_params * sizeof(float), st);
    if (!pc || !nrm || !feat || !idxs || !params || !grad_out || !grad_feat) return CPPF_EINVAL;
    const int n_parts = n_parts_for(n_pairs);
    const BwdLayout Lw = bwd_layout(n_pairs, n_points, dims, n_res, out_dim);
    if (!workspace || workspace_bytes < Lw.total) return CPPF_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    hipError_t e = hipMemsetAsync(ws + Lw.parts, 0, (size_t)n_parts * n_params * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    BwdArgs A;
    A.pc = pc; A.nrm = nrm; A.feat = feat; A.idxs = idxs; A.params = params; A.grad_out = grad_out;
    A.parts = (float*)(ws + Lw.parts); A.dx = (float*)(ws + Lw.dx); A.P = n_pairs; A.n_params = n_params;
    for (int i = 0; i < 20; ++i) A.offs[i] = offs[i];
    A.out_dim = out_dim; A.idx64 = idx_is_i64; A.n_parts = n_parts;
    pair_mlp_bwd_kernel<<<n_parts, 64, 0, st>>>(A);
    bwd_reduce_kernel<<<(int)((n_params + 63) / 64), 256, 0, st>>>(A.parts, n_parts, n_params, grad_params);
    // d/d(feat): stable sort of the 2P (point, entry) keys, segment starts, ordered row sums
    const int64_t M = 2 * n_pairs;
    int32_t* keys = (int32_t*)(ws + Lw.keys);
    int32_t *vals = keys + M, *skeys = keys + 2 * M, *svals = keys + 3 * M, *seg = (int32_t*)(ws + Lw.seg);
    bwd_keys_kernel<<<(int)((M + 255) / 256), 256, 0, st>>>(idxs, idx_is_i64, n_pairs, keys, vals);
    int end_bit = 1;
    while (end_bit < 31 && (1ll << end_bit) < n_points) ++end_bit;
    size_t need = 0;
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, skeys, vals, svals, (int)M, 0, end_bit, st);
    if (e != hipSuccess) return (int)e;
    if (need > Lw.temp_bytes) return CPPF_EWORKSPACE;
    need = Lw.temp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(ws + Lw.temp, need, keys, skeys, vals, svals, (int)M, 0, end_bit, st);
    if (e != hipSuccess) return (int)e;
    bwd_seg_kernel<<<(int)((M + 256) / 256), 256, 0, st>>>(skeys, M, n_points, seg);
    bwd_gather_kernel<<<(int)((n_points + 3) / 4), 256, 0, st>>>(A.dx, svals, seg, n_pairs, n_points, grad_feat);
    return (int)hipGetLastError();
}

}  // extern "C"
