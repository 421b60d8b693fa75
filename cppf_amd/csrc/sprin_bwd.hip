// Backward of the SPRIN point encoder for gfx950: parameter gradients of one SparseSO3Conv + GlobalInfoProp stage in the
// configuration train.py:34 trains (k <= 64 neighbours, kernel-MLP 6->32->64->32->32->32 with LayerNorm + ReLU, rank 32,
// two neighbour features, 32 + 8 outputs).  C ABI in include/cppf.h.
//
// The reference has no backward code: train.py:91 calls loss.backward() and autograd differentiates models/model.py:46-61
// and models/sprin.py:40-107; points and normals carry no gradient (train.py:58-60), so the result is d/d(parameter).
//
// One wavefront owns a point (like the forward, sprin.hip) and nothing is shared between wavefronts, so there is not a
// single barrier in the kernel.  It runs ONE wavefront per SIMD with the full 512-register budget: on gfx950 the fp32
// MFMA and the VALU share a pipe, so a second wavefront could only hide latency -- and the latencies that matter here
// (LDS operand reads) are prefetched explicitly.  Per point:
//   pass 1   the forward of sprin.hip for the four 16-row blocks -> kernel values -> contraction with the neighbour features
//   point    outnet + LayerNorm + GlobalInfoProp forward and backward (lanes = outputs), giving d(contraction)
//   pass 2   per 16-row block: the forward again (its activations are needed now and 4 x 48 registers would not fit),
//            backward-data as transposed MFMA chains (transposed lane-ordered images of W5..W2), LayerNorm backward on the
//            4-lane rows, and the weight gradients: deltas and activations pass through a wave-private LDS staging area as
//            [row][16 features] blocks and come back with rows on the k axis; 26 gradient tiles (16 x 16) accumulate in
//            registers over every row the wavefront ever sees.
// Every sum has a fixed order, restated in oracle/sprin_bwd_oracle.c: the result is bit-identical to the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cppf.h"
#include "cppf_math.h"
#include "sprin_layout.h"

using namespace cppf;
using namespace sprin;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NG = 8;                          // n_glob
constexpr int OUTW = SP_NOUT + NG;             // 40
constexpr int CMIX = SP_RANK * 2;              // 64
// natural offsets behind the kernel-MLP: outnet (transposed [64][32]) | bias | LN gamma | LN beta | aggr W [8][32] | bias
constexpr int NAT_WO = SP_NAT_KERNEL, NAT_BO = NAT_WO + CMIX * SP_NOUT, NAT_GO = NAT_BO + SP_NOUT, NAT_EO = NAT_GO + SP_NOUT,
              NAT_WA = NAT_EO + SP_NOUT, NAT_BA = NAT_WA + NG * SP_NOUT, NAT_TOTAL = NAT_BA + NG;   // 9 256
constexpr int SPB_WAVES = 4;
constexpr int STG_BLK = 16 * 16;               // one staged block: [16 rows][16 features]
// per-wave LDS scratch (floats)
constexpr int PW_X6 = 0, PW_NF = PW_X6 + 64 * 8, PW_RR = PW_NF + 64 * 2, PW_KERN = PW_RR + 64 * 3, PW_MIX = PW_KERN + 16 * SP_KSTRIDE,
              PW_V = PW_MIX + CMIX, PW_DMIX = PW_V + 128, PW_STG = PW_DMIX + CMIX, PW_FLOATS = PW_STG + 6 * STG_BLK;
constexpr int SPB_LDS_FLOATS = SPW_FLOATS + SPT_FLOATS + SPB_WAVES * PW_FLOATS;

struct SpbArgs {
    const float* pc;
    const float* nrm;
    const int32_t* nbrs;
    const float* nat;       // natural parameters of the layer (device)
    const float* wimg;      // forward image (SPW_FLOATS)
    const float* timg;      // transposed image (SPT_FLOATS)
    const float* grad_out;  // [N][40]
    const float* share;     // [8] d(pooled)[c] / number of points attaining the maximum
    const float* pooled;    // [8] the maxima (columns 32..39 of any row of the forward output)
    const float* mixed;     // optional [N][64]: the contraction the forward kept (cppf_point_encoder_forward_train); null = recompute
    float* parts;           // [n_parts][NAT_TOTAL]
    int N, k, n_parts;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// LayerNorm + ReLU on a row spread over 4 lanes (as sp_ln_relu4), keeping what the backward needs: xh = normalised value,
// a = relu(xh * gamma + beta), inv.  Sums: per-lane (ob, r) order, combined (p0 + p1) + (p2 + p3) through the LDS crossbar.
template <int NOB>
__device__ __forceinline__ void ln_fwd(const f32x4 (&y)[NOB], const float* __restrict__ gamma, const float* __restrict__ beta, int lane,
                                       int g, f32x4 (&xh)[NOB], f32x4 (&a)[NOB], float& inv)
{
    constexpr float H = 16.f * NOB;
    float p = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) p = p + y[ob][r];
    p = p + sp_xor16(p);
    p = p + sp_xor32(p, lane);
    const float mean = p / H;
    float q = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = y[ob][r] - mean; q = q + d * d; }
    q = q + sp_xor16(q);
    q = q + sp_xor32(q, lane);
    inv = 1.0f / sqrtf(q / H + 1e-5f);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * ob + 4 * g);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * ob + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[ob][r] = (y[ob][r] - mean) * inv;
            const float z = xh[ob][r] * gm[r] + bt[r];
            a[ob][r] = z > 0.f ? z : 0.f;
        }
    }
}
// d(pre-LN) from d(a):  dz = relu'(.) d(a);  gd = dz * gamma;  dy = ((gd - mean(gd)) - xh * mean(gd * xh)) * inv.
// dz is accumulated into the gamma / beta gradient slots of the lane (row slot j, features 4g + r): dG += dz * xh, dE += dz.
template <int NOB>
__device__ __forceinline__ void ln_bwd(const f32x4 (&da)[NOB], const f32x4 (&xh)[NOB], const f32x4 (&a)[NOB], float inv,
                                       const float* __restrict__ gamma, int lane, int g, f32x4 (&dy)[NOB], f32x4 (&dG)[NOB],
                                       f32x4 (&dE)[NOB])
{
    constexpr float H = 16.f * NOB;
    f32x4 gd[NOB];
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * ob + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dz = a[ob][r] > 0.f ? da[ob][r] : 0.f;
            dG[ob][r] = dG[ob][r] + dz * xh[ob][r];
            dE[ob][r] = dE[ob][r] + dz;
            gd[ob][r] = dz * gm[r];
        }
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p1 = p1 + gd[ob][r]; p2 = p2 + gd[ob][r] * xh[ob][r]; }
    p1 = p1 + sp_xor16(p1); p1 = p1 + sp_xor32(p1, lane);
    p2 = p2 + sp_xor16(p2); p2 = p2 + sp_xor32(p2, lane);
    const float m1 = p1 / H, m2 = p2 / H;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) dy[ob][r] = ((gd[ob][r] - m1) - xh[ob][r] * m2) * inv;
}
// One wavefront per SIMD hides nothing by itself: every block of MFMAs gets its A operands (a layer's lane-ordered
// weights, or the staged deltas / activations of a gradient tile) loaded into registers one block AHEAD, and
// sched_barrier pins "next block's ds_reads | this block's MFMAs" (see pair_mlp_bwd.hip).
#define SPB_SB() __builtin_amdgcn_sched_barrier(0)
template <int NW>
__device__ __forceinline__ void ldw(const float* __restrict__ base, int lane, float (&w)[NW])
{
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = base[i * 64 + lane];
}
template <int NIB, int NOB>   // y = bias + W x with the layer's weights already in registers (image order [ob][s])
__device__ __forceinline__ void layer_pre(const float (&w)[NOB * 4 * NIB], const float* __restrict__ bias, const f32x4 (&x)[NIB],
                                          f32x4 (&y)[NOB], int g)
{
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) y[ob] = *reinterpret_cast<const f32x4*>(bias + 16 * ob + 4 * g);
#pragma unroll
    for (int s = 0; s < 4 * NIB; ++s)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) y[ob] = mfma4(w[ob * 4 * NIB + s], x[s / 4][s % 4], y[ob]);
}
// transposed chain: d(in)[16*ib + 4g + r] of the lane's row = chain over the outputs in khid order of W[o][i] * d[o], from 0;
// the transposed weights are already in registers (image order [ib][s])
template <int NIB, int NOB>
__device__ __forceinline__ void t_pre(const float (&w)[NIB * 4 * NOB], const f32x4 (&d)[NOB], f32x4 (&dx)[NIB])
{
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib) dx[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4 * NOB; ++s)
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) dx[ib] = mfma4(w[ib * 4 * NOB + s], d[s / 4][s % 4], dx[ib]);
}

__global__ __launch_bounds__(SPB_WAVES * 64, 1) void sprin_bwd_kernel(SpbArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Wl = lds;
    float* const Tl = lds + SPW_FLOATS;
    for (int i = threadIdx.x; i < SPW_FLOATS / 4; i += blockDim.x)
        reinterpret_cast<f32x4*>(Wl)[i] = reinterpret_cast<const f32x4*>(A.wimg)[i];
    for (int i = threadIdx.x; i < SPT_FLOATS / 4; i += blockDim.x)
        reinterpret_cast<f32x4*>(Tl)[i] = reinterpret_cast<const f32x4*>(A.timg)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    __builtin_assume(g >= 0 && g < 4);
    float* const pw = lds + SPW_FLOATS + SPT_FLOATS + wave * PW_FLOATS;
    float* const x6l = pw + PW_X6;
    float* const nf = pw + PW_NF;
    float* const rr = pw + PW_RR;
    float* const kern = pw + PW_KERN;
    float* const mix = pw + PW_MIX;
    float* const vv = pw + PW_V;       // [0,32) y | [32,64) gd | [64,96) gd * xh | [96,104) lin / dlin
    float* const dmix = pw + PW_DMIX;
    float* const stg = pw + PW_STG;
    const int w = blockIdx.x * SPB_WAVES + wave;
    if (w >= A.n_parts) return;
    const int k = A.k;
    const float* const P = A.nat;

    // ---- persistent accumulators -------------------------------------------------------------------------------
    // gradient tiles (register r of lane (j, g) = d(W)[row 16*rb + 4g + r][col 16*cb + j]) and bias sub-sums of lane (m = j, kk = g)
    f32x4 tW5[2][2], tW4[2][2], tW3[2][4], tW2[4][2], tW1[2];
    float sb5[2] = {0.f, 0.f}, sb4[2] = {0.f, 0.f}, sb3[2] = {0.f, 0.f}, sb2[4] = {0.f, 0.f, 0.f, 0.f}, sb1[2] = {0.f, 0.f};
    // LayerNorm gamma / beta gradient slots of lane (row slot j, features 16*ob + 4g + r)
    f32x4 dG1[2], dE1[2], dG2[4], dE2[4], dG3[2], dE3[2], dG4[2], dE4[2];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        tW1[a] = z4; dG1[a] = dE1[a] = dG3[a] = dE3[a] = dG4[a] = dE4[a] = z4;
#pragma unroll
        for (int b = 0; b < 2; ++b) { tW5[a][b] = tW4[a][b] = z4; }
#pragma unroll
        for (int b = 0; b < 4; ++b) tW3[a][b] = z4;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) { tW2[a][0] = tW2[a][1] = z4; dG2[a] = dE2[a] = z4; }
    // per-point parameters: lane l owns output o = l & 31; outnet weight columns c = 32*(l >> 5) + q; aggr rows 4*(l >> 5) + q
    float aWo[32], aWa[4] = {0.f, 0.f, 0.f, 0.f}, aBo = 0.f, aGo = 0.f, aEo = 0.f, aBa = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) aWo[q] = 0.f;
    const int o_ = lane & 31, hi = lane >> 5;

    // staging: the lane's 16 bytes of block 0 (chunk index XOR-swizzled, see pair_mlp_bwd.hip) and the read offsets
    float* const my = stg + j * 16 + 4 * (g ^ ((j >> 1) & 3));
    const int rd0 = 16 * g + 4 * ((j >> 2) ^ (g >> 1)) + (j & 3);
    auto put = [&](int blk, f32x4 v) { *reinterpret_cast<f32x4*>(my + blk * STG_BLK) = v; };
    auto rdw = [&](int blk, int s) -> float { return stg[blk * STG_BLK + 64 * s + (rd0 ^ ((s & 1) << 3))]; };
    // gradient tiles of one layer: NA delta blocks (staged 0..NA-1) x NB input blocks (staged NA..NA+NB-1).  WG_FETCH
    // reads the 4 x (NA + NB) operands (rows on the k axis), WG_RUN issues the NA x NB x 4 MFMAs; bias sub-sums += deltas
#define WG_FETCH(NA, NB, OPS)                                                                     \
    do {                                                                                          \
        wave_lds_fence();                                                                         \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                             \
            _Pragma("unroll") for (int q = 0; q < NA + NB; ++q) OPS[s][q] = rdw(q, s);            \
    } while (0)
#define WG_RUN(NA, NB, OPS, TILE, SB)                                                             \
    do {                                                                                          \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                             \
            _Pragma("unroll") for (int a = 0; a < NA; ++a) {                                      \
                SB[a] = SB[a] + OPS[s][a];                                                        \
                _Pragma("unroll") for (int b = 0; b < NB; ++b) TILE[a][b] = mfma4(OPS[s][a], OPS[s][NA + b], TILE[a][b]); \
            }                                                                                     \
    } while (0)

    for (int n = w; n < A.N; n += A.n_parts) {
        asm volatile("" ::: "memory");
        // ---- rows: neighbour gather, rifeat, neighbour features (lane = row) ---------------------------------------
        const int jc = lane < k ? lane : k - 1;
        const int nb = A.nbrs[(size_t)n * k + jc];
        const float rx = A.pc[3 * nb], ry = A.pc[3 * nb + 1], rz = A.pc[3 * nb + 2];
        const float sx = A.pc[3 * n], sy = A.pc[3 * n + 1], sz = A.pc[3 * n + 2];
        rr[3 * lane] = rx; rr[3 * lane + 1] = ry; rr[3 * lane + 2] = rz;
        wave_lds_fence();
        float mx = 0.f, my_ = 0.f, mz = 0.f;
#pragma unroll 4
        for (int q = 0; q < k; ++q) { mx = mx + rr[3 * q]; my_ = my_ + rr[3 * q + 1]; mz = mz + rr[3 * q + 2]; }
        mx = mx / (float)k; my_ = my_ / (float)k; mz = mz / (float)k;
        {
            const float l1x = mx - rx, l1y = my_ - ry, l1z = mz - rz;
            const float l2x = rx - sx, l2y = ry - sy, l2z = rz - sz;
            const float l3x = sx - mx, l3y = sy - my_, l3z = sz - mz;
            const float l1n = norm3(l1x, l1y, l1z), l2n = norm3(l2x, l2y, l2z), l3n = norm3(l3x, l3y, l3z);
            x6l[lane * 8 + 0] = l1n; x6l[lane * 8 + 1] = l2n; x6l[lane * 8 + 2] = l3n;
            x6l[lane * 8 + 3] = ((l1x * l2x + l1y * l2y) + l1z * l2z) / (l1n * l2n + 1e-7f);
            x6l[lane * 8 + 4] = ((l2x * l3x + l2y * l3y) + l2z * l3z) / (l2n * l3n + 1e-7f);
            x6l[lane * 8 + 5] = ((l3x * l1x + l3y * l1y) + l3z * l1z) / (l3n * l1n + 1e-7f);
            x6l[lane * 8 + 6] = 0.f; x6l[lane * 8 + 7] = 0.f;
            const float nax = A.nrm[3 * nb], nay = A.nrm[3 * nb + 1], naz = A.nrm[3 * nb + 2];
            const float nsx = A.nrm[3 * n], nsy = A.nrm[3 * n + 1], nsz = A.nrm[3 * n + 2];
            nf[lane * 2] = l2n;
            nf[lane * 2 + 1] = (nax * nsx + nay * nsy) + naz * nsz;
        }
        wave_lds_fence();
        // ---- pass 1: kernel values of the four row blocks, contraction (lane t = r * 2 + i) ------------------------
        float contr = 0.f;
        if (A.mixed) contr = A.mixed[(size_t)n * CMIX + lane];
#pragma unroll 1
        for (int rb = 0; rb < (A.mixed ? 0 : 4); ++rb) {
            f32x4 a1[2], a2[4], a3[2], a4[2], kr[2];
            float w1[4], wa[32], wb[32], wc[16];
            ldw<4>(Wl + SPW_L1, lane, w1);
            ldw<32>(Wl + SPW_L2, lane, wa);
            const float bx0 = x6l[(16 * rb + j) * 8 + g], bx1 = x6l[(16 * rb + j) * 8 + 4 + g];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) a1[ob] = *reinterpret_cast<const f32x4*>(Wl + SPW_B1 + 16 * ob + 4 * g);
            SPB_SB();
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) { a1[ob] = mfma4(w1[ob * 2], bx0, a1[ob]); a1[ob] = mfma4(w1[ob * 2 + 1], bx1, a1[ob]); }
            ldw<32>(Wl + SPW_L3, lane, wb);
            SPB_SB();
            sp_ln_relu4<2>(a1, Wl + SPW_B1 + 32, Wl + SPW_B1 + 64, lane, g);
            layer_pre<2, 4>(wa, Wl + SPW_B2, a1, a2, g);
            ldw<16>(Wl + SPW_L4, lane, wc);
            SPB_SB();
            sp_ln_relu4<4>(a2, Wl + SPW_B2 + 64, Wl + SPW_B2 + 128, lane, g);
            layer_pre<4, 2>(wb, Wl + SPW_B3, a2, a3, g);
            {
                float wd[16];
                ldw<16>(Wl + SPW_L5, lane, wd);
                SPB_SB();
                sp_ln_relu4<2>(a3, Wl + SPW_B3 + 32, Wl + SPW_B3 + 64, lane, g);
                layer_pre<2, 2>(wc, Wl + SPW_B4, a3, a4, g);
                sp_ln_relu4<2>(a4, Wl + SPW_B4 + 32, Wl + SPW_B4 + 64, lane, g);
                layer_pre<2, 2>(wd, Wl + SPW_B5, a4, kr, g);
            }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) kern[j * SP_KSTRIDE + 16 * ob + 4 * g + r] = kr[ob][r];
            wave_lds_fence();
            const int jn = min(16, k - 16 * rb);
            const int r_ = lane >> 1, i_ = lane & 1;
            for (int jj = 0; jj < jn; ++jj) contr = fmaf(kern[jj * SP_KSTRIDE + r_], nf[(16 * rb + jj) * 2 + i_], contr);
            wave_lds_fence();
        }
        mix[lane] = contr;
        wave_lds_fence();
        // ---- point level: outnet + LayerNorm + GlobalInfoProp, forward and backward (lane = output o_, both halves) ----
        float yo = P[NAT_BO + o_];
#pragma unroll 8
        for (int c = 0; c < CMIX; ++c) yo = fmaf(P[NAT_WO + c * SP_NOUT + o_], mix[c], yo);
        vv[o_] = yo;
        wave_lds_fence();
        float sm = 0.f;
#pragma unroll
        for (int q = 0; q < SP_NOUT; ++q) sm = sm + vv[q];
        const float mean = sm / (float)SP_NOUT;
        float var = 0.f;
#pragma unroll
        for (int q = 0; q < SP_NOUT; ++q) { const float d = vv[q] - mean; var = var + d * d; }
        const float inv = 1.0f / sqrtf(var / (float)SP_NOUT + 1e-5f);
        const float xh = (yo - mean) * inv;
        const float xo = xh * P[NAT_GO + o_] + P[NAT_EO + o_];
        wave_lds_fence();
        vv[o_] = xo;                                   // x of the point, read below by every lane
        wave_lds_fence();
        {   // lin[c] = ba[c] + chain_o Wa[c][o] x[o]  (lanes 0..7 matter; every lane computes c = lane & 7)
            const int c = lane & 7;
            float lin = P[NAT_BA + c];
#pragma unroll 8
            for (int q = 0; q < SP_NOUT; ++q) lin = fmaf(P[NAT_WA + c * SP_NOUT + q], vv[q], lin);
            const float dl = lin == A.pooled[c] ? A.share[c] : 0.f;
            if (lane < NG) { vv[96 + lane] = dl; aBa = aBa + dl; }
        }
        wave_lds_fence();
        float dxo;
        {
            float t = 0.f;
#pragma unroll
            for (int c = 0; c < NG; ++c) t = fmaf(P[NAT_WA + c * SP_NOUT + o_], vv[96 + c], t);
            dxo = A.grad_out[(size_t)n * OUTW + o_] + t;
#pragma unroll
            for (int q = 0; q < 4; ++q) aWa[q] = fmaf(vv[96 + 4 * hi + q], xo, aWa[q]);
        }
        const float gdo = dxo * P[NAT_GO + o_];
        if (hi == 0) { aGo = aGo + dxo * xh; aEo = aEo + dxo; }
        vv[32 + o_] = gdo;
        vv[64 + o_] = gdo * xh;
        wave_lds_fence();
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < SP_NOUT; ++q) { s1 = s1 + vv[32 + q]; s2 = s2 + vv[64 + q]; }
        const float dyo = ((gdo - s1 / (float)SP_NOUT) - xh * (s2 / (float)SP_NOUT)) * inv;
        if (hi == 0) aBo = aBo + dyo;
#pragma unroll
        for (int q = 0; q < 32; ++q) aWo[q] = fmaf(dyo, mix[32 * hi + q], aWo[q]);
        wave_lds_fence();
        vv[o_] = dyo;
        wave_lds_fence();
        {   // d(mixed)[c] (lane = c) = chain_o Wo_t[c][o] dy[o]
            float acc = 0.f;
            const float* wr = P + NAT_WO + lane * SP_NOUT;
#pragma unroll 8
            for (int q = 0; q < SP_NOUT; ++q) acc = fmaf(wr[q], vv[q], acc);
            dmix[lane] = acc;
        }
        wave_lds_fence();
        // ---- pass 2: per row block, forward again, backward, weight gradients -----------------------------------------
#pragma unroll 1
        for (int rb = 0; rb < 4; ++rb) {
            const int row = 16 * rb + j;
            f32x4 y1[2], y2[4], y3[2], y4[2], xh1[2], xh2[4], xh3[2], xh4[2], a1[2], a2[4], a3[2], a4[2];
            float inv1, inv2, inv3, inv4;
            float wa[32], wb[32], wc[16], ops[4][6];
            {
                float w1[4];
                ldw<4>(Wl + SPW_L1, lane, w1);
                ldw<32>(Wl + SPW_L2, lane, wa);
                const float bx0 = x6l[row * 8 + g], bx1 = x6l[row * 8 + 4 + g];
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) y1[ob] = *reinterpret_cast<const f32x4*>(Wl + SPW_B1 + 16 * ob + 4 * g);
                SPB_SB();
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) { y1[ob] = mfma4(w1[ob * 2], bx0, y1[ob]); y1[ob] = mfma4(w1[ob * 2 + 1], bx1, y1[ob]); }
            }
            ldw<32>(Wl + SPW_L3, lane, wb);
            SPB_SB();
            ln_fwd<2>(y1, Wl + SPW_B1 + 32, Wl + SPW_B1 + 64, lane, g, xh1, a1, inv1);
            layer_pre<2, 4>(wa, Wl + SPW_B2, a1, y2, g);
            ldw<16>(Wl + SPW_L4, lane, wc);
            SPB_SB();
            ln_fwd<4>(y2, Wl + SPW_B2 + 64, Wl + SPW_B2 + 128, lane, g, xh2, a2, inv2);
            layer_pre<4, 2>(wb, Wl + SPW_B3, a2, y3, g);
            // d(kernel value)[f] of the row = chain_i d(mixed)[2f + i] * nf[row][i]; rows past k contribute nothing
            f32x4 dk[2];
            {
                const float n0 = nf[row * 2], n1 = nf[row * 2 + 1];
                const bool live = row < k;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 16 * ob + 4 * g + r;
                        const float v = fmaf(dmix[2 * f + 1], n1, dmix[2 * f] * n0);
                        dk[ob][r] = live ? v : 0.f;
                    }
            }
            float w5[16];
            ldw<16>(Tl + SPT_5, lane, w5);
            SPB_SB();
            ln_fwd<2>(y3, Wl + SPW_B3 + 32, Wl + SPW_B3 + 64, lane, g, xh3, a3, inv3);
            layer_pre<2, 2>(wc, Wl + SPW_B4, a3, y4, g);
            ln_fwd<2>(y4, Wl + SPW_B4 + 32, Wl + SPW_B4 + 64, lane, g, xh4, a4, inv4);
            // ---- layer 5: d(W5) += dk x a4;  d(a4) = W5^T dk
            put(0, dk[0]); put(1, dk[1]); put(2, a4[0]); put(3, a4[1]);
            WG_FETCH(2, 2, ops);
            ldw<16>(Tl + SPT_4, lane, wc);
            SPB_SB();
            f32x4 da4[2], dy4[2];
            t_pre<2, 2>(w5, dk, da4);
            WG_RUN(2, 2, ops, tW5, sb5);
            ln_bwd<2>(da4, xh4, a4, inv4, Wl + SPW_B4 + 32, lane, g, dy4, dG4, dE4);
            // ---- layer 4
            put(0, dy4[0]); put(1, dy4[1]); put(2, a3[0]); put(3, a3[1]);
            WG_FETCH(2, 2, ops);
            ldw<32>(Tl + SPT_3, lane, wa);
            SPB_SB();
            f32x4 da3[2], dy3[2];
            t_pre<2, 2>(wc, dy4, da3);
            WG_RUN(2, 2, ops, tW4, sb4);
            ln_bwd<2>(da3, xh3, a3, inv3, Wl + SPW_B3 + 32, lane, g, dy3, dG3, dE3);
            // ---- layer 3 (64 -> 32)
            put(0, dy3[0]); put(1, dy3[1]); put(2, a2[0]); put(3, a2[1]); put(4, a2[2]); put(5, a2[3]);
            WG_FETCH(2, 4, ops);
            ldw<32>(Tl + SPT_2, lane, wb);
            SPB_SB();
            f32x4 da2[4], dy2[4];
            t_pre<4, 2>(wa, dy3, da2);
            WG_RUN(2, 4, ops, tW3, sb3);
            ln_bwd<4>(da2, xh2, a2, inv2, Wl + SPW_B2 + 64, lane, g, dy2, dG2, dE2);
            // ---- layer 2 (32 -> 64)
            put(0, dy2[0]); put(1, dy2[1]); put(2, dy2[2]); put(3, dy2[3]); put(4, a1[0]); put(5, a1[1]);
            WG_FETCH(4, 2, ops);
            SPB_SB();
            f32x4 da1[2], dy1[2];
            t_pre<2, 4>(wb, dy2, da1);
            WG_RUN(4, 2, ops, tW2, sb2);
            ln_bwd<2>(da1, xh1, a1, inv1, Wl + SPW_B1 + 32, lane, g, dy1, dG1, dE1);
            // ---- layer 1 (6 -> 32): inputs as one block [row][x6 (6) | zeros]
            f32x4 x6b = {0.f, 0.f, 0.f, 0.f};
            if (g < 2) x6b = *reinterpret_cast<const f32x4*>(x6l + row * 8 + 4 * g);
            put(0, dy1[0]); put(1, dy1[1]); put(2, x6b);
            WG_FETCH(2, 1, ops);
            {
                f32x4 (&t1)[2][1] = *reinterpret_cast<f32x4 (*)[2][1]>(&tW1);
                WG_RUN(2, 1, ops, t1, sb1);
            }
            wave_lds_fence();
        }
    }
#undef WG_FETCH
#undef WG_RUN

    // ---- epilogue: this wavefront's partial gradient ------------------------------------------------------------
    float* part = A.parts + (size_t)w * NAT_TOTAL;
    auto put_tile = [&](int off, int ld, int rb, int cb, int cols, f32x4 v) {
        const int c = 16 * cb + j;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c < cols) part[off + (16 * rb + 4 * g + r) * ld + c] = v[r];
    };
    auto put_bias = [&](int off, int rb, float v) {
        const float s0 = __shfl(v, j), s1_ = __shfl(v, j + 16), s2_ = __shfl(v, j + 32), s3_ = __shfl(v, j + 48);
        const float t = ((s0 + s1_) + s2_) + s3_;
        if (g == 0) part[off + 16 * rb + j] = t;
    };
    // LayerNorm vectors: sequential sum over the 16 row slots j of the lane's slot value (through the wave's scratch)
    auto put_ln = [&](int off, int ob, f32x4 v) {
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[j * 16 + 4 * g + r] = v[r];   // [slot j][feature 4g + r]
        wave_lds_fence();
        if (lane < 16) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = acc + stg[s * 16 + lane];
            part[off + 16 * ob + lane] = acc;
        }
    };
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) { put_tile(NAT_W5, 32, a, b, 32, tW5[a][b]); put_tile(NAT_W4, 32, a, b, 32, tW4[a][b]); }
#pragma unroll
        for (int b = 0; b < 4; ++b) put_tile(NAT_W3, 64, a, b, 64, tW3[a][b]);
        put_tile(NAT_W1, 6, a, 0, 6, tW1[a]);
        put_bias(NAT_V5, a, sb5[a]); put_bias(NAT_V4, a, sb4[a]); put_bias(NAT_V3, a, sb3[a]); put_bias(NAT_V1, a, sb1[a]);
        put_ln(NAT_V1 + 32, a, dG1[a]); put_ln(NAT_V1 + 64, a, dE1[a]);
        put_ln(NAT_V3 + 32, a, dG3[a]); put_ln(NAT_V3 + 64, a, dE3[a]);
        put_ln(NAT_V4 + 32, a, dG4[a]); put_ln(NAT_V4 + 64, a, dE4[a]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        put_tile(NAT_W2, 32, a, 0, 32, tW2[a][0]); put_tile(NAT_W2, 32, a, 1, 32, tW2[a][1]);
        put_bias(NAT_V2, a, sb2[a]);
        put_ln(NAT_V2 + 64, a, dG2[a]); put_ln(NAT_V2 + 128, a, dE2[a]);
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) part[NAT_WO + (32 * hi + q) * SP_NOUT + o_] = aWo[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) part[NAT_WA + (4 * hi + q) * SP_NOUT + o_] = aWa[q];
    if (hi == 0) { part[NAT_BO + o_] = aBo; part[NAT_GO + o_] = aGo; part[NAT_EO + o_] = aEo; }
    if (lane < NG) part[NAT_BA + lane] = aBa;
}

// grad[q] = sum over groups of 32 consecutive partials (ascending) of the group's sum (ascending)
__global__ __launch_bounds__(256) void spb_reduce_kernel(const float* __restrict__ parts, int n_parts, int n_params, float* __restrict__ grad)
{
    __shared__ float gs[32][64];
    const int qi = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + qi;
    const int n_groups = (n_parts + 31) / 32;
    if (q < n_params) {
        for (int gi = slot; gi < n_groups; gi += 4) {
            const int w0 = gi * 32, w1 = min(w0 + 32, n_parts);
            float acc = 0.f;
#pragma unroll 8
            for (int w = w0; w < w1; ++w) acc = acc + parts[(size_t)w * n_params + q];
            gs[gi][qi] = acc;
        }
    }
    __syncthreads();
    if (slot == 0 && q < n_params) {
        float acc = 0.f;
        for (int gi = 0; gi < n_groups; ++gi) acc = acc + gs[gi][qi];
        grad[q] = acc;
    }
}

// forward + transposed images of the layer from its natural parameters (device -> device)
__global__ __launch_bounds__(256) void spb_pack_kernel(const float* __restrict__ nat, float* __restrict__ wimg, float* __restrict__ timg)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (wimg && i < SPW_FLOATS) wimg[i] = sp_image_elem(i, nat);
    if (timg && i < SPT_FLOATS) timg[i] = sp_timage_elem(i, nat);
}

// d(pooled) bookkeeping.  chunk kernel: one workgroup per 64 points; thread (c, lane p): lin[n][c] recomputed from the
// saved output, tie count (integer atomics: exact) and the chunk's sum of grad_out[n][32 + c] in ascending n
__global__ __launch_bounds__(64) void spb_pool_chunk_kernel(const float* __restrict__ out_fwd, const float* __restrict__ grad_out, int N,
                                                            const float* __restrict__ P, float* __restrict__ chunk_sums, int* __restrict__ cnt)
{
    __shared__ float gsh[64][NG + 1];
    const int n = blockIdx.x * 64 + threadIdx.x;
    const bool in = n < N;
#pragma unroll
    for (int c = 0; c < NG; ++c) gsh[threadIdx.x][c] = in ? grad_out[(size_t)n * OUTW + SP_NOUT + c] : 0.f;
    if (in) {
        float x[SP_NOUT];
#pragma unroll
        for (int q = 0; q < SP_NOUT; ++q) x[q] = out_fwd[(size_t)n * OUTW + q];
        for (int c = 0; c < NG; ++c) {
            float lin = P[NAT_BA + c];
#pragma unroll
            for (int q = 0; q < SP_NOUT; ++q) lin = fmaf(P[NAT_WA + c * SP_NOUT + q], x[q], lin);
            if (lin == out_fwd[(size_t)n * OUTW + SP_NOUT + c]) atomicAdd(&cnt[c], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < NG) {
        const int c = threadIdx.x;
        float acc = 0.f;
        for (int q = 0; q < 64 && blockIdx.x * 64 + q < N; ++q) acc = acc + gsh[q][c];
        chunk_sums[blockIdx.x * NG + c] = acc;
    }
}
__global__ __launch_bounds__(64) void spb_pool_final_kernel(const float* __restrict__ chunk_sums, int n_chunks, const int* __restrict__ cnt,
                                                            const float* __restrict__ out_fwd, float* __restrict__ share, float* __restrict__ pooled)
{
    const int c = threadIdx.x;
    if (c >= NG) return;
    float acc = 0.f;
    for (int q = 0; q < n_chunks; ++q) acc = acc + chunk_sums[q * NG + c];
    share[c] = acc / (float)cnt[c];
    pooled[c] = out_fwd[SP_NOUT + c];
}

bool spb_std(const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob, int num_layers)
{
    return hidden && n_hidden == 4 && hidden[0] == 32 && hidden[1] == 64 && hidden[2] == 32 && hidden[3] == 32 && rank == SP_RANK &&
           n_out == SP_NOUT && n_glob == NG && n_nbr_feats == 2 && num_layers == 1;
}
int spb_parts(int n_points) { return n_points < CPPF_SPRIN_BWD_MAX_PARTS ? n_points : CPPF_SPRIN_BWD_MAX_PARTS; }

}  // namespace

extern "C" {

// workspace: [transposed image][share 8 | pooled 8 | counts 8][chunk sums][partial gradients]
struct SpbLayout { size_t timg, small, chunks, parts, total; int n_chunks; };
static SpbLayout spb_layout(int n_points)
{
    SpbLayout L;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L.n_chunks = (n_points + 63) / 64;
    L.timg = 0;
    L.small = up((size_t)SPT_FLOATS * sizeof(float));
    L.chunks = L.small + 256;
    L.parts = L.chunks + up((size_t)L.n_chunks * NG * sizeof(float));
    L.total = L.parts + up((size_t)spb_parts(n_points) * NAT_TOTAL * sizeof(float));
    return L;
}

size_t cppf_point_encoder_backward_workspace_bytes(int n_points)
{
    return n_points < 1 ? 0 : spb_layout(n_points).total;
}

int cppf_point_encoder_pack_device(const float* natural, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out,
                                   int n_glob, int num_layers, float* packed, void* stream)
{
    if (!natural || !packed) return CPPF_EINVAL;
    if (!spb_std(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers)) return CPPF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(packed, natural, (size_t)NAT_TOTAL * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    spb_pack_kernel<<<(SPW_FLOATS + 255) / 256, 256, 0, st>>>(natural, packed + NAT_TOTAL, nullptr);
    return (int)hipGetLastError();
}

int cppf_point_encoder_backward(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k, const float* packed,
                                const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob,
                                int num_layers, const float* out_fwd, const float* contraction, const float* grad_out,
                                float* grad_packed, void* workspace, size_t workspace_bytes, void* stream)
{
    if (n_points < 0 || k <= 0) return CPPF_EINVAL;
    if (!spb_std(hidden, n_hidden, rank, n_nbr_feats, n_out, n_glob, num_layers) || k > 64 || (n_points > 0 && k > n_points))
        return CPPF_EUNSUPPORTED;
    if (!grad_packed) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_points == 0) return (int)hipMemsetAsync(grad_packed, 0, (size_t)NAT_TOTAL * sizeof(float), st);
    if (!pc || !nrm || !nbrs || !packed || !out_fwd || !grad_out) return CPPF_EINVAL;
    const SpbLayout L = spb_layout(n_points);
    if (!workspace || workspace_bytes < L.total) return CPPF_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    float* timg = (float*)(ws + L.timg);
    float* share = (float*)(ws + L.small);
    float* pooled = share + 8;
    int* cnt = (int*)(share + 16);
    float* chunk_sums = (float*)(ws + L.chunks);
    float* parts = (float*)(ws + L.parts);
    const int n_parts = spb_parts(n_points);
    hipError_t e = hipMemsetAsync(cnt, 0, 8 * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    spb_pack_kernel<<<(SPW_FLOATS + 255) / 256, 256, 0, st>>>(packed, nullptr, timg);
    spb_pool_chunk_kernel<<<L.n_chunks, 64, 0, st>>>(out_fwd, grad_out, n_points, packed, chunk_sums, cnt);
    spb_pool_final_kernel<<<1, 64, 0, st>>>(chunk_sums, L.n_chunks, cnt, out_fwd, share, pooled);
    SpbArgs A{pc, nrm, nbrs, packed, packed + NAT_TOTAL, timg, grad_out, share, pooled, contraction, parts, n_points, k, n_parts};
    static bool attr_done = false;
    if (!attr_done) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sprin_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SPB_LDS_FLOATS * sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    sprin_bwd_kernel<<<(n_parts + SPB_WAVES - 1) / SPB_WAVES, SPB_WAVES * 64, SPB_LDS_FLOATS * sizeof(float), st>>>(A);
    spb_reduce_kernel<<<(NAT_TOTAL + 63) / 64, 256, 0, st>>>(parts, n_parts, NAT_TOTAL, grad_packed);
    return (int)hipGetLastError();
}

}  // extern "C"
