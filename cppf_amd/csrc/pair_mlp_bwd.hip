// Backward of PPFEncoder.forward_with_idx for gfx950 (C ABI in include/cppf.h).
//
// The reference has no backward code: train.py:91 calls loss.backward() and autograd differentiates
// models/model.py:117-137.  This file computes the same gradients -- every parameter of the three ResLayers and the
// final linear, and d/d(feat) -- for the standard shape ppffcs = [84, 32, 32, 16] (train.py:35) and any out_dim, on
// the fp32 matrix cores, recomputing the forward (nothing but the inputs is saved).
//
// Three observations shape it (MI355X-first, not a transcription of autograd's op list):
//   1. forward and backward-data are the transposed MFMA chains of pair_mlp.hip: D[feature][pair] of one product is
//      already the B operand of the next, so a 16-pair block runs forward (72 MFMAs) and back (104 MFMAs, transposed
//      weights as A operands) with no data movement;
//   2. a weight gradient contracts over PAIRS, i.e. needs pairs on the k axis of both operands: deltas and inputs go
//      through LDS once as [pair][16 features] blocks (written 16 bytes per lane, read back one float per lane at an
//      address linear in the lane) and a 16 x 16 gradient tile accumulates in registers over every pair the workgroup
//      ever sees.  The 30 tiles are cut into four phases of eight staged blocks (32 KB), two tiles per wavefront per
//      phase, written out once at the end.  The upstream gradient is staged from the registers the backward-data chain
//      already holds it in -- it is read from memory exactly once;
//   3. layer 0 is linear in cat(feat[a], feat[b], ppf), so everything that touches the 80 feature columns moves from
//      pairs to POINTS: the kernel emits one 64-float row [d(h0) | d(x1)] per pair, the rows are summed per point and
//      role (a / b) in a fixed order, and d(feat) and the 2 x 32 x 80 feature columns of d(fc1.weight), d(fc0.weight)
//      are small per-point products of those sums (49x fewer rows than pairs at K = 128... and no [P,80] buffer).
// Every sum has a fixed order, restated in oracle/backward_oracle.c: all results are bit-identical to the oracle.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "../../include/cppf.h"
#include "cppf_math.h"
#include "pair_layout.h"

using namespace cppf;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BW_F = 40, BW_D0 = 84, BW_D1 = 32, BW_D2 = 32, BW_D3 = 16;
constexpr int BW_THREADS = 256;              // 4 wavefronts x 16 pairs = one tile of 64 pairs
constexpr int BW_ROW = 64;                   // per-pair row [d(h0) (32) | d(x1) (32)]
constexpr int BW_BLK = 64 * 16;              // one staged block: [64 pairs][16 features]
// LDS image: forward weights [0, OFF_WF) | hidden biases [OFF_B0B, OFF_BF) | transposed weights | 8 staging blocks
constexpr int L_BIAS = OFF_WF;
constexpr int L_BWD = L_BIAS + (OFF_BF - OFF_B0B);
constexpr int L_STG = L_BWD + STD_BWD_FLOATS;
constexpr int BW_LDS_FLOATS = L_STG + 8 * BW_BLK;   // 19 600 floats = 78 400 B: two workgroups per CU
constexpr int FW_CHUNK = 64;                 // points per chunk of the feature-column weight gradients

struct BwdArgs {
    const float* pc;
    const float* nrm;
    const void* idxs;
    const float* packed;   // device image (pair_layout.h)
    const float* table;    // [N][128] layer-0 projections of every point
    const float* grad_out;
    float* parts;          // [n_parts][n_params]
    float* rows;           // [P][64]
    int64_t P;
    int64_t n_params;
    const int64_t* offs;   // device copy of the 20 parameter offsets (written behind the image by bwd_pack_kernel)
    int out_dim, idx64, n_parts;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ldb4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 relu4(f32x4 v)
{
    f32x4 r;
    r[0] = v[0] > 0.f ? v[0] : 0.f; r[1] = v[1] > 0.f ? v[1] : 0.f;
    r[2] = v[2] > 0.f ? v[2] : 0.f; r[3] = v[3] > 0.f ? v[3] : 0.f;
    return r;
}
__device__ __forceinline__ f32x4 mask4(f32x4 d, f32x4 h)   // ReLU'(pre) = (relu(pre) > 0)
{
    f32x4 r;
    r[0] = h[0] > 0.f ? d[0] : 0.f; r[1] = h[1] > 0.f ? d[1] : 0.f;
    r[2] = h[2] > 0.f ? d[2] : 0.f; r[3] = h[3] > 0.f ? d[3] : 0.f;
    return r;
}
// PPF component `g` of one pair from its loaded points / normals (models/model.py:118-129; fp32 `+ 1e-7`, true divisions)
__device__ __forceinline__ float ppf_lane(f3 pa, f3 pb, f3 na, f3 nb, int g)
{
    const f3 xy = sub3(pa, pb);
    const float d = sqrtf((xy.x * xy.x + xy.y * xy.y) + xy.z * xy.z);
    const float den = d + 1e-7f;
    const f3 u = {xy.x / den, xy.y / den, xy.z / den};
    const float p0 = (na.x * u.x + na.y * u.y) + na.z * u.z;
    const float p1 = (nb.x * u.x + nb.y * u.y) + nb.z * u.z;
    const float p2 = (na.x * nb.x + na.y * nb.y) + na.z * nb.z;
    return g == 0 ? p0 : (g == 1 ? p1 : (g == 2 ? p2 : d));
}

// Layer-0 projections of every point (the table of pair_mlp.hip:point_proj_kernel, same arithmetic)
__global__ __launch_bounds__(256) void bwd_point_proj_kernel(const float* __restrict__ feat, const float* __restrict__ packed,
                                                             float* __restrict__ T, int64_t N)
{
    __shared__ float f[2][STD_F];
    const int half = threadIdx.x >> 7, r = threadIdx.x & 127;
    const int64_t n = (int64_t)blockIdx.x * 2 + half;
    if (r < STD_F && n < N) f[half][r] = feat[n * STD_F + r];
    __syncthreads();
    if (n >= N) return;
    float acc = r < 64 ? packed[OFF_BPT + r] : 0.f;
#pragma unroll 8
    for (int k = 0; k < STD_F; ++k) acc = fmaf(packed[OFF_WPT + k * PROJ_COLS + r], f[half][k], acc);
    T[n * PROJ_COLS + r] = acc;
}

struct PackOffs { int64_t o[20]; };
__global__ __launch_bounds__(256) void bwd_pack_kernel(const float* __restrict__ params, PackOffs offs, int out_dim,
                                                       float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < STD_PACKED) out[i] = std_pack_elem(i, params, offs.o, out_dim);
    if (i < 20) reinterpret_cast<int64_t*>(out + STD_PACKED)[i] = offs.o[i];   // the epilogue reads them from memory (40 fewer SGPRs)
}

// two gradient tiles over the 64 staged pairs: c_t += A_t^T B_t (k = pairs, ascending); s_t += the A operand (bias sums).
// stg0 / stg1: the lane's read bases for even / odd steps (swizzled staging, see `my` in the kernel).  The operands of the
// next four steps are fetched before the current four steps' MFMAs are issued.
__device__ __forceinline__ void wgrad2(const float* __restrict__ stg0, const float* __restrict__ stg1, int a0, int a1, int b0,
                                       int b1, f32x4& c0, f32x4& c1, float& s0, float& s1)
{
    // scalar block offsets, opaque per call: otherwise the 8 lane addresses of every phase are hoisted out of the tile
    // loop as invariants (32 registers) and spilled
    a0 *= BW_BLK; a1 *= BW_BLK; b0 *= BW_BLK; b1 *= BW_BLK;
    asm volatile("" : "+s"(a0), "+s"(a1), "+s"(b0), "+s"(b1));
    float v[2][4][4];
    auto fetch = [&](int buf, int s4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* base = ((s & 1) ? stg1 : stg0) + 64 * (s4 + s);
            v[buf][s][0] = base[a0]; v[buf][s][1] = base[a1];
            v[buf][s][2] = base[b0]; v[buf][s][3] = base[b1];
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b < 3) fetch((b + 1) & 1, 4 * b + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0 = mfma4(v[b & 1][s][0], v[b & 1][s][2], c0);
            c1 = mfma4(v[b & 1][s][1], v[b & 1][s][3], c1);
            s0 = s0 + v[b & 1][s][0];
            s1 = s1 + v[b & 1][s][1];
        }
    }
}

// OD_T: out_dim at compile time (141 = train.py's heads: the block tests below fold away), 0 = read it from the arguments
template <int OD_T>
__global__ __launch_bounds__(BW_THREADS, 2) void pair_mlp_bwd_kernel(BwdArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float W[];
    {   // weights: forward part, hidden biases, transposed part
        const f32x4* src = reinterpret_cast<const f32x4*>(A.packed);
        f32x4* dst = reinterpret_cast<f32x4*>(W);
        for (int k = threadIdx.x; k < L_STG / 4; k += BW_THREADS) {
            int from = k;
            if (k >= L_BWD / 4) from = OFF_T0B / 4 + (k - L_BWD / 4);
            else if (k >= L_BIAS / 4) from = OFF_B0B / 4 + (k - L_BIAS / 4);
            dst[k] = src[from];
        }
    }
    const float* const Bias = W + L_BIAS - OFF_B0B;   // Bias + OFF_B.. addresses a hidden bias
    const float* const Tw = W + L_BWD - OFF_T0B;      // Tw + OFF_T.. addresses a transposed weight block
    float* const stg = W + L_STG;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: block choices below are SALU work
    const int j = lane & 15, g = lane >> 4;
    __builtin_assume(g >= 0 && g < 4);
    const int OD = OD_T ? OD_T : A.out_dim;
    const int ob_full = OD >> 4;                       // output blocks that lie entirely inside out_dim
    const int64_t n_tiles = (A.P + 63) / 64;

    // Gradient tiles of this wavefront -- two per phase (G, A, B, C), see the table at the epilogue -- and the bias
    // sub-sums of lane (m = j, kk = g)
    f32x4 acc[4][2];
    float bs[4][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        acc[ph][0] = acc[ph][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        bs[ph][0] = bs[ph][1] = 0.f;
    }
    __syncthreads();

    // Inputs of a tile are loaded AHEAD and kept RAW in registers (any arithmetic on them would make the wavefront wait
    // for the load right there): the upstream gradient (cold HBM lines) while the previous tile's phases B and C run,
    // the L2-resident gathers (table rows, points, normals) during its phase C, the pair indices two tiles ahead.
    // Without this every workgroup on the chip starts a tile with the same burst of cold loads and sits through it
    // (measured with s_memtime: 11-14k cycles from tile start to the end of the forward, 4k with the prefetch).
    // (two 4-byte loads of the low words, no i32 / i64 branch: a branch would end in a wait for ALL outstanding loads)
    const int istr = A.idx64 ? 4 : 2;
    auto load_idx = [&](int64_t tile, int& ia, int& ib) {
        int64_t p = tile * 64 + wave * 16 + j;
        p = p < A.P ? p : A.P - 1;
        const int* q = reinterpret_cast<const int*>(A.idxs) + p * istr;
        ia = q[0];
        ib = q[istr >> 1];
    };
    f32x4 gR[STD_NOB], vR;   // raw: gR[ob] = g[16*ob + 4*g ..] for whole blocks, vR = the ragged block (clamped columns)
    f32x4 tA[4], tB[4];      // raw table rows TA[a], TB[b]
    int ia_c, ib_c;          // the current tile's pair (its points and normals are L1 / L2 hits, loaded at the tile's top:
                             // three-float loads carried across the loop edge get copied, and a copy is a wait)
    auto load_g = [&](int64_t tile) {
        int gg = g;
        asm volatile("" : "+v"(gg));   // recomputed per tile: as loop invariants the column offsets below get hoisted and spilled
        const int64_t p = tile * 64 + wave * 16 + j;
        const float* gp = A.grad_out + (p < A.P ? p : A.P - 1) * OD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * ob_full + 4 * gg + r;
            vR[r] = gp[c < OD ? c : OD - 1];
        }
#pragma unroll
        for (int ob = 0; ob < STD_NOB; ++ob)
            if (ob < ob_full) gR[ob] = *reinterpret_cast<const f32x4u*>(gp + 16 * ob + 4 * gg);   // wave-uniform test
    };
    auto load_gathers = [&](int ia, int ib) {
        int gg = g;
        asm volatile("" : "+v"(gg));
        const f32x4* pa = reinterpret_cast<const f32x4*>(A.table + (int64_t)ia * PROJ_COLS + 4 * gg);
        const f32x4* pb = reinterpret_cast<const f32x4*>(A.table + (int64_t)ib * PROJ_COLS + 64 + 4 * gg);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) { tA[ob] = pa[4 * ob]; tB[ob] = pb[4 * ob]; }
        ia_c = ia; ib_c = ib;
    };
    int ia_n, ib_n;
    {
        int ia0, ib0;
        load_idx(blockIdx.x, ia0, ib0);
        const int64_t t1 = (int64_t)blockIdx.x + A.n_parts;
        load_idx(t1 < n_tiles ? t1 : blockIdx.x, ia_n, ib_n);
        load_g(blockIdx.x);
        load_gathers(ia0, ib0);
    }

    // Two wavefronts per SIMD cannot hide LDS latency by themselves, so every block of MFMAs has its A operands
    // (weights, staged deltas) fetched one block AHEAD: "issue the next block's ds_reads | sched_barrier | this block's
    // MFMAs".  SB() pins that order (left alone, the scheduler sinks each read to just before its use and the
    // wavefront eats the full LDS round trip every four MFMAs -- measured: 41 % MFMA utilisation).
#define SB() __builtin_amdgcn_sched_barrier(0)
    auto ld8 = [&](const float* base, f32x2 (&w)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) w[s] = *reinterpret_cast<const f32x2*>(base + (s * 64 + lane) * 2);
    };
    // staging: the lane's 16 bytes of block 0; the 16-byte chunk index is XORed with (j >> 1) & 3 so that eight
    // consecutive lanes of a ds_write_b128 (rows 64 bytes apart) cover all 32 banks
    float* const my = stg + (wave * 16 + j) * 16 + 4 * (g ^ ((j >> 1) & 3));
    auto put = [&](int blk, f32x4 v) { *reinterpret_cast<f32x4*>(my + blk * BW_BLK) = v; };
    // read side (A / B layout: lane (m = j, kk = g) wants pair 4*s + kk, feature m): word 64*s + (rd0 ^ 8*(s & 1))
    const int rd0 = 16 * g + 4 * ((j >> 2) ^ (g >> 1)) + (j & 3);
    const float* const stg0 = stg + rd0;          // even steps
    const float* const stg1 = stg + (rd0 ^ 8);    // odd steps

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += A.n_parts) {
        asm volatile("" ::: "memory");   // keep the loop-invariant LDS weights out of registers (see pair_mlp.hip)
        const int64_t p = tile * 64 + wave * 16 + j;
        const bool live = p < A.P;
        // this tile's inputs from the raw prefetched registers
        f32x4 gB[STD_NOB];   // gB[ob][r] = g[16*ob + 4*g + r] of the lane's pair (0 beyond out_dim / P)
        f32x4 pre[4];        // TA[a] + TB[b]: the layer-0 accumulators before the PPF k-step
#pragma unroll
        for (int ob = 0; ob < STD_NOB; ++ob) {
            if (ob < ob_full) gB[ob] = gR[ob];
            else if (ob == ob_full) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gB[ob][r] = 16 * ob + 4 * g + r < OD ? vR[r] : 0.f;
            } else gB[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if ((tile + 1) * 64 > A.P) {   // the ragged last tile only: pairs past the end contribute nothing
#pragma unroll
            for (int ob = 0; ob < STD_NOB; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) gB[ob][r] = live ? gB[ob][r] : 0.f;
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) pre[ob] = tA[ob] + tB[ob];
        const float xp0 = ppf_lane(ld3(A.pc, ia_c), ld3(A.pc, ib_c), ld3(A.nrm, ia_c), ld3(A.nrm, ib_c), g);   // ppf[g]
        // ---------------------------------------------------------------- forward (pair_mlp.hip, oracle order 1)
        f32x4 h0[2], x1[2], h1[2], x2[2], h2, x3;
        f32x4 dy3, dh2, dy2[2], dh1[2], dy1[2], dh0[2];
        f32x2 wA[8], wB[8];
        float t2b[4];
        f32x4 t2[4];
        {
            const f32x4 w0p = ldb4(W + OFF_W0P + lane * 4);
            ld8(W + OFF_W0B, wA);
            f32x4 a2[2] = {ldb4(Bias + OFF_B0B + 4 * g), ldb4(Bias + OFF_B0B + 16 + 4 * g)};
            SB();
            f32x4 ac[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) ac[ob] = mfma4(w0p[ob], xp0, pre[ob]);
            ld8(W + OFF_W1A, wB);
            f32x4 a1[2] = {ldb4(Bias + OFF_B1A + 4 * g), ldb4(Bias + OFF_B1A + 16 + 4 * g)};
            SB();
            h0[0] = relu4(ac[0]); h0[1] = relu4(ac[1]);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a2[0] = mfma4(wA[s][0], h0[s >> 2][s & 3], a2[0]);
                a2[1] = mfma4(wA[s][1], h0[s >> 2][s & 3], a2[1]);
            }
            x1[0] = a2[0] + ac[2]; x1[1] = a2[1] + ac[3];
            ld8(W + OFF_W1B, wA);
            a2[0] = ldb4(Bias + OFF_B1B + 4 * g); a2[1] = ldb4(Bias + OFF_B1B + 16 + 4 * g);
            SB();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a1[0] = mfma4(wB[s][0], x1[s >> 2][s & 3], a1[0]);
                a1[1] = mfma4(wB[s][1], x1[s >> 2][s & 3], a1[1]);
            }
            h1[0] = relu4(a1[0]); h1[1] = relu4(a1[1]);
            ld8(W + OFF_W2, wB);
            a1[0] = ldb4(Bias + OFF_B2 + 4 * g); a1[1] = ldb4(Bias + OFF_B2 + 16 + 4 * g);   // layer 2: fc1 | fc0
            SB();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a2[0] = mfma4(wA[s][0], h1[s >> 2][s & 3], a2[0]);
                a2[1] = mfma4(wA[s][1], h1[s >> 2][s & 3], a2[1]);
            }
            x2[0] = a2[0] + x1[0]; x2[1] = a2[1] + x1[1];
            float w2b[4], tfa[18];
#pragma unroll
            for (int s = 0; s < 4; ++s) w2b[s] = W[OFF_W2B + s * 64 + lane];
            f32x4 a3 = ldb4(Bias + OFF_B2B + 4 * g);
            SB();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a1[0] = mfma4(wB[s][0], x2[s >> 2][s & 3], a1[0]);
                a1[1] = mfma4(wB[s][1], x2[s >> 2][s & 3], a1[1]);
            }
            h2 = relu4(a1[0]);
            // d(x3) = Wf^T g over the 144 padded outputs: two independent half chains (k-steps 0..17 | 18..35), then added;
            // the A operands come in two groups: steps {0..8, 18..26}, then {9..17, 27..35}
#pragma unroll
            for (int s = 0; s < 9; ++s) { tfa[s] = Tw[OFF_TF + s * 64 + lane]; tfa[9 + s] = Tw[OFF_TF + (18 + s) * 64 + lane]; }
            SB();
#pragma unroll
            for (int s = 0; s < 4; ++s) a3 = mfma4(w2b[s], h2[s], a3);
            x3 = a3 + a1[1];
            // ------------------------------------------------------------ backward-data (transposed chains, seeds 0)
            float tfb[18];
#pragma unroll
            for (int s = 0; s < 9; ++s) { tfb[s] = Tw[OFF_TF + (9 + s) * 64 + lane]; tfb[9 + s] = Tw[OFF_TF + (27 + s) * 64 + lane]; }
            SB();
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                c0 = mfma4(tfa[s], gB[s >> 2][s & 3], c0);
                c1 = mfma4(tfa[9 + s], gB[(s + 18) >> 2][(s + 18) & 3], c1);
            }
            SB();
#pragma unroll
            for (int s = 9; s < 18; ++s) {
                c0 = mfma4(tfb[s - 9], gB[s >> 2][s & 3], c0);
                c1 = mfma4(tfb[s], gB[(s + 18) >> 2][(s + 18) & 3], c1);
            }
            dy3 = c0 + c1;
        }
        // ---------------------------------------------------------------- phase G: final layer, output blocks 0..6
        //   staged: 0 x3 | 1 + k: g block k.  The upstream gradient is already in the staging layout (gB), so it is
        //   never read from memory a second time.  Wavefront v: blocks 2v, 2v + 1 (v = 3: block 6; its second tile idles)
        __syncthreads();                               // the previous tile's phase C has been read
        put(0, x3);
#pragma unroll
        for (int k = 0; k < 7; ++k) put(1 + k, gB[k]);
        __syncthreads();
        {
            const int k0 = 2 * wave, k1 = wave < 3 ? 2 * wave + 1 : 6;
            wgrad2(stg0, stg1, 1 + k0, 1 + k1, 0, 0, acc[0][0], acc[0][1], bs[0][0], bs[0][1]);
        }
        // ---------------------------------------------------------------- backward-data, continued
        {
#pragma unroll
            for (int s = 0; s < 4; ++s) { t2b[s] = Tw[OFF_T2B + s * 64 + lane]; t2[s] = ldb4(Tw + OFF_T2 + (s * 64 + lane) * 4); }
            ld8(Tw + OFF_T1B, wA);
            SB();
            {   // layer 2: d(h2) = relu'(.) fc2^T d(x3);  d(x2) = fc0^T d(x3) then fc1^T d(h2) on the same accumulators
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                dy2[0] = dy2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    c = mfma4(t2b[s], dy3[s], c);
                    dy2[0] = mfma4(t2[s][0], dy3[s], dy2[0]);
                    dy2[1] = mfma4(t2[s][1], dy3[s], dy2[1]);
                }
                dh2 = mask4(c, h2);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    dy2[0] = mfma4(t2[s][2], dh2[s], dy2[0]);
                    dy2[1] = mfma4(t2[s][3], dh2[s], dy2[1]);
                }
            }
            ld8(Tw + OFF_T1A, wB);
            SB();
            {   // layer 1: d(h1) = relu'(.) fc2^T d(x2);  d(x1) = d(x2) (identity skip, the seed) + fc1^T d(h1)
                f32x4 c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    c[0] = mfma4(wA[s][0], dy2[s >> 2][s & 3], c[0]);
                    c[1] = mfma4(wA[s][1], dy2[s >> 2][s & 3], c[1]);
                }
                dh1[0] = mask4(c[0], h1[0]); dh1[1] = mask4(c[1], h1[1]);
                ld8(Tw + OFF_T0B, wA);
                SB();
                dy1[0] = dy2[0]; dy1[1] = dy2[1];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    dy1[0] = mfma4(wB[s][0], dh1[s >> 2][s & 3], dy1[0]);
                    dy1[1] = mfma4(wB[s][1], dh1[s >> 2][s & 3], dy1[1]);
                }
            }
            {   // layer 0: d(h0) = relu'(.) fc2^T d(x1); the row [d(h0) | d(x1)] is all the point-level kernels need
                f32x4 c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    c[0] = mfma4(wA[s][0], dy1[s >> 2][s & 3], c[0]);
                    c[1] = mfma4(wA[s][1], dy1[s >> 2][s & 3], c[1]);
                }
                dh0[0] = mask4(c[0], h0[0]); dh0[1] = mask4(c[1], h0[1]);
            }
            if (live) {
                float* r = A.rows + p * BW_ROW + 4 * g;
                *reinterpret_cast<f32x4*>(r) = dh0[0];
                *reinterpret_cast<f32x4*>(r + 16) = dh0[1];
                *reinterpret_cast<f32x4*>(r + 32) = dy1[0];
                *reinterpret_cast<f32x4*>(r + 48) = dy1[1];
            }
        }
        // ---------------------------------------------------------------- phase A: final layer blocks 7, 8 and layer 2
        //   staged: 0 x3 (still there) | 1 h2 | 2,3 x2 | 4 d(x3) | 5 d(h2) | 6,7 g blocks 7, 8
        //   wavefront 0: g7 x x3, g8 x x3 | 1: d(x3) x h2, d(x3) x x2[0] | 2: d(x3) x x2[1], d(h2) x x2[0] | 3: d(h2) x x2[1], idle
        __syncthreads();                               // phase G has been read
        put(1, h2); put(2, x2[0]); put(3, x2[1]); put(4, dy3); put(5, dh2); put(6, gB[7]); put(7, gB[8]);
        __syncthreads();
        {
            const int a0 = wave == 0 ? 6 : (wave == 3 ? 5 : 4), a1 = wave == 0 ? 7 : (wave == 1 ? 4 : 5);
            const int b0 = wave == 0 ? 0 : (wave == 1 ? 1 : 3), b1 = wave == 0 ? 0 : (wave == 3 ? 3 : 2);
            wgrad2(stg0, stg1, a0, a1, b0, b1, acc[1][0], acc[1][1], bs[1][0], bs[1][1]);
        }
        {   // the next tile's upstream gradient: in flight during phases B and C
            const int64_t t1 = tile + A.n_parts;
            load_g(t1 < n_tiles ? t1 : tile);
        }
        __syncthreads();
        // ---------------------------------------------------------------- phase B: layer 1
        //   staged: 0,1 d(x2) | 2,3 d(h1) | 4,5 h1 | 6,7 x1;  wavefronts 0, 1: d(x2)[v] x h1;  2, 3: d(h1)[v - 2] x x1
        put(0, dy2[0]); put(1, dy2[1]); put(2, dh1[0]); put(3, dh1[1]); put(4, h1[0]); put(5, h1[1]); put(6, x1[0]); put(7, x1[1]);
        __syncthreads();
        {
            const int a = wave, b = wave < 2 ? 4 : 6;
            wgrad2(stg0, stg1, a, a, b, b + 1, acc[2][0], acc[2][1], bs[2][0], bs[2][1]);
        }
        __syncthreads();
        // ---------------------------------------------------------------- phase C: layer 0
        //   staged: 0,1 d(x1) | 2,3 d(h0) | 4,5 h0 | 6 [ppf (4) | zeros]
        //   wavefronts 0, 1: d(x1)[v] x h0 | 2: d(h0)[0,1] x ppf | 3: d(x1)[0,1] x ppf
        put(0, dy1[0]); put(1, dy1[1]); put(2, dh0[0]); put(3, dh0[1]); put(4, h0[0]); put(5, h0[1]);
        put(6, f32x4{0.f, 0.f, 0.f, 0.f});
        stg[6 * BW_BLK + (wave * 16 + j) * 16 + 4 * ((j >> 1) & 3) + g] = xp0;   // chunk 0 swizzled; same wavefront, LDS is in order: lands after the zeros
        {   // the next tile's gathers (its indices are here): in flight during phase C; and the indices of the tile after it
            load_gathers(ia_n, ib_n);
            const int64_t t2i = tile + 2 * (int64_t)A.n_parts;
            load_idx(t2i < n_tiles ? t2i : tile, ia_n, ib_n);
        }
        __syncthreads();
        {
            const int a0 = wave < 2 ? wave : (wave == 2 ? 2 : 0), a1 = wave < 2 ? wave : (wave == 2 ? 3 : 1);
            const int b0 = wave < 2 ? 4 : 6, b1 = wave < 2 ? 5 : 6;
            wgrad2(stg0, stg1, a0, a1, b0, b1, acc[3][0], acc[3][1], bs[3][0], bs[3][1]);
        }
    }
#undef SB

    // ---------------------------------------------------------------- epilogue: this workgroup's partial gradients
    // tile D: register r of lane (j, g) = d(W)[row 16*rb + 4*g + r][col 16*cb + j]; bias sub-sums: lane (m = j, kk = g),
    // total = ((s0 + s1) + s2) + s3
    float* part = A.parts + (size_t)blockIdx.x * A.n_params;
    auto put_tile = [&](int64_t offW, int ld, int rb, int cb, int rows, int cols, f32x4 v) {
        const int c = 16 * cb + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * rb + 4 * g + r;
            if (o < rows && c < cols) part[offW + (int64_t)o * ld + c] = v[r];
        }
    };
    auto bias_total = [&](float v) {
        const float s0 = __shfl(v, j), s1 = __shfl(v, j + 16), s2 = __shfl(v, j + 32), s3 = __shfl(v, j + 48);
        return ((s0 + s1) + s2) + s3;
    };
    auto put_bias = [&](int64_t offB, int rb, int rows, float v) {
        const float t = bias_total(v);
        if (g == 0 && 16 * rb + j < rows) part[offB + 16 * rb + j] = t;
    };
    auto put_ppf_cols = [&](int64_t offW, int rb, f32x4 v) {   // layer 0, PPF columns 80..83 = tile columns 0..3
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (j < 4) part[offW + (int64_t)(16 * rb + 4 * g + r) * BW_D0 + 2 * BW_F + j] = v[r];
    };
    const int64_t* __restrict__ O = A.offs;
    // phase G: final.weight / final.bias blocks 2v, 2v + 1 (v = 3: block 6 only)
    put_tile(O[18], BW_D3, 2 * wave, 0, OD, BW_D3, acc[0][0]); put_bias(O[19], 2 * wave, OD, bs[0][0]);
    {
        const float t = bias_total(bs[0][1]);
        if (wave < 3) {
            put_tile(O[18], BW_D3, 2 * wave + 1, 0, OD, BW_D3, acc[0][1]);
            if (g == 0 && 16 * (2 * wave + 1) + j < OD) part[O[19] + 16 * (2 * wave + 1) + j] = t;
        }
    }
    // phases B and C: layer 1 (fc2 rows 16v | fc1 rows 16(v-2)), layer 0 (fc2 rows 16v | PPF columns of fc1 | of fc0)
    {
        const float tB = bias_total(bs[2][0]), tC0 = bias_total(bs[3][0]), tC1 = bias_total(bs[3][1]);
        const float tA0 = bias_total(bs[1][0]), tA1 = bias_total(bs[1][1]);
        const bool lead = g == 0;
        if (wave < 2) {
            put_tile(O[8], BW_D2, wave, 0, BW_D2, BW_D2, acc[2][0]); put_tile(O[8], BW_D2, wave, 1, BW_D2, BW_D2, acc[2][1]);
            if (lead) part[O[9] + 16 * wave + j] = tB;
            put_tile(O[2], BW_D1, wave, 0, BW_D1, BW_D1, acc[3][0]); put_tile(O[2], BW_D1, wave, 1, BW_D1, BW_D1, acc[3][1]);
            if (lead) { part[O[3] + 16 * wave + j] = tC0; part[O[5] + 16 * wave + j] = tC0; }   // fc2.bias and fc0.bias: both sum d(x1)
        } else {
            put_tile(O[6], BW_D1, wave - 2, 0, BW_D2, BW_D1, acc[2][0]); put_tile(O[6], BW_D1, wave - 2, 1, BW_D2, BW_D1, acc[2][1]);
            if (lead) part[O[7] + 16 * (wave - 2) + j] = tB;
            put_ppf_cols(wave == 2 ? O[0] : O[4], 0, acc[3][0]); put_ppf_cols(wave == 2 ? O[0] : O[4], 1, acc[3][1]);
            if (wave == 2 && lead) { part[O[1] + j] = tC0; part[O[1] + 16 + j] = tC1; }        // fc1.bias: d(h0)
        }
        // phase A
        if (wave == 0) {
            put_tile(O[18], BW_D3, 7, 0, OD, BW_D3, acc[1][0]); put_tile(O[18], BW_D3, 8, 0, OD, BW_D3, acc[1][1]);
            if (lead && 112 + j < OD) part[O[19] + 112 + j] = tA0;
            if (lead && 128 + j < OD) part[O[19] + 128 + j] = tA1;
        } else if (wave == 1) {
            put_tile(O[14], BW_D3, 0, 0, BW_D3, BW_D3, acc[1][0]);                              // layer 2 fc2: d(x3) x h2
            put_tile(O[16], BW_D2, 0, 0, BW_D3, BW_D2, acc[1][1]);                              // layer 2 fc0 columns 0..15: d(x3) x x2[0]
            if (lead) { part[O[15] + j] = tA0; part[O[17] + j] = tA0; }                        // fc2.bias, fc0.bias: d(x3)
        } else if (wave == 2) {
            put_tile(O[16], BW_D2, 0, 1, BW_D3, BW_D2, acc[1][0]);                              // layer 2 fc0 columns 16..31
            put_tile(O[12], BW_D2, 0, 0, BW_D3, BW_D2, acc[1][1]);                              // layer 2 fc1 columns 0..15: d(h2) x x2[0]
            if (lead) part[O[13] + j] = tA1;                                                    // fc1.bias: d(h2)
        } else {
            put_tile(O[12], BW_D2, 0, 1, BW_D3, BW_D2, acc[1][0]);                              // layer 2 fc1 columns 16..31
        }
    }
}

// grad[q] = sum over groups of 32 consecutive partials (ascending) of the group's sum (ascending): a fixed two-level order
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

// ---- per-point sums of the pair rows, in a fixed order --------------------------------------------------------------
// entry e in [0, 2P): e < P is pair e in role a (key idx[e][0]), e >= P is pair e - P in role b (key idx[e-P][1]).
// Keys are uint16 when the cloud has at most 65 536 points: rocprim then takes its one-sweep radix path (two 8-bit
// passes here) instead of a 9-pass merge sort -- both are stable, which is all the fixed summation order needs.
template <typename KT>
__global__ __launch_bounds__(256) void bwd_keys_kernel(const void* __restrict__ idxs, int idx64, int64_t P, KT* __restrict__ keys,
                                                       int32_t* __restrict__ vals)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * P) return;
    const int64_t p = e < P ? e : e - P;
    const int half = e < P ? 0 : 1;
    keys[e] = (KT)(idx64 ? reinterpret_cast<const int64_t*>(idxs)[2 * p + half]
                         : (int64_t)reinterpret_cast<const int32_t*>(idxs)[2 * p + half]);
    vals[e] = (int32_t)e;
}
// seg[n] = first sorted position with key >= n (seg[N] = 2P)
template <typename KT>
__global__ __launch_bounds__(256) void bwd_seg_kernel(const KT* __restrict__ skeys, int64_t M, int64_t N, int32_t* __restrict__ seg)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > M) return;
    const int64_t lo = i == 0 ? 0 : (int64_t)skeys[i - 1] + 1;
    const int64_t hi = i == M ? N : (int64_t)skeys[i];
    for (int64_t n = lo; n <= hi && n <= N; ++n) seg[n] = (int32_t)i;   // (every n in (key[i-1], key[i]] starts at i)
}
// keys, vals, sorted keys, sorted vals live in `kbuf` (4 x M x 4 bytes reserved); returns the sorted entry ids
template <typename KT>
int sort_entries(const void* idxs, int idx64, int64_t P, int64_t N, char* kbuf, void* temp, size_t temp_bytes, int32_t* seg,
                 const int32_t** svals_out, hipStream_t st)
{
    const int64_t M = 2 * P;
    int32_t* vals = reinterpret_cast<int32_t*>(kbuf);
    int32_t* svals = vals + M;
    KT* keys = reinterpret_cast<KT*>(svals + M);
    KT* skeys = keys + M;
    bwd_keys_kernel<KT><<<(int)((M + 255) / 256), 256, 0, st>>>(idxs, idx64, P, keys, vals);
    int end_bit = 1;
    while (end_bit < (int)(8 * sizeof(KT)) - (sizeof(KT) == 4 ? 1 : 0) && (1ll << end_bit) < N) ++end_bit;
    size_t need = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, skeys, vals, svals, (int)M, 0, end_bit, st);
    if (e != hipSuccess) return (int)e;
    if (need > temp_bytes) return CPPF_EWORKSPACE;
    need = temp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(temp, need, keys, skeys, vals, svals, (int)M, 0, end_bit, st);
    if (e != hipSuccess) return (int)e;
    bwd_seg_kernel<KT><<<(int)((M + 256) / 256), 256, 0, st>>>(skeys, M, N, seg);
    *svals_out = svals;
    return 0;
}
// One wavefront per point n, lanes = the 64 row columns:
//   S[n][c] = rows of the pairs with a == n summed in pair order, S[n][64 + c] = the same for b == n   (stable sort order)
//   grad_feat[n][k] += chain over c = 0..63 of fmaf(Wa[c][k], S[n][c], .), continued over fmaf(Wb[c][k], S[n][64 + c], .)
// with Wa[c] = fc1.weight[c][0:40] (c < 32) | fc0.weight[c - 32][0:40], Wb the same rows, columns 40:80.
__global__ __launch_bounds__(256) void bwd_point_kernel(const float* __restrict__ rows, const int32_t* __restrict__ svals,
                                                        const int32_t* __restrict__ seg, int64_t P, int64_t N,
                                                        const float* __restrict__ w1, const float* __restrict__ w0,
                                                        float* __restrict__ S, float* __restrict__ grad_feat)
{
    __shared__ float Ssh[4][2 * BW_ROW];
    const int wv = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + wv;
    if (n >= N) return;
    const int b = seg[n], e_ = seg[n + 1];
    float sa = 0.f, sb = 0.f;
    int i = b;
    for (; i + 8 <= e_; i += 8) {   // eight independent row loads in flight, added in order
        float v[8];
        int e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            e[u] = svals[i + u];
            v[u] = rows[(size_t)(e[u] < P ? e[u] : e[u] - P) * BW_ROW + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (e[u] < P) sa = sa + v[u]; else sb = sb + v[u];
        }
    }
    for (; i < e_; ++i) {
        const int e = svals[i];
        const float v = rows[(size_t)(e < P ? e : e - P) * BW_ROW + c];
        if (e < P) sa = sa + v; else sb = sb + v;
    }
    S[n * (2 * BW_ROW) + c] = sa;
    S[n * (2 * BW_ROW) + BW_ROW + c] = sb;
    Ssh[wv][c] = sa;
    Ssh[wv][BW_ROW + c] = sb;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wavefront wrote it
    if (c < BW_F) {
        float acc = 0.f;
#pragma unroll 8
        for (int q = 0; q < BW_ROW; ++q) acc = fmaf((q < 32 ? w1 + q * BW_D0 : w0 + (q - 32) * BW_D0)[c], Ssh[wv][q], acc);
#pragma unroll 8
        for (int q = 0; q < BW_ROW; ++q) acc = fmaf((q < 32 ? w1 + q * BW_D0 : w0 + (q - 32) * BW_D0)[BW_F + c], Ssh[wv][BW_ROW + q], acc);
        grad_feat[n * BW_F + c] = grad_feat[n * BW_F + c] + acc;
    }
}
// Feature columns of d(fc1.weight), d(fc0.weight) of layer 0:  G[r][k] = sum_n S[n][r] feat[n][k]  (r < 128, k < 40).
// One workgroup per chunk of 64 points: chunk[r][k] = fmaf chain over its points in ascending order.
__global__ __launch_bounds__(256) void bwd_featw_kernel(const float* __restrict__ S, const float* __restrict__ feat, int64_t N,
                                                        float* __restrict__ chunks)
{
    __shared__ float Ssh[FW_CHUNK][2 * BW_ROW];
    __shared__ float Fsh[FW_CHUNK][BW_F];
    const int64_t n0 = (int64_t)blockIdx.x * FW_CHUNK;
    const int cnt = (int)min((int64_t)FW_CHUNK, N - n0);
    for (int i = threadIdx.x; i < cnt * 2 * BW_ROW; i += 256) Ssh[i >> 7][i & 127] = S[n0 * (2 * BW_ROW) + i];
    for (int i = threadIdx.x; i < cnt * BW_F; i += 256) Fsh[i / BW_F][i % BW_F] = feat[n0 * BW_F + i];
    __syncthreads();
    const int r = threadIdx.x >> 1, k0 = 20 * (threadIdx.x & 1);
    float acc[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) acc[q] = 0.f;
    for (int n = 0; n < cnt; ++n) {
        const float s = Ssh[n][r];
#pragma unroll
        for (int q = 0; q < 20; ++q) acc[q] = fmaf(s, Fsh[n][k0 + q], acc[q]);
    }
    float* out = chunks + (size_t)blockIdx.x * (2 * BW_ROW * BW_F) + r * BW_F + k0;
#pragma unroll
    for (int q = 0; q < 20; ++q) out[q] = acc[q];
}
// total over the chunks in ascending order, written to the feature columns of the two layer-0 weight gradients
__global__ __launch_bounds__(256) void bwd_featw_reduce_kernel(const float* __restrict__ chunks, int n_chunks, int64_t off_w1,
                                                               int64_t off_w0, float* __restrict__ grad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * BW_ROW * BW_F) return;
    float acc = 0.f;
#pragma unroll 4
    for (int c = 0; c < n_chunks; ++c) acc = acc + chunks[(size_t)c * (2 * BW_ROW * BW_F) + i];
    const int r = i / BW_F, k = i % BW_F;
    const int role = r >> 6, o = r & 63;        // role 0: columns 0..39 (feat[a]), role 1: columns 40..79 (feat[b])
    grad[(o < 32 ? off_w1 + (int64_t)o * BW_D0 : off_w0 + (int64_t)(o - 32) * BW_D0) + BW_F * role + k] = acc;
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
// number of partial accumulators = workgroups: every workgroup gets the same number of tiles (+-1), at most
// CPPF_BWD_MAX_PARTS of them (two resident workgroups per CU)
int n_parts_for(int64_t P)
{
    const int64_t t = (P + 63) / 64;
    if (t <= 1) return 1;
    const int64_t per = (t + CPPF_BWD_MAX_PARTS - 1) / CPPF_BWD_MAX_PARTS;
    return (int)((t + per - 1) / per);
}

}  // namespace

extern "C" {

// workspace: [image][table N*128][partials][rows P*64][keys, vals, sorted keys, sorted vals: 4 x 2P i32][seg N+1 i32]
//            [S N*128][chunk sums][sort temp]
struct BwdLayout { size_t image, table, parts, rows, keys, seg, S, chunks, temp, temp_bytes, total; int n_chunks; };
static BwdLayout bwd_layout(int64_t n_pairs, int64_t n_points, const int* dims, int n_res, int out_dim)
{
    BwdLayout L;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L.n_chunks = (int)((n_points + FW_CHUNK - 1) / FW_CHUNK);
    L.image = 0;
    L.table = up((size_t)STD_PACKED * sizeof(float) + 20 * sizeof(int64_t));
    L.parts = L.table + up((size_t)n_points * PROJ_COLS * sizeof(float));
    L.rows = L.parts + up((size_t)n_parts_for(n_pairs) * (size_t)count_params(dims, n_res, out_dim) * sizeof(float));
    L.keys = L.rows + up((size_t)n_pairs * BW_ROW * sizeof(float));
    L.seg = L.keys + up((size_t)8 * n_pairs * sizeof(int32_t));
    L.S = L.seg + up((size_t)(n_points + 1) * sizeof(int32_t));
    L.chunks = L.S + up((size_t)n_points * 2 * BW_ROW * sizeof(float));
    L.temp = L.chunks + up((size_t)L.n_chunks * 2 * BW_ROW * BW_F * sizeof(float));
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
    if (!std_shape(F, dims, n_res) || out_dim > 16 * STD_NOB) return CPPF_EUNSUPPORTED;
    if (!grad_params) return CPPF_EINVAL;
    if (offs[4] < 0 || offs[10] >= 0 || offs[16] < 0) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_params = count_params(dims, n_res, out_dim);
    if (n_pairs == 0) return (int)hipMemsetAsync(grad_params, 0, n_params * sizeof(float), st);
    if (!pc || !nrm || !feat || !idxs || !params || !grad_out || !grad_feat || n_points < 1) return CPPF_EINVAL;
    const int n_parts = n_parts_for(n_pairs);
    const BwdLayout Lw = bwd_layout(n_pairs, n_points, dims, n_res, out_dim);
    if (!workspace || workspace_bytes < Lw.total) return CPPF_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    float* image = (float*)(ws + Lw.image);
    float* table = (float*)(ws + Lw.table);
    hipError_t e = hipMemsetAsync(ws + Lw.parts, 0, (size_t)n_parts * n_params * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    PackOffs po;
    for (int i = 0; i < 20; ++i) po.o[i] = offs[i];
    bwd_pack_kernel<<<(STD_PACKED + 255) / 256, 256, 0, st>>>(params, po, out_dim, image);
    bwd_point_proj_kernel<<<(unsigned)((n_points + 1) / 2), 256, 0, st>>>(feat, image, table, n_points);
    BwdArgs A;
    A.pc = pc; A.nrm = nrm; A.idxs = idxs; A.packed = image; A.table = table; A.grad_out = grad_out;
    A.parts = (float*)(ws + Lw.parts); A.rows = (float*)(ws + Lw.rows); A.P = n_pairs; A.n_params = n_params;
    A.offs = reinterpret_cast<const int64_t*>(image + STD_PACKED);
    A.out_dim = out_dim; A.idx64 = idx_is_i64; A.n_parts = n_parts;
    static bool attr_done = false;
    if (!attr_done) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mlp_bwd_kernel<141>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                BW_LDS_FLOATS * sizeof(float));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mlp_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    BW_LDS_FLOATS * sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (out_dim == 141) pair_mlp_bwd_kernel<141><<<n_parts, BW_THREADS, BW_LDS_FLOATS * sizeof(float), st>>>(A);
    else pair_mlp_bwd_kernel<0><<<n_parts, BW_THREADS, BW_LDS_FLOATS * sizeof(float), st>>>(A);
    bwd_reduce_kernel<<<(int)((n_params + 63) / 64), 256, 0, st>>>(A.parts, n_parts, n_params, grad_params);
    // per-point sums: stable sort of the 2P (point, entry) keys, segment starts, ordered row sums
    int32_t* seg = (int32_t*)(ws + Lw.seg);
    const int32_t* svals = nullptr;
    const int rc = n_points <= 65536
                       ? sort_entries<uint16_t>(idxs, idx_is_i64, n_pairs, n_points, ws + Lw.keys, ws + Lw.temp, Lw.temp_bytes, seg, &svals, st)
                       : sort_entries<int32_t>(idxs, idx_is_i64, n_pairs, n_points, ws + Lw.keys, ws + Lw.temp, Lw.temp_bytes, seg, &svals, st);
    if (rc != 0) return rc;
    float* S = (float*)(ws + Lw.S);
    float* chunks = (float*)(ws + Lw.chunks);
    bwd_point_kernel<<<(int)((n_points + 3) / 4), 256, 0, st>>>(A.rows, svals, seg, n_pairs, n_points, params + offs[0],
                                                                 params + offs[4], S, grad_feat);
    bwd_featw_kernel<<<Lw.n_chunks, 256, 0, st>>>(S, feat, n_points, chunks);
    bwd_featw_reduce_kernel<<<(2 * BW_ROW * BW_F + 255) / 256, 256, 0, st>>>(chunks, Lw.n_chunks, offs[0], offs[4], grad_params);
    return (int)hipGetLastError();
}

}  // extern "C"
