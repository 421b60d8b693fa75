// Shared by the SPRIN forward (sprin.hip) and backward (sprin_bwd.hip) kernels: shapes of the standard encoder
// (train.py:34), the lane-ordered MFMA images of its kernel-MLP, and the device helpers both directions use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sprin {

constexpr int SP_WAVES_MAX = 8;   // points per workgroup: 8 when the per-wave LDS is small (first layer), else 4
constexpr int SP_RANK = 32, SP_NOUT = 32;
constexpr int SP_KSTRIDE = SP_RANK + 1;  // kern[j][r] row stride in LDS (odd: conflict-free column walks)

// ---- kernel-MLP on v_mfma_f32_16x16x4_f32, transposed: D[out][row] = W[out][k] * X^T[k][row].
// One instruction covers 16 outputs x 16 neighbour rows x 4 inputs; lane l = (j = l & 15 -> row, g = l >> 4).
// A operand = one weight from the lane-ordered LDS image, B operand = one activation register, D = f32x4 =
// outputs 16*ob + 4*g + r of row j.  The D layout of a layer is the B layout of the next one if that layer walks
// its inputs as k(s, g) = 16*(s/4) + 4*g + s%4, so the five layers chain with no data movement (same scheme as
// csrc/pair_mlp.hip); the exact-fp32 MFMA reproduces the oracle's fmaf chain in that order (order = 1).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SPW_L1 = 0;                         // [2 ob][2 s][64]   (inputs 6, 7 are zero columns)
constexpr int SPW_L2 = SPW_L1 + 2 * 2 * 64;       // [4 ob][8 s][64]
constexpr int SPW_L3 = SPW_L2 + 4 * 8 * 64;       // [2 ob][16 s][64]
constexpr int SPW_L4 = SPW_L3 + 2 * 16 * 64;      // [2 ob][8 s][64]
constexpr int SPW_L5 = SPW_L4 + 2 * 8 * 64;       // [2 ob][8 s][64]
constexpr int SPW_VEC = SPW_L5 + 2 * 8 * 64;      // bias/gamma/beta in natural order: b1 g1 be1 b2 g2 be2 b3 g3 be3 b4 g4 be4 b5
constexpr int SPW_B1 = SPW_VEC, SPW_B2 = SPW_B1 + 96, SPW_B3 = SPW_B2 + 192, SPW_B4 = SPW_B3 + 96, SPW_B5 = SPW_B4 + 96;
constexpr int SPW_FLOATS = SPW_B5 + 32;           // 6 912 floats = 27 KB, one copy per workgroup
constexpr int SP_NAT_KERNEL = 6 * 32 + 3 * 32 + 32 * 64 + 3 * 64 + 64 * 32 + 3 * 32 + 32 * 32 + 3 * 32 + 32 * 32 + 32;  // natural floats

__device__ __forceinline__ float sp_xor16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401f)); }
__device__ __forceinline__ float sp_xor32(float v, int lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v))); }

template <int NIB, int NOB>   // inputs 16*NIB (held as NIB f32x4), outputs 16*NOB
__device__ __forceinline__ void sp_mfma_layer(const float* __restrict__ Wl, const float* __restrict__ bias, const f32x4 (&x)[NIB],
                                              f32x4 (&y)[NOB], int lane, int g)
{
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) y[ob] = *reinterpret_cast<const f32x4*>(bias + 16 * ob + 4 * g);
#pragma unroll
    for (int s = 0; s < 4 * NIB; ++s) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
            y[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(ob * 4 * NIB + s) * 64 + lane], x[s / 4][s % 4], y[ob], 0, 0, 0);
    }
}
// the same layer for TWO row blocks at once: every weight element is read from LDS once and feeds both blocks' MFMAs -- two
// independent accumulator chains per output block (a chain of 4 NIB dependent fp32 MFMAs otherwise waits for itself) and half the
// LDS reads.  Per block the operations and their order are those of sp_mfma_layer: the same bits.
template <int NIB, int NOB>
__device__ __forceinline__ void sp_mfma_layer2(const float* __restrict__ Wl, const float* __restrict__ bias, const f32x4 (&xa)[NIB],
                                               const f32x4 (&xb)[NIB], f32x4 (&ya)[NOB], f32x4 (&yb)[NOB], int lane, int g)
{
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) { ya[ob] = *reinterpret_cast<const f32x4*>(bias + 16 * ob + 4 * g); yb[ob] = ya[ob]; }
#pragma unroll
    for (int s = 0; s < 4 * NIB; ++s) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const float w = Wl[(ob * 4 * NIB + s) * 64 + lane];
            ya[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xa[s / 4][s % 4], ya[ob], 0, 0, 0);
            yb[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xb[s / 4][s % 4], yb[ob], 0, 0, 0);
        }
    }
}
// LayerNorm (eps 1e-5, affine) + ReLU on a row spread over the 4 lanes g: per-lane partial sums in (ob, r) order,
// combined as (p0 + p1) + (p2 + p3) through the LDS crossbar (oracle/sprin_oracle.c:layer_norm_ord)
template <int NOB>
__device__ __forceinline__ void sp_ln_relu4(f32x4 (&y)[NOB], const float* __restrict__ gamma, const float* __restrict__ beta, int lane,
                                            int g)
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
    const float inv = cppf::inv_sqrt_rn(q / H + 1e-5f);   // = 1.0f / sqrtf(.), bit for bit (cppf_math.h)
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * ob + 4 * g);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * ob + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = ((y[ob][r] - mean) * inv) * gm[r] + bt[r];
            y[ob][r] = z > 0.f ? z : 0.f;
        }
    }
}

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }


// ---- natural (packed) layout of one standard layer's kernel-MLP, floats: W1 b1 g1 e1 | W2 b2 g2 e2 | W3 .. | W4 .. | W5 b5
constexpr int NAT_W1 = 0, NAT_V1 = NAT_W1 + 32 * 6, NAT_W2 = NAT_V1 + 96, NAT_V2 = NAT_W2 + 64 * 32, NAT_W3 = NAT_V2 + 192,
              NAT_V3 = NAT_W3 + 32 * 64, NAT_W4 = NAT_V3 + 96, NAT_V4 = NAT_W4 + 32 * 32, NAT_W5 = NAT_V4 + 96,
              NAT_V5 = NAT_W5 + 32 * 32;
static_assert(NAT_V5 + 32 == SP_NAT_KERNEL, "natural layout");
__host__ __device__ inline int sp_khid(int s, int g) { return 16 * (s / 4) + 4 * g + (s % 4); }

// float `idx` of the forward image (SPW_* layout) from the layer's natural parameters
__host__ __device__ inline float sp_image_elem(int idx, const float* __restrict__ q)
{
    if (idx < SPW_L2) {                                   // layer 1: k = 4*s + g, inputs 6 and 7 are zero columns
        const int ob = idx >> 7, s = (idx >> 6) & 1, ln = idx & 63;
        const int o = 16 * ob + (ln & 15), k = 4 * s + (ln >> 4);
        return k < 6 ? q[NAT_W1 + o * 6 + k] : 0.f;
    }
    if (idx < SPW_VEC) {                                  // layers 2..5: k(s, g) = 16*(s/4) + 4*g + s%4
        const int L = idx < SPW_L3 ? 0 : (idx < SPW_L4 ? 1 : (idx < SPW_L5 ? 2 : 3));
        const int off = L == 0 ? SPW_L2 : (L == 1 ? SPW_L3 : (L == 2 ? SPW_L4 : SPW_L5));
        const int nat = L == 0 ? NAT_W2 : (L == 1 ? NAT_W3 : (L == 2 ? NAT_W4 : NAT_W5));
        const int n_in = L == 1 ? 64 : 32, S = n_in / 4;
        const int i = idx - off, ob = i / (S * 64), s = (i >> 6) % S, ln = i & 63;
        return q[nat + (16 * ob + (ln & 15)) * n_in + sp_khid(s, ln >> 4)];
    }
    const int v = idx - SPW_VEC;                          // bias | gamma | beta blocks, natural order
    if (v < 96) return q[NAT_V1 + v];
    if (v < 288) return q[NAT_V2 + v - 96];
    if (v < 384) return q[NAT_V3 + v - 288];
    if (v < 480) return q[NAT_V4 + v - 384];
    return q[NAT_V5 + v - 480];
}

// ---- backward image: TRANSPOSED weights of layers 5..2 as A operands, A[input row 16*ib + j][k = output khid(s, g)]
constexpr int SPT_5 = 0;                       // [2 ib][8 s][64]   W5^T
constexpr int SPT_4 = SPT_5 + 2 * 8 * 64;      // [2 ib][8 s][64]   W4^T
constexpr int SPT_3 = SPT_4 + 2 * 8 * 64;      // [4 ib][8 s][64]   W3^T (64 inputs, 32 outputs)
constexpr int SPT_2 = SPT_3 + 4 * 8 * 64;      // [2 ib][16 s][64]  W2^T (32 inputs, 64 outputs)
constexpr int SPT_FLOATS = SPT_2 + 2 * 16 * 64;   // 6 144 floats
__host__ __device__ inline float sp_timage_elem(int idx, const float* __restrict__ q)
{
    const int L = idx < SPT_4 ? 5 : (idx < SPT_3 ? 4 : (idx < SPT_2 ? 3 : 2));
    const int off = L == 5 ? SPT_5 : (L == 4 ? SPT_4 : (L == 3 ? SPT_3 : SPT_2));
    const int nat = L == 5 ? NAT_W5 : (L == 4 ? NAT_W4 : (L == 3 ? NAT_W3 : NAT_W2));
    const int n_in = L == 3 ? 64 : 32, S = (L == 2 ? 64 : 32) / 4;
    const int i = idx - off, ib = i / (S * 64), s = (i >> 6) % S, ln = i & 63;
    return q[nat + sp_khid(s, ln >> 4) * n_in + 16 * ib + (ln & 15)];
}

}  // namespace sprin
