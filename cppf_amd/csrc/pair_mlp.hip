// Pair encoder for gfx950: PPF construction + feature gather + ResLayer chain + (optionally) the
// bin decode, one kernel, logits never leave the CU.
//
// Reference: PPFEncoder.forward_with_idx, models/model.py:117-137; ResLayer.forward :27-31;
// decode nocs/inference.py:185-188, 238-256.
//
// MFMA path (the architecture the reference trains: F = 40, ppffcs = [84,32,32,16], train.py:35).
// Everything is computed transposed, D[out][pair] = W[out][k] * X^T[k][pair], with
// v_mfma_f32_16x16x4_f32 (exact fp32, fmaf-chain numerics):
//   lane l = (j = l & 15, g = l >> 4) of a wave serves pair j of a 16-pair block;
//   A operand = one weight  W[16*ob + j][k(s,g)]   (pre-packed in lane order, read from LDS)
//   B operand = one input   X[pair j][k(s,g)]      (a register of lane l)
//   D         = f32x4: outputs 16*ob + 4*g + r, r = 0..3, of pair j.
// Because D of one layer leaves output (16*ob + 4*g + r) in the lane that will need it as the
// B operand of k-step s = 4*ob + r of the next layer, layers chain with no data movement at all:
// the k order of a hidden layer is k(s,g) = 16*(s/4) + 4*g + s%4.
//
// The first layer (84 -> 32|32) is linear in cat(feat[a], feat[b], ppf), so its two 40-wide blocks are
// hoisted out of the pair loop: a small kernel projects every POINT once,
//     TA[n][o] = b[o] + sum_k W[o][k]    feat[n][k]        TB[n][o] = sum_k W[o][40+k] feat[n][k]
// (fmaf chains, k = 0..39), and a pair's pre-activation is (TA[a][o] + TB[b][o]) plus one MFMA k-step for
// the 4 PPF inputs.  That removes 80 of the 188 MFMAs per 16 pairs (the per-point work is N*128*40 MAC,
// 0.2 % of the pair work at K = 128) and turns the 2x160-byte feature gather into 2x256 bytes of an
// L2-resident 2 MB table.  The oracle follows the same association (oracle/cppf_oracle.c, order 1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/cppf.h"
#include "cppf_math.h"

using namespace cppf;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at 4-byte alignment

#define CPPF_CHECK_LAUNCH()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

#include "pair_layout.h"   // packed image: offsets, khid, dec_col, std_pack_elem

static bool is_std(int F, const int* dims, int n_res, int out_dim)
{
    return F == STD_F && n_res == 3 && dims[0] == 84 && dims[1] == 32 && dims[2] == 32 && dims[3] == 16 &&
           out_dim >= 1 && out_dim <= 16 * STD_NOB;
}

#define GEN_MAX_RES 8
#define GEN_MAX_DIM 128
static bool gen_ok(int F, const int* dims, int n_res, int out_dim)
{
    if (n_res < 0 || n_res > GEN_MAX_RES || F < 1 || out_dim < 1 || out_dim > 1024) return false;
    if (dims[0] != 2 * F + 4) return false;
    for (int i = 0; i <= n_res; ++i)
        if (dims[i] < 1 || dims[i] > GEN_MAX_DIM) return false;
    return true;
}
static size_t gen_floats(const int* dims, int n_res, int out_dim)
{
    size_t n = 0;
    for (int i = 0; i < n_res; ++i) {
        size_t K = dims[i], M = dims[i + 1];
        n += M * K + M + M * M + M;
        if (K != M) n += M * K + M;
    }
    n += (size_t)out_dim * dims[n_res] + out_dim;
    return n;
}

extern "C" size_t cppf_pair_mlp_packed_floats(int F, const int* dims, int n_res, int out_dim)
{
    if (!dims) return 0;
    if (is_std(F, dims, n_res, out_dim)) return STD_PACKED;
    if (gen_ok(F, dims, n_res, out_dim)) return gen_floats(dims, n_res, out_dim);
    return 0;
}

extern "C" int cppf_pair_mlp_pack(const float* params, const int64_t* offs, int F, const int* dims, int n_res,
                                  int out_dim, float* out)
{
    if (!params || !offs || !dims || !out) return CPPF_EINVAL;
    if (is_std(F, dims, n_res, out_dim)) {
        if (offs[4] < 0 || offs[10] >= 0 || offs[16] < 0) return CPPF_EINVAL;  // fc0 on layers 0 and 2 only
        for (int i = 0; i < STD_PACKED; ++i) out[i] = std_pack_elem(i, params, offs, out_dim);
        return 0;
    }
    if (gen_ok(F, dims, n_res, out_dim)) {
        // canonical order: per layer fc1.w fc1.b fc2.w fc2.b [fc0.w fc0.b], then final.w final.b
        size_t n = 0;
        for (int i = 0; i < n_res; ++i) {
            size_t K = dims[i], M = dims[i + 1];
            const int64_t* o = offs + 6 * i;
            if ((K != M) != (o[4] >= 0)) return CPPF_EINVAL;
            memcpy(out + n, params + o[0], sizeof(float) * M * K); n += M * K;
            memcpy(out + n, params + o[1], sizeof(float) * M); n += M;
            memcpy(out + n, params + o[2], sizeof(float) * M * M); n += M * M;
            memcpy(out + n, params + o[3], sizeof(float) * M); n += M;
            if (K != M) {
                memcpy(out + n, params + o[4], sizeof(float) * M * K); n += M * K;
                memcpy(out + n, params + o[5], sizeof(float) * M); n += M;
            }
        }
        const int64_t* o = offs + 6 * n_res;
        memcpy(out + n, params + o[0], sizeof(float) * out_dim * dims[n_res]); n += (size_t)out_dim * dims[n_res];
        memcpy(out + n, params + o[1], sizeof(float) * out_dim);
        return 0;
    }
    return CPPF_EUNSUPPORTED;
}

// The same image built on the device from parameters that live there (training: the weights change every step and a
// host pack would cost a device -> host -> device round trip with a synchronisation).
struct PackOffs { int64_t o[20]; };
__global__ __launch_bounds__(256) void pair_pack_kernel(const float* __restrict__ params, PackOffs offs, int out_dim,
                                                        float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < STD_PACKED) out[i] = std_pack_elem(i, params, offs.o, out_dim);
}
extern "C" int cppf_pair_mlp_pack_device(const float* params, const int64_t* offs, int F, const int* dims, int n_res,
                                         int out_dim, float* out, void* stream)
{
    if (!params || !offs || !dims || !out) return CPPF_EINVAL;
    if (!is_std(F, dims, n_res, out_dim)) return CPPF_EUNSUPPORTED;
    if (offs[4] < 0 || offs[10] >= 0 || offs[16] < 0) return CPPF_EINVAL;
    PackOffs po;
    for (int i = 0; i < 20; ++i) po.o[i] = offs[i];
    hipLaunchKernelGGL(pair_pack_kernel, dim3((STD_PACKED + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, po,
                       out_dim, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ----------------------------------------------------------------------------- MFMA kernel
#define MLP_THREADS 1024
#define MLP_WAVES_PER_SIMD 4
#define PB 1  // 16-pair blocks per wave tile

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// max(v, 0) as ONE v_max_f32.  On a value the compiler cannot prove to be no signalling NaN (an MFMA result, a crossbar
// exchange) fmaxf / `v > 0 ? v : 0` is preceded by a canonicalising v_max_f32 v, v -- 20 extra instructions per tile on the
// pipe the MFMAs share -- unless the file is built with -fno-honor-nans (Makefile).  Not inline assembly: the compiler
// does not count an asm statement as a VALU instruction when it places the wait states an MFMA result needs before its
// first VALU reader, so an asm v_max_f32 on an accumulator reads it early (seen: run-to-run differences in the logits).
__device__ __forceinline__ float relu1(float v) { return fmaxf(v, 0.f); }
__device__ __forceinline__ f32x4 relu4(f32x4 v)
{
    f32x4 r;
    r[0] = relu1(v[0]); r[1] = relu1(v[1]); r[2] = relu1(v[2]); r[3] = relu1(v[3]);
    return r;
}
__device__ __forceinline__ f32x4 ldb4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// PPF of one pair (models/model.py:118-129); returns component `g`.
__device__ __forceinline__ float ppf_component(const float* __restrict__ pc, const float* __restrict__ nrm, int a, int b,
                                               int g)
{
    const f3 pa = ld3(pc, a), pb = ld3(pc, b), na = ld3(nrm, a), nb = ld3(nrm, b);
    const f3 xy = sub3(pa, pb);
    const float d = sqrtf((xy.x * xy.x + xy.y * xy.y) + xy.z * xy.z);
    const float den = d + 1e-7f;                       // fp32 add (torch), unlike the vote kernels
    const f3 u = {xy.x / den, xy.y / den, xy.z / den};
    const float p0 = (na.x * u.x + na.y * u.y) + na.z * u.z;
    const float p1 = (nb.x * u.x + nb.y * u.y) + nb.z * u.z;
    const float p2 = (na.x * nb.x + na.y * nb.y) + na.z * nb.z;
    return g == 0 ? p0 : (g == 1 ? p1 : (g == 2 ? p2 : d));
}

// ---- decode helpers (semantics: oracle/cppf_oracle.c:orc_sample_bin) --------------------------
// Logit 16*R + 4*g + r of the lane's pair lives in L[R][r] of lane group g = lane >> 4: one "chunk"
// of 4 consecutive logits per lane per MFMA output block ("row") R.  Cross-lane traffic inside the
// 4 lanes of a pair goes through the LDS crossbar (ds_swizzle / ds_bpermute: no LDS memory, and no VALU
// issue slots -- a v_permlane*_swap exchange costs 2 moves + swap (2 slots) + select = 5 slots on the pipe
// the fp32 MFMAs also need; measured 3 % faster, profiles/microbench/valu_bench.hip).
__device__ __forceinline__ unsigned xor16u(unsigned v, int lane)
{
    (void)lane;
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401f);  // bit mode: and 0x1f, or 0, xor 0x10
}
__device__ __forceinline__ unsigned xor32u(unsigned v, int lane)
{
    return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)v);
}
__device__ __forceinline__ float xor16f(float v, int lane) { return __uint_as_float(xor16u(__float_as_uint(v), lane)); }
__device__ __forceinline__ float xor32f(float v, int lane) { return __uint_as_float(xor32u(__float_as_uint(v), lane)); }

// One head whose NL consecutive bins NL*g .. NL*g + NL-1 sit in v[] of lane group g (dec_col layout).  Semantics:
// oracle/cppf_oracle.c:orc_sample_bin.  Returns true in exactly one of the pair's 4 lanes -- the one that owns
// the sampled bin -- with the bin index.
template <int NL>
__device__ __forceinline__ bool sample_seg(const float (&v)[NL], float u, int g, int lane, int& bin)
{
    static_assert(NL == 8 || NL == 9, "the draw below searches 7 running sums + at most one more");
    float m = v[0];
#pragma unroll
    for (int k = 1; k < NL; ++k) m = fmaxf(m, v[k]);   // (v_max3_f32 pairs; no canonicalising moves under -fno-honor-nans)
    float mall = fmaxf(m, xor16f(m, lane));
    mall = fmaxf(mall, xor32f(mall, lane));
    // softmax weights 2^(l log2e - max log2e): the max subtraction rides on the exponent's fma (cppf_math.h:det_exp2w)
    const float c = -(mall * CPPF_LOG2E);
    float e[NL];
    // (scalar instructions on purpose: beside the fp32 MFMAs, which run on the same datapath, a v_pk_fma_f32 costs as much as
    //  the two v_fma_f32 it replaces or more -- packed exponentials and the compiler's SLP packing measured 2.5-3 % slower,
    //  profiles/r2_pair_mlp_phases.txt; the Makefile passes -fno-slp-vectorize)
#pragma unroll
    for (int k = 0; k < NL; ++k) e[k] = det_exp2w(v[k], c);
    // running sums of the lane's segment: the last one is the segment total, and the draw compares them with the
    // threshold moved into the segment (t - off) -- one chain of NL - 1 additions serves both
    float b[NL];
    b[0] = e[0];
#pragma unroll
    for (int k = 1; k < NL; ++k) b[k] = b[k - 1] + e[k];
    const float T = b[NL - 1];
    const float Tp = xor16f(T, lane);           // the other lane of my half
    const float half = T + Tp;                  // T0 + T1 in lanes g = 0,1; T2 + T3 in lanes g = 2,3 (a+b == b+a)
    const float oth = xor32f(half, lane);
    const float off = ((g & 2) ? oth : 0.f) + ((g & 1) ? Tp : 0.f);
    const float t = u * ((g & 2) ? oth + half : half + oth) - off;   // u * ((T0 + T1) + (T2 + T3)) - off_g
    // first k with b[k] > t, NL - 1 when there is none.  The weights are >= 0, so the running sums never decrease and the
    // oracle's scan in k order finds what a bisection finds: three compares over b[0..6] instead of seven (+ one for NL = 9).
    const bool c1 = b[3] > t;
    const bool c2 = (c1 ? b[1] : b[5]) > t;
    const float lo = c2 ? b[0] : b[2], hi = c2 ? b[4] : b[6];
    const bool c3 = (c1 ? lo : hi) > t;
    int kk = (c1 ? 0 : 4) + (c2 ? 0 : 2) + (c3 ? 0 : 1);
    if (NL == 9) kk = b[7] > t ? kk : 8;
    // the pair's first segment with a hit owns the draw (none: the last bin, which lane 3's search has found by itself).
    // The ballot is wave-uniform, so "first of the four lanes j, j + 16, j + 32, j + 48" is scalar arithmetic on its four
    // 16-bit quarters, and the owner's predicate goes straight back into a lane mask: no VALU instruction after the compare
    // (the per-lane shift / and / find-first-bit form took thirteen).
    const unsigned long long hm = __ballot(b[NL - 1] > t);
    const unsigned h0 = (unsigned)hm & 0xffffu, h1 = (unsigned)(hm >> 16) & 0xffffu, h2 = (unsigned)(hm >> 32) & 0xffffu;
    const unsigned o1 = h1 & ~h0, o2 = h2 & ~(h0 | h1), o3 = ~(h0 | h1 | h2) & 0xffffu;
    const unsigned long long own = (unsigned long long)(h0 | (o1 << 16)) | ((unsigned long long)(o2 | (o3 << 16)) << 32);
    bool owner = __builtin_amdgcn_inverse_ballot_w64(own);
    bin = NL * g + kk;
    if (__any(u < 0.f)) {  // arg-max mode (rare, wave-uniform test so the common path really skips it)
        asm volatile("" ::: "memory");
        int ak = 64;   // (position inside the segment first, its base added once: `am = NL * g + k` per k made the compiler keep
#pragma unroll     //  NL loop-invariant registers per head for this rarely taken path -- the all-heads variant spilled)
        for (int k = NL - 1; k >= 0; --k)
            if (v[k] == mall) ak = k;
        const int am = NL * g + ak;   // a lane without the maximum: >= 64, above every bin
        int best = min(am, (int)xor16u((unsigned)am, lane));
        best = min(best, (int)xor32u((unsigned)best, lane));
        if (u < 0.f) { bin = best; owner = am == best; }
    }
    return owner;
}

struct MlpArgs {
    const float* pc;
    const float* nrm;
    const float* feat;
    const void* idxs;
    const float* packed;
    const float* u_tr;
    const float* u_rot;
    const float* table;  // [N][128] per-point layer-0 projections (point_proj_kernel)
    float* out;      // logits [P,out_dim]   (LOGITS)
    float* outputs;  // [P,2]                (DECODE)
    float* heads;    // [P,8] or null        (DECODE)
    int64_t P;
    int out_dim;
    int idx64;
    float vr0, vr1;
    // SEL variant (second MLP pass of nocs/inference.py:236 on the pairs that survived the back-vote): slot i of the launch
    // works on pair sel[i], i < min(*n_sel, P); everything per pair (indices, uniforms, results) stays at the pair's own row
    const int32_t* sel;
    const int32_t* n_sel;
};

// Addresses inside the tile loop are a uniform base (SGPR pair) + an unsigned 32-bit byte offset (one VGPR): the
// `global_load v, v_off, s[base]` form.  64-bit pointer arithmetic per lane (v_lshl_add_u64, v_lshlrev_b64 ...) was ~50 VALU
// per tile on the pipe the fp32 MFMAs share.  Hence the limits of the MFMA path: N < 2^23 points, P < 2^27 pairs (launch_std).
template <typename T>
__device__ __forceinline__ const T& at_off(const void* base, unsigned byte_off)
{
    return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ T& at_off(void* base, unsigned byte_off)
{
    return *reinterpret_cast<T*>(static_cast<char*>(base) + byte_off);
}
__device__ __forceinline__ f3 ld3o(const float* __restrict__ base, int i)
{
    const unsigned o = __umul24((unsigned)i, 12u);   // (N < 2^23 on this path: the full-rate 24-bit multiply; v_mul_lo_u32 takes four issue slots)
    // (the +4 / +8 ride on the uniform base, i.e. in the instruction's immediate offset: added to the 32-bit lane offset they are
    //  two v_add_u32 per index, because unsigned wrap-around has to be preserved)
    return {at_off<float>(base, o), at_off<float>(base + 1, o), at_off<float>(base + 2, o)};
}

// slot of the launch -> row of the pair arrays (identity unless SEL)
// (slots, tiles and pair counts are 32-bit inside the kernel -- P < 2^27, launch_std -- : 64-bit compares and selects were a
//  dozen instructions per tile)
template <bool SEL>
__device__ __forceinline__ unsigned pair_row(const MlpArgs& A, int slot, int Pn)
{
    const unsigned s = (unsigned)min(slot, Pn - 1);
    return SEL ? (unsigned)at_off<int>(A.sel, s * 4u) : s;
}

template <bool SEL>
__device__ __forceinline__ void load_pair_idx(const MlpArgs& A, int slot, int Pn, int& ia, int& ib)
{
    // two 4-byte loads of the low words whatever the index width: an i32 / i64 branch around the loads ends in a
    // wait for ALL outstanding loads (the gathers of the next tile that are in flight at this point)
    const unsigned p = pair_row<SEL>(A, slot, Pn);
    const unsigned o = p * (A.idx64 ? 16u : 8u);
    ia = at_off<int>(A.idxs, o);
    ib = at_off<int>(static_cast<const char*>(A.idxs) + (A.idx64 ? 8 : 4), o);   // (second column through the uniform base)
}

// PPF of one pair from already loaded points/normals (models/model.py:118-129); component `g`.
__device__ __forceinline__ float ppf_from(f3 pa, f3 pb, f3 na, f3 nb, int g)
{
    const f3 xy = sub3(pa, pb);
    const float d = sqrt_rn((xy.x * xy.x + xy.y * xy.y) + xy.z * xy.z);
    const float den = d + 1e-7f;                       // fp32 add (torch), unlike the vote kernels
    // three IEEE divisions by one denominator (den in [1e-7, ~2], |xy| <= d: no rescaling or fix-up would apply): div_by()
    const float rden = refined_rcp(den);
    const f3 u = {div_by(xy.x, den, rden), div_by(xy.y, den, rden), div_by(xy.z, den, rden)};
    const float p0 = (na.x * u.x + na.y * u.y) + na.z * u.z;
    const float p1 = (nb.x * u.x + nb.y * u.y) + nb.z * u.z;
    const float p2 = (na.x * nb.x + na.y * nb.y) + na.z * nb.z;
    // 4-way select by lane group: g = lane >> 4, so "g == k" is a constant lane mask -- three v_cndmask on scalar masks
    // (the and / or form on per-lane masks took eleven; a ?: chain on g itself becomes divergent branches that split the
    // MFMA schedule)
    (void)g;
    const bool g0 = __builtin_amdgcn_inverse_ballot_w64(0x000000000000ffffull), g1 = __builtin_amdgcn_inverse_ballot_w64(0x00000000ffff0000ull),
               g2 = __builtin_amdgcn_inverse_ballot_w64(0x0000ffff00000000ull);
    const float s23 = g2 ? p2 : d, s123 = g1 ? p1 : s23;
    return g0 ? p0 : s123;
}

// Layer-0 projections of every point: T[n][r] = (r < 64 ? bias[r] : 0) + sum_{k<40} WPT[k][r] * feat[n][k],
// one fmaf chain per output in k order.  One block per PROJ_PPB points, a thread per column and PROJ_PPB / 2 points: a
// weight is loaded once for the thread's points (one block per 2 points re-read the 20 KB of weights 2 048 times at N = 4096).
#define PROJ_PPB 8
__device__ __forceinline__ void point_proj_body(const float* __restrict__ feat, const float* __restrict__ packed, float* __restrict__ T,
                                                int64_t N)
{
    __shared__ float f[PROJ_PPB][STD_F];
    const int half = threadIdx.x >> 7, r = threadIdx.x & 127;
    const int64_t n0 = (int64_t)blockIdx.x * PROJ_PPB;
    for (int i = threadIdx.x; i < PROJ_PPB * STD_F; i += 256) {
        const int64_t n = n0 + i / STD_F;
        f[i / STD_F][i % STD_F] = n < N ? feat[n * STD_F + i % STD_F] : 0.f;
    }
    __syncthreads();
    float acc[PROJ_PPB / 2];
    const float b = r < 64 ? packed[OFF_BPT + r] : 0.f;
#pragma unroll
    for (int q = 0; q < PROJ_PPB / 2; ++q) acc[q] = b;
#pragma unroll 8
    for (int k = 0; k < STD_F; ++k) {
        const float w = packed[OFF_WPT + k * PROJ_COLS + r];
#pragma unroll
        for (int q = 0; q < PROJ_PPB / 2; ++q) acc[q] = fmaf(w, f[half + 2 * q][k], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < PROJ_PPB / 2; ++q) {
        const int64_t n = n0 + half + 2 * q;
        if (n < N) T[n * PROJ_COLS + r] = acc[q];
    }
}
__global__ __launch_bounds__(256) void point_proj_kernel(const float* __restrict__ feat, const float* __restrict__ packed,
                                                         float* __restrict__ T, int64_t N)
{
    point_proj_body(feat, packed, T, N);
}
// the projections of several clouds in one launch (cppf_pair_mlp_decode_batch): cloud blockIdx.y; as wide as the largest needs
struct ProjBatch { const float* feat[8]; const float* packed[8]; float* table[8]; int64_t N[8]; };
__global__ __launch_bounds__(256) void point_proj_batch_kernel(ProjBatch B)
{
    const int i = blockIdx.y;
    if ((int64_t)blockIdx.x * PROJ_PPB >= B.N[i]) return;
    point_proj_body(B.feat[i], B.packed[i], B.table[i], B.N[i]);
}

// The kernel's body for workgroup `wg` of the `n_wg` that share the pair list of `A` (one launch = one list: wg = blockIdx.x of
// gridDim.x; a batched launch = several lists, each with its own workgroups: pair_mlp_batch_kernel).
template <bool LOGITS, bool DECODE, bool HEADS, bool SEL>
__device__ __forceinline__ void pair_mlp_body(const MlpArgs& A, const int wg, const int n_wg)
{
    extern __shared__ __attribute__((aligned(16))) float W[];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(A.packed);
        f32x4* dst = reinterpret_cast<f32x4*>(W);
        // the decode variants take the final layer (weights and bias) from the copy whose output columns are in
        // dec_col order; everything else is common
        for (int k = threadIdx.x; k < STD_LDS / 4; k += MLP_THREADS) {
            int from = k;
            if (DECODE && k >= OFF_WF / 4 && k < OFF_B0B / 4) from = OFF_WFD / 4 + (k - OFF_WF / 4);
            if (DECODE && k >= OFF_BF / 4) from = OFF_BFD / 4 + (k - OFF_BF / 4);
            dst[k] = src[from];
        }
    }
    // bin -> value tables (nocs/inference.py:187-188,252,256; fp32, left to right, true division): a
    // correctly rounded divide is ~12 VALU, a table read is one LDS access
    float* lut = W + STD_LDS;  // [0,32) mu, [32,64) nu, [64,100) theta
    int* tile_ctr = reinterpret_cast<int*>(W + STD_LDS + 112);   // next tile of this workgroup's range, see below
    if (threadIdx.x == 0) *tile_ctr = 0;
    if (DECODE && threadIdx.x < 100) {
        const int k = threadIdx.x;
        float v;
        if (k < 32) v = ((float)k / 31.0f * 2.0f) * A.vr0 - A.vr0;
        else if (k < 64) v = (float)(k - 32) / 31.0f * A.vr1;
        else v = (float)(k - 64) / 35.0f * (float)CPPF_PI;
        lut[k] = v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    __builtin_assume(g >= 0 && g < 4);
    int Pn = (int)A.P;
    if (SEL) Pn = min(max(*A.n_sel, 0), Pn);
    const int n_tiles = (Pn + 16 * PB - 1) / (16 * PB);
    // Tiles are handed out per workgroup through an LDS counter.  With a fixed share (8 tiles per wave at the benchmark's
    // size) the waves of a SIMD -- arbitrated oldest first -- ran the same work in 119k ... 211k cycles and the kernel
    // waited for the slowest with a fifth of its issue slots idle (s_memtime trace, profiles/r2_pair_mlp_phases.txt).
    // A workgroup owns a contiguous range of tiles; a wave claims tile t+2 (whose pair indices it prefetches) while
    // tile t runs, so the claim's latency is never waited for.
    const int per_wg = (n_tiles + n_wg - 1) / n_wg;
    const int wg_begin = wg * per_wg;
    const int wg_end = min(wg_begin + per_wg, n_tiles);
    auto claim = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(tile_ctr, 1);
        return wg_begin + __builtin_amdgcn_readfirstlane(v);
    };
    int cur = claim();
    if (cur >= wg_end) return;
    int nxt = claim();

    // Software pipeline over the tiles this wave claims: while tile t runs through the
    // MFMA chain, the gathers of tile t+1 are in flight (indices were fetched one tile earlier still),
    // so a tile never starts with a dependent idx -> feature round trip to L2/HBM.
    // ta/tb[pb][ob] = the lane's 4 outputs (16*ob + 4*g ..) of TA[a] and TB[b]; xp = its PPF input (ppf[g])
    // The layer-0 accumulators of the NEXT tile (16 registers) are what is carried across the loop, not the
    // 32 gathered table registers: they are formed right after the current tile's last MFMA, before the
    // decode, which is where register pressure peaks.
    f32x4 acc[PB][4];
    int ia1[PB], ib1[PB];
    {
        f32x4 ta[PB][4], tb[PB][4];
        float xp[PB];
        int ia[PB], ib[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) load_pair_idx<SEL>(A, cur * (16 * PB) + pb * 16 + j, Pn, ia[pb], ib[pb]);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const unsigned oa = (unsigned)ia[pb] * (PROJ_COLS * 4u) + 16u * g, ob_ = (unsigned)ib[pb] * (PROJ_COLS * 4u) + 256u + 16u * g;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) { ta[pb][ob] = at_off<f32x4>(A.table, oa + 64u * ob); tb[pb][ob] = at_off<f32x4>(A.table, ob_ + 64u * ob); }
            xp[pb] = ppf_from(ld3o(A.pc, ia[pb]), ld3o(A.pc, ib[pb]), ld3o(A.nrm, ia[pb]), ld3o(A.nrm, ib[pb]), g);
        }
        const int nt = nxt < wg_end ? nxt : cur;   // (no next tile: reload this one, the values are never used)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) load_pair_idx<SEL>(A, nt * (16 * PB) + pb * 16 + j, Pn, ia1[pb], ib1[pb]);
        const f32x4 w = ldb4(W + OFF_W0P + lane * 4);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[pb][ob] = mfma4(w[ob], xp[pb], ta[pb][ob] + tb[pb][ob]);
    }

    for (;;) {
        const int tile = cur;
        // The packed weights are loop-invariant LDS data: without this compiler barrier LICM hoists
        // all ~200 weight registers out of the tile loop and the kernel spills.
        asm volatile("" ::: "memory");
        int pair[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) pair[pb] = tile * (16 * PB) + pb * 16 + j;   // slot of the launch
        unsigned row[PB];                                                              // row of the pair arrays
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) row[pb] = pair_row<SEL>(A, pair[pb], Pn);

        // ---- next tile: points / normals in flight during layer 0 ---------------------------------
        f3 npa[PB], npb[PB], nna[PB], nnb[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            npa[pb] = ld3o(A.pc, ia1[pb]); npb[pb] = ld3o(A.pc, ib1[pb]);
            nna[pb] = ld3o(A.nrm, ia1[pb]); nnb[pb] = ld3o(A.nrm, ib1[pb]);
        }
        f32x2 ut[PB], ur[PB];
        if (DECODE) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const unsigned po = row[pb] * 8u;
                if (!SEL) ut[pb] = at_off<f32x2>(A.u_tr, po);
                if (HEADS) ur[pb] = at_off<f32x2>(A.u_rot, po);
            }
        }

        // ---- layer 0 (84 -> 32 | 32) is already in `acc`: (TA[a] + TB[b]) + one MFMA step over the 4 PPF inputs,
        //      formed at the end of the previous trip (or in the prologue) ------------------------------
        // ---- layer 0: fc2 (32 -> 32) on relu(fc1), + fc0 ---------------------------------------
        f32x4 y[PB][2];
        {
            f32x4 h[PB][2], a2[PB][2];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) { h[pb][0] = relu4(acc[pb][0]); h[pb][1] = relu4(acc[pb][1]); }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const f32x4 b = ldb4(W + OFF_B0B + 16 * ob + 4 * g);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) a2[pb][ob] = b;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(W + OFF_W0B + (s * 64 + lane) * 2);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) a2[pb][ob] = mfma4(w[ob], h[pb][s >> 2][s & 3], a2[pb][ob]);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) { y[pb][0] = a2[pb][0] + acc[pb][2]; y[pb][1] = a2[pb][1] + acc[pb][3]; }
        }
        // ---- layer 1: 32 -> 32 -> 32, identity skip --------------------------------------------
        {
            f32x4 a1[PB][2], h[PB][2], a2[PB][2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const f32x4 b1 = ldb4(W + OFF_B1A + 16 * ob + 4 * g), b2 = ldb4(W + OFF_B1B + 16 * ob + 4 * g);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) { a1[pb][ob] = b1; a2[pb][ob] = b2; }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(W + OFF_W1A + (s * 64 + lane) * 2);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) a1[pb][ob] = mfma4(w[ob], y[pb][s >> 2][s & 3], a1[pb][ob]);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) { h[pb][0] = relu4(a1[pb][0]); h[pb][1] = relu4(a1[pb][1]); }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(W + OFF_W1B + (s * 64 + lane) * 2);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) a2[pb][ob] = mfma4(w[ob], h[pb][s >> 2][s & 3], a2[pb][ob]);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) { y[pb][0] = a2[pb][0] + y[pb][0]; y[pb][1] = a2[pb][1] + y[pb][1]; }
        }
        // ---- next tile: PPF from the landed points, table gathers (consumed after the final layer),
        //      and the indices of the tile after it ---------------------------------------------------
        f32x4 ta[PB][4], tb[PB][4];
        float xp[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            xp[pb] = ppf_from(npa[pb], npb[pb], nna[pb], nnb[pb], g);
            const unsigned oa = (unsigned)ia1[pb] * (PROJ_COLS * 4u) + 16u * g, ob_ = (unsigned)ib1[pb] * (PROJ_COLS * 4u) + 256u + 16u * g;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) { ta[pb][ob] = at_off<f32x4>(A.table, oa + 64u * ob); tb[pb][ob] = at_off<f32x4>(A.table, ob_ + 64u * ob); }
        }
        const int nxt2 = claim();
        {
            const int nt = nxt2 < wg_end ? nxt2 : tile;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) load_pair_idx<SEL>(A, nt * (16 * PB) + pb * 16 + j, Pn, ia1[pb], ib1[pb]);
        }

        // ---- layer 2: fc1 | fc0 (32 -> 16 | 16), fc2 (16 -> 16) ---------------------------------
        f32x4 z[PB];
        {
            f32x4 a1[PB][2], a2[PB];
            {
                const f32x4 b1 = ldb4(W + OFF_B2 + 4 * g), b0 = ldb4(W + OFF_B2 + 16 + 4 * g), b2 = ldb4(W + OFF_B2B + 4 * g);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) { a1[pb][0] = b1; a1[pb][1] = b0; a2[pb] = b2; }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(W + OFF_W2 + (s * 64 + lane) * 2);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) a1[pb][ob] = mfma4(w[ob], y[pb][s >> 2][s & 3], a1[pb][ob]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float w = W[OFF_W2B + s * 64 + lane];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    const f32x4 h = relu4(a1[pb][0]);
                    a2[pb] = mfma4(w, h[s], a2[pb]);
                }
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) z[pb] = a2[pb] + a1[pb][1];
        }
        // ---- layer 0 of the next tile from the gathers that have been in flight since layer 1 --------
        {
            const f32x4 w = ldb4(W + OFF_W0P + lane * 4);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[pb][ob] = mfma4(w[ob], xp[pb], ta[pb][ob] + tb[pb][ob]);
        }
        // ---- final 16 -> 144 (9 x 16) and epilogue, one 16-pair block at a time (halves the live
        //      logit registers; the 12 weight reads per block are cheap) ----------------------------
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            // a decode without the rotation heads consumes only the two centre heads = output blocks 0..3 in dec_col order
            constexpr int NOB_USED = (DECODE && !HEADS) ? 4 : STD_NOB;
            f32x4 L[STD_NOB];
#pragma unroll
            for (int ob = 0; ob < NOB_USED; ++ob) L[ob] = ldb4(W + OFF_BF + 16 * ob + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* wp = W + OFF_WF + (s * 64 + lane) * STD_NOBP;
                const f32x4 w0 = ldb4(wp), w1 = ldb4(wp + 4), w2 = ldb4(wp + 8);
#pragma unroll
                for (int ob = 0; ob < NOB_USED; ++ob) {
                    const float w = ob < 4 ? w0[ob & 3] : (ob < 8 ? w1[ob & 3] : w2[ob & 3]);
                    L[ob] = mfma4(w, z[pb][s], L[ob]);
                }
            }
            const bool live = pair[pb] < Pn;
            if (!LOGITS && !DECODE) {  // profiling variant: the MFMA chain alone, results kept alive by a never-true store
                float sacc = 0.f;
#pragma unroll
                for (int ob = 0; ob < STD_NOB; ++ob) sacc += (L[ob][0] + L[ob][1]) + (L[ob][2] + L[ob][3]);
                if (sacc == 1.2345e30f) A.outputs[0] = sacc;
            }
            if (LOGITS) {
                if (live) {
                    // the lane holds 4 consecutive logits per output block: one 16-byte store each (rows are
                    // only 4-byte aligned when out_dim % 4 != 0; global dwordx4 stores allow that)
                    float* o = A.out + (int64_t)pair[pb] * A.out_dim + 4 * g;
#pragma unroll
                    for (int ob = 0; ob < STD_NOB; ++ob) {
                        const int c0 = 16 * ob + 4 * g;
                        if (c0 + 3 < A.out_dim) {
                            *reinterpret_cast<f32x4u*>(o + 16 * ob) = L[ob];
                        } else {
#pragma unroll
                            for (int r = 0; r < 3; ++r)
                                if (c0 + r < A.out_dim) o[16 * ob + r] = L[ob][r];
                        }
                    }
                }
            }
            if (DECODE) {
                int k;
                // nocs/inference.py:187-188 (fp32, left to right); the owning lane stores its value
                if (!SEL) {   // (the second pass only needs the rotation / sign / scale heads: the centre was decoded in the first)
                    {
                        const float v[8] = {L[0][0], L[0][1], L[0][2], L[0][3], L[1][0], L[1][1], L[1][2], L[1][3]};
                        if (sample_seg<8>(v, ut[pb][0], g, lane, k) && live) at_off<float>(A.outputs, row[pb] * 8u) = lut[k];
                    }
                    {
                        const float v[8] = {L[2][0], L[2][1], L[2][2], L[2][3], L[3][0], L[3][1], L[3][2], L[3][3]};
                        if (sample_seg<8>(v, ut[pb][1], g, lane, k) && live) at_off<float>(A.outputs, row[pb] * 8u + 4u) = lut[32 + k];
                    }
                }
                if (HEADS) {
                    const unsigned ho = row[pb] * 32u;
                    {
                        const float v[9] = {L[4][0], L[4][1], L[4][2], L[4][3], L[5][0], L[5][1], L[5][2], L[5][3], L[8][0]};
                        if (sample_seg<9>(v, ur[pb][0], g, lane, k) && live) at_off<float>(A.heads, ho) = lut[64 + k];
                    }
                    {
                        const float v[9] = {L[6][0], L[6][1], L[6][2], L[6][3], L[7][0], L[7][1], L[7][2], L[7][3], L[8][1]};
                        if (sample_seg<9>(v, ur[pb][1], g, lane, k) && live) at_off<float>(A.heads, ho + 4u) = lut[64 + k];
                    }
                    // block 8, registers 2..3: aux_up aux_right | sx sy | sz - (dec_col)
                    if (live && g < 2) { f32x2 w; w[0] = L[8][2]; w[1] = L[8][3]; at_off<f32x2>(A.heads, ho + 8u + 8u * g) = w; }
                    if (live && g == 2) { f32x2 w; w[0] = L[8][2]; w[1] = 0.f; at_off<f32x2>(A.heads, ho + 24u) = w; }
                }
            }
        }
        cur = nxt;
        nxt = nxt2;
        if (cur >= wg_end) break;
    }
}

template <bool LOGITS, bool DECODE, bool HEADS, bool SEL = false>
__global__ __launch_bounds__(MLP_THREADS, MLP_WAVES_PER_SIMD) void pair_mlp_kernel(MlpArgs A)
{
    pair_mlp_body<LOGITS, DECODE, HEADS, SEL>(A, (int)blockIdx.x, (int)gridDim.x);
}

// Several pair lists in ONE launch (cppf_pair_mlp_decode_batch): workgroups [wg_begin[i], wg_begin[i + 1]) work on list i, so every
// base pointer -- and the weight image: the lists may belong to different networks -- stays workgroup-uniform and the tile loop is
// the single-list one.  What it buys: the ~9 us before a workgroup's first MFMA (weights -> LDS, the first cold index -> gather
// chain) are paid per LAUNCH, not per list.
#define MLP_BATCH_MAX 8
struct MlpBatch {
    MlpArgs item[MLP_BATCH_MAX];
    int wg_begin[MLP_BATCH_MAX + 1];
    int n;
    int per_xcd;   // > 0: lists of (nearly) equal length, 8 % n == 0: list i on XCDs [i per_xcd, (i + 1) per_xcd) -- see below
};
static_assert(sizeof(ProjBatch) <= 4096, "ProjBatch travels by value: kernel arguments are limited to 4 KB");
static_assert(sizeof(MlpBatch) <= 4096, "MlpBatch travels by value: kernel arguments are limited to 4 KB");
template <bool HEADS, bool SEL = false>
__global__ __launch_bounds__(MLP_THREADS, MLP_WAVES_PER_SIMD) void pair_mlp_batch_kernel(MlpBatch B)
{
    if (!SEL && B.per_xcd > 0) {
        // Workgroups go to XCD blockIdx mod 8 and every XCD has its own 4 MB L2: with a list's workgroups spread over all XCDs each L2
        // sees every list's 2 MB table and thrashes (PMC: 4.5x the HBM / MALL fetches of one launch per list).  With 1, 2, 4 or 8
        // lists of equal length a list keeps to its own XCDs.
        const int x = (int)blockIdx.x & 7, r = (int)blockIdx.x >> 3;
        const int i = x / B.per_xcd;
        pair_mlp_body<false, true, HEADS, false>(B.item[i], r * B.per_xcd + x % B.per_xcd, ((int)gridDim.x >> 3) * B.per_xcd);
        return;
    }
    int i = 0;
    while (i + 1 < B.n && (int)blockIdx.x >= B.wg_begin[i + 1]) ++i;
    pair_mlp_body<false, true, HEADS, SEL>(B.item[i], (int)blockIdx.x - B.wg_begin[i], B.wg_begin[i + 1] - B.wg_begin[i]);
}

// ----------------------------------------------------------------------------- generic kernel
// Any ResLayer stack up to 128 units wide: one pair per lane, activations in LDS ([k][lane], no
// bank conflicts), weights read through the scalar path (wave-uniform addresses), natural k order
// (oracle order 0), fmaf chain seeded by the bias.
struct GenArgs {
    const float* pc;
    const float* nrm;
    const float* feat;
    const void* idxs;
    const float* packed;
    float* out;
    int64_t P;
    int F, n_res, out_dim, idx64;
    int dims[GEN_MAX_RES + 1];
};
#define GEN_THREADS 64
__global__ __launch_bounds__(GEN_THREADS) void pair_mlp_generic_kernel(GenArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float S[];
    int maxd = 0;
    for (int i = 0; i <= A.n_res; ++i) maxd = max(maxd, A.dims[i]);
    float* X = S;                          // [maxd][64]
    float* H = S + maxd * GEN_THREADS;     // [maxd][64]
    const int t = threadIdx.x;
    for (int64_t p = (int64_t)blockIdx.x * GEN_THREADS + t; p < (A.P + GEN_THREADS - 1) / GEN_THREADS * GEN_THREADS;
         p += (int64_t)gridDim.x * GEN_THREADS) {
        const bool live = p < A.P;
        const int64_t pc_ = live ? p : A.P - 1;
        int ia, ib;
        if (A.idx64) {
            const longlong2 v = reinterpret_cast<const longlong2*>(A.idxs)[pc_];
            ia = (int)v.x; ib = (int)v.y;
        } else {
            const int2 v = reinterpret_cast<const int2*>(A.idxs)[pc_];
            ia = v.x; ib = v.y;
        }
        const int F = A.F;
        for (int k = 0; k < F; ++k) {
            X[k * GEN_THREADS + t] = A.feat[(int64_t)ia * F + k];
            X[(F + k) * GEN_THREADS + t] = A.feat[(int64_t)ib * F + k];
        }
        for (int c = 0; c < 4; ++c) X[(2 * F + c) * GEN_THREADS + t] = ppf_component(A.pc, A.nrm, ia, ib, c);
        const float* w = A.packed;
        for (int l = 0; l < A.n_res; ++l) {
            const int K = A.dims[l], M = A.dims[l + 1];
            const float *w1 = w, *b1 = w1 + M * K, *w2 = b1 + M, *b2 = w2 + M * M;
            const float *w0 = b2 + M, *b0 = w0 + M * K;
            const bool has0 = K != M;
            // H = relu(fc1 X)
            for (int o = 0; o < M; ++o) {
                float acc = b1[o];
                for (int k = 0; k < K; ++k) acc = fmaf(w1[o * K + k], X[k * GEN_THREADS + t], acc);
                H[o * GEN_THREADS + t] = acc > 0.f ? acc : 0.f;
            }
            // Y = fc2 H + (fc0 X | X): every output needs all of X, so Y is a third buffer
            for (int o = 0; o < M; ++o) {
                float acc = b2[o];
                for (int k = 0; k < M; ++k) acc = fmaf(w2[o * M + k], H[k * GEN_THREADS + t], acc);
                float skip;
                if (has0) {
                    float a0 = b0[o];
                    for (int k = 0; k < K; ++k) a0 = fmaf(w0[o * K + k], X[k * GEN_THREADS + t], a0);
                    skip = a0;
                } else {
                    skip = X[o * GEN_THREADS + t];
                }
                S[(2 * maxd + o) * GEN_THREADS + t] = acc + skip;
            }
            for (int o = 0; o < M; ++o) X[o * GEN_THREADS + t] = S[(2 * maxd + o) * GEN_THREADS + t];
            w = has0 ? b0 + M : b2 + M;
        }
        const int K = A.dims[A.n_res];
        const float *wf = w, *bf = wf + A.out_dim * K;
        for (int o = 0; o < A.out_dim; ++o) {
            float acc = bf[o];
            for (int k = 0; k < K; ++k) acc = fmaf(wf[o * K + k], X[k * GEN_THREADS + t], acc);
            if (live) A.out[p * A.out_dim + o] = acc;
        }
    }
}

// ----------------------------------------------------------------------------- decode from memory
// One lane per (pair, head): same arithmetic as sample_head / oracle orc_sample_bin.
__device__ int sample_bin_mem(const float* __restrict__ l, int nb, float u, int col0)
{
    (void)col0;
    float m = l[0];
    int am = 0;
    for (int k = 1; k < nb; ++k)
        if (l[k] > m) { m = l[k]; am = k; }
    if (u < 0.f) return am;
    const int NL = (nb + 3) / 4;
    const float c = -(m * CPPF_LOG2E);
    float T[4];
    for (int g = 0; g < 4; ++g) {
        float acc = 0.f;
        for (int k = g * NL; k < (g + 1) * NL && k < nb; ++k) {
            const float ek = det_exp2w(l[k], c);
            acc = k == g * NL ? ek : acc + ek;
        }
        T[g] = acc;
    }
    const float s01 = T[0] + T[1], s23 = T[2] + T[3];
    const float tt = u * (s01 + s23);
    const float off[4] = {0.f, T[0], s01, s01 + T[2]};
    for (int g = 0; g < 4; ++g) {
        const float t = tt - off[g];
        float b = 0.f;
        for (int k = g * NL; k < (g + 1) * NL && k < nb; ++k) {
            const float ek = det_exp2w(l[k], c);
            b = k == g * NL ? ek : b + ek;
            if (b > t) return k;
        }
    }
    return nb - 1;
}


__global__ __launch_bounds__(256) void decode_center_kernel(const float* __restrict__ logits, int64_t P, int ld, int nb,
                                                            float vr0, float vr1, const float* __restrict__ u,
                                                            float* __restrict__ outputs)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (pair, head)
    if (i >= 2 * P) return;
    const int64_t p = i >> 1;
    const int h = (int)(i & 1);
    const int k = sample_bin_mem(logits + p * ld + h * nb, nb, u[i], h * nb);
    const float d = (float)(nb - 1);
    outputs[i] = h == 0 ? ((float)k / d * 2.0f) * vr0 - vr0 : (float)k / d * vr1;
}

__global__ __launch_bounds__(256) void decode_rot_kernel(const float* __restrict__ logits, int64_t P, int ld, int out_dim,
                                                         int tb, int rb, const float* __restrict__ u,
                                                         float* __restrict__ heads)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * P) return;
    const int64_t p = i >> 1;
    const int h = (int)(i & 1);
    const float* l = logits + p * ld;
    const int k = sample_bin_mem(l + 2 * tb + h * rb, rb, u[i], 2 * tb + h * rb);
    float* o = heads + 8 * p;
    o[h] = (float)k / (float)(rb - 1) * (float)CPPF_PI;
    if (h == 0) { o[2] = l[out_dim - 5]; o[3] = l[out_dim - 4]; o[4] = l[out_dim - 3]; }
    else { o[5] = l[out_dim - 2]; o[6] = l[out_dim - 1]; o[7] = 0.f; }
}

// ----------------------------------------------------------------------------- entry points
static int mlp_grid(int64_t P)
{
    const int64_t tiles = (P + 16 * PB - 1) / (16 * PB);
    int64_t nb = (tiles + MLP_THREADS / 64 - 1) / (MLP_THREADS / 64);
    const int64_t resident = 256 * (MLP_WAVES_PER_SIMD * 4 / (MLP_THREADS / 64));  // workgroups the chip holds at once
    if (nb > resident) nb = resident;
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" size_t cppf_pair_mlp_workspace_bytes(int64_t N, int F, const int* dims, int n_res, int out_dim)
{
    if (!dims || N < 0) return 0;
    return is_std(F, dims, n_res, out_dim) ? (size_t)N * PROJ_COLS * sizeof(float) : 0;
}

template <bool LOGITS, bool DECODE, bool HEADS, bool SEL = false>
static int launch_std(MlpArgs& A, int64_t N, void* workspace, size_t workspace_bytes, hipStream_t st)
{
    if (N < 1) return CPPF_EINVAL;
    if (N >= (1ll << 23) || A.P >= (1ll << 27)) return CPPF_EUNSUPPORTED;   // 32-bit byte offsets inside the kernel
    if (!workspace || workspace_bytes < (size_t)N * PROJ_COLS * sizeof(float)) return CPPF_EWORKSPACE;
    float* table = static_cast<float*>(workspace);
    if (!SEL) {   // (the SEL pass reuses the table the first pass left in the same workspace)
        hipLaunchKernelGGL(point_proj_kernel, dim3((unsigned)((N + PROJ_PPB - 1) / PROJ_PPB)), dim3(256), 0, st, A.feat, A.packed, table, N);
        CPPF_CHECK_LAUNCH();
    }
    A.table = table;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mlp_kernel<LOGITS, DECODE, HEADS, SEL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (STD_LDS + 128) * sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((pair_mlp_kernel<LOGITS, DECODE, HEADS, SEL>), dim3(mlp_grid(A.P)), dim3(MLP_THREADS),
                       (STD_LDS + 128) * sizeof(float), st, A);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_pair_mlp_forward(const float* pc, const float* nrm, const float* feat, const void* idxs,
                                     int idx_is_i64, const float* packed, int64_t N, int F, const int* dims, int n_res,
                                     int64_t P, int out_dim, float* out, void* workspace, size_t workspace_bytes,
                                     void* stream)
{
    if (P < 0 || !dims) return CPPF_EINVAL;
    if (P == 0) return 0;  // empty pair list: nothing to do, pointers may be null
    if (!pc || !nrm || !feat || !idxs || !packed || !out) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (is_std(F, dims, n_res, out_dim)) {
        MlpArgs A = {};
        A.pc = pc; A.nrm = nrm; A.feat = feat; A.idxs = idxs; A.packed = packed; A.out = out; A.P = P;
        A.out_dim = out_dim; A.idx64 = idx_is_i64;
        return launch_std<true, false, false>(A, N, workspace, workspace_bytes, st);
    }
    if (!gen_ok(F, dims, n_res, out_dim)) return CPPF_EUNSUPPORTED;
    GenArgs G = {};
    G.pc = pc; G.nrm = nrm; G.feat = feat; G.idxs = idxs; G.packed = packed; G.out = out; G.P = P;
    G.F = F; G.n_res = n_res; G.out_dim = out_dim; G.idx64 = idx_is_i64;
    int maxd = 0;
    for (int i = 0; i <= n_res; ++i) { G.dims[i] = dims[i]; if (dims[i] > maxd) maxd = dims[i]; }
    const size_t lds = (size_t)3 * maxd * GEN_THREADS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mlp_generic_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 3 * GEN_MAX_DIM * GEN_THREADS * sizeof(float));
        attr_done = true;
    }
    int64_t nb = (P + GEN_THREADS - 1) / GEN_THREADS;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pair_mlp_generic_kernel, dim3((unsigned)nb), dim3(GEN_THREADS), lds, st, G);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_pair_mlp_decode(const float* pc, const float* nrm, const float* feat, const void* idxs,
                                    int idx_is_i64, const float* packed, int64_t N, int F, const int* dims, int n_res,
                                    int64_t P, int out_dim, int tr_bins, int rot_bins, float vr0, float vr1,
                                    const float* u_tr, const float* u_rot, float* outputs, float* heads,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    if (P < 0 || !dims) return CPPF_EINVAL;
    if (!is_std(F, dims, n_res, out_dim) || tr_bins != 32 || rot_bins != 36 || out_dim != 141) return CPPF_EUNSUPPORTED;
    if (P == 0) return 0;
    if (!pc || !nrm || !feat || !idxs || !packed || !u_tr || !outputs) return CPPF_EINVAL;
    if ((heads != nullptr) != (u_rot != nullptr)) return CPPF_EINVAL;
    MlpArgs A = {};
    A.pc = pc; A.nrm = nrm; A.feat = feat; A.idxs = idxs; A.packed = packed; A.P = P; A.out_dim = out_dim;
    A.idx64 = idx_is_i64; A.u_tr = u_tr; A.u_rot = u_rot; A.outputs = outputs; A.heads = heads; A.vr0 = vr0; A.vr1 = vr1;
    return heads ? launch_std<false, true, true>(A, N, workspace, workspace_bytes, (hipStream_t)stream)
                 : launch_std<false, true, false>(A, N, workspace, workspace_bytes, (hipStream_t)stream);
}

// The launch geometry of cppf_pair_mlp_decode_batch for lists of n_pairs[i] pairs: workgroups in proportion to the lists' tiles, at
// least one each, as many in all as one list alone would get (wg_begin[n + 1]); with 1, 2, 4 or 8 lists within 10 % of each other
// and a grid of at least 8 the XCD-aware form (per_xcd = 8 / n > 0, the grid rounded down to a multiple of 8).
static void batch_plan(int n_items, const int64_t* n_pairs, int* wg_begin, int* per_xcd, int* grid)
{
    int64_t tiles_all = 0, tmin = INT64_MAX, tmax = 0;
    for (int i = 0; i < n_items; ++i) {
        const int64_t tiles = (n_pairs[i] + 15) / 16;
        tiles_all += tiles;
        tmin = tiles < tmin ? tiles : tmin; tmax = tiles > tmax ? tiles : tmax;
    }
    const int total = mlp_grid(tiles_all * 16) < n_items ? n_items : mlp_grid(tiles_all * 16);
    int given = 0;
    for (int i = 0; i < n_items; ++i) {
        const int64_t tiles = (n_pairs[i] + 15) / 16;
        int w = (int)((tiles * total + tiles_all - 1) / tiles_all);
        const int left = total - given - (n_items - 1 - i);
        if (w > left) w = left;
        if (w < 1) w = 1;
        wg_begin[i] = given;
        given += w;
    }
    wg_begin[n_items] = given;
    *per_xcd = 0;
    if (8 % n_items == 0 && tmax * 10 <= tmin * 11 && given >= 8) {
        *per_xcd = 8 / n_items;
        given = given / 8 * 8;
    }
    *grid = given;
}

extern "C" int cppf_pair_mlp_batch_plan(int n_items, const int64_t* n_pairs, int* per_xcd, int* grid, int* wg_begin)
{
    if (n_items < 1 || n_items > MLP_BATCH_MAX || !n_pairs || !per_xcd || !grid) return CPPF_EINVAL;
    for (int i = 0; i < n_items; ++i)
        if (n_pairs[i] < 1 || n_pairs[i] >= (1ll << 27)) return CPPF_EINVAL;
    int wb[MLP_BATCH_MAX + 1];
    batch_plan(n_items, n_pairs, wb, per_xcd, grid);
    if (wg_begin)
        for (int i = 0; i <= n_items; ++i) wg_begin[i] = wb[i];
    return 0;
}

extern "C" int cppf_pair_mlp_decode_batch(int n_items, const CppfPairMlpItem* items, int F, const int* dims, int n_res, int out_dim,
                                          int tr_bins, int rot_bins, void* stream)
{
    if (n_items < 1 || n_items > MLP_BATCH_MAX || !items || !dims) return CPPF_EINVAL;
    if (!is_std(F, dims, n_res, out_dim) || tr_bins != 32 || rot_bins != 36 || out_dim != 141) return CPPF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    MlpBatch B = {};
    B.n = n_items;
    const bool heads = items[0].heads != nullptr;
    int64_t n_pairs[MLP_BATCH_MAX];
    int given = 0;
    for (int i = 0; i < n_items; ++i) {
        const CppfPairMlpItem& it = items[i];
        n_pairs[i] = it.n_pairs;
        if (it.n_pairs < 1 || it.n_points < 1) return CPPF_EINVAL;
        if (!it.pc || !it.nrm || !it.feat || !it.idxs || !it.packed || !it.u_tr || !it.outputs) return CPPF_EINVAL;
        if ((it.heads != nullptr) != heads || (it.heads != nullptr) != (it.u_rot != nullptr)) return CPPF_EINVAL;
        if (it.n_points >= (1ll << 23) || it.n_pairs >= (1ll << 27)) return CPPF_EUNSUPPORTED;
        if (!it.workspace || it.workspace_bytes < (size_t)it.n_points * PROJ_COLS * sizeof(float)) return CPPF_EWORKSPACE;
    }
    batch_plan(n_items, n_pairs, B.wg_begin, &B.per_xcd, &given);
    ProjBatch PJ = {};
    int64_t proj_blocks = 1;
    for (int i = 0; i < n_items; ++i) {
        const CppfPairMlpItem& it = items[i];
        float* table = static_cast<float*>(it.workspace);
        PJ.feat[i] = it.feat; PJ.packed[i] = it.packed; PJ.table[i] = table; PJ.N[i] = it.n_points;
        const int64_t nb = (it.n_points + PROJ_PPB - 1) / PROJ_PPB;
        proj_blocks = nb > proj_blocks ? nb : proj_blocks;
        MlpArgs& A = B.item[i];
        A.pc = it.pc; A.nrm = it.nrm; A.feat = it.feat; A.idxs = it.idxs; A.packed = it.packed; A.P = it.n_pairs; A.out_dim = out_dim;
        A.idx64 = it.idx_is_i64; A.u_tr = it.u_tr; A.u_rot = it.u_rot; A.outputs = it.outputs; A.heads = it.heads;
        A.vr0 = it.vr0; A.vr1 = it.vr1; A.table = table;
    }
    // the per-point projections of all lists in ONE launch (round 5: n launches of ~5 us each in front of the pair kernel before)
    hipLaunchKernelGGL(point_proj_batch_kernel, dim3((unsigned)proj_blocks, (unsigned)n_items), dim3(256), 0, st, PJ);
    CPPF_CHECK_LAUNCH();
    static bool attr_done[2] = {false, false};
    const void* fn = heads ? reinterpret_cast<const void*>(&pair_mlp_batch_kernel<true>) : reinterpret_cast<const void*>(&pair_mlp_batch_kernel<false>);
    if (!attr_done[heads]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (STD_LDS + 128) * sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_done[heads] = true;
    }
    if (heads) hipLaunchKernelGGL((pair_mlp_batch_kernel<true>), dim3((unsigned)given), dim3(MLP_THREADS), (STD_LDS + 128) * sizeof(float), st, B);
    else hipLaunchKernelGGL((pair_mlp_batch_kernel<false>), dim3((unsigned)given), dim3(MLP_THREADS), (STD_LDS + 128) * sizeof(float), st, B);
    CPPF_CHECK_LAUNCH();
    return 0;
}

// The second pass (cppf_pair_mlp_decode_sel) for several pair lists in ONE launch: workgroups [w_i, w_{i+1}) work on list i's
// survivors; every list's per-point table is the one its first pass left in item.workspace.  Items with max_sel == 0 are skipped.
extern "C" int cppf_pair_mlp_decode_sel_batch(int n_items, const CppfPairMlpItem* items, int F, const int* dims, int n_res, int out_dim,
                                              int tr_bins, int rot_bins, void* stream)
{
    if (n_items < 1 || n_items > MLP_BATCH_MAX || !items || !dims) return CPPF_EINVAL;
    if (!is_std(F, dims, n_res, out_dim) || tr_bins != 32 || rot_bins != 36 || out_dim != 141) return CPPF_EUNSUPPORTED;
    MlpBatch B = {};
    int given = 0;
    for (int i = 0; i < n_items; ++i) {
        const CppfPairMlpItem& it = items[i];
        if (it.n_pairs < 0 || it.max_sel < 0) return CPPF_EINVAL;
        if (it.n_pairs == 0 || it.max_sel == 0) continue;
        if (it.n_pairs >= (1ll << 27) || it.n_points >= (1ll << 23)) return CPPF_EUNSUPPORTED;
        if (!it.pc || !it.nrm || !it.feat || !it.idxs || !it.packed || !it.u_rot || !it.sel || !it.n_sel_dev || !it.heads) return CPPF_EINVAL;
        if (!it.workspace || it.workspace_bytes < (size_t)it.n_points * PROJ_COLS * sizeof(float)) return CPPF_EWORKSPACE;
        MlpArgs& A = B.item[B.n];
        A.pc = it.pc; A.nrm = it.nrm; A.feat = it.feat; A.idxs = it.idxs; A.packed = it.packed; A.out_dim = out_dim;
        A.P = it.max_sel < it.n_pairs ? it.max_sel : it.n_pairs;
        A.idx64 = it.idx_is_i64; A.u_rot = it.u_rot; A.heads = it.heads; A.sel = it.sel; A.n_sel = it.n_sel_dev;
        A.table = static_cast<const float*>(it.workspace);
        B.wg_begin[B.n] = given;
        // (a launch cannot know the survivor counts: every list gets the workgroups of its capacity, at most a quarter of the chip's
        // round when several lists share it; surplus workgroups exit at once)
        int w = mlp_grid(A.P);
        const int cap = mlp_grid(1ll << 26) / (n_items > 1 ? 2 : 1);
        if (w > cap) w = cap;
        given += w;
        ++B.n;
    }
    if (B.n == 0) return 0;
    B.wg_begin[B.n] = given;
    static bool attr_done = false;
    const void* fn = reinterpret_cast<const void*>(&pair_mlp_batch_kernel<true, true>);
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (STD_LDS + 128) * sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((pair_mlp_batch_kernel<true, true>), dim3((unsigned)given), dim3(MLP_THREADS), (STD_LDS + 128) * sizeof(float),
                       (hipStream_t)stream, B);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_pair_mlp_decode_sel(const float* pc, const float* nrm, const float* feat, const void* idxs,
                                        int idx_is_i64, const float* packed, int64_t N, int F, const int* dims, int n_res,
                                        int64_t P, int out_dim, int tr_bins, int rot_bins, const float* u_rot,
                                        const int32_t* sel, const int32_t* n_sel_dev, int64_t max_sel, float* heads,
                                        void* workspace, size_t workspace_bytes, void* stream)
{
    if (P < 0 || max_sel < 0 || !dims) return CPPF_EINVAL;
    if (!is_std(F, dims, n_res, out_dim) || tr_bins != 32 || rot_bins != 36 || out_dim != 141) return CPPF_EUNSUPPORTED;
    if (P == 0 || max_sel == 0) return 0;
    if (P >= (1ll << 27)) return CPPF_EUNSUPPORTED;   // rows come from sel[] and range over the whole pair list: 32-bit offsets
    if (!pc || !nrm || !feat || !idxs || !packed || !u_rot || !sel || !n_sel_dev || !heads) return CPPF_EINVAL;
    MlpArgs A = {};
    A.pc = pc; A.nrm = nrm; A.feat = feat; A.idxs = idxs; A.packed = packed; A.out_dim = out_dim;
    A.P = max_sel < P ? max_sel : P;   // slots of the launch (capacity); the device count bounds what runs
    A.idx64 = idx_is_i64; A.u_rot = u_rot; A.heads = heads; A.sel = sel; A.n_sel = n_sel_dev;
    return launch_std<false, true, true, true>(A, N, workspace, workspace_bytes, (hipStream_t)stream);
}

#ifdef CPPF_DEBUG_ENTRY
// Profiling aid (built with -DCPPF_DEBUG_ENTRY only, not in the shipped library): the PPF + gather + MFMA chain with no epilogue.
extern "C" int cppf_debug_mlp_chain_only(const float* pc, const float* nrm, const float* feat, const void* idxs,
                                         int idx_is_i64, const float* packed, int64_t N, int64_t P, float* scratch,
                                         void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pc || !nrm || !feat || !idxs || !packed || !scratch || P < 1) return CPPF_EINVAL;
    MlpArgs A = {};
    A.pc = pc; A.nrm = nrm; A.feat = feat; A.idxs = idxs; A.packed = packed; A.P = P; A.out_dim = 141;
    A.idx64 = idx_is_i64; A.outputs = scratch;
    return launch_std<false, false, false>(A, N, workspace, workspace_bytes, (hipStream_t)stream);
}
#endif

extern "C" int cppf_decode_center(const float* logits, int64_t P, int ld, int tr_bins, float vr0, float vr1,
                                  const float* u_tr, float* outputs, void* stream)
{
    if (P < 0 || tr_bins < 2 || ld < 2 * tr_bins) return CPPF_EINVAL;
    if (P == 0) return 0;
    if (!logits || !u_tr || !outputs) return CPPF_EINVAL;
    hipLaunchKernelGGL(decode_center_kernel, dim3((unsigned)((2 * P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, P, ld, tr_bins, vr0, vr1, u_tr, outputs);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_decode_rot(const float* logits, int64_t P, int ld, int out_dim, int tr_bins, int rot_bins,
                               const float* u_rot, float* heads, void* stream)
{
    if (P < 0 || rot_bins < 2 || out_dim < 2 * tr_bins + 2 * rot_bins + 5 || ld < out_dim) return CPPF_EINVAL;
    if (P == 0) return 0;
    if (!logits || !u_rot || !heads) return CPPF_EINVAL;
    hipLaunchKernelGGL(decode_rot_kernel, dim3((unsigned)((2 * P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, P, ld, out_dim, tr_bins, rot_bins, u_rot, heads);
    CPPF_CHECK_LAUNCH();
    return 0;
}
