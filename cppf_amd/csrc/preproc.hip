// Per-instance pre-processing in front of the CPPF path, on device (C ABI in include/cppf.h).
//
// Replaces two third-party host calls of the reference's instance loop:
//   voxel de-duplication  ME.utils.sparse_quantize(pc, return_index=True, quantization_size=res)   nocs/inference.py:140
//   normals               open3d estimate_normals(KDTreeSearchParamKNN(knn))                        utils/util.py:61-65
// Neither library is in this image and neither result is fully specified (which point represents a voxel; the sign of
// a normal), so parity with the reference is UNPINNED; the definitions used here (lowest index per voxel, largest
// component positive) are restated in oracle/preproc_oracle.c and the kernels match that file bit for bit.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <string.h>
#include "../../include/cppf.h"
#include "compact.h"

namespace {

__global__ __launch_bounds__(256) void vox_keys_kernel(const float* __restrict__ pc, int64_t N, double res, int64_t* __restrict__ keys,
                                                       int32_t* __restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int64_t k = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t v = (int64_t)floor((double)pc[3 * i + c] / res) + (1 << 20);
        k = (k << 21) | (v & ((1 << 21) - 1));
    }
    keys[i] = k;
    vals[i] = (int32_t)i;
}
// the first entry of every run of equal keys (stable sort: the lowest index of the voxel) marks its point
__global__ __launch_bounds__(256) void vox_mark_kernel(const int64_t* __restrict__ skeys, const int32_t* __restrict__ svals, int64_t N,
                                                       uint8_t* __restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (i == 0 || skeys[i] != skeys[i - 1]) mask[svals[i]] = 1;
}

// utils/util.py:598-631 backproject: valid = mask & (depth > 0) (:609-610); pixels in row-major order (np.where, :612);
// xyz = inv(K) @ (u, v, 1) (:622); pts = xyz * z / xyz.z (:628); x and y negated (:629-630).  fp64 like numpy; the 3-term
// products as k0*u, then fma(k1, v, .), then + k2 (oracle/preproc_oracle.c:orc_backproject is the same sequence).
template <typename T>
__global__ __launch_bounds__(256) void bp_valid_kernel(const T* __restrict__ depth, const uint8_t* __restrict__ mask, int64_t n,
                                                       uint8_t* __restrict__ valid)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) valid[i] = (mask[i] != 0) && (depth[i] > (T)0);
}
struct Kinv { double k[9]; };
template <typename T>
__global__ __launch_bounds__(256) void bp_points_kernel(const T* __restrict__ depth, const int32_t* __restrict__ pix,
                                                        const int32_t* __restrict__ count, int W, Kinv K,
                                                        double* __restrict__ pts)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= *count) return;
    const int p = pix[i];
    const double u = (double)(p % W), v = (double)(p / W), z = (double)depth[p];
    double xyz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xyz[c] = fma(K.k[3 * c + 1], v, K.k[3 * c] * u) + K.k[3 * c + 2];
    pts[3 * i] = -(xyz[0] * z / xyz[2]);
    pts[3 * i + 1] = -(xyz[1] * z / xyz[2]);
    pts[3 * i + 2] = xyz[2] * z / xyz[2];
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 matrix by 8 cyclic Jacobi sweeps in fp64 (oracle/preproc_oracle.c)
__device__ void smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double* out)
{
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = r == 2 ? 1 : 0, q = r == 0 ? 1 : 2;
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    const double v[3] = {m == 0 ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]), m == 0 ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]),
                         m == 0 ? V[2][0] : (m == 1 ? V[2][1] : V[2][2])};
    const double n = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (!(n > 0.0)) { out[0] = 0; out[1] = 0; out[2] = 1; return; }
    int big = 0;
    if (fabs(v[1]) > fabs(v[big])) big = 1;
    if (fabs(v[2]) > fabs(v[big])) big = 2;
    const double vb = big == 0 ? v[0] : (big == 1 ? v[1] : v[2]);
    const double sg = vb < 0.0 ? -1.0 : 1.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = sg * (v[c] / n);
}

__global__ __launch_bounds__(256) void normals_kernel(const float* __restrict__ pc, const int32_t* __restrict__ nbrs, int64_t N, int k,
                                                      float* __restrict__ normals)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const float* p = pc + 3 * (int64_t)nbrs[n * k + j];
        const double x = p[0], y = p[1], z = p[2];
        c[0] += x; c[1] += y; c[2] += z;
        c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] /= (double)k;
    double v[3];
    smallest_eigvec(c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2], c[6] - c[1] * c[1], c[7] - c[1] * c[2],
                    c[8] - c[2] * c[2], v);
#pragma unroll
    for (int i = 0; i < 3; ++i) normals[3 * n + i] = (float)v[i];
}

// ---- the whole per-instance pre-processing as ONE count-driven stage (cppf_frame_cloud_dyn): no size ever visits the host.
// valid pixel = bit `bit` of the frame's label image set and depth > 0 (utils/util.py:609-610 with instance_mask = that bit)
template <typename T, typename LT>
__global__ __launch_bounds__(256) void fc_valid_kernel(const T* __restrict__ depth, const LT* __restrict__ labels, unsigned bit, int64_t n,
                                                       uint8_t* __restrict__ valid, const int32_t* __restrict__ bit_dev)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (bit_dev) bit = (unsigned)*bit_dev & (8u * (unsigned)sizeof(LT) - 1u);      // (a captured launch: this replay's bit is in memory)
    if (i < n) valid[i] = ((labels[i] >> bit) & 1) && (depth[i] > (T)0);
}
// back-projection of the compacted pixels (bp_points_kernel's arithmetic), nocs/inference.py:132 `pc = pts / 1000.0` (fp64), the axis
// flips of :136-137 (negations of utils/util.py:629-630's negations: exact), `.float()` of :140 -- and the voxel key of every slot:
// slots beyond the count get the all-ones key, which sorts behind every real one
// Voxel de-duplication without a sort (rocprim's radix sort inside a captured graph gave different results from the second
// replay on: ROCm 7.2): an open-addressing table of M = 2^m >= 2 n_cap slots {voxel key, lowest point index}.  Every point inserts
// its key (atomicCAS on the key word, linear probing) and atomicMin's its index into the slot; a point represents its voxel iff it
// IS that minimum -- cppf_voxel_dedupe's definition (lowest index per voxel, output in index order), whatever order the atomics land in.
#define FC_EMPTY (~0ull)
__device__ __forceinline__ unsigned fc_hash(unsigned long long k, unsigned mask)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k & mask;
}
__global__ __launch_bounds__(256) void fc_clear_kernel(unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tidx, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) { tkeys[i] = FC_EMPTY; tidx[i] = 0x7fffffff; }
}
// back-projection of the compacted pixels (bp_points_kernel's arithmetic), nocs/inference.py:132 `pc = pts / 1000.0` (fp64), the axis
// flips of :136-137 (negations of utils/util.py:629-630's negations: exact), `.float()` of :140, the voxel key (vox_keys_kernel) and
// the point's entry in the table
template <typename T>
__device__ __forceinline__ void fc_points_body(const T* __restrict__ depth, const int32_t* __restrict__ pix,
                                               const int32_t* __restrict__ count, int W, const Kinv& K, double divisor, double res,
                                               int n_cap, float* __restrict__ pcf, unsigned long long* __restrict__ keys,
                                               unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tidx, unsigned tmask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(*count, n_cap)) return;
    const int p = pix[i];
    const double u = (double)(p % W), v = (double)(p / W), z = (double)depth[p];
    double xyz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xyz[c] = fma(K.k[3 * c + 1], v, K.k[3 * c] * u) + K.k[3 * c + 2];
    const double pts[3] = {-(xyz[0] * z / xyz[2]), -(xyz[1] * z / xyz[2]), xyz[2] * z / xyz[2]};
    const float f[3] = {(float)-(pts[0] / divisor), (float)-(pts[1] / divisor), (float)(pts[2] / divisor)};
    unsigned long long k = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pcf[3 * i + c] = f[c];
        const int64_t q = (int64_t)floor((double)f[c] / res) + (1 << 20);     // (vox_keys_kernel)
        k = (k << 21) | (unsigned long long)(q & ((1 << 21) - 1));
    }
    keys[i] = k;
    for (unsigned h = fc_hash(k, tmask);; h = (h + 1) & tmask) {
        const unsigned long long prev = atomicCAS(&tkeys[h], FC_EMPTY, k);
        if (prev == FC_EMPTY || prev == k) { atomicMin(&tidx[h], i); break; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void fc_points_kernel(const T* __restrict__ depth, const int32_t* __restrict__ pix,
                                                        const int32_t* __restrict__ count, int W, Kinv K, double divisor, double res,
                                                        int n_cap, float* __restrict__ pcf, unsigned long long* __restrict__ keys,
                                                        unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tidx, unsigned tmask)
{
    fc_points_body<T>(depth, pix, count, W, K, divisor, res, n_cap, pcf, keys, tkeys, tidx, tmask);
}
// does point i represent its voxel (the lowest index among the points of its key)?
__device__ __forceinline__ bool fc_is_representative(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ tkeys,
                                                     const int32_t* __restrict__ tidx, unsigned tmask, int i)
{
    const unsigned long long k = keys[i];
    unsigned h = fc_hash(k, tmask);
    while (tkeys[h] != k) h = (h + 1) & tmask;      // (the key is in the table: this point put it there or found it there)
    return tidx[h] == i;
}
__global__ __launch_bounds__(256) void fc_mark_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ tkeys,
                                                      const int32_t* __restrict__ tidx, unsigned tmask, const int32_t* __restrict__ count,
                                                      int n_cap, uint8_t* __restrict__ mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_cap) return;
    uint8_t m = 0;
    if (i < min(*count, n_cap)) m = fc_is_representative(keys, tkeys, tidx, tmask, i);
    mask[i] = m;
}
// pc = pc[keep] (:141) into the pipeline's cloud buffer; the instance's point count N (0 when it is below k_min: the reference
// skips such instances, :121-123) goes to shape[0], where the kernels behind this stage read it
__global__ __launch_bounds__(256) void fc_gather_kernel(const float* __restrict__ pcf, const int32_t* __restrict__ keep,
                                                        const int32_t* __restrict__ count, int k_min, int n_cap, float* __restrict__ pc_out,
                                                        int32_t* __restrict__ shape)
{
    const int n = min(*count, n_cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) shape[0] = n >= k_min ? n : 0;
    if (i >= n) return;
    const int s = keep[i];
    pc_out[3 * i] = pcf[3 * s]; pc_out[3 * i + 1] = pcf[3 * s + 1]; pc_out[3 * i + 2] = pcf[3 * s + 2];
}
__device__ __forceinline__ void fc_normals_body(const float* __restrict__ pc, const int32_t* __restrict__ nbrs,
                                                const int32_t* __restrict__ n_dev, int k, float* __restrict__ normals)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= *n_dev) return;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const float* p = pc + 3 * (int64_t)nbrs[n * k + j];
        const double x = p[0], y = p[1], z = p[2];
        c[0] += x; c[1] += y; c[2] += z;
        c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] /= (double)k;
    double v[3];
    smallest_eigvec(c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2], c[6] - c[1] * c[1], c[7] - c[1] * c[2],
                    c[8] - c[2] * c[2], v);
#pragma unroll
    for (int i = 0; i < 3; ++i) normals[3 * n + i] = (float)v[i];
}
__global__ __launch_bounds__(256) void fc_normals_kernel(const float* __restrict__ pc, const int32_t* __restrict__ nbrs,
                                                         const int32_t* __restrict__ n_dev, int k, float* __restrict__ normals)
{
    fc_normals_body(pc, nbrs, n_dev, k, normals);
}
// nocs/inference.py:194-195 from the device count: corner = min(pc), dims = int32((max - min) / res) + 1 -> shape[1..3]
// (minima and maxima: exact in any order, whatever the block size)
__device__ __forceinline__ void fc_grid_body(const float* __restrict__ pc, float res, float* __restrict__ corner, int32_t* __restrict__ shape)
{
    __shared__ float slo[16][3], shi[16][3];
    const int N = shape[0];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < N; i += (int)blockDim.x)
        for (int j = 0; j < 3; ++j) {
            const float v = pc[3 * i + j];
            lo[j] = fminf(lo[j], v);
            hi[j] = fmaxf(hi[j], v);
        }
    for (int j = 0; j < 3; ++j)
        for (int off = 32; off > 0; off >>= 1) {
            lo[j] = fminf(lo[j], __shfl_xor(lo[j], off, 64));
            hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], off, 64));
        }
    if ((threadIdx.x & 63) == 0)
        for (int j = 0; j < 3; ++j) { slo[threadIdx.x >> 6][j] = lo[j]; shi[threadIdx.x >> 6][j] = hi[j]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int j = threadIdx.x;
        float l = slo[0][j], h = shi[0][j];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { l = fminf(l, slo[w][j]); h = fmaxf(h, shi[w][j]); }
        corner[j] = N > 0 ? l : 0.f;
        shape[1 + j] = N > 0 ? (int32_t)((h - l) / res) + 1 : 1;
    }
}
__global__ __launch_bounds__(1024) void fc_grid_kernel(const float* __restrict__ pc, float res, float* __restrict__ corner,
                                                       int32_t* __restrict__ shape)
{
    fc_grid_body(pc, res, corner, shape);
}
// idx = idx mod N (both columns): pairs drawn as full-range integers on the device before the instance's N is known anywhere but here
__global__ __launch_bounds__(256) void mod_pairs_kernel(long long* __restrict__ idx, int64_t n2, const int32_t* __restrict__ n_dev)
{
    const long long N = (long long)*n_dev;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const unsigned long long r = (unsigned long long)idx[i];
        idx[i] = N > 0 ? (long long)(r % (unsigned long long)N) : 0ll;
    }
}

struct FcLayout { size_t valid, cmp1, pix, count, pcf, keys, tkeys, tidx, mask2, cmp2, keep, nbrs, total; int M; };
FcLayout fc_layout(int H, int W, int n_cap, int k)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t n = (size_t)H * W;
    FcLayout L;
    L.M = 256;
    while (L.M < 2 * n_cap) L.M <<= 1;
    L.valid = 0;
    L.cmp1 = L.valid + up(n);
    L.pix = L.cmp1 + up(cppf_compact_workspace_bytes((int64_t)n));
    L.count = L.pix + up(n * sizeof(int32_t));
    L.pcf = L.count + 256;                                        // count1 at +0, count2 at +64
    L.keys = L.pcf + up((size_t)n_cap * 3 * sizeof(float));
    L.tkeys = L.keys + up((size_t)n_cap * sizeof(unsigned long long));
    L.tidx = L.tkeys + up((size_t)L.M * sizeof(unsigned long long));
    L.mask2 = L.tidx + up((size_t)L.M * sizeof(int32_t));
    L.cmp2 = L.mask2 + up((size_t)n_cap);
    L.keep = L.cmp2 + up(cppf_compact_workspace_bytes(n_cap));
    L.nbrs = L.keep + up((size_t)n_cap * sizeof(int32_t));
    L.total = L.nbrs + up((size_t)n_cap * (size_t)k * sizeof(int32_t));
    return L;
}

// ---- pair list + bin uniforms drawn on the device (cppf_sample_pairs): Philox-4x32-10 (Salmon et al., SC'11), counter = pair index,
// key = the caller's 64-bit seed; stateless, so a pair's draw depends on (seed, pair index) only -- whichever rank or stream draws it.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
        c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k.y, (unsigned)p0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
__global__ __launch_bounds__(256) void sample_pairs_kernel(long long* __restrict__ idx, float* __restrict__ u_tr, float* __restrict__ u_rot,
                                                           int64_t P, int64_t n_points, const int32_t* __restrict__ n_dev,
                                                           unsigned long long seed, const unsigned long long* __restrict__ seed_dev)
{
    const unsigned long long N = (unsigned long long)(n_dev ? (int64_t)*n_dev : n_points);
    if (seed_dev) seed = *seed_dev;      // (a captured launch: the seed of this replay is in memory)
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        const uint4 a = philox4x32_10(make_uint4((unsigned)p, (unsigned)(p >> 32), 0u, 0u), key);
        const uint4 b = philox4x32_10(make_uint4((unsigned)p, (unsigned)(p >> 32), 1u, 0u), key);
        // index = floor(r N / 2^32): uniform over [0, N) up to a bias of N / 2^32 (np.random.randint(0, N, (P, 2)), nocs/inference.py:177)
        reinterpret_cast<longlong2*>(idx)[p] = make_longlong2((long long)(((unsigned long long)a.x * N) >> 32),
                                                              (long long)(((unsigned long long)a.y * N) >> 32));
        // uniforms in [0, 1): the top 24 bits (stand-ins for torch.multinomial's draws, :186,250,254)
        if (u_tr) reinterpret_cast<float2*>(u_tr)[p] = make_float2((float)(a.z >> 8) * 0x1p-24f, (float)(a.w >> 8) * 0x1p-24f);
        if (u_rot) reinterpret_cast<float2*>(u_rot)[p] = make_float2((float)(b.x >> 8) * 0x1p-24f, (float)(b.y >> 8) * 0x1p-24f);
    }
}

// ---- cppf_stage_batch: the head of a captured chain for objects already on the device.  Workgroup roles by blockIdx.x, object =
// blockIdx.y: 0 = grid set-up (nocs/inference.py:194-195, the arithmetic of grid_setup_kernel), 1 .. STAGE_COPY_BLOCKS = copies of cloud,
// normals and features into the chain's buffers, the rest = the pair / uniform draws (sample_pairs_kernel's, bit for bit).
#define STAGE_MAX 8
#define STAGE_COPY_BLOCKS 16
struct StageBatch { CppfStageItem item[STAGE_MAX]; int n_sample_blocks; };
__global__ __launch_bounds__(256) void stage_batch_kernel(StageBatch B)
{
    const CppfStageItem& I = B.item[blockIdx.y];
    const CppfStageDesc D = *I.desc;
    const int64_t N = D.n_points < 0 ? 0 : (D.n_points > I.n_cap ? I.n_cap : D.n_points);
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        __shared__ float slo[4][3], shi[4][3];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int64_t i = tid; i < N; i += 256)
            for (int j = 0; j < 3; ++j) {
                const float v = D.pc_src[3 * i + j];
                lo[j] = fminf(lo[j], v);
                hi[j] = fmaxf(hi[j], v);
            }
        for (int j = 0; j < 3; ++j)
            for (int off = 32; off > 0; off >>= 1) {
                lo[j] = fminf(lo[j], __shfl_xor(lo[j], off, 64));
                hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], off, 64));
            }
        if ((tid & 63) == 0)
            for (int j = 0; j < 3; ++j) { slo[tid >> 6][j] = lo[j]; shi[tid >> 6][j] = hi[j]; }
        __syncthreads();
        if (tid < 3) {
            float l = slo[0][tid], h = shi[0][tid];
            for (int w = 1; w < 4; ++w) { l = fminf(l, slo[w][tid]); h = fmaxf(h, shi[w][tid]); }
            I.corner[tid] = N > 0 ? l : 0.f;
            if (I.shape) I.shape[1 + tid] = N > 0 ? (int32_t)((h - l) / I.res) + 1 : 1;
        }
        if (tid == 3 && I.shape) I.shape[0] = (int32_t)N;
        return;
    }
    if (blockIdx.x <= STAGE_COPY_BLOCKS) {
        const int64_t n3 = 3 * N, nf = I.feat && D.feat_src ? (int64_t)I.F * N : 0, total = 2 * n3 + nf;
        for (int64_t i = (int64_t)(blockIdx.x - 1) * 256 + tid; i < total; i += (int64_t)STAGE_COPY_BLOCKS * 256) {
            if (i < n3) I.pc[i] = D.pc_src[i];
            else if (i < 2 * n3) I.nrm[i - n3] = D.nrm_src[i - n3];
            else I.feat[i - 2 * n3] = D.feat_src[i - 2 * n3];
        }
        return;
    }
    if (!I.idx) return;
    const unsigned long long Nu = (unsigned long long)N;
    const uint2 key = make_uint2((unsigned)D.seed, (unsigned)(D.seed >> 32));
    const int64_t first = (int64_t)(blockIdx.x - 1 - STAGE_COPY_BLOCKS) * 256 + tid, step = (int64_t)B.n_sample_blocks * 256;
    for (int64_t p = first; p < I.n_pairs; p += step) {
        const uint4 a = philox4x32_10(make_uint4((unsigned)p, (unsigned)(p >> 32), 0u, 0u), key);
        const uint4 b = philox4x32_10(make_uint4((unsigned)p, (unsigned)(p >> 32), 1u, 0u), key);
        const unsigned long long i0 = ((unsigned long long)a.x * Nu) >> 32, i1 = ((unsigned long long)a.y * Nu) >> 32;
        if (I.idx_is_i64) reinterpret_cast<longlong2*>(I.idx)[p] = make_longlong2((long long)i0, (long long)i1);
        else reinterpret_cast<int2*>(I.idx)[p] = make_int2((int)i0, (int)i1);
        if (I.u_tr) reinterpret_cast<float2*>(I.u_tr)[p] = make_float2((float)(a.z >> 8) * 0x1p-24f, (float)(a.w >> 8) * 0x1p-24f);
        if (I.u_rot) reinterpret_cast<float2*>(I.u_rot)[p] = make_float2((float)(b.x >> 8) * 0x1p-24f, (float)(b.y >> 8) * 0x1p-24f);
    }
}

// ---- cppf_frame_cloud_dyn_batch: the frame stage of up to 8 instances of ONE frame in eight launches instead of sixteen each.
// A frame's chain is launch-bound on the host (a hipGraph launch costs the host per kernel node: three chains of two instances, 52
// nodes each, took 1.2 of a frame's 1.6 ms to enqueue) and its ~15 pre-processing kernels per instance run for ~5 us each.  Here a
// launch serves every member (blockIdx.y) and neighbouring steps share launches: the mask kernels count their own 1 024-byte chunks
// (no count + scan launches: compact.h), the first one clears the voxel table on the way, the normals launch also sets up the grid
// and draws the pairs.  Per member the arithmetic is that of cppf_frame_cloud_dyn + cppf_sample_pairs (the same device functions).
#define FCB_MAX 8
struct FcbItem {
    const int32_t* bit_dev; const unsigned long long* seed_dev;
    uint8_t *valid, *mask2;
    int32_t *cc1, *cc2, *pix, *count1, *count2, *tidx, *keep, *nbrs, *shape_out;
    float *pcf, *pc_out, *nrm_out, *corner_out, *u_tr, *u_rot;
    unsigned long long *keys, *tkeys;
    void* idx;
    double res;
    long long n_pairs;
    int knn_k, k_min, n_cap, M, idx_is_i64;
};
struct FcbBatch { FcbItem item[FCB_MAX]; const void* depth; const void* labels; long long n_pix; Kinv K; double divisor; int W, label_bytes, n_sample_blocks; };
static_assert(sizeof(FcbBatch) <= 4096, "FcbBatch travels by value: kernel arguments are limited to 4 KB");

// 1. valid pixels of every member's label bit + their chunk counts; the member's voxel table cleared on the way
template <typename T>
__global__ __launch_bounds__(CMP_BLOCK) void fcb_valid_kernel(FcbBatch B)
{
    const FcbItem& I = B.item[blockIdx.y];
    const long long i = (long long)blockIdx.x * CMP_BLOCK + threadIdx.x;
    for (long long k = i; k < I.M; k += (long long)gridDim.x * CMP_BLOCK) { I.tkeys[k] = FC_EMPTY; I.tidx[k] = 0x7fffffff; }
    bool f = false;
    if (i < B.n_pix) {
        const T d = static_cast<const T*>(B.depth)[i];
        unsigned lab;
        if (B.label_bytes == 1) lab = static_cast<const uint8_t*>(B.labels)[i];
        else if (B.label_bytes == 2) lab = static_cast<const uint16_t*>(B.labels)[i];
        else lab = static_cast<const uint32_t*>(B.labels)[i];
        const unsigned bit = (unsigned)*I.bit_dev & (8u * (unsigned)B.label_bytes - 1u);
        f = ((lab >> bit) & 1u) && (d > (T)0);
        I.valid[i] = f;
    }
    const int s = compact_chunk_count(f);
    if (threadIdx.x == 0) I.cc1[blockIdx.x] = s;
}
// 2. / 5. the compactions (np.where order, :612; pc[keep], :140-141)
__global__ __launch_bounds__(CMP_BLOCK) void fcb_scatter_kernel(FcbBatch B, int second)
{
    const FcbItem& I = B.item[blockIdx.y];
    if (second) compact_scatter_self_body(I.mask2, I.n_cap, I.cc2, I.keep, I.count2);
    else compact_scatter_self_body(I.valid, B.n_pix, I.cc1, I.pix, I.count1);
}
// 3. back-projection, voxel keys, table inserts
template <typename T>
__global__ __launch_bounds__(256) void fcb_points_kernel(FcbBatch B)
{
    const FcbItem& I = B.item[blockIdx.y];
    fc_points_body<T>(static_cast<const T*>(B.depth), I.pix, I.count1, B.W, B.K, B.divisor, I.res, I.n_cap, I.pcf, I.keys, I.tkeys, I.tidx,
                      (unsigned)I.M - 1u);
}
// 4. the representatives of the voxels + their chunk counts
__global__ __launch_bounds__(CMP_BLOCK) void fcb_mark_kernel(FcbBatch B)
{
    const FcbItem& I = B.item[blockIdx.y];
    const int i = blockIdx.x * CMP_BLOCK + threadIdx.x;
    if ((long long)blockIdx.x * CMP_BLOCK >= I.n_cap) return;      // (a shorter member: workgroup-uniform)
    bool f = false;
    if (i < I.n_cap) {
        if (i < min(*I.count1, I.n_cap)) f = fc_is_representative(I.keys, I.tkeys, I.tidx, (unsigned)I.M - 1u, i);
        I.mask2[i] = f;
    }
    const int s = compact_chunk_count(f);
    if (threadIdx.x == 0) I.cc2[blockIdx.x] = s;
}
// 6. pc = pc[keep], the point count
__global__ __launch_bounds__(256) void fcb_gather_kernel(FcbBatch B)
{
    const FcbItem& I = B.item[blockIdx.y];
    const int n = min(*I.count2, I.n_cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) I.shape_out[0] = n >= I.k_min ? n : 0;
    if (i >= n) return;
    const int sidx = I.keep[i];
    I.pc_out[3 * i] = I.pcf[3 * sidx]; I.pc_out[3 * i + 1] = I.pcf[3 * sidx + 1]; I.pc_out[3 * i + 2] = I.pcf[3 * sidx + 2];
}
// 8. (7 = the neighbour search, sprin.hip) normals on blocks [0, nbc), the grid set-up on block nbc, the pair / uniform draws behind it
__global__ __launch_bounds__(256) void fcb_finish_kernel(FcbBatch B, int nbc)
{
    const FcbItem& I = B.item[blockIdx.y];
    if ((int)blockIdx.x < nbc) { fc_normals_body(I.pc_out, I.nbrs, I.shape_out, I.knn_k, I.nrm_out); return; }
    if ((int)blockIdx.x == nbc) { fc_grid_body(I.pc_out, (float)I.res, I.corner_out, I.shape_out); return; }
    if (!I.idx) return;
    const unsigned long long N = (unsigned long long)(long long)I.shape_out[0], seed = *I.seed_dev;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    const long long first = (long long)((int)blockIdx.x - nbc - 1) * 256 + threadIdx.x, step = (long long)B.n_sample_blocks * 256;
    for (long long p = first; p < I.n_pairs; p += step) {
        const uint4 a = philox4x32_10(make_uint4((unsigned)p, (unsigned)(p >> 32), 0u, 0u), key);
        const uint4 b = philox4x32_10(make_uint4((unsigned)p, (unsigned)(p >> 32), 1u, 0u), key);
        const unsigned long long i0 = ((unsigned long long)a.x * N) >> 32, i1 = ((unsigned long long)a.y * N) >> 32;
        if (I.idx_is_i64) reinterpret_cast<longlong2*>(I.idx)[p] = make_longlong2((long long)i0, (long long)i1);
        else reinterpret_cast<int2*>(I.idx)[p] = make_int2((int)i0, (int)i1);
        if (I.u_tr) reinterpret_cast<float2*>(I.u_tr)[p] = make_float2((float)(a.z >> 8) * 0x1p-24f, (float)(a.w >> 8) * 0x1p-24f);
        if (I.u_rot) reinterpret_cast<float2*>(I.u_rot)[p] = make_float2((float)(b.x >> 8) * 0x1p-24f, (float)(b.y >> 8) * 0x1p-24f);
    }
}

// cppf_copy_words: a few 64-bit words moved by a kernel of the stream instead of by a copy engine.  hipMemcpyAsync hands small copies to
// the SDMA engines, whose queues are shared between streams and served in order: a 200-byte descriptor copy of lane 0 then waits
// behind the caller's stream's read-back, which waits for the previous batch -- a false dependency that (depending on which engine
// the runtime picked in this process) serialised consecutive batches.  Either pointer may be pinned host memory (device-accessible).
__global__ __launch_bounds__(256) void copy_words_kernel(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

// cppf_gather_words: row r of dst <- the first n_words 64-bit words at src[r] (up to 32 rows, one launch): the records / shape words of
// a chain's members, each in its own pipeline's buffers, into one array -- instead of one small copy per member
#define GATHER_MAX 32
struct GatherRows { const unsigned long long* src[GATHER_MAX]; };
__global__ __launch_bounds__(64) void gather_words_kernel(GatherRows G, unsigned long long* __restrict__ dst, int n_words)
{
    const unsigned long long* s_ = G.src[blockIdx.x];
    for (int t = threadIdx.x; t < n_words; t += 64) dst[(size_t)blockIdx.x * n_words + t] = s_[t];
}

struct VoxLayout { size_t keys, vals, mask, compact, temp, temp_bytes, total; };
VoxLayout vox_layout(int64_t N)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    VoxLayout L;
    L.keys = 0;
    L.vals = L.keys + up(2 * (size_t)N * sizeof(int64_t));   // keys + sorted keys
    L.mask = L.vals + up(2 * (size_t)N * sizeof(int32_t));   // vals + sorted vals
    L.compact = L.mask + up((size_t)N);
    L.temp = L.compact + up(cppf_compact_workspace_bytes(N));
    L.temp_bytes = up((size_t)48 * N + (1u << 20));
    L.total = L.temp + L.temp_bytes;
    return L;
}

}  // namespace

extern "C" {

size_t cppf_voxel_dedupe_workspace_bytes(int64_t n_points) { return n_points < 0 ? 0 : vox_layout(n_points).total; }

int cppf_voxel_dedupe(const float* pc, int64_t n_points, double res, int32_t* keep_idx, int32_t* count, void* workspace,
                      size_t workspace_bytes, void* stream)
{
    if (n_points < 0 || n_points > 0x7fffffffll || !(res > 0.0) || !count) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n_points == 0) return (int)hipMemsetAsync(count, 0, sizeof(int32_t), st);
    if (!pc || !keep_idx) return CPPF_EINVAL;
    const VoxLayout L = vox_layout(n_points);
    if (!workspace || workspace_bytes < L.total) return CPPF_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    int64_t *keys = (int64_t*)(ws + L.keys), *skeys = keys + n_points;
    int32_t *vals = (int32_t*)(ws + L.vals), *svals = vals + n_points;
    uint8_t* mask = (uint8_t*)(ws + L.mask);
    const int nb = (int)((n_points + 255) / 256);
    vox_keys_kernel<<<nb, 256, 0, st>>>(pc, n_points, res, keys, vals);
    size_t need = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, skeys, vals, svals, (int)n_points, 0, 63, st);
    if (e != hipSuccess) return (int)e;
    if (need > L.temp_bytes) return CPPF_EWORKSPACE;
    need = L.temp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(ws + L.temp, need, keys, skeys, vals, svals, (int)n_points, 0, 63, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(mask, 0, (size_t)n_points, st);
    if (e != hipSuccess) return (int)e;
    vox_mark_kernel<<<nb, 256, 0, st>>>(skeys, svals, n_points, mask);
    return cppf_compact_mask(mask, n_points, keep_idx, count, ws + L.compact, cppf_compact_workspace_bytes(n_points), stream);
}

size_t cppf_backproject_workspace_bytes(int H, int W)
{
    if (H < 1 || W < 1) return 0;
    const int64_t n = (int64_t)H * W;
    return (size_t)((n + 255) / 256 * 256) + cppf_compact_workspace_bytes(n);
}

int cppf_backproject(const void* depth, int depth_is_u16, const uint8_t* mask, int H, int W, const double* kinv_host,
                     double* pts, int32_t* pix, int32_t* count, void* workspace, size_t workspace_bytes, void* stream)
{
    if (H < 1 || W < 1 || (int64_t)H * W > 0x7fffffffll || !depth || !mask || !kinv_host || !pts || !pix || !count)
        return CPPF_EINVAL;
    if (!workspace || workspace_bytes < cppf_backproject_workspace_bytes(H, W)) return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)H * W;
    uint8_t* valid = static_cast<uint8_t*>(workspace);
    char* cws = static_cast<char*>(workspace) + (n + 255) / 256 * 256;
    const int nb = (int)((n + 255) / 256);
    if (depth_is_u16) bp_valid_kernel<uint16_t><<<nb, 256, 0, st>>>((const uint16_t*)depth, mask, n, valid);
    else bp_valid_kernel<float><<<nb, 256, 0, st>>>((const float*)depth, mask, n, valid);
    int rc = cppf_compact_mask(valid, n, pix, count, cws, cppf_compact_workspace_bytes(n), stream);
    if (rc) return rc;
    Kinv K;
    for (int i = 0; i < 9; ++i) K.k[i] = kinv_host[i];
    if (depth_is_u16) bp_points_kernel<uint16_t><<<nb, 256, 0, st>>>((const uint16_t*)depth, pix, count, W, K, pts);
    else bp_points_kernel<float><<<nb, 256, 0, st>>>((const float*)depth, pix, count, W, K, pts);
    return (int)hipGetLastError();
}

size_t cppf_frame_cloud_workspace_bytes(int H, int W, int n_cap, int knn_k)
{
    if (H < 1 || W < 1 || n_cap < 1 || knn_k < 1) return 0;
    return fc_layout(H, W, n_cap, knn_k).total;
}

static int frame_cloud_impl(const void* depth, int depth_is_u16, const void* labels, int label_bytes, int label_bit, const int32_t* bit_dev,
                            int H, int W, const double* kinv_host, double divisor, double res, int knn_k, int k_min, int n_cap, float* pc_out,
                            float* nrm_out, float* corner_out, int32_t* shape_out, int32_t* nbrs_out, void* workspace, size_t workspace_bytes,
                            void* stream)
{
    if (H < 1 || W < 1 || (int64_t)H * W > 0x7fffffffll || !depth || !labels || !kinv_host || !pc_out || !nrm_out || !corner_out || !shape_out)
        return CPPF_EINVAL;
    if ((label_bytes != 1 && label_bytes != 2 && label_bytes != 4) || label_bit < 0 || label_bit >= 8 * label_bytes) return CPPF_EINVAL;
    if (n_cap < 1 || knn_k < 1 || knn_k > 64 || knn_k > n_cap || k_min < knn_k || !(res > 0.0) || !(divisor > 0.0)) return CPPF_EINVAL;
    const FcLayout L = fc_layout(H, W, n_cap, knn_k);
    if (!workspace || workspace_bytes < L.total) return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = static_cast<char*>(workspace);
    const int64_t n = (int64_t)H * W;
    uint8_t* valid = (uint8_t*)(ws + L.valid);
    int32_t* pix = (int32_t*)(ws + L.pix);
    int32_t *count1 = (int32_t*)(ws + L.count), *count2 = (int32_t*)(ws + L.count + 64);
    float* pcf = (float*)(ws + L.pcf);
    unsigned long long *keys = (unsigned long long*)(ws + L.keys), *tkeys = (unsigned long long*)(ws + L.tkeys);
    int32_t* tidx = (int32_t*)(ws + L.tidx);
    uint8_t* mask2 = (uint8_t*)(ws + L.mask2);
    int32_t *keep = (int32_t*)(ws + L.keep), *nbrs = nbrs_out ? nbrs_out : (int32_t*)(ws + L.nbrs);
    const int nbp = (int)((n + 255) / 256), nbc = (n_cap + 255) / 256;
    const unsigned bit = (unsigned)label_bit, tmask = (unsigned)L.M - 1u;
#define FC_VALID(T)                                                                                                                       \
    do {                                                                                                                                  \
        if (label_bytes == 1) fc_valid_kernel<T, uint8_t><<<nbp, 256, 0, st>>>((const T*)depth, (const uint8_t*)labels, bit, n, valid, bit_dev);   \
        else if (label_bytes == 2) fc_valid_kernel<T, uint16_t><<<nbp, 256, 0, st>>>((const T*)depth, (const uint16_t*)labels, bit, n, valid, bit_dev); \
        else fc_valid_kernel<T, uint32_t><<<nbp, 256, 0, st>>>((const T*)depth, (const uint32_t*)labels, bit, n, valid, bit_dev);                  \
    } while (0)
    if (depth_is_u16) FC_VALID(uint16_t); else FC_VALID(float);
#undef FC_VALID
    int rc = cppf_compact_mask(valid, n, pix, count1, ws + L.cmp1, cppf_compact_workspace_bytes(n), stream);        // np.where order (:612)
    if (rc) return rc;
    Kinv K;
    for (int i = 0; i < 9; ++i) K.k[i] = kinv_host[i];
    fc_clear_kernel<<<(L.M + 255) / 256, 256, 0, st>>>(tkeys, tidx, L.M);
    if (depth_is_u16) fc_points_kernel<uint16_t><<<nbc, 256, 0, st>>>((const uint16_t*)depth, pix, count1, W, K, divisor, res, n_cap, pcf, keys, tkeys, tidx, tmask);
    else fc_points_kernel<float><<<nbc, 256, 0, st>>>((const float*)depth, pix, count1, W, K, divisor, res, n_cap, pcf, keys, tkeys, tidx, tmask);
    fc_mark_kernel<<<nbc, 256, 0, st>>>(keys, tkeys, tidx, tmask, count1, n_cap, mask2);
    rc = cppf_compact_mask(mask2, n_cap, keep, count2, ws + L.cmp2, cppf_compact_workspace_bytes(n_cap), stream);     // :140
    if (rc) return rc;
    fc_gather_kernel<<<nbc, 256, 0, st>>>(pcf, keep, count2, k_min, n_cap, pc_out, shape_out);                       // :141
    rc = cppf_knn_dyn(pc_out, n_cap, shape_out, knn_k, nbrs, stream);                                                 // :142
    if (rc) return rc;
    fc_normals_kernel<<<nbc, 256, 0, st>>>(pc_out, nbrs, shape_out, knn_k, nrm_out);
    fc_grid_kernel<<<1, 1024, 0, st>>>(pc_out, (float)res, corner_out, shape_out);                                    // :194-195
    return (int)hipGetLastError();
}

int cppf_frame_cloud_dyn(const void* depth, int depth_is_u16, const void* labels, int label_bytes, int label_bit, int H, int W,
                         const double* kinv_host, double divisor, double res, int knn_k, int k_min, int n_cap, float* pc_out,
                         float* nrm_out, float* corner_out, int32_t* shape_out, int32_t* nbrs_out, void* workspace, size_t workspace_bytes,
                         void* stream)
{
    return frame_cloud_impl(depth, depth_is_u16, labels, label_bytes, label_bit, nullptr, H, W, kinv_host, divisor, res, knn_k, k_min, n_cap,
                            pc_out, nrm_out, corner_out, shape_out, nbrs_out, workspace, workspace_bytes, stream);
}

int cppf_frame_cloud_dyn_bit(const void* depth, int depth_is_u16, const void* labels, int label_bytes, const int32_t* label_bit_dev, int H,
                             int W, const double* kinv_host, double divisor, double res, int knn_k, int k_min, int n_cap, float* pc_out,
                             float* nrm_out, float* corner_out, int32_t* shape_out, int32_t* nbrs_out, void* workspace,
                             size_t workspace_bytes, void* stream)
{
    if (!label_bit_dev) return CPPF_EINVAL;
    return frame_cloud_impl(depth, depth_is_u16, labels, label_bytes, 0, label_bit_dev, H, W, kinv_host, divisor, res, knn_k, k_min, n_cap,
                            pc_out, nrm_out, corner_out, shape_out, nbrs_out, workspace, workspace_bytes, stream);
}

int cppf_frame_cloud_dyn_batch(int n_items, const CppfFrameCloudItem* items, const void* depth, int depth_is_u16, const void* labels,
                               int label_bytes, int H, int W, const double* kinv_host, double divisor, void* stream)
{
    if (n_items < 1 || n_items > FCB_MAX || !items) return CPPF_EINVAL;
    if (H < 1 || W < 1 || (int64_t)H * W > 0x7fffffffll || !depth || !labels || !kinv_host || !(divisor > 0.0)) return CPPF_EINVAL;
    if (label_bytes != 1 && label_bytes != 2 && label_bytes != 4) return CPPF_EINVAL;
    const int64_t n = (int64_t)H * W;
    if ((n + CMP_BLOCK - 1) / CMP_BLOCK > CMP_SELF_MAX) return CPPF_EUNSUPPORTED;
    FcbBatch B;
    memset(&B, 0, sizeof(B));
    B.depth = depth; B.labels = labels; B.n_pix = n; B.divisor = divisor; B.W = W; B.label_bytes = label_bytes;
    for (int i = 0; i < 9; ++i) B.K.k[i] = kinv_host[i];
    CppfKnnBatchItem knn[FCB_MAX];
    int cap_max = 0;
    int64_t max_pairs = 0;
    for (int i = 0; i < n_items; ++i) {
        const CppfFrameCloudItem& it = items[i];
        if (!it.label_bit_dev || !it.pc_out || !it.nrm_out || !it.corner_out || !it.shape_out) return CPPF_EINVAL;
        if (it.n_cap < 1 || it.knn_k < 1 || it.knn_k > 64 || it.knn_k > it.n_cap || it.k_min < it.knn_k || !(it.res > 0.0)) return CPPF_EINVAL;
        if (it.n_pairs < 0 || (it.n_pairs > 0 && (!it.idx || !it.seed_dev || (reinterpret_cast<uintptr_t>(it.idx) & (it.idx_is_i64 ? 15 : 7)))))
            return CPPF_EINVAL;
        const FcLayout L = fc_layout(H, W, it.n_cap, it.knn_k);
        if (!it.workspace || it.workspace_bytes < L.total) return CPPF_EWORKSPACE;
        char* ws = static_cast<char*>(it.workspace);
        FcbItem& I = B.item[i];
        I.bit_dev = it.label_bit_dev; I.seed_dev = it.seed_dev;
        I.valid = (uint8_t*)(ws + L.valid); I.mask2 = (uint8_t*)(ws + L.mask2);
        I.cc1 = (int32_t*)(ws + L.cmp1); I.cc2 = (int32_t*)(ws + L.cmp2); I.pix = (int32_t*)(ws + L.pix);
        I.count1 = (int32_t*)(ws + L.count); I.count2 = (int32_t*)(ws + L.count + 64);
        I.tidx = (int32_t*)(ws + L.tidx); I.keep = (int32_t*)(ws + L.keep); I.nbrs = it.nbrs_out ? it.nbrs_out : (int32_t*)(ws + L.nbrs);
        I.shape_out = it.shape_out; I.pcf = (float*)(ws + L.pcf); I.pc_out = it.pc_out; I.nrm_out = it.nrm_out; I.corner_out = it.corner_out;
        I.u_tr = it.u_tr; I.u_rot = it.u_rot; I.keys = (unsigned long long*)(ws + L.keys); I.tkeys = (unsigned long long*)(ws + L.tkeys);
        I.idx = it.n_pairs > 0 ? it.idx : nullptr; I.res = it.res; I.n_pairs = it.n_pairs;
        I.knn_k = it.knn_k; I.k_min = it.k_min; I.n_cap = it.n_cap; I.M = L.M; I.idx_is_i64 = it.idx_is_i64;
        knn[i] = CppfKnnBatchItem{it.pc_out, I.nbrs, it.shape_out, it.n_cap, it.knn_k};
        cap_max = it.n_cap > cap_max ? it.n_cap : cap_max;
        max_pairs = it.n_pairs > max_pairs ? it.n_pairs : max_pairs;
    }
    hipStream_t st = (hipStream_t)stream;
    const unsigned ni = (unsigned)n_items;
    const unsigned nbp = (unsigned)((n + CMP_BLOCK - 1) / CMP_BLOCK), nbc = (unsigned)((cap_max + 255) / 256),
                   nbc4 = (unsigned)((cap_max + CMP_BLOCK - 1) / CMP_BLOCK);
    if (depth_is_u16) fcb_valid_kernel<uint16_t><<<dim3(nbp, ni), CMP_BLOCK, 0, st>>>(B);
    else fcb_valid_kernel<float><<<dim3(nbp, ni), CMP_BLOCK, 0, st>>>(B);
    fcb_scatter_kernel<<<dim3(nbp, ni), CMP_BLOCK, 0, st>>>(B, 0);
    if (depth_is_u16) fcb_points_kernel<uint16_t><<<dim3(nbc, ni), 256, 0, st>>>(B);
    else fcb_points_kernel<float><<<dim3(nbc, ni), 256, 0, st>>>(B);
    fcb_mark_kernel<<<dim3(nbc4, ni), CMP_BLOCK, 0, st>>>(B);
    fcb_scatter_kernel<<<dim3(nbc4, ni), CMP_BLOCK, 0, st>>>(B, 1);
    fcb_gather_kernel<<<dim3(nbc, ni), 256, 0, st>>>(B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rc = cppf_internal_knn_batch(n_items, knn, stream);
    if (rc) return rc;
    int64_t nsb = (max_pairs + 511) / 512;          // two pairs per thread at the longest list
    nsb = nsb > 2048 ? 2048 : nsb;
    B.n_sample_blocks = (int)(nsb < 1 ? 1 : nsb);
    fcb_finish_kernel<<<dim3(nbc + 1 + (max_pairs > 0 ? (unsigned)B.n_sample_blocks : 0u), ni), 256, 0, st>>>(B, (int)nbc);
    return (int)hipGetLastError();
}

int cppf_sample_pairs(long long* idx, float* u_tr, float* u_rot, int64_t n_pairs, int64_t n_points, const int32_t* n_dev,
                      unsigned long long seed, const unsigned long long* seed_dev, void* stream)
{
    if (n_pairs < 0 || (n_pairs > 0 && !idx) || (!n_dev && (n_points < 1 || n_points > 0x7fffffffll))) return CPPF_EINVAL;
    if (n_pairs == 0) return 0;
    int64_t nb = (n_pairs + 255) / 256;
    if (nb > 4096) nb = 4096;
    sample_pairs_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(idx, u_tr, u_rot, n_pairs, n_points, n_dev, seed, seed_dev);
    return (int)hipGetLastError();
}

int cppf_copy_words(void* dst, const void* src, int64_t n_words, void* stream)
{
    if (n_words < 0 || (n_words > 0 && (!dst || !src)) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 7)) return CPPF_EINVAL;
    if (n_words == 0) return 0;
    int64_t nb = (n_words + 255) / 256;
    if (nb > 1024) nb = 1024;
    copy_words_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(static_cast<unsigned long long*>(dst), static_cast<const unsigned long long*>(src), n_words);
    return (int)hipGetLastError();
}

int cppf_gather_words(int n_rows, const void* const* src_host, int64_t n_words, void* dst, void* stream)
{
    if (n_rows < 0 || n_rows > GATHER_MAX || n_words < 0 || n_words > 0x7fffffffll) return CPPF_EINVAL;
    if (n_rows == 0 || n_words == 0) return 0;
    if (!src_host || !dst || (reinterpret_cast<uintptr_t>(dst) & 7)) return CPPF_EINVAL;
    GatherRows G = {};
    for (int r = 0; r < n_rows; ++r) {
        if (!src_host[r] || (reinterpret_cast<uintptr_t>(src_host[r]) & 7)) return CPPF_EINVAL;
        G.src[r] = static_cast<const unsigned long long*>(src_host[r]);
    }
    gather_words_kernel<<<n_rows, 64, 0, (hipStream_t)stream>>>(G, static_cast<unsigned long long*>(dst), (int)n_words);
    return (int)hipGetLastError();
}

int cppf_stage_batch(int n_items, const CppfStageItem* items, void* stream)
{
    if (n_items < 1 || n_items > STAGE_MAX || !items) return CPPF_EINVAL;
    StageBatch B = {};
    int64_t max_pairs = 0;
    for (int i = 0; i < n_items; ++i) {
        const CppfStageItem& it = items[i];
        if (!it.desc || !it.pc || !it.nrm || !it.corner || it.n_cap < 1 || it.n_cap > 0x7fffffffll || it.n_pairs < 0 || !(it.res > 0.f) ||
            (it.feat && it.F < 1) || (it.n_pairs > 0 && it.idx && (reinterpret_cast<uintptr_t>(it.idx) & (it.idx_is_i64 ? 15 : 7))))
            return CPPF_EINVAL;
        B.item[i] = it;
        if (it.idx && it.n_pairs > max_pairs) max_pairs = it.n_pairs;
    }
    int64_t nb = (max_pairs + 511) / 512;          // two pairs per thread at the largest list
    if (nb > 2048) nb = 2048;
    B.n_sample_blocks = (int)(nb < 1 ? 1 : nb);
    const unsigned gx = 1 + STAGE_COPY_BLOCKS + (max_pairs > 0 ? (unsigned)B.n_sample_blocks : 0u);
    stage_batch_kernel<<<dim3(gx, (unsigned)n_items), 256, 0, (hipStream_t)stream>>>(B);
    return (int)hipGetLastError();
}

// nocs/inference.py:194-195 on a HOST cloud (no device involved): corners = [min(pc), max(pc)] (f32), dims = int32((max - min) / res) + 1
// with the quotient in fp32 like numpy's (cppf_amd.inference.grid_shape: 17 us of numpy per cloud, 2 us here)
int cppf_host_grid_shape(const float* pc_host, int64_t n_points, float res, float* corners_host, int32_t* dims_host)
{
    if (!pc_host || n_points < 1 || !(res > 0.f) || !corners_host || !dims_host) return CPPF_EINVAL;
    float lo[3] = {pc_host[0], pc_host[1], pc_host[2]}, hi[3] = {pc_host[0], pc_host[1], pc_host[2]};
    float bad = 0.f;                                   // v - v is 0 for a finite v, NaN for NaN and +-inf: one accumulator, no branch
    for (int64_t i = 0; i < n_points; ++i)
        for (int c = 0; c < 3; ++c) {
            const float v = pc_host[3 * i + c];
            bad += v - v;
            lo[c] = v < lo[c] ? v : lo[c];
            hi[c] = v > hi[c] ? v : hi[c];
        }
    if (bad != 0.f) return CPPF_ENONFINITE;            // (numpy's min / max would propagate the NaN: the caller fails loudly, as the reference)
    for (int c = 0; c < 3; ++c) {
        corners_host[c] = lo[c]; corners_host[3 + c] = hi[c];
        const volatile float q = (hi[c] - lo[c]) / res;
        dims_host[c] = (int32_t)q + 1;
    }
    return 0;
}

int cppf_mod_pairs_dyn(long long* idx, int64_t n_pairs, const int32_t* n_dev, void* stream)
{
    if (n_pairs < 0 || (n_pairs > 0 && (!idx || !n_dev))) return CPPF_EINVAL;
    if (n_pairs == 0) return 0;
    int64_t nb = (2 * n_pairs + 255) / 256;
    if (nb > 2048) nb = 2048;
    mod_pairs_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(idx, 2 * n_pairs, n_dev);
    return (int)hipGetLastError();
}

int cppf_estimate_normals(const float* pc, const int32_t* nbrs, int64_t n_points, int k, float* normals, void* stream)
{
    if (n_points < 0 || k < 1) return CPPF_EINVAL;
    if (n_points == 0) return 0;
    if (!pc || !nbrs || !normals) return CPPF_EINVAL;
    normals_kernel<<<(int)((n_points + 255) / 256), 256, 0, (hipStream_t)stream>>>(pc, nbrs, n_points, k, normals);
    return (int)hipGetLastError();
}

}  // extern "C"
