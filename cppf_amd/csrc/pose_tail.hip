// Everything after the centre vote's arg-max for gfx950 (MI355X): translation from the arg-max, back-vote filter, survivor
// compaction, orientation vote + sphere-bin count, axis sign / scale reductions, grid set-up (nocs/inference.py:194-195,209-339).
// C ABI in include/cppf.h; reference semantics cited per kernel.  (Split from vote.hip in round 5; the centre vote stays there.)
#include "vote_common.h"
#include "compact.h"

// nocs/inference.py:209-210: cand = unravel_index(argmax); T = corners[0] + cand * res in fp64;
// T32 is the float32 copy handed to backvote (:225).
__device__ __forceinline__ void center_from_argmax_body(const long long* __restrict__ idx, const float* __restrict__ corner, double res,
                                                        int gy, int gz, double* __restrict__ T64, float* __restrict__ T32,
                                                        const float* __restrict__ peak, double* __restrict__ idx_peak,
                                                        const int32_t* __restrict__ shape, uint4* __restrict__ zero16, int n_zero16)
{
    // cppf_pose_tail_begin: the accumulators of the launches that follow (sphere-bin counts, chunk counts, ticket, record)
    // start at zero; T64 / idx_peak may lie inside the region, so the zeroing comes first
    for (int k = threadIdx.x; k < n_zero16; k += blockDim.x) zero16[k] = make_uint4(0u, 0u, 0u, 0u);
    if (n_zero16) __syncthreads();
    if (shape) { gy = max(shape[2], 1); gz = max(shape[3], 1); }   // dims record in memory (*_dyn)
    const long long flat = *idx;
    const long long syz = (long long)gy * gz;
    const long long c[3] = {flat / syz, (flat % syz) / gz, (flat % syz) % gz};
    const int j = threadIdx.x;
    if (j < 3) {
        const double t = (double)corner[j] + (double)c[j] * res;
        if (T64) T64[j] = t;
        if (T32) T32[j] = (float)t;
    }
    if (j == 3 && idx_peak) {   // the arg-max index and its value as doubles, for the pose record
        idx_peak[0] = (double)flat;
        idx_peak[1] = peak ? (double)*peak : 0.0;
    }
}

__global__ void center_from_argmax_kernel(const long long* __restrict__ idx, const float* __restrict__ corner, double res,
                                          int gy, int gz, double* __restrict__ T64, float* __restrict__ T32,
                                          const float* __restrict__ peak, double* __restrict__ idx_peak,
                                          const int32_t* __restrict__ shape, uint4* __restrict__ zero16 = nullptr,
                                          int n_zero16 = 0)
{
    center_from_argmax_body(idx, corner, res, gy, gz, T64, T32, peak, idx_peak, shape, zero16, n_zero16);
}

// np.argmax(counts) (first maximum, nocs/inference.py:283) and best_dir = sphere_pts[argmax] (fp64) in one launch
__global__ __launch_bounds__(256) void counts_argmax_select_kernel(const int32_t* __restrict__ counts, int n,
                                                                   const double* __restrict__ sphere64,
                                                                   long long* __restrict__ best_idx, double* __restrict__ best_dir)
{
    __shared__ unsigned long long best[4];
    // key = count << 32 | ~index: the largest key is the largest count at the lowest index
    unsigned long long k = 0ull;
    for (int i = threadIdx.x; i < n; i += 256) {
        const unsigned long long ki = ((unsigned long long)(uint32_t)counts[i] << 32) | (uint32_t)(~(uint32_t)i);
        k = ki > k ? ki : k;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long b = best[0];
        for (int w = 1; w < 4; ++w) b = best[w] > b ? best[w] : b;
        const int bi = (int)(~(uint32_t)(b & 0xffffffffull));
        best_dir[threadIdx.x] = sphere64[3 * (size_t)bi + threadIdx.x];
        if (threadIdx.x == 0 && best_idx) *best_idx = bi;
    }
}

extern "C" int cppf_center_from_argmax(const long long* idx, const float* corner, double res, int gy, int gz,
                                       double* T64, float* T32, const float* peak, double* idx_peak_f64, void* stream)
{
    if (!idx || !corner || gy < 1 || gz < 1) return CPPF_EINVAL;
    hipLaunchKernelGGL(center_from_argmax_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, idx, corner, res, gy, gz,
                       T64, T32, peak, idx_peak_f64, (const int32_t*)nullptr);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_center_from_argmax_dyn(const long long* idx, const float* corner, double res, const int32_t* shape_dev,
                                           double* T64, float* T32, const float* peak, double* idx_peak_f64, void* stream)
{
    if (!idx || !corner || !shape_dev) return CPPF_EINVAL;
    hipLaunchKernelGGL(center_from_argmax_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, idx, corner, res, 1, 1,
                       T64, T32, peak, idx_peak_f64, shape_dev);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_pose_tail_begin(const long long* idx, const float* corner, double res, int gy, int gz,
                                    const int32_t* shape_dev, double* T64, float* T32, const float* peak,
                                    double* idx_peak_f64, void* zero_ptr, size_t zero_bytes, void* stream)
{
    if (!idx || !corner || (!shape_dev && (gy < 1 || gz < 1))) return CPPF_EINVAL;
    if (zero_bytes && (!zero_ptr || (zero_bytes & 15) || (reinterpret_cast<uintptr_t>(zero_ptr) & 15) || zero_bytes > (1u << 26)))
        return CPPF_EINVAL;
    hipLaunchKernelGGL(center_from_argmax_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, idx, corner, res,
                       shape_dev ? 1 : gy, shape_dev ? 1 : gz, T64, T32, peak, idx_peak_f64, shape_dev,
                       static_cast<uint4*>(zero_ptr), (int)(zero_bytes / 16));
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_counts_argmax_select(const int32_t* counts, int n, const double* sphere64, long long* best_idx,
                                         double* best_dir, void* stream)
{
    if (!counts || !sphere64 || !best_dir || n < 1) return CPPF_EINVAL;
    hipLaunchKernelGGL(counts_argmax_select_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, n, sphere64,
                       best_idx, best_dir);
    CPPF_CHECK_LAUNCH();
    return 0;
}

// ----------------------------------------------------------------------------- back-vote
// Reference: CUDA backvote, models/voting.py:74-112 (always adaptive, bounds [0, dim-1)).
// The reference's 13-argument shape has no workspace, so each (persistent, grid-stride) block
// builds the (cos,sin) table for every n <= n_rots in LDS when it fits.
__device__ __forceinline__ void backvote_body(const float* __restrict__ points,
                                                              const float* __restrict__ outputs,
                                                              float* __restrict__ out_offsets,
                                                              const int32_t* __restrict__ point_idxs,
                                                              const float* __restrict__ corner, float res, int64_t n_ppfs,
                                                              int n_rots, int gx, int gy, int gz,
                                                              const float* __restrict__ gt_center, float tol,
                                                              uint8_t* __restrict__ mask, const int32_t* __restrict__ shape,
                                                              const unsigned long long* __restrict__ vote_ws,
                                                              int32_t* __restrict__ chunk_counts,
                                                              const long long* __restrict__ idx64,
                                                              int32_t* __restrict__ idx32_out)
{
    // cppf_backvote_count64: the pair list as the caller holds it (int64, nocs/inference.py:177); the int32 copy the later
    // launches of the tail read is written on the way (this kernel touches every pair anyway)
    auto load_ij = [&](const int64_t c) -> int2 {
        if (idx64) {
            const longlong2 v = reinterpret_cast<const longlong2*>(idx64)[c];
            return make_int2((int)v.x, (int)v.y);
        }
        return reinterpret_cast<const int2*>(point_idxs)[c];
    };
    if (shape) { gx = shape[1]; gy = shape[2]; gz = shape[3]; }   // dims record in memory (*_dyn)
    // (cos,sin) table for every n <= n_rots, built per block in LDS when it fits.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float2* ltab = reinterpret_cast<float2*>(lds);
    const int entries = n_rots * (n_rots + 1) / 2;
    const bool in_lds = entries <= VOTE_TAB_LDS_MAX;
    if (in_lds) {
        // the vote that produced gt_center left the same table in its workspace (VOTE_TAB_STAMP): 21 KB to load instead of
        // 2 628 fp64 sincos per block
        const float2* wtab = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(vote_ws) + VOTE_WS_TAB);
        if (vote_ws && vote_ws[31] == (VOTE_TAB_STAMP ^ (unsigned long long)n_rots) && rot_table_intact(wtab, n_rots)) {
            for (int e = threadIdx.x; e < entries; e += blockDim.x) ltab[e] = wtab[e];
        } else {
            fill_rot_table(ltab, entries, threadIdx.x, blockDim.x);
        }
        __syncthreads();
    }
    const f3 cr = {corner[0], corner[1], corner[2]};
    const f3 gt = {gt_center[0], gt_center[1], gt_center[2]};
    const float bx = (float)(gx - 1), by = (float)(gy - 1), bz = (float)(gz - 1);
    const float rinv_res = 1.0f / res;
    // Two stages per wave.  Stage 1, one pair per lane: frame, rotation count, and the skip test -- every sample
    // lies at distance |offset| = rho (1 +- 1e-6) from cc, hence at least | |cc - gt| - rho | from gt; when that
    // exceeds tol (with a margin far above the rounding) no rotation can pass :101 and the pair is finished
    // (offset 0).  The other pairs go to a per-wave LDS queue and stage 2 runs their rotation loops 64 at a time,
    // so a wave never walks 72 rotations for the sake of one lane.
    uint32_t* q = reinterpret_cast<uint32_t*>(lds + (in_lds ? 2 * entries : 0)) + (threadIdx.x >> 6) * 128;
    const int lane = threadIdx.x & 63;
    int qn = 0;
    // cppf_backvote_count: survivors per chunk of CMP_BLOCK pairs (integer atomics: the counts do not depend on the
    // order), so the compaction that follows needs no counting pass.  The lanes that call this together mostly hold pairs
    // of one chunk: those send one atomic, the others their own.
    auto count_survivor = [&](const int64_t idx, const bool nz) {
        const int ch = (int)(idx >> 10);
        const int ch0 = __builtin_amdgcn_readfirstlane(ch);
        const unsigned long long m = __ballot(nz && ch == ch0);
        if (nz) {
            if (ch != ch0) atomicAdd(&chunk_counts[ch], 1);
            else if ((threadIdx.x & 63) == __builtin_ctzll(m)) atomicAdd(&chunk_counts[ch0], __popcll(m));
        }
    };
    auto finish = [&](const int64_t idx, const f3 found) {
        if (out_offsets) {
            float* oo = out_offsets + 3 * idx;
            oo[0] = found.x; oo[1] = found.y; oo[2] = found.z;
        }
        const bool nz = (found.x != 0.f) || (found.y != 0.f) || (found.z != 0.f);
        if (mask) mask[idx] = nz;
        if (chunk_counts) count_survivor(idx, nz);
    };
    // The reference's loop (:97-110) for one pair that passed stage 1, one pair per lane -- over the ARC of rotations that can
    // pass :101 only, in index order.  With offset = cos(t) x + sin(t) y, |x| = |y| = rho and x, y, ab orthogonal,
    //   |cc + offset - gt|^2 = |w|^2 + rho^2 - 2 (A cos t + B sin t),   w = gt - cc, A = w.x, B = w.y,
    // so the distance test holds exactly where cos(t - phi) >= K / M, phi = atan2(B, A), M = |(A, B)|,
    // K = (|w|^2 + rho^2 - tol^2) / 2: a run of indices around phi n / 2 pi, about 2 tol / res + 3 of them instead of n <= 72.
    // The run is a superset (approximate acos / atan2 with their error bounds, one index of margin on either side, K lowered
    // by the deviations of |offset|^2 from rho^2: roundings and the 1e-7 regulariser, which shortens ab by 1e-7 / L -- pairs
    // closer than 1e-3 scan everything); each candidate still goes through the reference's exact arithmetic and the first
    // one in index order wins, so the result is the reference's.  On inputs where every pair survives stage 1 (a trained
    // network) the full loop was the kernel: 100 us at C2, a lane walking on average 36 rotations and a wave its slowest lane's.
    auto rotations = [&](const int64_t idx) {
        const float2 o = reinterpret_cast<const float2*>(outputs)[idx];
        const int2 ij = load_ij(idx);
        f3 a, ab, xd;
        pair_frame(points, ij.x, ij.y, a, ab, xd);
        const float proj_len = o.x, odist = o.y;
        const f3 cc = sub3(a, scl3(ab, proj_len));
        const f3 x = scl3(xd, odist);
        const f3 y = cross3(x, ab);
        f3 found = {0.f, 0.f, 0.f};                                                   // :96
        const int n = min((int)((double)(odist / res) * (2 * CPPF_PI)), n_rots);      // :97
        const int tbase = n * (n - 1) / 2;
        int lo = 0, cnt = n;
        {
            const f3 pb_ = ld3(points, ij.y);
            const f3 dd = sub3(a, pb_);
            const float L2 = dot3(dd, dd);
            const f3 w = sub3(gt, cc);
            const float A_ = dot3(w, x), B_ = dot3(w, y), w2 = dot3(w, w), rho2 = dot3(x, x);
            const float M = __builtin_amdgcn_sqrtf(A_ * A_ + B_ * B_);
            const float K = 0.5f * (w2 + rho2 - tol * tol) - (3e-4f * rho2 + 4e-6f * (w2 + rho2 + tol * tol));
            if (n > 0 && L2 >= 1e-6f && M > 1e-30f) {
                const float c = K * __builtin_amdgcn_rcpf(M);
                if (c > 1.0005f) {
                    cnt = 0;                                   // no rotation comes within tol of the centre
                } else if (c > -0.9995f) {
                    const float alpha = acos_approx(fminf(c, 1.f)) + 8e-4f;        // acos / atan2 errors, rcp, table angles
                    const float phi = atan2_approx(B_, A_);
                    const float k = (float)n * 0.159154943f;                       // n / 2 pi
                    const float ic = phi * k, hw = fmaf(alpha, k, 1.0f);
                    const int i_lo = (int)floorf(ic - hw), i_hi = (int)ceilf(ic + hw);
                    if (i_hi - i_lo + 1 < n) {
                        cnt = i_hi - i_lo + 1;
                        lo = i_lo % n;
                        lo = lo < 0 ? lo + n : lo;
                    }
                }
            }
        }
        const int p1 = max(0, lo + cnt - n);     // candidates that wrap past n - 1 come first in index order: 0 .. p1 - 1
        for (int kk = 0; kk < cnt; ++kk) {
            const int i = kk < p1 ? kk : lo + (kk - p1);
            const float2 cs = in_lds ? ltab[tbase + i] : rot_cs(i, n);
            const f3 offset = add3(scl3(x, cs.x), scl3(y, cs.y));
            const f3 pc = add3(cc, offset);
            if (len3(sub3(pc, gt)) > tol) continue;                                   // :101
            const f3 g = div3(sub3(pc, cr), res);
            if (g.x < 0.f || g.y < 0.f || g.z < 0.f || g.x >= bx || g.y >= by || g.z >= bz) continue;  // :103-107
            found = neg3(offset);                                                      // :108
            break;
        }
        finish(idx, found);
    };
    // A wave takes BV_U x 64 consecutive pairs per trip and has all their loads in flight before it looks at any of them
    // (one pair per lane per trip was two dependent round trips to memory per pair with nothing else to do: 27 us for 9 MB).
    constexpr int BV_U = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * BV_U;
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * BV_U;; base += stride) {
        const bool more = base < n_ppfs;   // wave-uniform
        if (more) {
            float2 o_[BV_U];
            int2 ij_[BV_U];
            f3 pa_[BV_U], pb_[BV_U];
#pragma unroll
            for (int u = 0; u < BV_U; ++u) {
                const int64_t i = base + u * 64 + lane;
                const int64_t c = i < n_ppfs ? i : n_ppfs - 1;
                o_[u] = reinterpret_cast<const float2*>(outputs)[c];
                ij_[u] = load_ij(c);
                if (idx32_out && i < n_ppfs) reinterpret_cast<int2*>(idx32_out)[i] = ij_[u];
            }
#pragma unroll
            for (int u = 0; u < BV_U; ++u) { pa_[u] = ld3(points, ij_[u].x); pb_[u] = ld3(points, ij_[u].y); }
#pragma unroll
            for (int u = 0; u < BV_U; ++u) {
                const int64_t idx = base + u * 64 + lane;
                bool pass = false;
                if (idx < n_ppfs) {
                    const float2 o = o_[u];
                    const int2 ij = ij_[u];
                    const f3 pa = pa_[u], pb = pb_[u];
                    const f3 dd = sub3(pa, pb);
                    const float L2 = dot3(dd, dd);
                    if (L2 >= 1e-13f) {
                        // approximate arithmetic (v_sqrt / v_rcp, no exact divisions) and a slack far above its error: a pair
                        // that fails here cannot pass :101 for any rotation; the others get the reference's exact
                        // arithmetic in stage 2
                        const float L = __builtin_amdgcn_sqrtf(L2);
                        const float inv = __builtin_amdgcn_rcpf(L + 1e-7f);
                        const float proj_len = o.x, odist = o.y;
                        const f3 u = scl3(dd, inv);
                        const f3 cc = sub3(pa, scl3(u, proj_len));
                        // distance from gt to the vote CIRCLE (centre cc, axis u, radius |nu|): h along the axis, r in the
                        // circle's plane; every sample lies on that circle (to 1e-6), so none can be nearer than this
                        const f3 w = sub3(gt, cc);
                        const float h = dot3(w, u), w2 = dot3(w, w), rho = fabsf(odist);
                        const float r = __builtin_amdgcn_sqrtf(fmaxf(w2 - h * h, 0.f));
                        const float dist = __builtin_amdgcn_sqrtf((r - rho) * (r - rho) + h * h);
                        const float mag = __builtin_amdgcn_sqrtf(w2) + rho + tol;
                        // ab = (a-b)/(L + 1e-7) is shorter than a unit vector by 1e-7/L: the samples sit on an ellipse inside
                        // that circle and cc is off the ideal axis point, both by at most (|nu| + |mu|) * 1e-7 / L
                        const float squash = (rho + fabsf(proj_len)) * 2e-7f * inv;
                        pass = (odist * rinv_res * 6.2831855f >= 0.9999f) && !(dist > tol + 2e-4f * mag + 1e-6f + squash);
                        if (!pass) finish(idx, f3{0.f, 0.f, 0.f});
                    } else {   // (nearly) coincident points: the exact test decides what is degenerate
                        f3 a, ab, xd;
                        if (pair_frame(points, ij.x, ij.y, a, ab, xd)) {
                            const float proj_len = o.x, odist = o.y;
                            const f3 cc = sub3(a, scl3(ab, proj_len));
                            const f3 x = scl3(xd, odist);
                            const int n = min((int)((double)(odist / res) * (2 * CPPF_PI)), n_rots);
                            const float dc = len3(sub3(cc, gt)), rho = len3(x);
                            pass = n > 0 && !(fabsf(dc - rho) > tol + 1e-5f * (dc + rho + tol) + 1e-7f);
                            if (!pass) finish(idx, f3{0.f, 0.f, 0.f});
                        } else if (mask) {   // degenerate pair: out_offsets keeps the caller's value (:87 returns early)
                            const float* oo = out_offsets ? out_offsets + 3 * idx : nullptr;
                            const bool nz = oo ? ((oo[0] != 0.f) || (oo[1] != 0.f) || (oo[2] != 0.f)) : false;   // mask-only: as if zero-initialised (:220)
                            mask[idx] = nz;
                            if (chunk_counts) count_survivor(idx, nz);
                        }
                    }
                }
                const unsigned long long m = __ballot(pass);
                if (pass)
                    q[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = (uint32_t)idx;
                qn += __popcll(m);
                while (qn >= 64) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const uint32_t pidx = q[qn - 64 + lane];
                    qn -= 64;
                    rotations((int64_t)pidx);
                }
            }
        }
        if (!more) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (qn > 0) {
                const uint32_t pidx = lane < qn ? q[lane] : 0u;
                if (lane < qn) rotations((int64_t)pidx);
            }
            break;
        }
    }
}

__global__ __launch_bounds__(256) void backvote_kernel(const float* __restrict__ points, const float* __restrict__ outputs,
                                                       float* __restrict__ out_offsets, const int32_t* __restrict__ point_idxs,
                                                       const float* __restrict__ corner, float res, int64_t n_ppfs, int n_rots, int gx,
                                                       int gy, int gz, const float* __restrict__ gt_center, float tol,
                                                       uint8_t* __restrict__ mask, const int32_t* __restrict__ shape,
                                                       const unsigned long long* __restrict__ vote_ws, int32_t* __restrict__ chunk_counts,
                                                       const long long* __restrict__ idx64, int32_t* __restrict__ idx32_out)
{
    backvote_body(points, outputs, out_offsets, point_idxs, corner, res, n_ppfs, n_rots, gx, gy, gz, gt_center, tol, mask, shape, vote_ws,
                  chunk_counts, idx64, idx32_out);
}

static int backvote_impl(const float* points, const float* outputs, float* out_offsets,
                         const int32_t* point_idxs, const float* corner, float res, int64_t n_ppfs, int n_rots,
                         int gx, int gy, int gz, const float* gt_center, float tol, uint8_t* mask, void* stream,
                         const int32_t* shape_dev, const void* vote_workspace = nullptr, int32_t* chunk_counts = nullptr,
                         const long long* idx64 = nullptr, int32_t* idx32_out = nullptr)
{
    if (n_rots < 1 || n_rots > CPPF_MAX_ROTS || n_ppfs < 0 || n_ppfs > 0xffffffffll) return CPPF_EINVAL;
    if (n_ppfs == 0) return 0;
    if (!points || !outputs || (!out_offsets && !mask) || (!point_idxs && !idx64) || !corner || !gt_center) return CPPF_EINVAL;
    const int entries = tri(n_rots);
    const size_t lds = (entries <= VOTE_TAB_LDS_MAX ? (size_t)entries * sizeof(float2) : 0) + 4 * 128 * sizeof(uint32_t);
    int64_t nb = (n_ppfs + 4 * 256 - 1) / (4 * 256);   // BV_U = 4 pairs per thread and trip
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(backvote_kernel, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, points,
                       outputs, out_offsets, point_idxs, corner, res, n_ppfs, n_rots, gx, gy, gz, gt_center, tol,
                       mask, shape_dev, static_cast<const unsigned long long*>(vote_workspace), chunk_counts, idx64, idx32_out);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_backvote(const float* points, const float* outputs, float* out_offsets,
                             const int32_t* point_idxs, const float* corner, float res, int64_t n_ppfs, int n_rots,
                             int gx, int gy, int gz, const float* gt_center, float tol, uint8_t* mask, void* stream)
{
    return backvote_impl(points, outputs, out_offsets, point_idxs, corner, res, n_ppfs, n_rots, gx, gy, gz, gt_center, tol,
                         mask, stream, nullptr);
}

extern "C" int cppf_backvote_ws(const float* points, const float* outputs, float* out_offsets,
                                const int32_t* point_idxs, const float* corner, float res, int64_t n_ppfs, int n_rots,
                                int gx, int gy, int gz, const int32_t* shape_dev, const float* gt_center, float tol,
                                uint8_t* mask, const void* vote_workspace, void* stream)
{
    if (!shape_dev && (gx < 1 || gy < 1 || gz < 1)) return CPPF_EINVAL;
    return backvote_impl(points, outputs, out_offsets, point_idxs, corner, res, n_ppfs, n_rots, shape_dev ? 1 : gx,
                         shape_dev ? 1 : gy, shape_dev ? 1 : gz, gt_center, tol, mask, stream, shape_dev, vote_workspace);
}

extern "C" int cppf_backvote_count(const float* points, const float* outputs, const int32_t* point_idxs, const float* corner,
                                   float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz, const int32_t* shape_dev,
                                   const float* gt_center, float tol, uint8_t* mask, int32_t* chunk_counts,
                                   const void* vote_workspace, void* stream)
{
    if (!mask || !chunk_counts) return CPPF_EINVAL;
    if (!shape_dev && (gx < 1 || gy < 1 || gz < 1)) return CPPF_EINVAL;
    return backvote_impl(points, outputs, nullptr, point_idxs, corner, res, n_ppfs, n_rots, shape_dev ? 1 : gx,
                         shape_dev ? 1 : gy, shape_dev ? 1 : gz, gt_center, tol, mask, stream, shape_dev, vote_workspace,
                         chunk_counts);
}

extern "C" int cppf_backvote_count64(const float* points, const float* outputs, const long long* point_idxs64,
                                     int32_t* idx32_out, const float* corner, float res, int64_t n_ppfs, int n_rots, int gx,
                                     int gy, int gz, const int32_t* shape_dev, const float* gt_center, float tol, uint8_t* mask,
                                     int32_t* chunk_counts, const void* vote_workspace, void* stream)
{
    if (!mask || !chunk_counts || !point_idxs64) return CPPF_EINVAL;
    if (!shape_dev && (gx < 1 || gy < 1 || gz < 1)) return CPPF_EINVAL;
    return backvote_impl(points, outputs, nullptr, nullptr, corner, res, n_ppfs, n_rots, shape_dev ? 1 : gx,
                         shape_dev ? 1 : gy, shape_dev ? 1 : gz, gt_center, tol, mask, stream, shape_dev, vote_workspace,
                         chunk_counts, point_idxs64, idx32_out);
}

extern "C" int cppf_backvote_dyn(const float* points, const float* outputs, float* out_offsets,
                                 const int32_t* point_idxs, const float* corner, float res, int64_t n_ppfs, int n_rots,
                                 const int32_t* shape_dev, const float* gt_center, float tol, uint8_t* mask, void* stream)
{
    if (!shape_dev) return CPPF_EINVAL;
    return backvote_impl(points, outputs, out_offsets, point_idxs, corner, res, n_ppfs, n_rots, 1, 1, 1, gt_center, tol,
                         mask, stream, shape_dev);
}

// ----------------------------------------------------------------------------- compaction
// surv = nonzero(mask) in increasing order (point_idxs[mask], nocs/inference.py:231).
// Three small kernels: per-block counts, one-block scan of the counts, scatter.
__global__ __launch_bounds__(CMP_BLOCK) void compact_count_kernel(const uint8_t* __restrict__ mask, int64_t n,
                                                                   int32_t* __restrict__ block_counts)
{
    __shared__ int wsum[CMP_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * CMP_BLOCK + threadIdx.x;
    const bool f = i < n && mask[i] != 0;
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < CMP_BLOCK / 64; ++w) s += wsum[w];
        block_counts[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(1024) void compact_scan_kernel(int32_t* __restrict__ block_counts, int64_t nblocks,
                                                            int32_t* __restrict__ total)
{
    // exclusive scan in place, 1024 entries per sweep with a running carry
    __shared__ int buf[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblocks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int v = i < nblocks ? block_counts[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = buf[threadIdx.x];
        const int c = carry;
        if (i < nblocks) block_counts[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(CMP_BLOCK) void compact_scatter_kernel(const uint8_t* __restrict__ mask, int64_t n,
                                                                     const int32_t* __restrict__ block_offs,
                                                                     int32_t* __restrict__ surv)
{
    __shared__ int wsum[CMP_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * CMP_BLOCK + threadIdx.x;
    const bool f = i < n && mask[i] != 0;
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(b);
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    if (f) {
        const int rank = __popcll(b & ((1ull << lane) - 1ull));
        surv[block_offs[blockIdx.x] + woff + rank] = (int32_t)i;
    }
}

// cppf_compact_scatter: the scatter step alone, for chunk counts that already exist (cppf_backvote_count): compact.h
static_assert(CMP_BLOCK == 1024, "backvote_kernel counts survivors per chunk of 1 << 10 pairs");
__global__ __launch_bounds__(CMP_BLOCK) void compact_scatter_self_kernel(const uint8_t* __restrict__ mask, int64_t n,
                                                                          const int32_t* __restrict__ chunk_counts,
                                                                          int32_t* __restrict__ surv, int32_t* __restrict__ total)
{
    compact_scatter_self_body(mask, n, chunk_counts, surv, total);
}

extern "C" int cppf_compact_scatter(const uint8_t* mask, int64_t n, const int32_t* chunk_counts, int32_t* surv,
                                    int32_t* count, void* stream)
{
    if (n < 1 || !mask || !surv || !count || !chunk_counts) return CPPF_EINVAL;
    const int64_t nb = (n + CMP_BLOCK - 1) / CMP_BLOCK;
    if (nb > CMP_SELF_MAX) return CPPF_EUNSUPPORTED;   // use cppf_compact_mask
    hipLaunchKernelGGL(compact_scatter_self_kernel, dim3((unsigned)nb), dim3(CMP_BLOCK), 0, (hipStream_t)stream, mask, n,
                       chunk_counts, surv, count);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t cppf_compact_workspace_bytes(int64_t n)
{
    if (n < 0) return 0;
    return align_up((size_t)((n + CMP_BLOCK - 1) / CMP_BLOCK + 1) * sizeof(int32_t), 256);
}

extern "C" int cppf_compact_mask(const uint8_t* mask, int64_t n, int32_t* surv, int32_t* count, void* workspace,
                                 size_t workspace_bytes, void* stream)
{
    if ((n > 0 && (!mask || !surv)) || !count || n < 0 || n > 0x7fffffffll) return CPPF_EINVAL;
    if (!workspace || workspace_bytes < cppf_compact_workspace_bytes(n)) return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int32_t* bc = static_cast<int32_t*>(workspace);
    const int64_t nb = (n + CMP_BLOCK - 1) / CMP_BLOCK;
    if (nb > 0) {
        hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(CMP_BLOCK), 0, st, mask, n, bc);
        CPPF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, bc, nb, count);
    CPPF_CHECK_LAUNCH();
    if (nb > 0) {
        hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nb), dim3(CMP_BLOCK), 0, st, mask, n, bc, surv);
        CPPF_CHECK_LAUNCH();
    }
    return 0;
}

// ----------------------------------------------------------------------------- orientation vote
// Reference: CUDA rot_voting, models/voting.py:119-147.  A block takes ROT_PPB pairs: their frames
// are computed once into LDS, then the (pair, rotation) items are spread over the lanes so that the
// 12-byte candidates of consecutive lanes are consecutive in memory.
#define ROT_PPB 32
struct RotFrame { f3 x, y, base; float t; int ok; int pad; };  // 48 B: keeps the dynamic LDS base 16-B aligned

__device__ __forceinline__ RotFrame rot_frame(const float* __restrict__ points, int ia, int ib, float rot)
{
    RotFrame fr;
    f3 a, ab, xd;
    fr.pad = 0;   // 1 = slot without a pair (an `order` entry beyond the survivor list): contributes nothing
    fr.ok = pair_frame(points, ia, ib, a, ab, xd);
    if (fr.ok) {
        fr.x = xd;
        fr.y = cross3(xd, ab);                 // :135
        fr.t = det_tanf(rot);
        fr.base = fr.t > 0.f ? ab : neg3(ab);  // :142
    }
    return fr;
}
__device__ __forceinline__ f3 rot_candidate(const RotFrame& fr, float2 cs)
{
    const f3 offset = add3(scl3(fr.x, cs.x), scl3(fr.y, cs.y));                  // :141
    f3 up = add3(scl3(offset, fr.t), fr.base);                                     // :142
    return div3(up, (float)((double)len3(up) + 1e-7));                            // :143
}

__global__ __launch_bounds__(256) void rot_voting_kernel(const float* __restrict__ points,
                                                         const float* __restrict__ preds_rot,
                                                         float* __restrict__ outputs_up,
                                                         const int32_t* __restrict__ point_idxs, int64_t n_ppfs,
                                                         int n_rots)
{
    __shared__ RotFrame frames[ROT_PPB];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float2* row = reinterpret_cast<float2*>(lds);  // (cos,sin) of the n_rots rotations
    const int64_t p0 = (int64_t)blockIdx.x * ROT_PPB;
    const int np = (int)min((int64_t)ROT_PPB, n_ppfs - p0);
    for (int i = threadIdx.x; i < n_rots; i += blockDim.x) row[i] = rot_cs(i, n_rots);
    if ((int)threadIdx.x < np) {
        const int2 ij = reinterpret_cast<const int2*>(point_idxs)[p0 + threadIdx.x];
        frames[threadIdx.x] = rot_frame(points, ij.x, ij.y, preds_rot[p0 + threadIdx.x]);
    }
    __syncthreads();
    const int items = np * n_rots;
    float* out = outputs_up + p0 * n_rots * 3;
    for (int k = threadIdx.x; k < items; k += blockDim.x) {
        const int pl = k / n_rots, i = k - pl * n_rots;
        if (!frames[pl].ok) continue;  // caller's zeros stay (:131)
        const f3 up = rot_candidate(frames[pl], row[i]);
        out[3 * k] = up.x; out[3 * k + 1] = up.y; out[3 * k + 2] = up.z;
    }
}

extern "C" int cppf_rot_voting(const float* points, const float* preds_rot, float* outputs_up,
                               const int32_t* point_idxs, int64_t n_ppfs, int n_rots, void* stream)
{
    if (n_rots < 1 || n_rots > 4096 || n_ppfs < 0) return CPPF_EINVAL;
    if (n_ppfs == 0) return 0;
    if (!points || !preds_rot || !outputs_up || !point_idxs) return CPPF_EINVAL;
    const int64_t nb = (n_ppfs + ROT_PPB - 1) / ROT_PPB;
    hipLaunchKernelGGL(rot_voting_kernel, dim3((unsigned)nb), dim3(256), (size_t)n_rots * sizeof(float2),
                       (hipStream_t)stream, points, preds_rot, outputs_up, point_idxs, n_ppfs, n_rots);
    CPPF_CHECK_LAUNCH();
    return 0;
}

// Fused rot_voting + sphere count (nocs/inference.py:265-284): candidates of SPH_PPB pairs go to
// LDS, then every lane owns sphere bins and sweeps the block's candidates (broadcast LDS reads),
// cos = fma(c.z,s.z, fma(c.y,s.y, c.x*s.x)) > thr.  Integer atomics: deterministic counts.
#define SPH_PPB 16
#define SPH_THREADS 512
__global__ __launch_bounds__(SPH_THREADS) void rot_sphere_kernel(const float* __restrict__ points,
                                                                 const float* __restrict__ preds_rot, int rot_stride,
                                                                 const int32_t* __restrict__ point_idxs,
                                                                 const int32_t* __restrict__ sel,
                                                                 const int32_t* __restrict__ n_sel_dev,
                                                                 int64_t n_sel_host, int64_t max_pairs, int n_rots,
                                                                 const float* __restrict__ sphere, int n_sphere,
                                                                 float thr, int32_t* __restrict__ counts,
                                                                 int rot_dir_step, int counts_dir_step,
                                                                 const int32_t* __restrict__ order, int64_t n_order)
{
    preds_rot += (int64_t)blockIdx.y * rot_dir_step;
    counts += (int64_t)blockIdx.y * counts_dir_step;
    __shared__ RotFrame frames[SPH_PPB];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* cand = reinterpret_cast<float4*>(lds);                       // [SPH_PPB*n_rots]
    float2* row = reinterpret_cast<float2*>(lds + 4 * SPH_PPB * n_rots);  // [n_rots]
    const int64_t n_avail = n_sel_dev ? (int64_t)*n_sel_dev : n_sel_host;   // entries of sel (or pairs, sel == null)
    int64_t n_sel = order ? n_order : n_avail;                               // slots of this count
    if (n_sel > max_pairs) n_sel = max_pairs;
    const int64_t k0 = (int64_t)blockIdx.x * SPH_PPB;
    if (k0 >= n_sel) return;
    const int np = (int)min((int64_t)SPH_PPB, n_sel - k0);
    for (int i = threadIdx.x; i < n_rots; i += SPH_THREADS) row[i] = rot_cs(i, n_rots);
    if ((int)threadIdx.x < np) {
        const int64_t sl = order ? (int64_t)order[k0 + threadIdx.x] : k0 + threadIdx.x;   // position in the survivor list
        if (sl >= 0 && sl < n_avail) {
            const int p = sel ? sel[sl] : (int)sl;
            const int2 ij = reinterpret_cast<const int2*>(point_idxs)[p];
            frames[threadIdx.x] = rot_frame(points, ij.x, ij.y, preds_rot[(int64_t)p * rot_stride]);
        } else {
            frames[threadIdx.x].ok = 0;
            frames[threadIdx.x].pad = 1;
        }
    }
    __syncthreads();
    const int items = np * n_rots;
    for (int k = threadIdx.x; k < items; k += SPH_THREADS) {
        const int pl = k / n_rots, i = k - pl * n_rots;
        f3 up = {0.f, 0.f, 0.f};  // degenerate pair: the reference leaves zeros, which still count if thr < 0
        if (frames[pl].ok) up = rot_candidate(frames[pl], row[i]);
        cand[k] = make_float4(up.x, up.y, up.z, frames[pl].pad ? 1.f : 0.f);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_sphere; j += SPH_THREADS) {
        const float sx = sphere[3 * j], sy = sphere[3 * j + 1], sz = sphere[3 * j + 2];
        int cnt = 0;
        for (int k = 0; k < items; ++k) {
            const float4 c = cand[k];
            const float d = fmaf(c.z, sz, fmaf(c.y, sy, c.x * sx));
            cnt += (d > thr) && c.w == 0.f;
        }
        if (cnt) atomicAdd(&counts[j], cnt);
    }
}

// Same count when the sphere bins are unit vectors sorted by y (the Fibonacci sphere of
// utils/util.py:102-118 is): a candidate c can only match bins with |s.y - c.y| < sqrt(2 - 2 thr), so
// each lane takes candidates and tests only that band of bins (~14 of 480 at 1.5 deg) instead of every
// lane sweeping every candidate.  Same dot product, same threshold test -> identical counts.
#define SPHB_PPB 8   // most pairs per group; few survivors are taken 2 at a time, see the kernel
__device__ __forceinline__ void rot_sphere_band_body(const float* __restrict__ points,
                                                              const float* __restrict__ preds_rot, int rot_stride,
                                                              const int32_t* __restrict__ point_idxs,
                                                              const int32_t* __restrict__ sel,
                                                              const int32_t* __restrict__ n_sel_dev, int64_t n_sel_host,
                                                              int64_t max_pairs, int n_rots,
                                                              const float* __restrict__ sphere, int n_sphere, float thr,
                                                              int32_t* __restrict__ counts, int descending,
                                                              int rot_dir_step, int counts_dir_step,
                                                              const int32_t* __restrict__ order, int64_t n_order)
{
    preds_rot += (int64_t)blockIdx.y * rot_dir_step;   // cppf_rot_sphere_count_dirs: direction blockIdx.y of the launch
    counts += (int64_t)blockIdx.y * counts_dir_step;
    __shared__ RotFrame frames[SPHB_PPB];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sph = lds;                                             // [n_sphere][3]
    int* cnt = reinterpret_cast<int*>(lds + 3 * n_sphere);        // [n_sphere]
    float2* row = reinterpret_cast<float2*>(cnt + n_sphere);      // [n_rots]
    const int64_t n_avail = n_sel_dev ? (int64_t)*n_sel_dev : n_sel_host;   // entries of sel (or pairs, sel == null)
    int64_t n_sel = order ? n_order : n_avail;                               // slots of this count
    if (n_sel > max_pairs) n_sel = max_pairs;
    const int ppb = n_sel > 4096 ? SPHB_PPB : 2;   // (uniform over the launch)
    if ((int64_t)blockIdx.x * ppb >= n_sel) return;
    for (int i = threadIdx.x; i < 3 * n_sphere; i += 256) sph[i] = sphere[i];
    for (int i = threadIdx.x; i < n_sphere; i += 256) cnt[i] = 0;
    for (int i = threadIdx.x; i < n_rots; i += 256) row[i] = rot_cs(i, n_rots);
    float band = 2.f - 2.f * thr;
    band = sqrtf(fminf(fmaxf(band, 0.f), 4.f) + 1e-5f) + 1e-4f;
    // A block takes groups of ppb pairs, blockIdx, + gridDim, ...  A group's time is a chain (frame with an fp64 tangent ->
    // candidate -> binary search -> ~14 dependent band steps) that only more groups in flight hide, so few survivors go 2 to a
    // group (the ~2 000 of the benchmark object: 15.7 -> 12.2 us; 500: 13.4 -> 8.5), many go 8 to a group for throughput, and the
    // grid is bounded so that the block's set-up (bins, rotation row) is paid once when it has several groups.
    for (int64_t k0 = (int64_t)blockIdx.x * ppb; k0 < n_sel; k0 += (int64_t)gridDim.x * ppb) {
    const int np = (int)min((int64_t)ppb, n_sel - k0);
    __syncthreads();   // (previous group's frames are no longer read; first trip: the tables above are complete)
    if ((int)threadIdx.x < np) {
        const int64_t sl = order ? (int64_t)order[k0 + threadIdx.x] : k0 + threadIdx.x;   // position in the survivor list
        if (sl >= 0 && sl < n_avail) {
            const int p = sel ? sel[sl] : (int)sl;
            const int2 ij = reinterpret_cast<const int2*>(point_idxs)[p];
            frames[threadIdx.x] = rot_frame(points, ij.x, ij.y, preds_rot[(int64_t)p * rot_stride]);
        } else {
            frames[threadIdx.x].ok = 0;
            frames[threadIdx.x].pad = 1;
        }
    }
    __syncthreads();
    const int items = np * n_rots;
    for (int k = threadIdx.x; k < items; k += 256) {
        const int pl = k / n_rots, i = k - pl * n_rots;
        if (frames[pl].pad) continue;
        f3 up = {0.f, 0.f, 0.f};
        if (frames[pl].ok) up = rot_candidate(frames[pl], row[i]);
        // bins with y in [up.y - band, up.y + band]: binary searches on the sorted y column
        const float ylo = up.y - band, yhi = up.y + band;
        int lo = 0, hi = n_sphere;  // first bin inside the band
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float y = sph[3 * mid + 1];
            const bool before = descending ? (y > yhi) : (y < ylo);
            if (before) lo = mid + 1; else hi = mid;
        }
        for (int j = lo; j < n_sphere; ++j) {
            const float sy = sph[3 * j + 1];
            if (descending ? (sy < ylo) : (sy > yhi)) break;
            const float d = fmaf(up.z, sph[3 * j + 2], fmaf(up.y, sy, up.x * sph[3 * j]));
            if (d > thr) atomicAdd(&cnt[j], 1);
        }
    }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_sphere; j += 256)
        if (cnt[j]) atomicAdd(&counts[j], cnt[j]);
}

// The same count for launches that serve several instances at once (cppf_pose_tail_batch).  rot_sphere_band_body's blocks take groups
// of 2 or 8 pairs -- tuned for ONE instance on an idle chip, where a group is a latency chain and only more groups in flight hide
// it -- and every block pays the set-up (the 480 bins, the rotation row in fp64, the flush) for 144 or 576 candidates: as much
// work as the candidates themselves.  In a chain of 4-8 instances beside other lanes the chip is full anyway and total work is what
// counts: here a block owns a CONTIGUOUS share of the survivors (<= 256 blocks per instance and direction), computes the frames of up
// to 64 pairs in one step and sweeps their candidates with full lanes; bins as float4 (one LDS read per band step instead of three).
// Same candidates, same dot products, same threshold test, integer counts: identical results.
#define SPHE_PAIRS 64
#define SPHE_BLOCKS 256
__device__ __forceinline__ void rot_sphere_band_even_body(const float* __restrict__ points, const float* __restrict__ preds_rot,
                                                          int rot_stride, const int32_t* __restrict__ point_idxs,
                                                          const int32_t* __restrict__ sel, const int32_t* __restrict__ n_sel_dev,
                                                          int64_t n_sel_host, int64_t max_pairs, int n_rots,
                                                          const float* __restrict__ sphere, int n_sphere, float thr,
                                                          int32_t* __restrict__ counts, int descending, int rot_dir_step,
                                                          int counts_dir_step)
{
    preds_rot += (int64_t)blockIdx.y * rot_dir_step;
    counts += (int64_t)blockIdx.y * counts_dir_step;
    __shared__ RotFrame frames[SPHE_PAIRS];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* sph4 = reinterpret_cast<float4*>(lds);                // [n_sphere] {x, y, z, -}
    int* cnt = reinterpret_cast<int*>(lds + 4 * n_sphere);        // [n_sphere]
    float2* row = reinterpret_cast<float2*>(cnt + n_sphere);      // [n_rots]
    const int64_t n_avail = n_sel_dev ? (int64_t)*n_sel_dev : n_sel_host;
    const int64_t n_sel = n_avail < max_pairs ? n_avail : max_pairs;
    const int64_t per = (n_sel + gridDim.x - 1) / gridDim.x;
    const int64_t k_begin = (int64_t)blockIdx.x * per, k_end = k_begin + per < n_sel ? k_begin + per : n_sel;
    if (k_begin >= k_end) return;
    for (int i = threadIdx.x; i < n_sphere; i += 256) { sph4[i] = make_float4(sphere[3 * i], sphere[3 * i + 1], sphere[3 * i + 2], 0.f); cnt[i] = 0; }
    for (int i = threadIdx.x; i < n_rots; i += 256) row[i] = rot_cs(i, n_rots);
    float band = 2.f - 2.f * thr;
    band = sqrtf(fminf(fmaxf(band, 0.f), 4.f) + 1e-5f) + 1e-4f;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += SPHE_PAIRS) {
        const int np = (int)min((int64_t)SPHE_PAIRS, k_end - k0);
        __syncthreads();   // (previous round's frames are no longer read; first trip: the tables above are complete)
        if ((int)threadIdx.x < np) {
            const int64_t sl = k0 + threadIdx.x;
            const int p = sel ? sel[sl] : (int)sl;
            const int2 ij = reinterpret_cast<const int2*>(point_idxs)[p];
            frames[threadIdx.x] = rot_frame(points, ij.x, ij.y, preds_rot[(int64_t)p * rot_stride]);
        }
        __syncthreads();
        const int items = np * n_rots;
        for (int k = threadIdx.x; k < items; k += 256) {
            const int pl = k / n_rots, i = k - pl * n_rots;
            f3 up = {0.f, 0.f, 0.f};
            if (frames[pl].ok) up = rot_candidate(frames[pl], row[i]);
            const float ylo = up.y - band, yhi = up.y + band;
            int lo = 0, hi = n_sphere;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float y = sph4[mid].y;
                const bool before = descending ? (y > yhi) : (y < ylo);
                if (before) lo = mid + 1; else hi = mid;
            }
            for (int j = lo; j < n_sphere; ++j) {
                const float4 sb = sph4[j];
                if (descending ? (sb.y < ylo) : (sb.y > yhi)) break;
                const float d = fmaf(up.z, sb.z, fmaf(up.y, sb.y, up.x * sb.x));
                if (d > thr) atomicAdd(&cnt[j], 1);
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_sphere; j += 256)
        if (cnt[j]) atomicAdd(&counts[j], cnt[j]);
}

__global__ __launch_bounds__(256) void rot_sphere_band_kernel(const float* __restrict__ points, const float* __restrict__ preds_rot,
                                                              int rot_stride, const int32_t* __restrict__ point_idxs,
                                                              const int32_t* __restrict__ sel, const int32_t* __restrict__ n_sel_dev,
                                                              int64_t n_sel_host, int64_t max_pairs, int n_rots,
                                                              const float* __restrict__ sphere, int n_sphere, float thr,
                                                              int32_t* __restrict__ counts, int descending, int rot_dir_step,
                                                              int counts_dir_step, const int32_t* __restrict__ order, int64_t n_order)
{
    rot_sphere_band_body(points, preds_rot, rot_stride, point_idxs, sel, n_sel_dev, n_sel_host, max_pairs, n_rots, sphere, n_sphere, thr,
                         counts, descending, rot_dir_step, counts_dir_step, order, n_order);
}

static int rot_sphere_impl(const float* points, const float* preds_rot, int rot_stride, int rot_dir_step, int n_dirs,
                           const int32_t* point_idxs, const int32_t* sel, const int32_t* n_sel_dev,
                           int64_t n_sel_host, int64_t max_pairs, int n_rots, const float* sphere,
                           int n_sphere, float thr, int sphere_sorted_by_y, int32_t* counts, int counts_dir_step, void* stream,
                           const int32_t* order = nullptr, int64_t n_order = 0)
{
    if (!points || !preds_rot || !point_idxs || !sphere || !counts) return CPPF_EINVAL;
    if (n_order < 0) return CPPF_EINVAL;
    if (!order) n_order = 0;
    if (n_rots < 1 || n_rots > 512 || n_sphere < 1 || max_pairs < 0 || n_sel_host < 0 || rot_stride < 1 || n_dirs < 1 ||
        n_dirs > 16)
        return CPPF_EINVAL;
    const int64_t slots = order ? n_order : n_sel_host;      // (with an order the slots are the order's)
    int64_t bound = slots < max_pairs ? slots : max_pairs;
    if (bound == 0) return 0;
    if (sphere_sorted_by_y != 0 && n_sphere <= 4096) {
        int64_t nb = (bound + 1) / 2;
        if (nb > 2048) nb = 2048;
        const size_t lds = (size_t)(4 * n_sphere + 2 * n_rots) * sizeof(float);
        hipLaunchKernelGGL(rot_sphere_band_kernel, dim3((unsigned)nb, (unsigned)n_dirs), dim3(256), lds, (hipStream_t)stream, points,
                           preds_rot, rot_stride, point_idxs, sel, n_sel_dev, n_sel_host, max_pairs, n_rots, sphere,
                           n_sphere, thr, counts, sphere_sorted_by_y > 0 ? 1 : 0, rot_dir_step, counts_dir_step, order, n_order);
        CPPF_CHECK_LAUNCH();
        return 0;
    }
    const int64_t nb = (bound + SPH_PPB - 1) / SPH_PPB;
    const size_t lds = (size_t)(4 * SPH_PPB * n_rots + 2 * n_rots) * sizeof(float);
    hipLaunchKernelGGL(rot_sphere_kernel, dim3((unsigned)nb, (unsigned)n_dirs), dim3(SPH_THREADS), lds, (hipStream_t)stream, points,
                       preds_rot, rot_stride, point_idxs, sel, n_sel_dev, n_sel_host, max_pairs, n_rots, sphere,
                       n_sphere, thr, counts, rot_dir_step, counts_dir_step, order, n_order);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_rot_sphere_count(const float* points, const float* preds_rot, int rot_stride,
                                     const int32_t* point_idxs, const int32_t* sel, const int32_t* n_sel_dev,
                                     int64_t n_sel_host, int64_t max_pairs, int n_rots, const float* sphere,
                                     int n_sphere, float thr, int sphere_sorted_by_y, int32_t* counts, void* stream)
{
    return rot_sphere_impl(points, preds_rot, rot_stride, 0, 1, point_idxs, sel, n_sel_dev, n_sel_host, max_pairs, n_rots,
                           sphere, n_sphere, thr, sphere_sorted_by_y, counts, 0, stream);
}

extern "C" int cppf_rot_sphere_count_dirs(const float* points, const float* preds_rot, int rot_stride, int rot_dir_step,
                                          int n_dirs, const int32_t* point_idxs, const int32_t* sel,
                                          const int32_t* n_sel_dev, int64_t n_sel_host, int64_t max_pairs, int n_rots,
                                          const float* sphere, int n_sphere, float thr, int sphere_sorted_by_y,
                                          int32_t* counts, int counts_dir_step, void* stream)
{
    if (n_dirs > 1 && (rot_dir_step < 1 || counts_dir_step < n_sphere)) return CPPF_EINVAL;
    return rot_sphere_impl(points, preds_rot, rot_stride, rot_dir_step, n_dirs, point_idxs, sel, n_sel_dev, n_sel_host,
                           max_pairs, n_rots, sphere, n_sphere, thr, sphere_sorted_by_y, counts, counts_dir_step, stream);
}

extern "C" int cppf_rot_sphere_count_dirs_order(const float* points, const float* preds_rot, int rot_stride, int rot_dir_step,
                                                int n_dirs, const int32_t* point_idxs, const int32_t* sel,
                                                const int32_t* n_sel_dev, int64_t n_sel_host, const int32_t* order,
                                                int64_t n_order, int64_t max_pairs, int n_rots, const float* sphere,
                                                int n_sphere, float thr, int sphere_sorted_by_y, int32_t* counts,
                                                int counts_dir_step, void* stream)
{
    if (n_dirs > 1 && (rot_dir_step < 1 || counts_dir_step < n_sphere)) return CPPF_EINVAL;
    if (!order) return CPPF_EINVAL;
    return rot_sphere_impl(points, preds_rot, rot_stride, rot_dir_step, n_dirs, point_idxs, sel, n_sel_dev, n_sel_host,
                           max_pairs, n_rots, sphere, n_sphere, thr, sphere_sorted_by_y, counts, counts_dir_step, stream,
                           order, n_order);
}

// ----------------------------------------------------------------------------- pose-tail reductions
#define RED_BLOCKS 256
#define RED_THREADS 256
extern "C" size_t cppf_reduce_workspace_bytes(void) { return (size_t)RED_BLOCKS * 4 * sizeof(double); }

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < RED_THREADS / 64; ++w) s += sh[w];
    return s;
}

// nocs/inference.py:287-301
__global__ __launch_bounds__(RED_THREADS) void axis_sign_kernel(const float* __restrict__ pc,
                                                                const float* __restrict__ nrm,
                                                                const int32_t* __restrict__ point_idxs,
                                                                const int32_t* __restrict__ sel,
                                                                const int32_t* __restrict__ n_sel_dev, int64_t n_sel_host,
                                                                const float* __restrict__ aux, int aux_stride,
                                                                const double* __restrict__ best_dir,
                                                                double* __restrict__ partial)
{
    __shared__ double sh[RED_THREADS / 64];
    const int64_t n_sel = n_sel_dev ? (int64_t)*n_sel_dev : n_sel_host;
    const double bx = best_dir[0], by = best_dir[1], bz = best_dir[2];
    double up = 0.0, down = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; k < n_sel; k += (int64_t)RED_BLOCKS * RED_THREADS) {
        const int p = sel ? sel[k] : (int)k;
        const int2 ij = reinterpret_cast<const int2*>(point_idxs)[p];
        const f3 ab = sub3(ld3(pc, ij.x), ld3(pc, ij.y));
        const float distsq = (ab.x * ab.x + ab.y * ab.y) + ab.z * ab.z;
        const float den = sqrtf(distsq) + 1e-7f;
        const f3 abn = {ab.x / den, ab.y / den, ab.z / den};
        f3 n = ld3(nrm, ij.x);
        const float d = (n.x * abn.x + n.y * abn.y) + n.z * abn.z;
        if (d < 0.f) n = neg3(n);
        const double proj = ((double)n.x * bx + (double)n.y * by) + (double)n.z * bz;
        const double t = proj > 0.0 ? 1.0 : 0.0;
        const double x = (double)aux[(int64_t)p * aux_stride];
        const double sp = (x > 0.0 ? x : 0.0) + log1p(exp(-fabs(x)));
        up += sp - x * t;
        down += sp - x * (1.0 - t);
    }
    const double su = block_sum(up, sh);
    const double sd = block_sum(down, sh);
    if (threadIdx.x == 0) {
        partial[4 * blockIdx.x] = su;
        partial[4 * blockIdx.x + 1] = sd;
    }
}

// nocs/inference.py:335 (sums; the caller finishes exp(mean)*scale_mean*2)
__global__ __launch_bounds__(RED_THREADS) void scale_sum_kernel(const float* __restrict__ scale_logits, int stride,
                                                                const int32_t* __restrict__ sel,
                                                                const int32_t* __restrict__ n_sel_dev, int64_t n_sel_host,
                                                                double* __restrict__ partial)
{
    __shared__ double sh[RED_THREADS / 64];
    const int64_t n_sel = n_sel_dev ? (int64_t)*n_sel_dev : n_sel_host;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; k < n_sel; k += (int64_t)RED_BLOCKS * RED_THREADS) {
        const int64_t p = sel ? sel[k] : k;
        const float* s = scale_logits + p * stride;
        s0 += (double)s[0]; s1 += (double)s[1]; s2 += (double)s[2];
    }
    const double a = block_sum(s0, sh), b = block_sum(s1, sh), c = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        partial[4 * blockIdx.x] = a;
        partial[4 * blockIdx.x + 1] = b;
        partial[4 * blockIdx.x + 2] = c;
    }
}

__global__ void reduce_final_kernel(const double* __restrict__ partial, int ncomp, const int32_t* n_sel_dev,
                                    int64_t n_sel_host, double* __restrict__ out)
{
    if (threadIdx.x < ncomp) {
        double s = 0.0;
        for (int b = 0; b < RED_BLOCKS; ++b) s += partial[4 * b + threadIdx.x];
        out[threadIdx.x] = s;
    }
    if (threadIdx.x == 0) out[ncomp] = (double)(n_sel_dev ? (int64_t)*n_sel_dev : n_sel_host);
}

extern "C" int cppf_axis_sign(const float* pc, const float* nrm, const int32_t* point_idxs, const int32_t* sel,
                              const int32_t* n_sel_dev, int64_t n_sel_host, const float* aux, int aux_stride,
                              const double* best_dir, double* out, void* workspace, size_t workspace_bytes,
                              void* stream)
{
    if (!pc || !nrm || !point_idxs || !aux || !best_dir || !out || aux_stride < 1 || n_sel_host < 0) return CPPF_EINVAL;
    if (!workspace || workspace_bytes < cppf_reduce_workspace_bytes()) return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* partial = static_cast<double*>(workspace);
    hipLaunchKernelGGL(axis_sign_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, st, pc, nrm, point_idxs, sel,
                       n_sel_dev, n_sel_host, aux, aux_stride, best_dir, partial);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, partial, 2, n_sel_dev, n_sel_host, out);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_scale_sum(const float* scale_logits, int stride, const int32_t* sel, const int32_t* n_sel_dev,
                              int64_t n_sel_host, double* out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!scale_logits || !out || stride < 3 || n_sel_host < 0) return CPPF_EINVAL;
    if (!workspace || workspace_bytes < cppf_reduce_workspace_bytes()) return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* partial = static_cast<double*>(workspace);
    hipLaunchKernelGGL(scale_sum_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, st, scale_logits, stride, sel,
                       n_sel_dev, n_sel_host, partial);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, st, partial, 3, n_sel_dev, n_sel_host, out);
    CPPF_CHECK_LAUNCH();
    return 0;
}

// One launch for what follows the orientation vote (nocs/inference.py:283-301,335): np.argmax of every direction's bin
// counts and best_dir = sphere_pts[argmax] (every block, from L2: 480 integers per direction), the two BCE sums per
// direction and the three scale sums over the surviving pairs, and -- in the block that draws the last ticket -- the sum
// of the per-block partials.  Replaces counts_argmax_select + (axis_sign + reduce_final) per direction + scale_sum +
// reduce_final: 4 n_dirs + 2 launches.  Per block and per pair the arithmetic is that of axis_sign_kernel and
// scale_sum_kernel; the partials are summed 8 per lane and then across 32 lanes (fixed order).
#define PS_MAX_DIRS 2
#define PS_COMP 8   // doubles per block partial: {up, down} x 2 directions, 3 scale sums, unused
struct PoseSumsArgs {
    const float *pc, *nrm, *aux, *scale_logits;
    const int32_t *point_idxs, *sel, *n_sel_dev, *counts;
    const double* sphere64;
    int64_t n_sel_host;
    int aux_stride, scale_stride, n_dirs, n_sphere, counts_dir_step;
    long long* best_idx;
    double *best_dir, *sign, *scale_out, *partial;
    unsigned* ticket;
    // the finished record (cppf.h: CppfPoseTailItem.record_out), assembled by the workgroup that draws the last ticket; NULL: not assembled
    double* record_out; const double* rec; const long long* object_id_dev; long long object_id_host; double scale_mean[3]; int regress_right;
};
// The host end of nocs/inference.py:299-339 (cppf_amd.inference._assemble, the same operations in the same order) from the sums one
// thread holds: fin = {up/down loss sums of direction 0, of direction 1, three scale sums}, bdir = the two best sphere bins.
__device__ __forceinline__ void assemble_record(const PoseSumsArgs& A, const double* fin, const double (*bdir)[3], const double n_surv)
{
    double* out = A.record_out;
    const double n = fmax(n_surv, 1.0);
    double dirs[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    for (int j = 0; j < 2; ++j) {
        if (j >= A.n_dirs) break;
        const bool flip = fin[2 * j + 1] / n < fin[2 * j] / n;              // down_loss < up_loss (:299-302)
        for (int c = 0; c < 3; ++c) dirs[j][c] = flip ? -bdir[j][c] : bdir[j][c];
    }
    const double* up = dirs[0];
    double r[3];
    if (A.regress_right && A.n_dirs > 1) {                                  // :305-312
        const double d = (up[0] * dirs[1][0] + up[1] * dirs[1][1]) + up[2] * dirs[1][2];
        for (int c = 0; c < 3; ++c) r[c] = dirs[1][c] - d * up[c];
    } else {
        r[0] = 0.0; r[1] = -up[2]; r[2] = up[1];
    }
    double nr = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + 1e-9;
    for (int c = 0; c < 3; ++c) r[c] = r[c] / nr;
    if (sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) < 1e-7) {           // :325-328; numpy.random.default_rng(0).standard_normal(3)
        r[0] = 0x1.017ed89db8441p-3; r[1] = -0x1.0e8cfe9bd45ccp-3; r[2] = 0x1.47e57a468b06dp-1;
        const double d = (r[0] * up[0] + r[1] * up[1]) + r[2] * up[2];
        for (int c = 0; c < 3; ++c) r[c] = r[c] - d * up[c];
        nr = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
        for (int c = 0; c < 3; ++c) r[c] = r[c] / nr;
    }
    for (int c = 0; c < 3; ++c) {
        out[c] = A.rec[c];
        out[3 + c] = up[c];
        out[6 + c] = r[c];
        const float mean = (float)(fin[4 + c] / n);                         // torch's mean is fp32 (:335)
        out[9 + c] = ((double)(float)exp((double)mean) * A.scale_mean[c]) * 2.0;
    }
    out[12] = A.rec[19]; out[13] = A.rec[20]; out[14] = n_surv;
    out[15] = (double)(A.object_id_dev ? *A.object_id_dev : A.object_id_host);
}
__device__ __forceinline__ void pose_sums_body(const PoseSumsArgs& A)
{
    __shared__ double shc[RED_THREADS / 64][PS_COMP];
    __shared__ unsigned long long best[PS_MAX_DIRS][RED_THREADS / 64];
    __shared__ double bdir[PS_MAX_DIRS][3];
    __shared__ unsigned drawn;
    const int tid = threadIdx.x;
    // A thread's first pair (with <= 65 536 survivors its only one) is fetched BEFORE the arg-max of the bin counts: three
    // levels of dependent loads (sel -> pair -> points) that do not depend on best_dir and used to start after two barriers.
    struct Item { f3 pa, pb, nn; float aux[PS_MAX_DIRS]; float sl[3]; };
    auto load_item = [&](const int64_t k) -> Item {
        Item it;
        const int p = A.sel ? A.sel[k] : (int)k;
        const int2 ij = reinterpret_cast<const int2*>(A.point_idxs)[p];
        it.pa = ld3(A.pc, ij.x); it.pb = ld3(A.pc, ij.y); it.nn = ld3(A.nrm, ij.x);
#pragma unroll
        for (int j = 0; j < PS_MAX_DIRS; ++j) it.aux[j] = j < A.n_dirs ? A.aux[(int64_t)p * A.aux_stride + j] : 0.f;
        it.sl[0] = it.sl[1] = it.sl[2] = 0.f;
        if (A.scale_logits) {
            const float* sl = A.scale_logits + (int64_t)p * A.scale_stride;
            it.sl[0] = sl[0]; it.sl[1] = sl[1]; it.sl[2] = sl[2];
        }
        return it;
    };
    const int64_t n_sel = A.n_sel_dev ? (int64_t)*A.n_sel_dev : A.n_sel_host;
    Item it = {};
    if ((int64_t)blockIdx.x * RED_THREADS + tid < n_sel) it = load_item((int64_t)blockIdx.x * RED_THREADS + tid);
    for (int j = 0; j < A.n_dirs; ++j) {   // key = count << 32 | ~index: the largest count at the lowest index (:283)
        const int32_t* cj = A.counts + (int64_t)j * A.counts_dir_step;
        unsigned long long k = 0ull;
        for (int i = tid; i < A.n_sphere; i += RED_THREADS) {
            const unsigned long long ki = ((unsigned long long)(uint32_t)cj[i] << 32) | (uint32_t)(~(uint32_t)i);
            k = ki > k ? ki : k;
        }
        k = wave_max_u64(k);
        if ((tid & 63) == 0) best[j][tid >> 6] = k;
    }
    __syncthreads();
    if (tid < 3 * A.n_dirs) {
        const int j = tid / 3, c = tid - 3 * j;
        unsigned long long b = best[j][0];
        for (int w = 1; w < RED_THREADS / 64; ++w) b = best[j][w] > b ? best[j][w] : b;
        const int bi = (int)(~(uint32_t)(b & 0xffffffffull));
        const double v = A.sphere64[3 * (size_t)bi + c];
        bdir[j][c] = v;
        if (blockIdx.x == 0) {
            A.best_dir[3 * j + c] = v;
            if (c == 0 && A.best_idx) A.best_idx[j] = bi;
        }
    }
    __syncthreads();
    double acc[PS_COMP] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t k = (int64_t)blockIdx.x * RED_THREADS + tid; k < n_sel; k += (int64_t)RED_BLOCKS * RED_THREADS) {
        if (k != (int64_t)blockIdx.x * RED_THREADS + tid) it = load_item(k);   // (the first one is already here)
        const f3 ab = sub3(it.pa, it.pb);
        const float distsq = (ab.x * ab.x + ab.y * ab.y) + ab.z * ab.z;
        const float den = sqrtf(distsq) + 1e-7f;
        const f3 abn = {ab.x / den, ab.y / den, ab.z / den};
        f3 n = it.nn;
        const float d = (n.x * abn.x + n.y * abn.y) + n.z * abn.z;
        if (d < 0.f) n = neg3(n);
#pragma unroll
        for (int j = 0; j < PS_MAX_DIRS; ++j) {
            if (j >= A.n_dirs) break;
            const double proj = ((double)n.x * bdir[j][0] + (double)n.y * bdir[j][1]) + (double)n.z * bdir[j][2];
            const double t = proj > 0.0 ? 1.0 : 0.0;
            const double x = (double)it.aux[j];
            const double sp = (x > 0.0 ? x : 0.0) + log1p(exp(-fabs(x)));
            acc[2 * j] += sp - x * t;
            acc[2 * j + 1] += sp - x * (1.0 - t);
        }
        if (A.scale_logits) { acc[4] += (double)it.sl[0]; acc[5] += (double)it.sl[1]; acc[6] += (double)it.sl[2]; }
    }
    double* mine = A.partial + (size_t)PS_COMP * blockIdx.x;
    // the seven block sums with one barrier: butterflies inside the wave, then thread c adds the four wave sums of component c
    // in wave order (the order block_sum uses)
#pragma unroll
    for (int c = 0; c < PS_COMP - 1; ++c) {
        double v = acc[c];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((tid & 63) == 0) shc[tid >> 6][c] = v;
    }
    __syncthreads();
    if (tid < PS_COMP - 1) {
        double v = 0.0;
        for (int w = 0; w < RED_THREADS / 64; ++w) v += shc[w][tid];
        // device-scope atomic stores and loads: the block that sums them runs on another CU, maybe another XCD
        __hip_atomic_store(mine + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // (workgroup scope: the seven stores are ordered before thread 0's release below)
    if (tid == 0) drawn = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (drawn != RED_BLOCKS - 1) return;
    static_assert(RED_THREADS == 32 * PS_COMP && RED_BLOCKS == 32 * 8, "final sum: 8 components x 32 lanes x 8 partials");
    const int c = tid >> 5, l = tid & 31;
    double s = 0.0;
    // every block's partials were stored (device scope) before its ticket; this block drew the last one: an acquire fence, then
    // plain loads, eight per lane, all in flight at once
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (c < PS_COMP - 1) {
        double v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) v[b] = A.partial[(size_t)PS_COMP * (8 * l + b) + c];
#pragma unroll
        for (int b = 0; b < 8; ++b) s += v[b];
    }
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (l == 0) {
        if (c < 4) { if (c < 2 * A.n_dirs) A.sign[3 * (c >> 1) + (c & 1)] = s; }
        else if (c < 7 && A.scale_logits) A.scale_out[c - 4] = s;
    }
    if (A.record_out) {            // (workgroup-uniform)
        double* fin = &shc[0][0];  // the block sums were consumed above: the LDS is free
        __syncthreads();
        if (l == 0) fin[c] = (c < 4 ? c < 2 * A.n_dirs : (c < 7 && A.scale_logits)) ? s : 0.0;
        __syncthreads();
        if (tid == 0) assemble_record(A, fin, bdir, (double)n_sel);
    }
    if (tid < A.n_dirs) A.sign[3 * tid + 2] = (double)n_sel;
    if (tid == 0 && A.scale_logits) A.scale_out[3] = (double)n_sel;
    if (tid == 0) __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
}

__global__ __launch_bounds__(RED_THREADS) void pose_sums_kernel(PoseSumsArgs A)
{
    pose_sums_body(A);
}

extern "C" size_t cppf_pose_sums_workspace_bytes(void) { return (size_t)RED_BLOCKS * PS_COMP * sizeof(double) + 16; }

extern "C" int cppf_pose_sums(const float* pc, const float* nrm, const int32_t* point_idxs, const int32_t* sel,
                              const int32_t* n_sel_dev, int64_t n_sel_host, const float* aux, int aux_stride, int n_dirs,
                              const int32_t* counts, int n_sphere, int counts_dir_step, const double* sphere64,
                              const float* scale_logits, int scale_stride, long long* best_idx, double* best_dir,
                              double* sign, double* scale_out, void* workspace, size_t workspace_bytes, unsigned* ticket,
                              void* stream)
{
    if (!pc || !nrm || !point_idxs || !aux || !counts || !sphere64 || !best_dir || !sign || !ticket) return CPPF_EINVAL;
    if (n_dirs < 1 || n_dirs > PS_MAX_DIRS || aux_stride < n_dirs || n_sphere < 1 || n_sel_host < 0 ||
        (n_dirs > 1 && counts_dir_step < n_sphere) || (scale_logits && (scale_stride < 3 || !scale_out)))
        return CPPF_EINVAL;
    if (!workspace || workspace_bytes < cppf_pose_sums_workspace_bytes()) return CPPF_EWORKSPACE;
    PoseSumsArgs A;
    A.pc = pc; A.nrm = nrm; A.aux = aux; A.scale_logits = scale_logits;
    A.point_idxs = point_idxs; A.sel = sel; A.n_sel_dev = n_sel_dev; A.counts = counts;
    A.sphere64 = sphere64; A.n_sel_host = n_sel_host;
    A.aux_stride = aux_stride; A.scale_stride = scale_stride; A.n_dirs = n_dirs; A.n_sphere = n_sphere;
    A.counts_dir_step = counts_dir_step;
    A.best_idx = best_idx; A.best_dir = best_dir; A.sign = sign; A.scale_out = scale_out;
    A.partial = static_cast<double*>(workspace); A.ticket = ticket;
    A.record_out = nullptr; A.rec = nullptr; A.object_id_dev = nullptr; A.object_id_host = 0; A.regress_right = 0;
    A.scale_mean[0] = A.scale_mean[1] = A.scale_mean[2] = 0.0;
    hipLaunchKernelGGL(pose_sums_kernel, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, (hipStream_t)stream, A);
    CPPF_CHECK_LAUNCH();
    return 0;
}

// ----------------------------------------------------------------------------- the tail of several objects, launch by launch
// cppf_pose_tail_batch: everything between the centre vote's arg-max and the pose record (nocs/inference.py:209-303,335) for up to 8
// objects -- the instances of a frame (:120 loops over them) -- in SIX launches instead of six per object: every kernel below is the
// single-object kernel's body run on item blockIdx.y (.z for the sphere count, whose .y is the direction) of a by-value item array,
// so every pointer stays workgroup-uniform and results are the single calls' bit for bit.  At the reference's own size (100 000
// pairs, :177) a launch is half prologue: the tail's ~6 launches were a third of an instance's time.
#define TAIL_BATCH_MAX 8
struct TailItem {
    // pose_tail_begin
    const long long* idx; const float* corner; double res64; const int32_t* shape; double* T64; float* T32; const float* peak;
    double* idx_peak; uint4* zero16; int n_zero16;
    // back-vote + compaction
    const float* points; const float* outputs; const long long* idx64; int32_t* idx32; float res; float tol; int64_t n_ppfs;
    int gx, gy, gz; uint8_t* mask; const unsigned long long* vote_ws; int32_t* chunk_counts; int32_t* surv; int32_t* count;
    // orientation vote + sphere count
    const float* heads; int n_dirs; int32_t* counts;
    // sums
    PoseSumsArgs sums;
};
struct TailBatch {
    TailItem item[TAIL_BATCH_MAX];
    int n, n_rots, n_sphere, descending, sphere_legacy;
    int64_t max_rot_pairs;
    const float* sphere32;
    float thr;
};
static_assert(sizeof(TailBatch) <= 4096, "TailBatch travels by value: kernel arguments are limited to 4 KB");
__global__ void tail_begin_batch_kernel(TailBatch B)
{
    const TailItem& I = B.item[blockIdx.x];
    center_from_argmax_body(I.idx, I.corner, I.res64, I.gy, I.gz, I.T64, I.T32, I.peak, I.idx_peak, I.shape, I.zero16, I.n_zero16);
}
__global__ __launch_bounds__(256) void backvote_batch_kernel(TailBatch B)
{
    const TailItem& I = B.item[blockIdx.y];
    // (idx64 null: the pair list was drawn as int32 -- idx32 is the input, nothing to convert)
    backvote_body(I.points, I.outputs, nullptr, I.idx64 ? nullptr : I.idx32, I.corner, I.res, I.n_ppfs, B.n_rots, I.gx, I.gy, I.gz, I.T32, I.tol,
                  I.mask, I.shape, I.vote_ws, I.chunk_counts, I.idx64, I.idx64 ? I.idx32 : nullptr);
}
__global__ __launch_bounds__(CMP_BLOCK) void compact_scatter_batch_kernel(TailBatch B)
{
    const TailItem& I = B.item[blockIdx.y];
    compact_scatter_self_body(I.mask, I.n_ppfs, I.chunk_counts, I.surv, I.count);
}
__global__ __launch_bounds__(256) void rot_sphere_band_batch_kernel(TailBatch B)
{
    const TailItem& I = B.item[blockIdx.z];
    if ((int)blockIdx.y >= I.n_dirs) return;
    if (B.sphere_legacy)
        rot_sphere_band_body(I.points, I.heads, 8, I.idx32, I.surv, I.count, I.n_ppfs, B.max_rot_pairs, B.n_rots, B.sphere32, B.n_sphere, B.thr,
                             I.counts, B.descending, 1, B.n_sphere, nullptr, 0);
    else
        rot_sphere_band_even_body(I.points, I.heads, 8, I.idx32, I.surv, I.count, I.n_ppfs, B.max_rot_pairs, B.n_rots, B.sphere32, B.n_sphere,
                                  B.thr, I.counts, B.descending, 1, B.n_sphere);
}
__global__ __launch_bounds__(RED_THREADS) void pose_sums_batch_kernel(TailBatch B)
{
    pose_sums_body(B.item[blockIdx.y].sums);
}

extern "C" int cppf_pose_tail_batch(int n_items, const CppfPoseTailItem* items, int F, const int* dims, int n_res, int out_dim, int tr_bins,
                                    int rot_bins, int n_rots, const float* sphere32, const double* sphere64, int n_sphere,
                                    int sphere_sorted_by_y, float thr, int64_t max_rot_pairs, void* stream)
{
    if (n_items < 1 || n_items > TAIL_BATCH_MAX || !items || !dims || !sphere32 || !sphere64) return CPPF_EINVAL;
    if (n_rots < 1 || n_rots > CPPF_MAX_ROTS || n_sphere < 1 || n_sphere > 4096 || sphere_sorted_by_y == 0 || max_rot_pairs < 0) return CPPF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    TailBatch B = {};
    B.n = n_items; B.n_rots = n_rots; B.n_sphere = n_sphere; B.descending = sphere_sorted_by_y > 0 ? 1 : 0;
    {   // development knob (A/B measurements): the single-instance grouping of the sphere count in the batched launch
        static int legacy = -1;
        if (legacy < 0) legacy = getenv("CPPF_SPHERE_LEGACY") ? 1 : 0;
        B.sphere_legacy = legacy;
    }
    B.max_rot_pairs = max_rot_pairs; B.sphere32 = sphere32; B.thr = thr;
    CppfPairMlpItem sel_items[TAIL_BATCH_MAX];
    int n_second = 0, max_dirs = 1;
    int64_t bv_blocks = 1, cmp_blocks = 1, rot_blocks = 1;
    for (int i = 0; i < n_items; ++i) {
        const CppfPoseTailItem& it = items[i];
        if (!it.pc || !it.nrm || !it.idx32 || !it.outputs || !it.heads || !it.corner || !it.argmax_idx || !it.rec || !it.T32 ||
            !it.mask || !it.chunk_counts || !it.surv || !it.count || !it.counts || !it.ticket || !it.sums_workspace)
            return CPPF_EINVAL;
        if (it.n_pairs < 1 || it.n_pairs > 8192ll * CMP_BLOCK || it.n_points < 1 || it.n_dirs < 1 || it.n_dirs > PS_MAX_DIRS) return CPPF_EINVAL;
        if (!it.shape_dev && (it.gx < 1 || it.gy < 1 || it.gz < 1)) return CPPF_EINVAL;
        if (it.tail0_bytes && (!it.tail0 || (it.tail0_bytes & 15) || (reinterpret_cast<uintptr_t>(it.tail0) & 15) || it.tail0_bytes > (1u << 26)))
            return CPPF_EINVAL;
        if (it.sums_workspace_bytes < cppf_pose_sums_workspace_bytes()) return CPPF_EWORKSPACE;
        TailItem& I = B.item[i];
        I.idx = it.argmax_idx; I.corner = it.corner; I.res64 = it.res64; I.shape = it.shape_dev;
        I.T64 = it.rec; I.T32 = it.T32; I.peak = it.peak; I.idx_peak = it.rec + 19;
        I.zero16 = static_cast<uint4*>(it.tail0); I.n_zero16 = (int)(it.tail0_bytes / 16);
        I.points = it.pc; I.outputs = it.outputs; I.idx64 = it.idx64; I.idx32 = it.idx32; I.res = it.res; I.tol = it.tol; I.n_ppfs = it.n_pairs;
        I.gx = it.shape_dev ? 1 : it.gx; I.gy = it.shape_dev ? 1 : it.gy; I.gz = it.shape_dev ? 1 : it.gz;
        I.mask = it.mask; I.vote_ws = static_cast<const unsigned long long*>(it.vote_workspace); I.chunk_counts = it.chunk_counts;
        I.surv = it.surv; I.count = it.count; I.heads = it.heads; I.n_dirs = it.n_dirs; I.counts = it.counts;
        PoseSumsArgs& S = I.sums;
        S.pc = it.pc; S.nrm = it.nrm; S.aux = it.heads + 2; S.scale_logits = it.heads + 4;
        S.point_idxs = it.idx32; S.sel = it.surv; S.n_sel_dev = it.count; S.counts = it.counts; S.sphere64 = sphere64; S.n_sel_host = it.n_pairs;
        S.aux_stride = 8; S.scale_stride = 8; S.n_dirs = it.n_dirs; S.n_sphere = n_sphere; S.counts_dir_step = n_sphere;
        S.best_idx = it.best_idx; S.best_dir = it.rec + 3; S.sign = it.rec + 9; S.scale_out = it.rec + 15;
        S.partial = static_cast<double*>(it.sums_workspace); S.ticket = it.ticket;
        S.record_out = it.record_out; S.rec = it.rec; S.object_id_dev = it.object_id_dev; S.object_id_host = it.object_id_host;
        S.scale_mean[0] = it.scale_mean[0]; S.scale_mean[1] = it.scale_mean[1]; S.scale_mean[2] = it.scale_mean[2];
        S.regress_right = it.regress_right;
        int64_t nb = (it.n_pairs + 4 * 256 - 1) / (4 * 256);
        nb = nb > 256 ? 256 : nb;      // (every block loads the 21 KB rotation table: at least two trips per wave for it; 1024: 1 % slower)
        bv_blocks = nb > bv_blocks ? nb : bv_blocks;
        nb = (it.n_pairs + CMP_BLOCK - 1) / CMP_BLOCK;
        cmp_blocks = nb > cmp_blocks ? nb : cmp_blocks;
        const int64_t bound = it.n_pairs < max_rot_pairs ? it.n_pairs : max_rot_pairs;
        nb = (bound + 7) / 8;                          // (rot_sphere_band_even_body: a block owns a contiguous share of the survivors)
        nb = nb > SPHE_BLOCKS ? SPHE_BLOCKS : nb;
        if (B.sphere_legacy) { nb = (bound + 1) / 2; nb = nb > 2048 ? 2048 : nb; }
        rot_blocks = nb > rot_blocks ? nb : rot_blocks;
        max_dirs = it.n_dirs > max_dirs ? it.n_dirs : max_dirs;
        if (it.second_pass) {
            if (!it.feat || !it.packed || !it.u_rot || !it.mlp_workspace) return CPPF_EINVAL;
            CppfPairMlpItem& M = sel_items[n_second++];
            M = CppfPairMlpItem{};
            M.pc = it.pc; M.nrm = it.nrm; M.feat = it.feat; M.idxs = it.idx64 ? (const void*)it.idx64 : (const void*)it.idx32;
            M.packed = it.packed; M.u_rot = it.u_rot; M.heads = it.heads;
            M.workspace = it.mlp_workspace; M.workspace_bytes = it.mlp_workspace_bytes; M.n_points = it.n_points; M.n_pairs = it.n_pairs;
            M.idx_is_i64 = it.idx64 ? 1 : 0; M.sel = it.surv; M.n_sel_dev = it.count; M.max_sel = it.n_pairs;
        }
    }
    // 1. T = corner + unravel(arg-max) * res (:209-210); zeroes the record, the bin counts, the chunk counts and the tickets
    hipLaunchKernelGGL(tail_begin_batch_kernel, dim3((unsigned)n_items), dim3(256), 0, st, B);
    CPPF_CHECK_LAUNCH();
    // 2. back-vote filter (:216-231): mask + survivors per chunk of 1024 pairs; writes the int32 pair list on the way
    const int entries = tri(n_rots);
    const size_t lds_bv = (entries <= VOTE_TAB_LDS_MAX ? (size_t)entries * sizeof(float2) : 0) + 4 * 128 * sizeof(uint32_t);
    hipLaunchKernelGGL(backvote_batch_kernel, dim3((unsigned)bv_blocks, (unsigned)n_items), dim3(256), lds_bv, st, B);
    CPPF_CHECK_LAUNCH();
    // 3. order-preserving compaction of the survivors
    hipLaunchKernelGGL(compact_scatter_batch_kernel, dim3((unsigned)cmp_blocks, (unsigned)n_items), dim3(CMP_BLOCK), 0, st, B);
    CPPF_CHECK_LAUNCH();
    // 4. second MLP pass on the survivors (:236-256) for the items in their split form
    if (n_second > 0) {
        const int rc = cppf_pair_mlp_decode_sel_batch(n_second, sel_items, F, dims, n_res, out_dim, tr_bins, rot_bins, stream);
        if (rc != 0) return rc;
    }
    // 5. orientation vote + sphere-bin count (:259-284), both directions
    if (max_rot_pairs > 0) {
        const size_t lds_rot = (size_t)(5 * n_sphere + 2 * n_rots) * sizeof(float);
        static bool lds_attr = false;      // (4 096 bins x 20 B + the rotation row: above the 64 KB a launch gets unasked)
        if (!lds_attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rot_sphere_band_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            lds_attr = true;
        }
        hipLaunchKernelGGL(rot_sphere_band_batch_kernel, dim3((unsigned)rot_blocks, (unsigned)max_dirs, (unsigned)n_items), dim3(256), lds_rot, st, B);
        CPPF_CHECK_LAUNCH();
    }
    // 6. arg-max of the counts -> best_dir, sign sums (:287-301), scale sums (:335)
    hipLaunchKernelGGL(pose_sums_batch_kernel, dim3(RED_BLOCKS, (unsigned)n_items), dim3(RED_THREADS), 0, st, B);
    CPPF_CHECK_LAUNCH();
    return 0;
}

// ----------------------------------------------------------------------------- grid setup
// nocs/inference.py:194-195: corners = [min(pc), max(pc)]; grid_res = int32((max-min)/res) + 1
__global__ __launch_bounds__(1024) void grid_setup_kernel(const float* __restrict__ pc, int64_t N, float res,
                                                          float* __restrict__ corner, int32_t* __restrict__ dims)
{
    __shared__ float slo[16][3], shi[16][3];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = threadIdx.x; i < N; i += 1024)
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
        for (int w = 1; w < 16; ++w) { l = fminf(l, slo[w][j]); h = fmaxf(h, shi[w][j]); }
        corner[j] = l;
        dims[j] = (int32_t)((h - l) / res) + 1;
    }
}

extern "C" int cppf_grid_setup(const float* pc, int64_t N, float res, float* corner, int32_t* dims, void* stream)
{
    if (!pc || !corner || !dims || N < 1) return CPPF_EINVAL;
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pc, N, res, corner, dims);
    CPPF_CHECK_LAUNCH();
    return 0;
}
