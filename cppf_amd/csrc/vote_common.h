// Shared by vote.hip (centre vote) and pose_tail.hip (back-vote, compaction, orientation vote, pose reductions): launch check, the
// (cos, sin) rotation table and where a vote leaves it in its workspace, small wave-level helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cppf.h"
#include "cppf_math.h"

using namespace cppf;

#define CPPF_CHECK_LAUNCH()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// ----------------------------------------------------------------------------- rotation table
// tab[n*(n-1)/2 + i] = (cos, sin) of rotation i of n, n = 1..n_rots: every workgroup that needs it
// builds it in LDS in its prologue (2 628 entries for n_rots = 72, ~3 fp64 sincos per thread).
__device__ __forceinline__ void fill_rot_table(float2* ltab, int entries, int tid, int nthreads)
{
    for (int e = tid; e < entries; e += nthreads) {
        int n = (int)((sqrtf(8.f * (float)e + 1.f) + 1.f) * 0.5f);
        while (n * (n - 1) / 2 > e) --n;
        while ((n + 1) * n / 2 <= e) ++n;
        ltab[e] = rot_cs(e - n * (n - 1) / 2, n);
    }
}

// A cached table is trusted only if its stamp matches AND its 72 known entries are in place: rotation 0 of every n is
// exactly (1, 0), at offsets n(n-1)/2 spread over the whole table.  The stamp alone would survive a caller that reuses one
// arena for several entry points (or an allocator that recycles the block) and overwrites table bytes but not byte 248.
// Wave-uniform result; every wave of a launch reads the same memory and reaches the same verdict.
__device__ __forceinline__ bool rot_table_intact(const float2* wtab, int n_rots)
{
    const int lane = threadIdx.x & 63;
    bool ok = true;
    for (int n = lane + 1; n <= n_rots; n += 64) {
        const float2 v = wtab[n * (n - 1) / 2];
        ok = ok && v.x == 1.0f && v.y == 0.0f;
    }
    return !__any(!ok);
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int tri(int n) { return n * (n + 1) / 2; }

#define VOTE_TAB_LDS_MAX 2628   // (cos,sin) pairs kept in LDS: the triangular table of n_rots <= 72; more rotations: see WIDE   
// Workspace layout: [0, 256) arg-max keys and tickets; [256, VOTE_WS_PART) the (cos, sin) rotation table of the LAST launch
// that used this workspace, stamped with its n_rots at byte 248 -- building the table (2 628 fp64 sincos for 72 rotations)
// cost every workgroup ~5 us of its prologue, so the first launch on a workspace builds it in LDS and workgroup 0
// also leaves a copy here; later launches with the same n_rots find the stamp and load the 21 KB instead.  Nothing is kept
// outside the caller's workspace.  [VOTE_WS_PART, VOTE_WS_PART + VOTE_WS_V3_STATE): the queue header and the extra plane (V3Hdr);
// behind them the tile queues and the partial tiles.
#define VOTE_WS_TAB 256
#define VOTE_WS_PART (256 + ((VOTE_TAB_LDS_MAX + 2) * 8 + 255) / 256 * 256)
#define VOTE_WS_V3_STATE (8704 + 64 * 30720 * 8)
#define VOTE_TAB_STAMP 0x43505046726f7400ull   // "CPPFrot\0" ^ n_rots: table valid
// The launch that builds the table must not be able to read it back: workgroups of that same launch that start late
// (more workgroups than the chip holds at once) would see workgroup 0's stamp without any guarantee of seeing its table
// (no release/acquire between workgroups of one kernel).  So the vote kernel leaves a PENDING stamp and the reduce kernel
// that follows it -- a kernel boundary later -- turns it into the valid one.
#define VOTE_TAB_PENDING 0x43505046726f5000ull

__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// ----------------------------------------------------------------------------- reduce + arg-max
// grid[cell] (+)= sum_c partials[c][cell] in a fixed order, and the arg-max of the result with
// numpy's tie rule (first maximum in C order, nocs/inference.py:208): key = ord(value) << 32 |
// ~index, max-reduced in the wave, then one returning atomicMax per block.  A block = 64 cells x 16
// chunk groups (chunk c goes to group c % 16) with 8 loads in flight per lane, so the ~100 partial
// grids stream at L2/HBM rate instead of one dependent load at a time.  The last block to take a ticket unpacks the key into out_idx / out_val: every access to
// the key and the ticket is a device-scope atomic whose result is consumed before the next one is
// issued, so no cache maintenance is needed.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    return k;
}

// approximate inverse trigonometry of the arc screens (vote.hip) and the back-vote's angular window (pose_tail.hip): callers budget the errors
__device__ __forceinline__ float atan01_approx(float t)   // atan on [0, 1], |error| < 2e-5
{
    const float t2 = t * t;
    float p = fmaf(t2, 0.0208351f, -0.0851330f);
    p = fmaf(t2, p, 0.1801410f);
    p = fmaf(t2, p, -0.3302995f);
    p = fmaf(t2, p, 0.9998660f);
    return p * t;
}
__device__ __forceinline__ float atan2_approx(float y, float x)   // (-pi, pi], |error| < 1e-4; atan2(0, 0) = 0
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = atan01_approx(mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1e-30f)));
    r = ay > ax ? 1.57079633f - r : r;
    r = x < 0.f ? 3.14159265f - r : r;
    return __builtin_copysignf(r, y);
}
__device__ __forceinline__ float acos_approx(float u)   // u in [-1, 1]; |error| < 1e-4 (callers budget 2e-4)
{
    // Abramowitz & Stegun 4.4.45: acos(x) = sqrt(1 - x) (a0 + a1 x + a2 x^2 + a3 x^3) on [0, 1], |error| <= 6.7e-5; acos(-x) = pi - acos(x).
    // The square root carries the singularity at 1, so the error bound holds up to the end points (round 3: 9 instructions
    // instead of the 16 of the atan form -- every arc mask takes two).
    const float ax = fabsf(u);
    float p = fmaf(ax, -0.0187293f, 0.0742610f);
    p = fmaf(p, ax, -0.2121144f);
    p = fmaf(p, ax, 1.5707288f);
    const float r = p * __builtin_amdgcn_sqrtf(fmaxf(1.f - ax, 0.f));
    return u < 0.f ? 3.14159265f - r : r;
}
