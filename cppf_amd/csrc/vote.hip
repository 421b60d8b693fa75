// Centre vote + arg-max for gfx950 (MI355X): plan, arc screen, run walk, v3_bin / v3_vote / v3_reduce kernels, the batched and the
// integer forms.  C ABI in include/cppf.h; reference semantics cited per kernel (models/voting.py:8-66, nocs/inference.py:192-208).
// The rest of the pose chain (back-vote, compaction, orientation vote, reductions) lives in pose_tail.hip.
#include "vote_common.h"

// ----------------------------------------------------------------------------- vote plan
// ONE tiled implementation serves every n_rots (1..360) and every grid of up to 64 LDS tiles: v3_bin_kernel / v3_vote_kernel /
// v3_reduce_kernel further down (round 4 removed round 2's vote_kernel / reduce_tiles_kernel, which had survived for n_rots > 72);
// beyond 64 tiles -- none of the reference's categories -- vote_global_kernel restates the reference's own kernel with global
// fp32 atomics.  The plan (tiling, workgroups, fixed-point bits) is a pure function of (n_ppfs, n_rots, grid dims): host code
// evaluates it for a by-value launch, the *_dyn kernels evaluate the SAME functions on the device from a dims record.
#define VOTE_MAX_TILES 64       // beyond this the grid goes to global atomics (measured: 32 tiles still beat them 4-9x)
#define VOTE_WIN 72             // rotations of a pair one launch serves (the arc masks are 96-bit words): n_rots > 72 takes ceil(n_rots / 72) passes

// fixed-point bits of the largest weight: a workgroup deposits at most chunk_pairs*n_rots*(2^kk + 4) in
// total, and every 2^32 of that is one carry-log entry (VOTE_CARRY_CAP of them)
__host__ __device__ inline int vote_fixed_bits_of(int64_t chunk_pairs, int n_rots)
{
    const double cap = 2048.0 * 4294967296.0 / ((double)(chunk_pairs > 0 ? chunk_pairs : 1) * n_rots) - 4.0;
    int kk = 24;
    while (kk > 8 && (double)(1u << kk) > cap) --kk;
    return kk;
}


static int v3_fixed_bits_bound(int64_t n_ppfs, int n_rots, int gx, int gy, int gz);
static bool v3_eligible(int64_t n_ppfs, int n_rots, int gx, int gy, int gz);
static size_t v3_workspace_bytes(int64_t n_ppfs, int gx, int gy, int gz);
static size_t v3_workspace_bytes_dyn(int many_tiles, int64_t n_ppfs);
static int v3_fused_tiles();
// fixed-point bits of the largest weight in the tiled vote of this launch: exact for < 4 tiles; a LOWER bound for the binned path,
// whose scale follows the queues' lengths (a finer quantum than reported, never a coarser one); 0: global fp32 atomics, no quantisation
extern "C" int cppf_vote_fixed_point_bits(int64_t n_ppfs, int n_rots, int gx, int gy, int gz)
{
    if (n_rots < 1 || n_rots > CPPF_MAX_ROTS || gx < 1 || gy < 1 || gz < 1 || n_ppfs < 0) return -1;
    return v3_eligible(n_ppfs, n_rots, gx, gy, gz) ? v3_fixed_bits_bound(n_ppfs, n_rots, gx, gy, gz) : 0;
}

extern "C" size_t cppf_vote_workspace_bytes(int64_t n_ppfs, int n_rots, int gx, int gy, int gz)
{
    if (n_rots < 1 || n_rots > CPPF_MAX_ROTS || gx < 1 || gy < 1 || gz < 1 || n_ppfs < 0) return 0;
    // (global-atomics path and empty pair lists: the arg-max keys only -- but never less than the state a later tiled call on the same
    // allocation would need to find zeroed, so that one allocation sized for its largest call is always valid)
    if (!v3_eligible(n_ppfs, n_rots, gx, gy, gz)) return cppf_vote_workspace_init_bytes();
    return v3_workspace_bytes(n_ppfs, gx, gy, gz);
}
// the *_dyn launch: room for the pair -> tile queues (n_ppfs pairs, every tile of the class) and one partial tile per workgroup
extern "C" size_t cppf_vote_workspace_bytes_dyn_pairs(int many_tiles, int64_t n_ppfs)
{
    if (n_ppfs < 0) return 0;
    return v3_workspace_bytes_dyn(many_tiles, n_ppfs);
}
// ----------------------------------------------------------------------------- centre vote
// Reference: CUDA ppf_voting, models/voting.py:8-66.
// LDS accumulation is 32-bit FIXED POINT: on gfx950 ds_add_f32 costs ~195 cycles per wave instruction (3 cycles/lane, any address
// pattern) while ds_add_rtn_u32 costs ~22 (profiles/microbench/atomics_bench.hip).  A weight w is deposited as rn(w * S),
// S = 2^kk / p2 with p2 = max(probs) rounded up to a power of two and kk <= 24 chosen so that the number of 32-bit wrap-arounds a
// workgroup can produce fits the carry log; a wrap-around (detected on the returned old value) appends the cell to that log and
// is added back as 2^32 quanta when the tile is reduced.  With kk = 24 the quantum is 2^-24 of the largest weight: each deposit
// is exact to fp32 precision and the sum is order-independent, i.e. at least as accurate as any order of the reference's fp32
// atomicAdd.  Negative / non-finite probs fall back to ds_add_f32.
#define VOTE_PAIRQ 128  // per-wave queue of culled pair offsets: <= 63 waiting + 64 pushed
#define VOTE_CARRY_CAP 2048
// floor(x + 0.5) evaluated exactly, one instruction (x in [0, 2^24]: checked exhaustively, profiles/r2_div_check.txt); the
// fixed-point deposit's rounding (ties go up; any nearest rounding keeps the half-quantum error bound)
__device__ __forceinline__ uint32_t rpi_u32(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return (uint32_t)r;
}

// (double)g < 0.01  <=>  g < smallest float >= 0.01 ;  (double)g >= d  <=>  g >= smallest float >= d
__device__ __forceinline__ float ceil_to_float(double d)
{
    float f = (float)d;
    if ((double)f < d) f = __uint_as_float(__float_as_uint(f) + (f > 0.f ? 1u : -1u));
    return f;
}

// ----------------------------------------------------------------------------- arc screen
// Which rotations of a pair can land in the tile?  Along axis k the sample coordinate is
//   q_k(theta) = c_k + x_k cos(theta) + y_k sin(theta) = c_k + A_k cos(theta - phi_k),
// so lo <= q_k <= hi holds on the set  { alpha2 <= |theta - phi_k| <= alpha1 }  with alpha1 = acos((lo - c_k)/A_k),
// alpha2 = acos((hi - c_k)/A_k): an arc around phi_k minus a smaller arc around phi_k.  Rotation i of n sits at
// theta_i = 2 pi i / n, so each arc is a cyclic run of indices: a 96-bit mask per axis (n <= 72), and the AND of the three
// axes is a superset of the rotations the exact test of the deposit stage accepts -- cheap approximations with explicit
// slack (1e-4 in the cosine, 2e-3 rad in the angle, far above their error), never a dropped vote.  A pair costs ~300
// instructions for its masks instead of ~45 per pair of rotations in a loop over all of them.
#define VOTE_BELOW_N 97   // BELOW[j] = bits [0, j) set, j = 0..96, as uint4 (x, y, z = three words)

struct Mask96 { uint32_t a, b, c; };

// bits of the cyclic index run [ia, ib] (mod n) -- all of [0, n) when it has n or more members, none when ib < ia.
// WIDE (n_rots > 72: a pair has up to 360 rotations, a launch serves the WINDOW [win_base, win_base + 72) of them, see v3_launch):
// the caller has shifted the run by -win_base, bit k stands for rotation win_base + k, and only the first L = min(n - win_base, 72)
// bits exist -- the run, still cyclic mod n, is clipped to [0, L).
template <bool WIDE = false>
__device__ __forceinline__ Mask96 arc_run(const uint4* __restrict__ below, int ia, int ib, int n, int L = 0)
{
    int len = ib - ia + 1;
    len = len < 0 ? 0 : len;
    const bool full = len >= n;
    int s0 = ia < 0 ? ia + n : ia;
    s0 = s0 >= n ? s0 - n : s0;
    s0 = full ? 0 : (s0 < 0 ? 0 : s0);
    const int e = full ? n : s0 + len;
    if (WIDE) {
        const uint4 B1 = below[min(e, L)], B0 = below[min(s0, L)], B2 = below[min(max(e - n, 0), L)];
        return {(B1.x & ~B0.x) | B2.x, (B1.y & ~B0.y) | B2.y, (B1.z & ~B0.z) | B2.z};
    }
    const uint4 B1 = below[e < n ? e : n], B0 = below[s0], B2 = below[e > n ? e - n : 0];
    return {(B1.x & ~B0.x) | B2.x, (B1.y & ~B0.y) | B2.y, (B1.z & ~B0.z) | B2.z};
}
// WIDE: the arc's centre in window-local index units, wrapped to (-n/2, n/2] like the unshifted one
__device__ __forceinline__ float arc_shift(float f, int n, int win_base)
{
    const float g = f - (float)win_base;
    return g <= -0.5f * (float)n ? g + (float)n : g;
}

// rotations i of n whose coordinate c + x cos(theta_i) + y sin(theta_i) can lie in [lo, hi]; nf = n / (2 pi)
template <bool WIDE = false>
__device__ __forceinline__ Mask96 axis_arc_mask(const uint4* __restrict__ below, float c, float x, float y, float lo, float hi,
                                                float nf, int n, int win_base = 0, int L = 0)
{
    const float A = __builtin_amdgcn_sqrtf(fmaf(x, x, y * y));
    const float rA = __builtin_amdgcn_rcpf(fmaxf(A, 1e-20f));
    const float u1 = fmaf(lo - c, rA, -1e-4f);   // cos(theta - phi) >= u1
    const float u2 = fmaf(hi - c, rA, 1e-4f);    // cos(theta - phi) <= u2
    const bool none = !(u1 <= 1.f) || !(u2 >= -1.f);   // (NaN-safe: a NaN coordinate drops the pair like the exact test does)
    const float a1 = u1 <= -1.f ? 3.14159265f : acos_approx(fminf(u1, 1.f));
    const float a2 = u2 >= 1.f ? 0.f : acos_approx(fmaxf(u2, -1.f));
    float f = atan2_approx(y, x) * nf;
    if (WIDE) f = arc_shift(f, n, win_base);
    const float W1 = (a1 + 2e-3f) * nf, W2 = (a2 - 2e-3f) * nf;
    const Mask96 O = arc_run<WIDE>(below, (int)ceilf(f - W1), (int)floorf(f + W1), n, L);
    // excluded: the integers strictly inside (f - W2, f + W2)
    const Mask96 I = arc_run<WIDE>(below, (int)floorf(f - W2) + 1, W2 > 0.f ? (int)ceilf(f + W2) - 1 : -(1 << 20), n, L);
    const uint32_t keep = none ? 0u : 0xffffffffu;
    return {O.a & ~I.a & keep, O.b & ~I.b & keep, O.c & ~I.c & keep};
}

// ----------------------------------------------------------------------------- run walk (see v3_vote_kernel)
// index of the lowest set bit of the 96-bit word (a, b, c), 96 when it is empty (v_ffbl_b32 returns -1 for 0: the OR keeps it)
__device__ __forceinline__ int ctz96(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t f0 = (uint32_t)(__ffs((int)a) - 1), f1 = (uint32_t)(__ffs((int)b) - 1) | 32u, f2 = (uint32_t)(__ffs((int)c) - 1) | 64u;
    return (int)min(min(f0, f1), min(f2, 96u));
}
// one past the highest set bit, 0 when empty
__device__ __forceinline__ int top96(uint32_t a, uint32_t b, uint32_t c)
{
    const int ea = 32 - __clz((int)a), tb = 32 - __clz((int)b), tc = 32 - __clz((int)c);
    return max(ea, max(tb ? tb + 32 : 0, tc ? tc + 64 : 0));
}
// A mask of rotation indices as up to three runs [s, e) in ascending order: the first two runs of set bits exactly, the
// third = the hull of everything above them.  Empty runs come out as s = e (or e < s for the third: callers clamp).
__device__ __forceinline__ void mask_runs(const uint4* __restrict__ below, uint32_t m0, uint32_t m1, uint32_t m2, int& s0, int& e0,
                                          int& s1, int& e1, int& s2, int& e2)
{
    s0 = ctz96(m0, m1, m2);
    uint4 B = below[s0];
    e0 = ctz96(~(m0 | B.x), ~(m1 | B.y), ~(m2 | B.z));   // first clear bit at or above s0 (bits >= n are clear: e0 <= n)
    B = below[e0];
    s1 = ctz96(m0 & ~B.x, m1 & ~B.y, m2 & ~B.z);
    B = below[s1];
    e1 = ctz96(~(m0 | B.x), ~(m1 | B.y), ~(m2 | B.z));
    B = below[e1];
    const uint32_t r0 = m0 & ~B.x, r1 = m1 & ~B.y, r2 = m2 & ~B.z;
    s2 = ctz96(r0, r1, r2);
    e2 = top96(r0, r1, r2);
}
// inclusive prefix sum over the 64 lanes: four row_shr steps inside each row of 16, then the row totals (row_bcast 15 / 31)
#define RED_GROUPS 16
#define RED_MAX_BLOCKS 256
__global__ __launch_bounds__(64 * RED_GROUPS) void reduce_argmax_kernel(float* __restrict__ grid,
                                                                        const float* __restrict__ partials, int chunks,
                                                                        int64_t G, unsigned long long* packed,
                                                                        int accumulate, int write_back,
                                                                        long long* out_idx, float* out_val)
{
    __shared__ float part[RED_GROUPS][64];
    const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int64_t ngroups = (G + 63) / 64;
    unsigned long long key = 0ull;
    // grid-stride over 64-cell groups: few blocks, so the two same-address atomics per block at the end
    // (~12 ns each, serialised chip-wide) stay negligible
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t cell = grp * 64 + lane;
        float s = 0.f;
        if (cell < G) {
            int c = cg;
            for (; c + 7 * RED_GROUPS < chunks; c += 8 * RED_GROUPS) {  // 8 independent loads in flight per lane
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = partials[(int64_t)(c + k * RED_GROUPS) * G + cell];
#pragma unroll
                for (int k = 0; k < 8; ++k) s = s + v[k];
            }
            for (; c < chunks; c += RED_GROUPS) s = s + partials[(int64_t)c * G + cell];
        }
        part[cg][lane] = s;
        __syncthreads();
        if (cg == 0 && cell < G) {
            float v = accumulate ? grid[cell] : 0.f;
#pragma unroll
            for (int k = 0; k < RED_GROUPS; ++k) v = v + part[k][lane];
            if (write_back) grid[cell] = v;
            const unsigned long long kk =
                ((unsigned long long)f2ord(v) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)cell);
            key = kk > key ? kk : key;
        }
        __syncthreads();
    }
    if (cg != 0) return;
    key = wave_max_u64(key);
    if (lane == 0) {
        const unsigned long long old = atomicMax(packed, key);          // returning: completes before the ticket
        unsigned d1 = (unsigned)old, d2;
        asm volatile("v_mov_b32 %0, %1" : "=v"(d2) : "v"(d1));
        unsigned* ticket = reinterpret_cast<unsigned*>(packed + 1);
        const unsigned tk = atomicAdd(ticket, 1u + (d1 ^ d2));           // d1 ^ d2 == 0, data-dependent on `old`
        if (tk == gridDim.x - 1) {
            const unsigned long long best = atomicMax(packed, 0ull);
            if (out_idx) *out_idx = (long long)(0xffffffffu - (uint32_t)(best & 0xffffffffull));
            if (out_val) *out_val = ord2f((uint32_t)(best >> 32));
        }
    }
}

#define RED_CELLS 256
#define RED_FANIN 8   // groups of the two-level arg-max (power of two)

// ============================================================================ binned tiled vote ("v3", round 3)
// The tiled vote for n_rots <= 72 and <= 64 tiles, rebuilt around three observations from round 2's counters (the kernel is
// bound by VALU issue at 97 %: every stage is priced by instruction count):
//   1. Per PAIR work -- the exact pair frame, the three arc masks -- was repeated by every (tile, chunk) workgroup that met
//      the pair, after a culling test that every workgroup of every tile ran over every pair of its chunk (T x P tests).
//      `v3_bin_kernel` now walks the pairs ONCE: frame, per-axis arc parameters, then the masks of only the tiles the
//      circle's bounding box touches (the x mask of a tile column, the y mask of a tile row, one z mask), turned into index
//      runs (mask_runs) and appended as a 12-byte record {pair, runs} to that tile's queue in HBM -- through an LDS staging
//      area, so that a workgroup spends one global atomic per tile per flush.
//   2. Tiles OWN candidates by their floor cell and carry a one-cell halo on their cut sides (x0 + tx, y0 + ty), so the owner
//      deposits all eight corners: no in-tile tests, no zeroed factors, no dummy words, and a candidate next to a cut is no
//      longer processed by two to four tiles.  The ownership test is folded into the bounds of the reference's in-grid test
//      (floor(g) >= x0  <=>  g >= x0): it costs nothing.
//   3. Partial tiles leave the chip as RAW 32-bit fixed point and `v3_reduce_kernel` adds them as 64-bit integers (halo
//      rows of the neighbours included, logged 2^32 wrap-arounds added back) before ONE conversion to fp32: the grid is the
//      exact sum of the quantised deposits whatever workgroup took whichever record -- results are bit-identical from run to
//      run and between the by-value, *_dyn and pair-sharded forms although the queues fill in a racy order.
// `v3_vote_kernel`, workgroup (tile t, chunk c), consumes records [c n_t / C, (c+1) n_t / C) of tile t's queue: exact frame
// again (12 B/record of HBM traffic instead of 52), then the run walk of vote_kernel.
#define V3_TILE_FLOATS 30720      // 120 KiB of LDS for the tile incl. its halo
#ifndef V3_STAGE
#define V3_STAGE 3072             // staged records per flush (16 B each in LDS: 48 KiB; 72 KiB with the rings and pair queues: two workgroups per CU)
#endif
#define V3_MAGIC 0x43503356u
#define V3_THREADS 1024
#define V3_BIN_THREADS 512
#define V3_BIN_SR 8               // blocks of 64 pairs a wave culls per super-round
#define V3_IRING 256              // per-wave ring of (lane, tile) items: <= 63 waiting + 64 pushed
#define V3_CNT_STRIDE 32
#define V3_RED_FANIN 16              // arg-max groups of v3_reduce_kernel (power of two, <= VOTE_MAX_TILES)
struct V3Hdr {                    // workspace + VOTE_WS_PART; all zero (or left by a previous call) when a launch starts
    unsigned int tile_count[VOTE_MAX_TILES * V3_CNT_STRIDE];   // records in each tile's queue, one counter per 128-byte line; words 2..5
                                  // of lines 0..V3_RED_FANIN-1: {key, count} of the reduce kernel's arg-max groups (zero between launches)
    unsigned int flags;           // 1: the workspace was not initialised / a capacity was exceeded -> arg-max -1, peak NaN
    unsigned int any_extra;       // some workgroup added to the extra plane in this launch
    unsigned int magic;           // 0 (fresh, zeroed) or V3_MAGIC
    unsigned int fmt;             // partial tiles: 0 = raw fixed point, 1 = fp32 (negative / non-finite probs)
    float quantum;                // value of one fixed-point unit (p2 / 2^kk)
    unsigned int done;            // ticket of the reduce blocks (the last one re-zeroes this header)
    unsigned int pad[2];
};
// Behind the header: the EXTRA PLANE, one u64 per grid cell, for what a workgroup's partial tile cannot hand to the reduce
// kernel in place: the words of its HALO (cells the neighbour tile owns; non-zero ones only, a few hundred per workgroup) and
// 2^32 per logged wrap-around of a 32-bit cell.  Added with device-scope 64-bit atomics when the tile is dumped; zero between
// launches: the reduce kernel reads it with the cell and clears what it finds.  (fp32 partial tiles: the low word is a float.)
#define V3_HDR_BYTES ((sizeof(V3Hdr) + 255) / 256 * 256)
#define V3_PLANE_BYTES ((size_t)VOTE_MAX_TILES * V3_TILE_FLOATS * sizeof(unsigned long long))
static_assert(V3_HDR_BYTES + V3_PLANE_BYTES <= VOTE_WS_V3_STATE, "VOTE_WS_V3_STATE too small");

struct V3Tiling { int tx, ty, ntx, nty, T, hx, hy; };   // hx / hy: the grid is cut along x / y (tiles carry a halo column / row)
// fewest tiles, then least cut area between tiles (a vote circle is a curve: the tiles it passes through grow with the area of the cuts
// it can cross: 26x19x52 rather than 7x76x52 for the 52x152x52 grid of config 5); a tile of tx x ty owned cells occupies (tx + hx)(ty + hy) gz words
__host__ __device__ inline V3Tiling v3_tiling(int gx, int gy, int gz)
{
    V3Tiling p = {0, 0, 0, 0, 1 << 30, 0, 0};
    int64_t best_cut = 0;
    if (gz > V3_TILE_FLOATS) return p;
    for (int nty0 = 1; nty0 <= gy; ++nty0) {
        const int ty = (gy + nty0 - 1) / nty0;
        const int nty = (gy + ty - 1) / ty;
        const int hy = nty > 1 ? 1 : 0;
        if ((int64_t)(ty + hy) * gz > V3_TILE_FLOATS) continue;
        const int cols = (int)(V3_TILE_FLOATS / ((int64_t)(ty + hy) * gz));   // columns of the LDS tile, halo included
        int tx, ntx, hx;
        if (cols >= gx) { tx = gx; ntx = 1; hx = 0; }
        else {
            if (cols < 2) continue;
            ntx = (gx + cols - 2) / (cols - 1);
            tx = (gx + ntx - 1) / ntx;
            ntx = (gx + tx - 1) / tx;
            hx = 1;
        }
        const int T = ntx * nty;
        const int64_t cut = ((int64_t)(ntx - 1) * gy + (int64_t)(nty - 1) * gx) * gz;
        if (T < p.T || (T == p.T && cut < best_cut)) {
            p.T = T; p.tx = tx; p.ty = ty; p.ntx = ntx; p.nty = nty; p.hx = hx; p.hy = hy;
            best_cut = cut;
        }
        if (ntx == 1) break;   // more y cuts can only add tiles
    }
    return p;
}
__host__ __device__ inline int v3_slot_words(const V3Tiling& t, int gz) { return ((t.tx + t.hx) * (t.ty + t.hy) * gz + 3) & ~3; }
// workgroups of the vote launch: one per CU (a workgroup fills a CU's LDS, so 256 of them run as ONE round: with two to four rounds
// the per-workgroup prologue / dump and the quantisation of the last round cost 7 % (known-answer) to 19 % (uniform bins) of the
// C5 vote, profiles/r3_vote_phases.txt), never more than one per 512 pairs and tile
#define V3_WGS 256
__host__ __device__ inline int v3_wgs(int64_t n_ppfs, int T)
{
    int64_t w = V3_WGS;
    const int64_t wmax = ((n_ppfs + 511) / 512) * T;
    if (w > wmax) w = wmax;
    return (int)(w < T ? T : w);
}
// record range of chunk c of a tile holding n records cut into C chunks: boundaries at multiples of 64
__host__ __device__ inline unsigned v3_bound(unsigned n, int c, int C)
{
    if (c >= C) return n;
    const unsigned b = (unsigned)(((unsigned long long)n * (unsigned)c / (unsigned)C + 63ull) & ~63ull);
    return b < n ? b : n;
}
// fixed-point bits: a workgroup deposits at most (records of its chunk) x n_rots weights <= 1; every 2^32 quanta of that is one
// carry-log entry (VOTE_CARRY_CAP of them per workgroup)
// (a launch deposits at most VOTE_WIN rotations of a pair: the passes of n_rots > 72 each choose their own scale)
__host__ __device__ inline int v3_bits(unsigned chunk_records, int n_rots) { return vote_fixed_bits_of((int64_t)chunk_records + 64, n_rots > VOTE_WIN ? VOTE_WIN : n_rots); }

// fused form: the pair list is dealt to a tile's C workgroups in blocks of 64, block b to workgroup b mod C (neighbouring pairs share
// their first point, so contiguous chunks would inherit the cloud's unevenness); the most pairs a workgroup can get
__host__ __device__ inline int64_t v3_fused_chunk_pairs(int64_t n_ppfs, int C)
{
    const int64_t nb = (n_ppfs + 63) / 64;
    return (nb + C - 1) / (C > 0 ? C : 1) * 64;
}
// The fixed-point scale of the fused vote does NOT follow the launch's width: it is the scale of the NARROWEST launch an object can
// get (64 workgroups: cppf_vote_argmax's smallest CPPF_VOTE_WORKGROUPS and the width of an object in a batch of four or more) or of
// the actual one if that is narrower still (short pair lists).  Every width a caller can choose then quantises every deposit alike,
// and since the grid is the exact integer sum of the quantised deposits it is the SAME BITS whether an object is voted alone on 256
// workgroups, on 128 beside its neighbours or as one of eight in a batched launch (round 4: one bit per halving of the width, so a
// pose record's peak depended on how the batch was cut).  Cost: two bits at full width -- 2^-22 instead of 2^-24 of the largest
// weight per deposit at C2, still an order of magnitude below the fp32 rounding of the reference's own atomicAdd sums.
#define V3_BITS_WGS 64
__host__ __device__ inline int64_t v3_fused_bits_pairs(int64_t n_ppfs, int C, int T)
{
    const int Cc = V3_BITS_WGS / (T > 0 ? T : 1) > 0 ? V3_BITS_WGS / (T > 0 ? T : 1) : 1;
    return v3_fused_chunk_pairs(n_ppfs, C < Cc ? C : Cc);
}

struct V3Plan { V3Tiling t; int wgs, slot; size_t pool_off, frames_off, part_off, total; int64_t pool_cap; };
// pool: T queues of `cap` records (12 B) each -- every pair can visit every tile, and the HBM is there (288 GB) -- and, behind them,
// one 48-byte FRAME per pair (V3_FRAME_BYTES: the bin kernel computes a pair's exact frame once, every tile's consumer loads it)
#define V3_FRAME_BYTES 48
static V3Plan v3_plan(int64_t n_ppfs, const V3Tiling& t, int gz, int wgs, int64_t cap, int64_t cells)
{
    V3Plan p;
    p.t = t; p.wgs = wgs; p.slot = v3_slot_words(t, gz); p.pool_cap = cap;
    // (the extra plane has a FIXED place and size -- the most cells a tiled grid can have -- whatever the grid: a workspace serves
    // calls with different grids, and the plane's "zero between launches" invariant must not depend on the previous layout)
    p.pool_off = VOTE_WS_PART + V3_HDR_BYTES + V3_PLANE_BYTES;
    (void)cells;
    p.frames_off = p.pool_off + align_up((size_t)t.T * (size_t)cap * 12, 256);
    p.part_off = p.frames_off + (cap > 0 ? align_up((size_t)n_ppfs * V3_FRAME_BYTES, 256) : 0);
    p.total = p.part_off + (size_t)wgs * p.slot * sizeof(uint32_t);   // one partial tile per workgroup
    return p;
}

struct V3Args {
    const float* points;
    const float* outputs;
    const float* probs;
    const void* point_idxs;
    int idx64;
    const float* corner;
    float res;
    int64_t n_ppfs, n_points;
    int n_rots, adaptive, gx, gy, gz;
    V3Tiling t;
    int wgs;               // workgroups of the vote launch = partial tiles; the tiles share them in proportion to their queues
    int t_cap;             // *_dyn: most tiles this launch serves
    const int32_t* shape;  // {n_points, gx, gy, gz} in device memory, or null
    int64_t grid_cap;
    V3Hdr* hdr;
    unsigned long long* plane;   // extra plane, one u64 per grid cell
    uint32_t* pool;        // [T][pool_cap][3]
    int64_t pool_cap;
    float4* frames;        // binned: [n_ppfs][3] = {cc.xyz, x.x | x.yz, y.xy | y.z, n, prob, -}: a pair's exact frame, written by the bin kernel
    uint32_t* partials;    // [wgs][slot]: workgroup b's tile
    unsigned long long* packed;
    float* grid;
    int accumulate;
    long long* out_idx;
    float* out_val;
    int tab_entries;
    int bin_sr;            // bin kernel: blocks of 64 pairs a wave culls per super-round
    int fused;             // < 4 tiles: no queues, workgroup (tile, chunk) culls and screens its own pairs (chunks = wgs / T)
    int kk;                // fused, by-value launches: fixed-point bits (the chunks are static)
    int win_base;          // n_rots > 72 (WIDE kernels): this launch serves rotations [win_base, win_base + 72) of every pair
    int kk_force;          // != 0: the caller fixes the fixed-point bits (pair-sharded votes: every rank must quantise alike)
    long long* grid_raw;   // != null: the reduce kernel also stores every cell's exact sum of quanta (i64[gx*gy*gz]; grid may be null)
    float* quantum_out;    // != null: value of one quantum of grid_raw (0: fp32 partial tiles, grid_raw is not valid)
};

// *_dyn: the plan from the dims record, identically in all three kernels; false = the record does not fit the launch
__device__ __forceinline__ bool v3_resolve(const V3Args& A, int& gx, int& gy, int& gz, int64_t& n_points, V3Tiling& t)
{
    gx = A.gx; gy = A.gy; gz = A.gz; n_points = A.n_points; t = A.t;
    if (!A.shape) return true;
    n_points = A.shape[0]; gx = A.shape[1]; gy = A.shape[2]; gz = A.shape[3];
    if (n_points < 1 || n_points > A.n_points || gx < 1 || gy < 1 || gz < 1 || (int64_t)gx * gy * gz > A.grid_cap) return false;
    t = v3_tiling(gx, gy, gz);
    return t.T <= A.t_cap && t.T <= A.wgs;
}

// The tiles share the launch's workgroups in proportion to their queues: tile t gets C_t = 1 + floor(n_t E / W) chunks (E =
// workgroups minus non-empty tiles, W = all records; an empty tile gets none), workgroup b = base_t + c works on chunk c of tile t
// and owns partial tile b.  Evaluated by the first wavefront of a workgroup into LDS (sp[0..63] = C_t, [64..127] = base_t,
// [128..191] = n_t, [192] = largest chunk in records); the vote and the reduce kernel evaluate the same function of the same counters.
__device__ __forceinline__ void v3_split(const V3Args& A, int T, int* sp)
{
    const int lane = threadIdx.x & 63;
    if (A.fused) {   // static chunks of the pair list, the same number for every tile; "records" = pairs of a chunk (never empty)
        if (threadIdx.x < 64) {
            const int C = lane < T ? A.wgs / T : 0;
            sp[lane] = C; sp[64 + lane] = lane * C; sp[128 + lane] = 0x7fffffff;
            if (lane == 0) sp[192] = (int)v3_fused_bits_pairs(A.n_ppfs, A.wgs / T, T);   // (what the fixed-point scale follows)
        }
        return;
    }
    if (threadIdx.x < 64) {
        const unsigned n = lane < T ? min(A.hdr->tile_count[lane * V3_CNT_STRIDE], (unsigned)A.pool_cap) : 0u;
        const int nonempty = __popcll(__ballot(n > 0u));
        unsigned long long W = n;
        for (int off = 32; off > 0; off >>= 1) W += __shfl_xor(W, off, 64);
        const unsigned long long E = (unsigned long long)max(A.wgs - nonempty, 0);
        const int C = n > 0u ? 1 + (int)((unsigned long long)n * E / (W > 0ull ? W : 1ull)) : 0;
        const int incl = wave_incl_scan(C);
        // the scale follows the largest chunk of the NARROWEST launch an object can get (V3_BITS_WGS workgroups, as for the fused
        // vote: see v3_fused_bits_pairs), or of this one if it is narrower still: whatever width a caller or a batch chooses, every
        // deposit is quantised alike and the grid -- the exact integer sum -- is the same bits
        const unsigned long long Eref = (unsigned long long)max(min(A.wgs, V3_BITS_WGS) - nonempty, 0);
        const int Cref = n > 0u ? 1 + (int)((unsigned long long)n * Eref / (W > 0ull ? W : 1ull)) : 0;
        int big = Cref > 0 ? (int)((n + (unsigned)Cref - 1u) / (unsigned)Cref) : 0;
        for (int off = 32; off > 0; off >>= 1) big = max(big, __shfl_xor(big, off, 64));
        sp[lane] = C; sp[64 + lane] = incl - C; sp[128 + lane] = (int)n;
        if (lane == 0) sp[192] = big;
    }
}
// (chunk c of a tile of n records in C chunks is empty unless its boundaries differ; workgroup 0 always dumps its tile)
__device__ __forceinline__ bool v3_chunk_live(unsigned n, int c, int C, int b) { return b == 0 || v3_bound(n, c, C) < v3_bound(n, c + 1, C); }

__device__ __forceinline__ int2 v3_pair_idx(const V3Args& A, int64_t p)
{
    if (A.idx64) {
        const longlong2 v = reinterpret_cast<const longlong2*>(A.point_idxs)[p];
        return make_int2((int)v.x, (int)v.y);
    }
    return reinterpret_cast<const int2*>(A.point_idxs)[p];
}

// the part of axis_arc_mask that does not depend on the tile: amplitude, its reciprocal, phase in index units
struct AxisArc { float c, rA, f; };
template <bool WIDE = false>
__device__ __forceinline__ AxisArc axis_arc_prep(float c, float x, float y, float nf, int n = 0, int win_base = 0)
{
    const float A = __builtin_amdgcn_sqrtf(fmaf(x, x, y * y));
    AxisArc a;
    a.c = c;
    a.rA = __builtin_amdgcn_rcpf(fmaxf(A, 1e-20f));
    a.f = atan2_approx(y, x) * nf;
    if (WIDE) a.f = arc_shift(a.f, n, win_base);
    return a;
}
template <bool WIDE = false>
__device__ __forceinline__ Mask96 axis_arc_eval(const uint4* __restrict__ below, const AxisArc& a, float lo, float hi, float nf, int n,
                                                int L = 0)
{
    const float u1 = fmaf(lo - a.c, a.rA, -1e-4f);   // cos(theta - phi) >= u1
    const float u2 = fmaf(hi - a.c, a.rA, 1e-4f);    // cos(theta - phi) <= u2
    const bool none = !(u1 <= 1.f) || !(u2 >= -1.f);
    const float a1 = u1 <= -1.f ? 3.14159265f : acos_approx(fminf(u1, 1.f));
    const float a2 = u2 >= 1.f ? 0.f : acos_approx(fmaxf(u2, -1.f));
    const float W1 = (a1 + 2e-3f) * nf, W2 = (a2 - 2e-3f) * nf;
    const Mask96 O = arc_run<WIDE>(below, (int)ceilf(a.f - W1), (int)floorf(a.f + W1), n, L);
    const Mask96 I = arc_run<WIDE>(below, (int)floorf(a.f - W2) + 1, W2 > 0.f ? (int)ceilf(a.f + W2) - 1 : -(1 << 20), n, L);
    const uint32_t keep = none ? 0u : 0xffffffffu;
    return {O.a & ~I.a & keep, O.b & ~I.b & keep, O.c & ~I.c & keep};
}

__device__ __forceinline__ int wave_max_i32(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// ---------------------------------------------------------------------------- v3_bin_kernel
// One pass over the pairs, in super-rounds of bin_sr x 512 per workgroup: cull against the whole grid and queue the survivors (per
// wave, LDS); then in batches of 64 survivors: frame in approximate arithmetic, per-axis arc parameters, z mask; the tiles of the
// circle's bounding box walked with a cheap plane / shell test, the (lane, tile) items that pass compacted through a per-wave ring;
// per item the x / y masks of its column / row, AND, runs, and a record into the tile's queue.  Records are staged in LDS (slot
// within the tile's share of the flush from an LDS atomic) and flushed with one global atomic per tile.
template <bool WIDE>
__global__ __launch_bounds__(V3_BIN_THREADS) void v3_bin_kernel(V3Args A)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    uint4* stage = reinterpret_cast<uint4*>(lds_u);                          // [V3_STAGE] {pair, runs a, runs b | tile << 24, slot}
    uint4* below = reinterpret_cast<uint4*>(stage + V3_STAGE);               // [VOTE_BELOW_N]
    uint32_t* cnt = reinterpret_cast<uint32_t*>(below + VOTE_BELOW_N);       // [64] records of each tile in this flush
    uint32_t* gbase = cnt + VOTE_MAX_TILES;                                   // [64] their place in the tile's queue
    uint32_t* ctl = gbase + VOTE_MAX_TILES;                                   // [0] staged, [1] a wave could not stage
    uint16_t* iring = reinterpret_cast<uint16_t*>(ctl + 16) + (threadIdx.x >> 6) * V3_IRING;   // this wave's (lane, tile) items
    uint32_t* pq = reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(ctl + 16) + (V3_BIN_THREADS / 64) * V3_IRING) + (threadIdx.x >> 6) * (64 * V3_BIN_SR + 64);   // this wave's culled pairs
    const int tid = threadIdx.x, lane = tid & 63;
    int gx, gy, gz;
    int64_t n_points;
    V3Tiling t;
    if (blockIdx.x == 0 && tid < 20) A.packed[tid] = 0ull;   // arg-max keys + tickets of the reduce kernel, its error-path ticket [18]
    if (blockIdx.x == 0 && tid == 20 && A.win_base == 0) A.packed[20] = 0ull;   // "an earlier pass of this call gave up" (n_rots > 72)
    if (!v3_resolve(A, gx, gy, gz, n_points, t)) return;   // (the reduce kernel reports it)
    {
        const unsigned m = A.hdr->magic;
        if (m != 0u && m != V3_MAGIC) { if (tid == 0) atomicOr(&A.hdr->flags, 1u); return; }
    }
    if (tid < VOTE_BELOW_N) {
        const int j = tid;
        auto w = [](int c) { return c <= 0 ? 0u : (c >= 32 ? 0xffffffffu : ((1u << c) - 1u)); };
        below[j] = make_uint4(w(j), w(j - 32), w(j - 64), 0u);
    }
    if (tid < VOTE_MAX_TILES) cnt[tid] = 0u;
    if (tid == 0) { ctl[0] = 0u; ctl[1] = 0u; ctl[2] = 0u; }
    for (int i = tid; i < V3_STAGE; i += V3_BIN_THREADS) stage[i].w = 0xffffffffu;   // "empty" mark of a staging slot
    const f3 cr = {A.corner[0], A.corner[1], A.corner[2]};
    const float res = A.res, rinv = 1.0f / res;
    const float ptxf = (float)t.tx, ptyf = (float)t.ty;
    const float rtx = 1.0f / ptxf, rty = 1.0f / ptyf;
    __syncthreads();
    // per-lane state of the batch a wave is working on (one pair per lane), set by setup()
    int64_t p = 0;
    bool valid = false;
    f3 Fcc = {0.f, 0.f, 0.f}, Fx = Fcc, Fy = Fcc, Fu = Fcc, cq = Fcc, xq = Fcc, yq = Fcc;
    float Rq = 0.f, ex = 0.f, ey = 0.f, ez = 0.f, nf = 0.f, slq = 0.f;   // Rq: the circle's radius in cells
    int n = 0, ix0 = 0, ix1 = -1, iy0 = 0, iy1 = -1, nxw = 0, nyw = 0, nsteps = 0, step = 0, wdx = 0, wdy = 0, qhead = 0, qtail = 0;
    const int wb = WIDE ? A.win_base : 0;   // WIDE: this launch bins rotations [wb, wb + 72) of every pair (bit k of a mask = rotation wb + k)
    AxisArc ax = {0.f, 0.f, 0.f}, ay = ax;
    Mask96 mz = {0u, 0u, 0u};
    // ---- cull: a pair whose circle cannot reach the grid at all (a random-weight network: half of them at C5) is dropped before any
    // of the above is computed -- bounding box, plane and shell against the whole grid (vote_kernel's test, ~60 instructions) -- and the
    // survivors are compacted through a per-wave queue so that the frame, the arc parameters and the tile walk run with full lanes.
    // On a trained network's inputs nearly every circle passes: a wave whose last culled block kept >= 48 of 64 takes its next seven
    // blocks as they are (v3_vote_kernel<true> does the same).
    const float gbx = 0.5f * ((float)gx - 1.f), gby = 0.5f * ((float)gy - 1.f), gbz = 0.5f * ((float)gz - 1.f);       // box centre
    const float ghx = 0.5f * ((float)gx - 1.02f), ghy = 0.5f * ((float)gy - 1.02f), ghz = 0.5f * ((float)gz - 1.02f);   // half extents
    auto cull = [&](const int64_t q) -> bool {
        const float2 o = reinterpret_cast<const float2*>(A.outputs)[q];
        const int2 ij = v3_pair_idx(A, q);
        const f3 a = ld3(A.points, ij.x), b = ld3(A.points, ij.y);
        const f3 d = sub3(a, b);
        const float L = __builtin_amdgcn_sqrtf(fmaf(d.z, d.z, fmaf(d.y, d.y, d.x * d.x)));
        const f3 u = scl3(d, __builtin_amdgcn_rcpf(L + 1e-7f));
        const f3 cc = {fmaf(-u.x, o.x, a.x), fmaf(-u.y, o.x, a.y), fmaf(-u.z, o.x, a.z)};
        const float R = fabsf(o.y) * rinv;
        const float ax_ = R * __builtin_amdgcn_sqrtf(fmaxf(0.f, 1.f - u.x * u.x)), ay_ = R * __builtin_amdgcn_sqrtf(fmaxf(0.f, 1.f - u.y * u.y)),
                    az_ = R * __builtin_amdgcn_sqrtf(fmaxf(0.f, 1.f - u.z * u.z));
        const float sl = fmaf(R, 1.1e-3f, 4e-3f);
        const float sx = sl + 8e-6f * (fabsf(cc.x) + fabsf(cr.x)) * rinv, sy = sl + 8e-6f * (fabsf(cc.y) + fabsf(cr.y)) * rinv,
                    sz = sl + 8e-6f * (fabsf(cc.z) + fabsf(cr.z)) * rinv;
        const float qx = (cc.x - cr.x) * rinv, qy = (cc.y - cr.y) * rinv, qz = (cc.z - cr.z) * rinv;
        const float dx = gbx - qx, dy = gby - qy, dz = gbz - qz, sall = sl + (sx - sl) + (sy - sl) + (sz - sl);
        const float off_plane = fabsf((dx * u.x + dy * u.y) + dz * u.z);
        const float reach = (fabsf(u.x) * ghx + fabsf(u.y) * ghy) + fabsf(u.z) * ghz;
        const float nx = fmaxf(fabsf(dx) - ghx, 0.f), ny = fmaxf(fabsf(dy) - ghy, 0.f), nz = fmaxf(fabsf(dz) - ghz, 0.f);
        const float fx = fabsf(dx) + ghx, fy = fabsf(dy) + ghy, fz = fabsf(dz) + ghz;
        const float dmin2 = (nx * nx + ny * ny) + nz * nz, dmax2 = (fx * fx + fy * fy) + fz * fz;
        const float r_hi = R + sall, r_lo = fmaxf(R - sall, 0.f);
        return (L >= 0.9e-7f) & (!A.adaptive | (R >= 0.15f)) & (!WIDE | !A.adaptive | (fmaf(R, 6.2832f, 1.f) > (float)wb)) &   // (n > win_base, generously)
               (qx + ax_ + sx >= 0.01f) & (qx - ax_ - sx < (float)gx - 1.01f) & (qy + ay_ + sy >= 0.01f) & (qy - ay_ - sy < (float)gy - 1.01f) &
               (qz + az_ + sz >= 0.01f) & (qz - az_ - sz < (float)gz - 1.01f) &
               (off_plane <= reach + sall) & (dmin2 <= r_hi * r_hi * 1.0001f) & (dmax2 * 1.0001f >= r_lo * r_lo);
    };
    auto setup = [&]() {
            // ---- per pair: the EXACT frame (pair_frame: :20-28 with their IEEE divisions and square roots, ~180 instructions), once:
            // it feeds the arc screen here (whose acceptance boxes stay widened as for round 3-5's approximate frame) and is left in
            // `frames` for the consumers -- a pair visits 3.4 tiles at C5 and 4-5 on a posed object, and every visit used to recompute
            // it from the pair list, the points and (mu, nu): 48 bytes of dependent gathers replaced by 48 bytes of one load level.
            Fcc = {0.f, 0.f, 0.f}; Fx = Fcc; Fy = Fcc; Fu = Fcc;
            Rq = 0.f;
            n = 0;
            if (valid) {
                const float2 o = reinterpret_cast<const float2*>(A.outputs)[p];
                const int2 ij = v3_pair_idx(A, p);
                f3 a, ab, xd;
                float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0, f2 = f0;       // (a degenerate pair, :21: n = 0, never queued)
                if (pair_frame(A.points, ij.x, ij.y, a, ab, xd)) {                   // :20-28, the consumer's arithmetic
                    Fu = ab;
                    Rq = fabsf(o.y) * rinv;
                    Fcc = sub3(a, scl3(ab, o.x));                                    // :23
                    Fx = scl3(xd, o.y);                                              // :28
                    Fy = cross3(Fx, ab);                                             // :29
                    n = A.n_rots;
                    if (A.adaptive) n = min((int)((double)(o.y / res) * (2 * CPPF_PI)), A.n_rots);   // :31
                    n = max(n, 0);
                    const float prob = A.probs ? fmaxf(A.probs[ij.x], A.probs[ij.y]) : 1.f;
                    f0 = make_float4(Fcc.x, Fcc.y, Fcc.z, Fx.x); f1 = make_float4(Fx.y, Fx.z, Fy.x, Fy.y);
                    f2 = make_float4(Fy.z, __int_as_float(n), prob, 0.f);
                    if (WIDE && n <= wb) n = 0;   // no rotation of this pair in the window
                }
                float4* fr = A.frames + 3 * p;
                fr[0] = f0; fr[1] = f1; fr[2] = f2;
            }
            const int Lw = WIDE ? max(min(n - wb, 72), 0) : n;   // bits of this pair's masks
            cq = scl3(sub3(Fcc, cr), rinv); xq = scl3(Fx, rinv); yq = scl3(Fy, rinv);
            // slack of the screen: the exact-frame form (1e-6 of the terms + 1e-3 cells, see vote_kernel) plus the approximate frame's
            // own error -- rcp / sqrt are good to 1 ulp, so u, x, y carry a few 1e-7 relative: <= ~5e-7 m on a sample for |mu|, nu up to
            // 0.3 m (1e-6 m at the 1.9 m of the SUN RGB-D categories), i.e. <= 2.5e-4 cells at res 2e-3 -- hence 4e-6 and 2e-3
            ex = fmaf((fabsf(Fcc.x) + fabsf(cr.x) + fabsf(Fx.x) + fabsf(Fy.x)) * rinv, 4e-6f, 2e-3f);
            ey = fmaf((fabsf(Fcc.y) + fabsf(cr.y) + fabsf(Fx.y) + fabsf(Fy.y)) * rinv, 4e-6f, 2e-3f);
            ez = fmaf((fabsf(Fcc.z) + fabsf(cr.z) + fabsf(Fx.z) + fabsf(Fy.z)) * rinv, 4e-6f, 2e-3f);
            nf = (float)n * 0.159154943f;
            ax = axis_arc_prep<WIDE>(cq.x, xq.x, yq.x, nf, n, wb); ay = axis_arc_prep<WIDE>(cq.y, xq.y, yq.y, nf, n, wb);
            const AxisArc az = axis_arc_prep<WIDE>(cq.z, xq.z, yq.z, nf, n, wb);
            mz = axis_arc_eval<WIDE>(below, az, 0.01f - ez, (float)gz - 1.01f + ez, nf, n, Lw);
            // tiles the circle's bounding box touches: coordinate range [c - A, c + A] (+ slack) against owned ranges [x0, x0 + tx)
            const float Ax = __builtin_amdgcn_sqrtf(fmaf(xq.x, xq.x, yq.x * yq.x)) * 1.0001f + ex + 2e-3f;
            const float Ay = __builtin_amdgcn_sqrtf(fmaf(xq.y, xq.y, yq.y * yq.y)) * 1.0001f + ey + 2e-3f;
            const float xlo = fmaxf(cq.x - Ax, 0.f), xhi = fminf(cq.x + Ax, (float)gx - 1.f);
            const float ylo = fmaxf(cq.y - Ay, 0.f), yhi = fminf(cq.y + Ay, (float)gy - 1.f);
            const bool alive = n > 0 && (mz.a | mz.b | mz.c) != 0u && xlo <= xhi && ylo <= yhi;   // (NaN coordinates: not alive)
            ix0 = 0; ix1 = -1; iy0 = 0; iy1 = -1;
            if (alive) {
                // (a float quotient may land one tile off at a boundary: one extra tile either side costs an empty mask, never a vote)
                ix0 = max((int)(xlo * rtx) - 1, 0); ix1 = min((int)(xhi * rtx) + 1, t.ntx - 1);
                iy0 = max((int)(ylo * rty) - 1, 0); iy1 = min((int)(yhi * rty) + 1, t.nty - 1);
                // trim: a column / row the box does not reach
                ix0 += ((float)((ix0 + 1) * t.tx) <= xlo) ? 1 : 0;
                ix1 -= ((float)(ix1 * t.tx) > xhi) ? 1 : 0;
                iy0 += ((float)((iy0 + 1) * t.ty) <= ylo) ? 1 : 0;
                iy1 -= ((float)(iy1 * t.ty) > yhi) ? 1 : 0;
            }
            nxw = __builtin_amdgcn_readfirstlane(wave_max_i32(ix1 - ix0 + 1));       // this wave's walk (wave-uniform: scalar registers,
            nyw = __builtin_amdgcn_readfirstlane(wave_max_i32(iy1 - iy0 + 1));       // scalar branches in the walk below)
            nsteps = nxw * nyw; wdx = 0; wdy = 0;
            slq = fmaf(Rq, 1.1e-3f, 4e-3f);
            step = 0; qhead = 0; qtail = 0;
    };
    // ---- (pair, tile) items.  A circle's bounding box covers many tiles while the curve passes through few (all of them against
    // 0.75 on uniform-bin inputs at C5, 12 against 3.4 on known-answer inputs), and a per-lane loop over the box runs every lane
    // to the wave's longest range.  So the box is walked with a CHEAP test only -- the tile's box must reach the circle's plane,
    // and the radius must lie between the box's nearest and farthest distance from the centre (vote_kernel's plane / shell test,
    // ~35 instructions) -- and the (lane, tile) pairs that pass go through a per-wave ring in LDS; whenever 64 are queued every
    // lane takes one, pulls the owning lane's arc parameters (ds_bpermute) and does the expensive part with full lanes: the x and
    // y masks of that tile, AND with the z mask, runs, record.  Records are staged in LDS across rounds and flushed -- one global
    // atomic per tile -- when the area is half full, when a wave finds it full (it retries its batch after the flush), and after
    // the workgroup's last round.
    auto produce = [&]() {
        const int ix = ix0 + wdx, iy = iy0 + wdy;   // step = wdx nyw + wdy
        bool pass = false;
        if (ix <= ix1 && iy <= iy1) {
            const float x0f = (float)(ix * t.tx), y0f = (float)(iy * t.ty);
            const float bx0 = fmaxf(0.01f, x0f), bx1 = fminf((float)gx - 1.01f, x0f + ptxf);
            const float by0 = fmaxf(0.01f, y0f), by1 = fminf((float)gy - 1.01f, y0f + ptyf);
            const float hx_ = 0.5f * (bx1 - bx0) + ex, hy_ = 0.5f * (by1 - by0) + ey, hz_ = 0.5f * ((float)gz - 1.02f) + ez;
            const float ddx = 0.5f * (bx0 + bx1) - cq.x, ddy = 0.5f * (by0 + by1) - cq.y, ddz = 0.5f * ((float)gz - 1.f) - cq.z;
            const float off_plane = fabsf(fmaf(ddz, Fu.z, fmaf(ddy, Fu.y, ddx * Fu.x)));
            const float reach = fmaf(fabsf(Fu.z), hz_, fmaf(fabsf(Fu.y), hy_, fabsf(Fu.x) * hx_));
            const float nx_ = fmaxf(fabsf(ddx) - hx_, 0.f), ny_ = fmaxf(fabsf(ddy) - hy_, 0.f), nz_ = fmaxf(fabsf(ddz) - hz_, 0.f);
            const float fx_ = fabsf(ddx) + hx_, fy_ = fabsf(ddy) + hy_, fz_ = fabsf(ddz) + hz_;
            const float dmin2 = fmaf(nz_, nz_, fmaf(ny_, ny_, nx_ * nx_)), dmax2 = fmaf(fz_, fz_, fmaf(fy_, fy_, fx_ * fx_));
            const float r_hi = Rq + slq, r_lo = fmaxf(Rq - slq, 0.f);
            pass = (bx0 < bx1) & (by0 < by1) & (off_plane <= reach + slq) & (dmin2 <= r_hi * r_hi * 1.0001f) &
                   (dmax2 * 1.0001f >= r_lo * r_lo);
        }
        const unsigned long long m = __ballot(pass);
        if (pass) {
            const int pos = qtail + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            iring[pos & (V3_IRING - 1)] = (uint16_t)(lane | ((ix * t.nty + iy) << 6));
        }
        qtail += __popcll(m);
        ++step;
        if (++wdy == nyw) { wdy = 0; ++wdx; }
    };
    // one item per lane: masks of its tile, runs, record; false (nothing consumed) when the staging area cannot take the batch
    auto consume = [&](const int count) -> bool {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned code = iring[(qhead + lane) & (V3_IRING - 1)];
        const int src = (int)(code & 63u), tile = (int)(code >> 6);
        // (all 64 lanes execute the pulls)
        AxisArc bx_, by_;
        bx_.c = __shfl(ax.c, src, 64); bx_.rA = __shfl(ax.rA, src, 64); bx_.f = __shfl(ax.f, src, 64);
        by_.c = __shfl(ay.c, src, 64); by_.rA = __shfl(ay.rA, src, 64); by_.f = __shfl(ay.f, src, 64);
        const float sex = __shfl(ex, src, 64), sey = __shfl(ey, src, 64);
        const int sn = __shfl(n, src, 64);
        const uint32_t z0 = (uint32_t)__shfl((int)mz.a, src, 64), z1 = (uint32_t)__shfl((int)mz.b, src, 64), z2 = (uint32_t)__shfl((int)mz.c, src, 64);
        const uint32_t sp_ = (uint32_t)__shfl((int)(uint32_t)p, src, 64);
        uint32_t m0 = 0u, m1 = 0u, m2 = 0u;
        if (lane < count) {
            const int ix = tile / t.nty, iy = tile - ix * t.nty;
            const float x0f = (float)(ix * t.tx), y0f = (float)(iy * t.ty);
            const float snf = (float)sn * 0.159154943f;
            const int sL = WIDE ? max(min(sn - wb, 72), 0) : sn;
            const Mask96 mx = axis_arc_eval<WIDE>(below, bx_, fmaxf(0.01f, x0f) - sex, fminf((float)gx - 1.01f, x0f + ptxf) + sex, snf, sn, sL);
            const Mask96 my = axis_arc_eval<WIDE>(below, by_, fmaxf(0.01f, y0f) - sey, fminf((float)gy - 1.01f, y0f + ptyf) + sey, snf, sn, sL);
            m0 = mx.a & my.a & z0; m1 = mx.b & my.b & z1; m2 = mx.c & my.c & z2;
        }
        const bool have = (m0 | m1 | m2) != 0u;
        const unsigned long long hm = __ballot(have);
        const int nh = __popcll(hm);
        if (nh == 0) return true;
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(&ctl[0], (uint32_t)nh);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + nh > V3_STAGE) {   // full: ask for a flush and retry this batch afterwards (the slots it drew stay empty: the
            if (lane == 0) ctl[1] = 1u;   // flush copies marked entries only)
            return false;
        }
        if (have) {
            int s0, e0, s1, e1, s2, e2;
            mask_runs(below, m0, m1, m2, s0, e0, s1, e1, s2, e2);
            const int l2 = max(e2 - s2, 0);
            const uint32_t slot = atomicAdd(&cnt[tile], 1u);
            const uint32_t ra = (uint32_t)s0 | ((uint32_t)(e0 - s0) << 8) | ((uint32_t)s1 << 16) | ((uint32_t)(e1 - s1) << 24);
            const uint32_t rb = (uint32_t)s2 | ((uint32_t)l2 << 8) | ((uint32_t)tile << 24);
            const int fi = base + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0));
            stage[fi] = make_uint4(sp_, ra, rb, slot);
        }
        return true;
    };
    // Super-rounds of bin_sr x 512 pairs per workgroup: every wave first culls its 64 bin_sr pairs into its queue (independent
    // load chains back to back: the latency is paid once, not per 64 pairs), then the workgroup runs batch rounds -- one full batch of
    // the queue per wave and round, the flush protocol between them -- until no wave has 64 left (none at all after the last super-round).
    int qn = 0;
    const int srb = A.bin_sr;   // blocks of 64 pairs per wave and super-round (<= V3_BIN_SR; fewer when the pair list is short)
    const int64_t n_sr = (A.n_ppfs + (int64_t)V3_BIN_THREADS * srb - 1) / ((int64_t)V3_BIN_THREADS * srb);
    for (int64_t sr = blockIdx.x; sr < n_sr; sr += gridDim.x) {
        const bool last_sr = sr + gridDim.x >= n_sr;
        const int64_t base = sr * ((int64_t)V3_BIN_THREADS * srb) + (int64_t)(tid >> 6) * (64 * srb);
        int dense = 0;   // the super-round's first block kept >= 48 of 64: the others are queued as they are
#pragma unroll 2
        for (int blk = 0; blk < srb; ++blk) {
            const int64_t pin = base + blk * 64 + lane;
            if (base + blk * 64 >= A.n_ppfs) break;
            bool pass = pin < A.n_ppfs && (dense || cull(pin));
            if (blk == 0 && __popcll(__ballot(pass)) >= 48) { dense = 1; pass = pin < A.n_ppfs; }   // (whole blocks: the batches stay full)
            const unsigned long long m = __ballot(pass);
            if (pass) pq[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = (uint32_t)pin;
            qn += __popcll(m);
        }
        const int nb = (qn >> 6) + ((last_sr && (qn & 63)) ? 1 : 0);   // this wave's batches
        if (lane == 0) atomicMax(&ctl[2], (uint32_t)nb);
        __syncthreads();
        const int nbmax = (int)ctl[2];
        for (int round = 0; round < nbmax; ++round) {
        const bool last_round = last_sr && round == nbmax - 1;
        bool todo = round < nb, busy = false;
        for (;;) {
            // this wave: its batch of the round, walked and consumed until its items are done or the staging area is full
            bool stuck = false;
            while (!stuck) {
                if (!busy) {
                    if (!todo) break;
                    todo = false;
                    const int take = qn < 64 ? qn : 64;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    valid = lane < take;
                    p = valid ? (int64_t)pq[qn - take + lane] : 0;
                    qn -= take;
                    setup();
                    busy = true;
                }
                const int queued = qtail - qhead;
                if (queued >= 64 || (step >= nsteps && queued > 0)) {
                    const int take = queued < 64 ? queued : 64;
                    if (consume(take)) qhead += take; else stuck = true;
                } else if (step < nsteps) {
                    produce();
                } else {
                    busy = false;
                }
            }
            __syncthreads();
            const uint32_t staged = min(ctl[0], (uint32_t)V3_STAGE);
            const bool again = ctl[1] != 0u;
            if (again || last_round || staged >= V3_STAGE / 2) {
                if (tid < t.T && cnt[tid] != 0u) {
                    const uint32_t b = atomicAdd(&A.hdr->tile_count[tid * V3_CNT_STRIDE], cnt[tid]);
                    gbase[tid] = b;
                    if ((int64_t)b + cnt[tid] > A.pool_cap) atomicOr(&A.hdr->flags, 1u);
                }
                __syncthreads();
                for (uint32_t i = tid; i < staged; i += V3_BIN_THREADS) {
                    const uint4 rec = stage[i];
                    if (rec.w == 0xffffffffu) continue;   // a slot a refused batch drew
                    const uint32_t tile = rec.z >> 24;
                    const int64_t pos = (int64_t)gbase[tile] + rec.w;
                    if (pos < A.pool_cap) {
                        uint32_t* dst = A.pool + ((int64_t)tile * A.pool_cap + pos) * 3;
                        dst[0] = rec.x; dst[1] = rec.y; dst[2] = rec.z & 0xffffffu;
                    }
                    stage[i].w = 0xffffffffu;
                }
                __syncthreads();
                if (tid < VOTE_MAX_TILES) cnt[tid] = 0u;
                if (tid == 0) { ctl[0] = 0u; ctl[1] = 0u; }
            }
            __syncthreads();
            if (!again) break;
        }
        }
        if (tid == 0) ctl[2] = 0u;   // (read by everybody before the rounds' barriers; the next super-round's atomicMax comes after them)
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------- v3_vote_kernel
struct V3Tile {
    uint32_t* tile;        // LDS tile, (tx + hx) x (ty + hy) x gz words
    uint32_t* carry_log;
    int* carry_n;
    int x0, y0, gz, ltyz;  // ltyz: words per x column of the LDS tile
    float res, rres, S;    // S > 0: fixed-point scale; S == 0: fp32 atomics
    double rres64;         // 1 / res in fp64
    int unit_probs;
    float lox, hix, loy, hiy, loz, hiz;   // the reference's in-grid bounds intersected with the cells this tile owns
};

// models/voting.py:35-63 for one sample whose floor cell this tile owns: every corner lies in the LDS tile (halo included)
__device__ __forceinline__ void v3_deposit(const V3Tile& T, f3 v, float prob)
{
    // :35, the IEEE fp32 quotient through ONE fp64 product: fl32(fl64(v * fl64(1 / res))) -- for fp32 operands the exact quotient is at
    // least 2^-49 (relative) away from every fp32 rounding boundary while the two fp64 roundings err by < 2^-52, so the result is the
    // correctly rounded v / res; v_cvt_f64_f32 + v_mul_f64 + v_cvt_f32_f64 issue in 15 cycles against the 20 of div_by's mul + 4 fma
    // (profiles/r3_valu_rates.txt; exhaustive comparison with `/`: profiles/microbench/div_check.hip)
    const f3 g = {(float)((double)v.x * T.rres64), (float)((double)v.y * T.rres64), (float)((double)v.z * T.rres64)};
    // :36-39 (fp64 tests folded to fp32 thresholds) and the ownership of the floor cell (floor(g) >= x0 <=> g >= x0) in one test
    if (!(g.x >= T.lox && g.x < T.hix && g.y >= T.loy && g.y < T.hiy && g.z >= T.loz && g.z < T.hiz)) return;
    const int fx = (int)g.x, fy = (int)g.y, fz = (int)g.z;         // :40
    const float rx = __builtin_amdgcn_fractf(g.x), ry = __builtin_amdgcn_fractf(g.y), rz = __builtin_amdgcn_fractf(g.z);
    const float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
    const float ll = w0x * w0y, lh = w0x * ry, hl = rx * w0y, hh = rx * ry;
    const int b = __mul24(fx - T.x0, T.ltyz) + (__mul24(fy - T.y0, T.gz) + fz);
    if (T.S > 0.f) {
        // fixed point: inc = rn(weight * S), S a power of two riding on the last factor (see vote_deposit)
        float lll, llh, lhl, lhh, hll, hlh, hhl, hhh;
        if (T.unit_probs) {
            const float z0 = w0z * T.S, z1 = rz * T.S;
            lll = ll * z0; llh = ll * z1; lhl = lh * z0; lhh = lh * z1;
            hll = hl * z0; hlh = hl * z1; hhl = hh * z0; hhh = hh * z1;
        } else {
            const float ps = prob * T.S;
            lll = ll * w0z * ps; llh = ll * rz * ps; lhl = lh * w0z * ps; lhh = lh * rz * ps;
            hll = hl * w0z * ps; hlh = hl * rz * ps; hhl = hh * w0z * ps; hhh = hh * rz * ps;
        }
        uint32_t* t0 = T.tile + b;
        uint32_t* t2 = t0 + T.gz;
        uint32_t* t4 = t0 + T.ltyz;
        uint32_t* t6 = t4 + T.gz;
        const uint32_t i0 = rpi_u32(lll), i1 = rpi_u32(llh), i2 = rpi_u32(lhl), i3 = rpi_u32(lhh);
        const uint32_t i4 = rpi_u32(hll), i5 = rpi_u32(hlh), i6 = rpi_u32(hhl), i7 = rpi_u32(hhh);
        const uint32_t o0 = atomicAdd(t0, i0), o1 = atomicAdd(t0 + 1, i1), o2 = atomicAdd(t2, i2), o3 = atomicAdd(t2 + 1, i3),
                       o4 = atomicAdd(t4, i4), o5 = atomicAdd(t4 + 1, i5), o6 = atomicAdd(t6, i6), o7 = atomicAdd(t6 + 1, i7);
        const uint32_t om = max(max(max(o0, o1), max(o2, o3)), max(max(o4, o5), max(o6, o7)));
        if (om >= 0xfe000000u) {   // a wrap needs old >= 2^32 - inc with inc <= 2^24
            auto wrapped = [&](uint32_t o, uint32_t inc, int a) {
                if (o + inc < o) {
                    const int slot = atomicAdd(T.carry_n, 1);
                    if (slot < VOTE_CARRY_CAP) T.carry_log[slot] = (uint32_t)a;
                }
            };
            wrapped(o0, i0, b); wrapped(o1, i1, b + 1); wrapped(o2, i2, b + T.gz); wrapped(o3, i3, b + T.gz + 1);
            wrapped(o4, i4, b + T.ltyz); wrapped(o5, i5, b + T.ltyz + 1); wrapped(o6, i6, b + T.ltyz + T.gz);
            wrapped(o7, i7, b + T.ltyz + T.gz + 1);
        }
        return;
    }
    float* t = reinterpret_cast<float*>(T.tile) + b;
    atomicAdd(t, ll * w0z * prob); atomicAdd(t + 1, ll * rz * prob);
    atomicAdd(t + T.gz, lh * w0z * prob); atomicAdd(t + T.gz + 1, lh * rz * prob);
    atomicAdd(t + T.ltyz, hl * w0z * prob); atomicAdd(t + T.ltyz + 1, hl * rz * prob);
    atomicAdd(t + T.ltyz + T.gz, hh * w0z * prob); atomicAdd(t + T.ltyz + T.gz + 1, hh * rz * prob);
}

#define V3_LDS_HEAD (VOTE_CARRY_CAP * 4 + (V3_THREADS / 64) * VOTE_PAIRQ * 4 + 64 + VOTE_BELOW_N * 16)
// `bid`: this workgroup's index within ITS launch -- blockIdx.x, or blockIdx.x minus the first workgroup of its item when the votes
// of several objects share one launch (v3_vote_batch_kernel)
template <bool FUSED, bool WIDE>
__device__ __forceinline__ void v3_vote_body(const V3Args& A, const int bid)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: [carry log 8 KiB][pair queues 16 x 128 x u32 = 8 KiB (fused mode)][ctrl 64 B][arc-mask table 97 x 16 B]
    //      [rotation table (+2 spare)][tile incl. halo]
    uint32_t* carry_log = reinterpret_cast<uint32_t*>(lds);
    uint32_t* pairq = carry_log + VOTE_CARRY_CAP + (threadIdx.x >> 6) * VOTE_PAIRQ;
    int* ctrl = reinterpret_cast<int*>(carry_log + VOTE_CARRY_CAP + (V3_THREADS / 64) * VOTE_PAIRQ);   // [0] carry count, [4] next batch
    uint4* below = reinterpret_cast<uint4*>(ctrl + 16);
    float2* ltab = reinterpret_cast<float2*>(below + VOTE_BELOW_N);
    uint32_t* tile = reinterpret_cast<uint32_t*>(ltab + A.tab_entries + 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef V3_TRACE   // development aid (profiles/microbench/vote_trace.py): wall-clock stamps per workgroup in the unused tail of the extra plane
    unsigned long long* trace = A.plane + 1500000 + (int64_t)bid * 32;
    if (tid == 0) trace[0] = wall_clock64();
#endif
    int gx, gy, gz;
    int64_t n_points;
    V3Tiling pt;
    // Everything the prologue reads from HBM is requested HERE, in one round trip (header words, the cached rotation table's stamp,
    // probe entries and -- speculatively -- the table itself, the queue counters inside v3_split): read one after the other behind the
    // branches that use them they were four dependent round trips, 5 us of a 38 us launch.
    float2* wtab = reinterpret_cast<float2*>(reinterpret_cast<char*>(A.packed) + VOTE_WS_TAB);
    const unsigned hmagic = A.hdr->magic, hflags = A.hdr->flags;
    const unsigned long long tstamp = A.packed[31];
    float2 probe[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int n_ = lane + 1 + 64 * u; probe[u] = (!WIDE && n_ <= A.n_rots) ? wtab[n_ * (n_ - 1) / 2] : make_float2(1.f, 0.f); }
    // (WIDE, n_rots > 72: the table of every (n, i) would not fit LDS -- 520 KB at 360 -- so tab_entries = 0 and a sample's (cos, sin)
    // comes from rot_cs(), the same fp64 evaluation the table is filled with; a rarely used knob, nocs/inference.py:39)
    float2 tab_in[(VOTE_TAB_LDS_MAX + V3_THREADS - 1) / V3_THREADS];
#pragma unroll
    for (int u = 0; u < (VOTE_TAB_LDS_MAX + V3_THREADS - 1) / V3_THREADS; ++u)
        tab_in[u] = tid + u * V3_THREADS < A.tab_entries ? wtab[tid + u * V3_THREADS] : make_float2(0.f, 0.f);
    const int wb = WIDE ? A.win_base : 0;
    if (FUSED && bid == 0 && tid < 20) A.packed[tid] = 0ull;   // arg-max keys + tickets of the reduce kernel (binned: the bin kernel did)
    if (FUSED && bid == 0 && tid == 20 && wb == 0) A.packed[20] = 0ull;   // "an earlier pass of this call gave up" (n_rots > 72)
    if (!v3_resolve(A, gx, gy, gz, n_points, pt)) return;
    int* sp = reinterpret_cast<int*>(carry_log);   // (the carry log is unused until the main loop)
    v3_split(A, pt.T, sp);
    if (FUSED && hmagic != 0u && hmagic != V3_MAGIC) { if (tid == 0) atomicOr(&A.hdr->flags, 1u); return; }   // workspace never initialised (binned: the bin kernel checked)
    if (hflags & 1u) return;
    __syncthreads();
    // this workgroup's tile and chunk (a uniform scan of <= 64 tiles)
    int t = -1, c = 0, Ct = 0;
    unsigned n_t = 0u;
    for (int k = 0; k < pt.T; ++k) {
        const int ck = sp[k], bk = sp[64 + k];
        if (bid >= bk && bid < bk + ck) { t = k; c = bid - bk; Ct = ck; n_t = (unsigned)sp[128 + k]; }
    }
    // every workgroup must use the same scale: the launch's largest chunk (fused, by-value: fixed on the host, like the plan)
    const int kk = A.kk_force ? A.kk_force : ((FUSED && !A.shape) ? A.kk : v3_bits((unsigned)sp[192], A.n_rots));
    __syncthreads();   // (sp is the carry log: everybody has read it)
    if (t < 0 && bid != 0) return;            // more workgroups than chunks
    // binned: records [r0, r1) of the tile's queue; fused: pairs [p0, p1) of the pair list
    unsigned r0 = 0u, r1 = 0u;
    int64_t p0 = 0, p1 = 0;
    if (FUSED) {   // blocks c, c + Ct, c + 2 Ct, ... of 64 pairs (v3_fused_chunk_pairs)
        p0 = 0; p1 = A.n_ppfs;
        if ((int64_t)c * 64 >= p1 && bid != 0) return;
    } else {
        if (t >= 0) { r0 = v3_bound(n_t, c, Ct); r1 = v3_bound(n_t, c + 1, Ct); }
        if (r0 >= r1 && bid != 0) return;     // nothing queued for this chunk: no tile to zero or dump (the reduce kernel skips it too)
    }
    if (t < 0) t = 0;                                // (workgroup 0 of a launch without a single record: an empty tile, the launch-wide duties)
    const int tix = t / pt.nty, tiy = t - tix * pt.nty;
    const int x0 = tix * pt.tx, y0 = tiy * pt.ty;
    const int tx = min(pt.tx, gx - x0), ty = min(pt.ty, gy - y0);
    const int tyh = pt.ty + pt.hy;                 // LDS rows per column: uniform over the tiles (the slot layout of the partials)
    const int ltyz = tyh * gz;
    const int nwords = (tx + pt.hx) * ltyz;

    // prologue (as vote_kernel): probs scan and rotation table loads first, tile zeroed meanwhile, one barrier
    float pm = A.probs ? 0.f : 1.f;
    int bad = 0, nonunit = 0;
    if (A.probs) {
        for (int64_t k0 = tid; k0 < n_points; k0 += 4 * V3_THREADS) {
            float pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pv[u] = k0 + u * V3_THREADS < n_points ? A.probs[k0 + u * V3_THREADS] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool there = k0 + u * V3_THREADS < n_points;
                bad |= !(pv[u] >= 0.f) || !(pv[u] < INFINITY);
                nonunit |= there && pv[u] != 1.0f;
                pm = fmaxf(pm, pv[u]);
            }
        }
    }
    // (the cached table is trusted if its stamp matches and rotation 0 of every n is exactly (1, 0): rot_table_intact's test)
    const bool probe_ok = probe[0].x == 1.0f && probe[0].y == 0.0f && probe[1].x == 1.0f && probe[1].y == 0.0f;
    const bool tab_cached = WIDE || (tstamp == (VOTE_TAB_STAMP ^ (unsigned long long)A.n_rots) && !__any(!probe_ok));   // (WIDE: no table at all)
    if (tid < 16) ctrl[tid] = 0;
    for (int k = tid; k < (nwords + 3) >> 2; k += V3_THREADS) reinterpret_cast<uint4*>(tile)[k] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < VOTE_BELOW_N) {
        const int j = tid;
        auto w = [](int c_) { return c_ <= 0 ? 0u : (c_ >= 32 ? 0xffffffffu : ((1u << c_) - 1u)); };
        below[j] = make_uint4(w(j), w(j - 32), w(j - 64), 0u);
    }
    if (tab_cached) {
#pragma unroll
        for (int u = 0; u < (VOTE_TAB_LDS_MAX + V3_THREADS - 1) / V3_THREADS; ++u)
            if (tid + u * V3_THREADS < A.tab_entries) ltab[tid + u * V3_THREADS] = tab_in[u];
    } else {
        fill_rot_table(ltab, A.tab_entries, tid, V3_THREADS);
        if (bid == 0) {   // leave a copy for the next launch (visible to it: kernel boundary)
            __syncthreads();
            for (int e = tid; e < A.tab_entries; e += V3_THREADS) wtab[e] = ltab[e];
            if (tid == 0) A.packed[31] = VOTE_TAB_PENDING ^ (unsigned long long)A.n_rots;
        }
    }
    if (tid < 2) ltab[A.tab_entries + tid] = make_float2(0.f, 0.f);
    // the waves' summaries of the probs go to the (still unused) carry log: two words per wave
    for (int off = 32; off > 0; off >>= 1) pm = fmaxf(pm, __shfl_xor(pm, off, 64));
    {
        const int flags = (__any(bad) ? 1 : 0) | (__any(nonunit) ? 2 : 0);
        if (lane == 0) { reinterpret_cast<float*>(carry_log)[2 * wave] = pm; carry_log[2 * wave + 1] = (uint32_t)flags; }
    }
    __syncthreads();
    float S = 0.f;
    int unit_probs = 0;
    {
        float pmax = 0.f;
        int flags = 0;
        for (int w = 0; w < V3_THREADS / 64; ++w) {
            pmax = fmaxf(pmax, reinterpret_cast<const float*>(carry_log)[2 * w]);
            flags |= (int)carry_log[2 * w + 1];
        }
        unit_probs = !(flags & 2);
        if (!(flags & 1)) {
            int e = 127;
            if (pmax > 0.f) {
                const unsigned bits = __float_as_uint(pmax);
                e = (int)(bits >> 23) + ((bits & 0x7fffffu) ? 1 : 0);
                if (e < 1) e = 1;
            }
            const int se = 127 + kk - (e - 127);
            S = (se >= 1 && se <= 254) ? __uint_as_float((unsigned)se << 23) : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();   // (the summaries live in the carry log: nobody logs a wrap before everybody has read them)
    }
    if (bid == 0 && tid == 0) {   // what the partial tiles hold, for the reduce kernel
        A.hdr->fmt = S > 0.f ? 0u : 1u;
        A.hdr->quantum = S > 0.f ? 1.0f / S : 0.f;
    }

#ifdef V3_TRACE
    if (tid == 0) trace[1] = wall_clock64();
#endif
    const f3 cr = {A.corner[0], A.corner[1], A.corner[2]};
    const float res = A.res, rinv = 1.0f / res;
    V3Tile VT;
    VT.tile = tile; VT.carry_log = carry_log; VT.carry_n = ctrl;
    VT.x0 = x0; VT.y0 = y0; VT.gz = gz; VT.ltyz = ltyz; VT.res = res; VT.rres = refined_rcp(res); VT.S = S;
    VT.rres64 = 1.0 / (double)res;
    VT.unit_probs = unit_probs;
    // (double)g < 0.01 <=> g < ceil_to_float(0.01); integers are exact in fp32: max / min with the owned range is exact
    VT.lox = fmaxf(ceil_to_float(0.01), (float)x0); VT.hix = fminf(ceil_to_float((double)gx - 1.01), (float)(x0 + tx));
    VT.loy = fmaxf(ceil_to_float(0.01), (float)y0); VT.hiy = fminf(ceil_to_float((double)gy - 1.01), (float)(y0 + ty));
    VT.loz = ceil_to_float(0.01); VT.hiz = ceil_to_float((double)gz - 1.01);
    const uint32_t* queue = A.pool + (int64_t)t * A.pool_cap * 3;
    const char* ltab_b = reinterpret_cast<const char*>(ltab);
    // acceptance box of the coordinates whose floor cell this tile owns (fused mode's cull and arc screen)
    const float blx = fmaxf(0.01f, (float)x0), bhx = fminf((float)gx - 1.01f, (float)(x0 + tx));
    const float bly = fmaxf(0.01f, (float)y0), bhy = fminf((float)gy - 1.01f, (float)(y0 + ty));
    const float blz = 0.01f, bhz = (float)gz - 1.01f;
    const float bcx = 0.5f * (blx + bhx), bcy = 0.5f * (bly + bhy), bcz = 0.5f * (blz + bhz);
    const float bhx_ = 0.5f * (bhx - blx), bhy_ = 0.5f * (bhy - bly), bhz_ = 0.5f * (bhz - blz);

    // one pair per lane (`valid` lanes): exact frame, then its candidate runs -- from the record (binned) or from the arc
    // screen against this tile (fused) -- and the run walk
    auto process = [&](const int64_t p, const bool valid, const uint32_t ra_in, const uint32_t rb_in) {
        f3 Fcc = {0.f, 0.f, 0.f}, Fx = Fcc, Fy = Fcc;
        float Fprob = 1.f;
        int n = 0;
        if (valid && !FUSED) {   // the frame the bin kernel left for this pair (the same arithmetic as below, once per pair, not per tile)
            const float4* fr = A.frames + 3 * p;
            const float4 f0 = fr[0], f1 = fr[1], f2 = fr[2];
            Fcc = {f0.x, f0.y, f0.z}; Fx = {f0.w, f1.x, f1.y}; Fy = {f1.z, f1.w, f2.x};
            n = __float_as_int(f2.y);
            Fprob = f2.z;
            if (WIDE && n <= wb) n = 0;
        }
        if (valid && FUSED) {
            const float2 o = reinterpret_cast<const float2*>(A.outputs)[p];
            const int2 ij = v3_pair_idx(A, p);
            f3 a, ab, xd;
            if (pair_frame(A.points, ij.x, ij.y, a, ab, xd)) {
                Fcc = sub3(a, scl3(ab, o.x));
                Fprob = A.probs ? fmaxf(A.probs[ij.x], A.probs[ij.y]) : 1.f;
                Fx = scl3(xd, o.y);
                Fy = cross3(Fx, ab);
                n = A.n_rots;
                if (A.adaptive) n = min((int)((double)(o.y / res) * (2 * CPPF_PI)), A.n_rots);
                n = max(n, 0);
                if (WIDE && n <= wb) n = 0;   // no rotation of this pair in the launch's window
            }
        }
        const int Lw = WIDE ? max(min(n - wb, 72), 0) : n;
        int s0, l0, s1, l1, s2, l2;
        if (FUSED) {
            const f3 cq = scl3(sub3(Fcc, cr), rinv), xq = scl3(Fx, rinv), yq = scl3(Fy, rinv);
            const float ex = fmaf((fabsf(Fcc.x) + fabsf(cr.x) + fabsf(Fx.x) + fabsf(Fy.x)) * rinv, 1e-6f, 1e-3f);
            const float ey = fmaf((fabsf(Fcc.y) + fabsf(cr.y) + fabsf(Fx.y) + fabsf(Fy.y)) * rinv, 1e-6f, 1e-3f);
            const float ez = fmaf((fabsf(Fcc.z) + fabsf(cr.z) + fabsf(Fx.z) + fabsf(Fy.z)) * rinv, 1e-6f, 1e-3f);
            const float nf = (float)n * 0.159154943f;
            const Mask96 mx = axis_arc_mask<WIDE>(below, cq.x, xq.x, yq.x, blx - ex, bhx + ex, nf, n, wb, Lw);
            const Mask96 my = axis_arc_mask<WIDE>(below, cq.y, xq.y, yq.y, bly - ey, bhy + ey, nf, n, wb, Lw);
            const Mask96 mz = axis_arc_mask<WIDE>(below, cq.z, xq.z, yq.z, blz - ez, bhz + ez, nf, n, wb, Lw);
            const uint32_t live = n > 0 ? 0xffffffffu : 0u;
            int e0, e1, e2;
            mask_runs(below, mx.a & my.a & mz.a & live, mx.b & my.b & mz.b & live, mx.c & my.c & mz.c & live, s0, e0, s1, e1, s2, e2);
            l0 = e0 - s0; l1 = e1 - s1; l2 = max(e2 - s2, 0);
        } else {
            s0 = (int)(ra_in & 0xffu); l0 = (int)((ra_in >> 8) & 0xffu); s1 = (int)((ra_in >> 16) & 0xffu); l1 = (int)(ra_in >> 24);
            s2 = (int)(rb_in & 0xffu); l2 = (int)((rb_in >> 8) & 0xffu);
        }
        // The walk's unit is a SLOT = two consecutive candidates of a lane's concatenated runs (round 5; one candidate before): a
        // step pulls the source lane's frame ONCE for both -- thirteen ds_bpermute and the source / position bookkeeping were a
        // third of a step's LDS time and a seventh of its VALU (profiles/r1_issue_microbench.txt: a ds_bpermute occupies the LDS
        // pipe like three ds_read_b32) -- and deposits twice.  Only a lane's LAST slot can be half empty (odd candidate count).
        const int cnt = n > 0 ? l0 + l1 + l2 : 0;          // candidates
        const int slots = (cnt + 1) >> 1;
        const int incl = wave_incl_scan(slots);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total > 0) {
            const unsigned long long nz = __ballot(cnt > 0);
            const unsigned long long above = nz & ~((2ull << lane) - 1ull);
            const int nxt = above ? __builtin_ctzll(above) : 64;
            const uint32_t wA = (uint32_t)nxt | ((uint32_t)l0 << 8) | ((uint32_t)(l0 + l1) << 16) | ((uint32_t)cnt << 24);
            const uint32_t wB = ((uint32_t)max(s1 - s0 - l0, 0) << 3) | ((uint32_t)max(s2 - s1 - l1, 0) << 19);
            // where the run walk finds (cos, sin) of the lane's first candidate: a byte offset into the LDS table, row n(n-1)/2;
            // WIDE: {n, rotation index} packed as n << 16 | index << 3 -- the walk's increments are multiples of 8 either way
            const int tabS = WIDE ? ((n << 16) | ((wb + s0) << 3)) : (((__mul24(n, n - 1) >> 1) + s0) << 3);
            const int Bn = (total + 63) >> 6;
#ifdef V3_TRACE
            if (lane == 0) { atomicAdd(&ctrl[8], Bn); atomicAdd(&ctrl[9], 1); }
#endif
            const int q0 = __mul24(lane, Bn);
            const int mine = min(max(total - q0, 0), Bn);
            int src = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) src += (__shfl(incl, src + step - 1, 64) <= q0) ? step : 0;
            src = min(src, 63);
            int k = q0 - (__shfl(incl, src, 64) - __shfl(slots, src, 64));      // slot within the source lane
            struct Pulled { f3 cc, x, y; float prob; uint32_t a, b; int tab; };
            auto pull = [&](const int from) {
                Pulled q;
                q.cc.x = __shfl(Fcc.x, from, 64); q.cc.y = __shfl(Fcc.y, from, 64); q.cc.z = __shfl(Fcc.z, from, 64);
                q.x.x = __shfl(Fx.x, from, 64); q.x.y = __shfl(Fx.y, from, 64); q.x.z = __shfl(Fx.z, from, 64);
                q.y.x = __shfl(Fy.x, from, 64); q.y.y = __shfl(Fy.y, from, 64); q.y.z = __shfl(Fy.z, from, 64);
                q.prob = unit_probs ? 1.0f : __shfl(Fprob, from, 64);
                q.a = (uint32_t)__shfl((int)wA, from, 64);
                q.b = (uint32_t)__shfl((int)wB, from, 64);
                q.tab = __shfl(tabS, from, 64);
                return q;
            };
            // two steps per trip, the frames ping-ponging between `cur` and `nx` (a single-step loop copies the 13 pulled registers
            // every step); each step requests the next step's frame (behind its own table reads) before it deposits
            auto step = [&](const int it, const Pulled& use, Pulled& next) {
                const uint32_t a = use.a, b = use.b;
                const int t1 = (int)((a >> 8) & 0xffu), t2 = (int)((a >> 16) & 0xffu), cn = (int)(a >> 24);
                const int g1 = (int)(b & 0xffffu), g2 = (int)(b >> 16);
                const int c0 = k << 1, c1 = c0 + 1;                     // the slot's candidates in the lane's concatenated runs
                const int toff0 = use.tab + (c0 << 3) + (g1 & -(int)(c0 >= t1)) + (g2 & -(int)(c0 >= t2));
                const int toff1 = use.tab + (c1 << 3) + (g1 & -(int)(c1 >= t1)) + (g2 & -(int)(c1 >= t2));
                const bool live = it < mine, two = live && c1 < cn;
                k += 1;
                const bool adv = (k << 1) >= cn;
                src = adv ? (int)(a & 0xffu) : src;
                k = adv ? 0 : k;
                float2 cs0, cs1;
                if (WIDE) {
                    cs0 = live ? rot_cs((toff0 & 0xffff) >> 3, toff0 >> 16) : make_float2(1.f, 0.f);
                    cs1 = two ? rot_cs((toff1 & 0xffff) >> 3, toff1 >> 16) : make_float2(1.f, 0.f);
                } else {
                    cs0 = *reinterpret_cast<const float2*>(ltab_b + (live ? toff0 : 0));
                    cs1 = *reinterpret_cast<const float2*>(ltab_b + (two ? toff1 : 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                next = pull(src);
                __builtin_amdgcn_sched_barrier(0);
                if (live) {
                    const f3 offset = add3(scl3(use.x, cs0.x), scl3(use.y, cs0.y));    // :34
                    const f3 v = sub3(add3(use.cc, offset), cr);                       // numerator of :35
                    v3_deposit(VT, v, use.prob);
                }
                if (two) {
                    const f3 offset = add3(scl3(use.x, cs1.x), scl3(use.y, cs1.y));
                    const f3 v = sub3(add3(use.cc, offset), cr);
                    v3_deposit(VT, v, use.prob);
                }
            };
            Pulled cur = pull(src), nx;
            for (int it = 0; it < Bn; it += 2) {
                step(it, cur, nx);
                if (it + 1 < Bn) step(it + 1, nx, cur);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    if (!FUSED) {
        for (;;) {
            int blk = 0;
            if (lane == 0) blk = atomicAdd(&ctrl[4], 1);
            const unsigned rb0 = r0 + 64u * (unsigned)__builtin_amdgcn_readfirstlane(blk);
            if (rb0 >= r1) break;
            const unsigned rq = rb0 + lane;
            uint32_t p = 0u, ra = 0u, rb = 0u;
            if (rq < r1) {
                const uint32_t* rec = queue + (int64_t)rq * 3;
                p = rec[0]; ra = rec[1]; rb = rec[2];
            }
            process((int64_t)p, rq < r1, ra, rb);
        }
    } else {
        // cull (vote_kernel's, against the owned box) + compaction through a per-wave queue, blocks of 64 pairs from an LDS counter
        // The cull pays when it removes pairs (a random-weight network: three quarters of them); on a trained network's inputs
        // nearly every circle passes and it is ~5 % of the kernel for nothing.  So a wave whose last culled block kept >= 48 of 64
        // pairs takes its next blocks as they are -- a pair the cull would have dropped only yields an empty mask -- and culls
        // every eighth block again to notice a change.
        int qn = 0, direct = 0, since = 0;
        for (;;) {
            int blk = 0;
            if (lane == 0) blk = atomicAdd(&ctrl[4], 1);
            const int64_t pb = 64 * ((int64_t)c + (int64_t)__builtin_amdgcn_readfirstlane(blk) * Ct);
            const bool more = pb < p1;
            if (more && direct && (++since & 7) != 0) {
                process(pb + lane, pb + lane < p1, 0u, 0u);
                continue;
            }
            if (more) {
                const int64_t p = pb + lane;
                bool pass = false;
                if (p < p1) {
                    const float2 o = reinterpret_cast<const float2*>(A.outputs)[p];
                    const int2 ij = v3_pair_idx(A, p);
                    const f3 a = ld3(A.points, ij.x), b = ld3(A.points, ij.y);
                    const f3 d = sub3(a, b);
                    const float L = sqrtf(dot3(d, d));
                    const float inv = __builtin_amdgcn_rcpf(L + 1e-7f);
                    const f3 u = scl3(d, inv);
                    const f3 cc = sub3(a, scl3(u, o.x));
                    const float R = fabsf(o.y) * rinv;
                    const float ex_ = R * __builtin_amdgcn_sqrtf(fmaxf(0.f, 1.f - u.x * u.x));
                    const float ey_ = R * __builtin_amdgcn_sqrtf(fmaxf(0.f, 1.f - u.y * u.y));
                    const float ez_ = R * __builtin_amdgcn_sqrtf(fmaxf(0.f, 1.f - u.z * u.z));
                    const float sl = fmaf(R, 1.1e-3f, 4e-3f);
                    const float sx = sl + 8e-6f * (fabsf(cc.x) + fabsf(cr.x)) * rinv, sy = sl + 8e-6f * (fabsf(cc.y) + fabsf(cr.y)) * rinv,
                                sz = sl + 8e-6f * (fabsf(cc.z) + fabsf(cr.z)) * rinv;
                    const float qx = (cc.x - cr.x) * rinv, qy = (cc.y - cr.y) * rinv, qz = (cc.z - cr.z) * rinv;
                    const float dx = bcx - qx, dy = bcy - qy, dz = bcz - qz, sall = sl + (sx - sl) + (sy - sl) + (sz - sl);
                    const float off_plane = fabsf((dx * u.x + dy * u.y) + dz * u.z);
                    const float reach = (fabsf(u.x) * bhx_ + fabsf(u.y) * bhy_) + fabsf(u.z) * bhz_;
                    const float nx = fmaxf(fabsf(dx) - bhx_, 0.f), ny = fmaxf(fabsf(dy) - bhy_, 0.f), nz = fmaxf(fabsf(dz) - bhz_, 0.f);
                    const float fx = fabsf(dx) + bhx_, fy = fabsf(dy) + bhy_, fz = fabsf(dz) + bhz_;
                    const float dmin2 = (nx * nx + ny * ny) + nz * nz, dmax2 = (fx * fx + fy * fy) + fz * fz;
                    const float r_hi = R + sall, r_lo = fmaxf(R - sall, 0.f);
                    pass = (L >= 9e-8f) & (!A.adaptive | (R >= 0.15f)) & (!WIDE | !A.adaptive | (fmaf(R, 6.2832f, 1.f) > (float)wb)) &
                           (qx + ex_ + sx >= blx) & (qx - ex_ - sx < bhx) & (qy + ey_ + sy >= bly) & (qy - ey_ - sy < bhy) &
                           (qz + ez_ + sz >= blz) & (qz - ez_ - sz < bhz) &
                           (off_plane <= reach + sall) & (dmin2 <= r_hi * r_hi * 1.0001f) & (dmax2 * 1.0001f >= r_lo * r_lo);
                }
                const unsigned long long m = __ballot(pass);
                if (pass)
                    pairq[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = (uint32_t)(p - p0);
                qn += __popcll(m);
                direct = __popcll(m) >= 48 ? 1 : 0;
            }
            while (qn >= 64 || (!more && qn > 0)) {
                const int take = qn < 64 ? qn : 64;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint32_t off = lane < take ? pairq[qn - take + lane] : 0u;
                qn -= take;
                process(p0 + off, lane < take, 0u, 0u);
            }
            if (!more) break;
        }
    }

    // ---- dump: raw fixed point (the reduce kernel adds the partial tiles as integers); the halo's non-zero words and the logged
    // wrap-arounds go to the extra plane instead (device-scope 64-bit atomics, a few hundred per workgroup)
#ifdef V3_TRACE
    if (lane == 0) { trace[8 + wave] = wall_clock64(); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); if (wave < 8) trace[24 + wave] = hw; }
#endif
    __syncthreads();
#ifdef V3_TRACE
    if (tid == 0) { trace[2] = wall_clock64(); trace[4] = (unsigned long long)t; trace[5] = (unsigned long long)c; trace[6] = (unsigned long long)ctrl[8]; trace[7] = (unsigned long long)ctrl[9]; }
#endif
    const int slot = v3_slot_words(pt, gz);
    uint4* part4 = reinterpret_cast<uint4*>(A.partials + (int64_t)bid * slot);
    const uint4* t4 = reinterpret_cast<const uint4*>(tile);
    for (int k = tid; k < (nwords + 3) >> 2; k += V3_THREADS) part4[k] = t4[k];
    bool wrote = false;
    auto to_plane = [&](const int w, const int lx, const int ly, const int z) {
        const uint32_t v = tile[w];
        if (v == 0u) return;
        unsigned long long* dst = &A.plane[((int64_t)(x0 + lx) * gy + (y0 + ly)) * gz + z];
        if (S > 0.f) atomicAdd(dst, (unsigned long long)v); else atomicAdd(reinterpret_cast<float*>(dst), __uint_as_float(v));
        wrote = true;
    };
    if (pt.hx && x0 + tx < gx)       // halo column lx = tx (rows 0..ty, the diagonal cell included)
        for (int k = tid; k < (ty + 1) * gz && k < ltyz; k += V3_THREADS) { const int ly = k / gz; to_plane(tx * ltyz + k, tx, ly, k - ly * gz); }
    if (pt.hy && y0 + ty < gy)       // halo row ly = ty of the owned columns
        for (int k = tid; k < tx * gz; k += V3_THREADS) { const int lx = k / gz, z = k - lx * gz; to_plane(lx * ltyz + ty * gz + z, lx, ty, z); }
    if (S > 0.f) {
        const int nc = min(ctrl[0], VOTE_CARRY_CAP);
        if (ctrl[0] > VOTE_CARRY_CAP && tid == 0) atomicOr(&A.hdr->flags, 1u);
        for (int k = tid; k < nc; k += V3_THREADS) {
            const int w = (int)carry_log[k];   // word of the LDS tile -> grid cell (a halo word: the neighbour's cell)
            const int lx = w / ltyz, rem = w - lx * ltyz, ly = rem / gz, z = rem - ly * gz;
            atomicAdd(&A.plane[((int64_t)(x0 + lx) * gy + (y0 + ly)) * gz + z], 1ull << 32);
            wrote = true;
        }
    }
    if (__any(wrote) && lane == 0) A.hdr->any_extra = 1u;
#ifdef V3_TRACE
    __syncthreads();
    if (tid == 0) trace[3] = wall_clock64();
#endif
}

template <bool FUSED, bool WIDE>
__global__ __launch_bounds__(V3_THREADS) void v3_vote_kernel(V3Args A)
{
    v3_vote_body<FUSED, WIDE>(A, (int)blockIdx.x);
}

// The votes of several objects in ONE launch (cppf_vote_argmax_batch): workgroups [wg_begin[i], wg_begin[i + 1]) are object i's vote
// launch -- its own pair list, grid, workspace (header, extra plane, partial tiles, cached rotation table) -- so every pointer stays
// workgroup-uniform and the body is the single-object one.  What it buys: one launch instead of n, each object on fewer, longer-lived
// workgroups (a workgroup zeroes, dumps and has read back its 113 KB tile whatever it deposits: a quarter of the partial-tile traffic
// per object with four objects on 64 workgroups each) without leaving three quarters of the chip idle.  Fused form only (< 4 tiles,
// n_rots <= 72: five of the six NOCS categories need one tile, the bottle two).
#define V3_BATCH_MAX 8
struct V3Batch {
    V3Args item[V3_BATCH_MAX];
    int wg_begin[V3_BATCH_MAX + 1];    // vote launch
    int red_begin[V3_BATCH_MAX + 1];   // reduce launch
    int bps[V3_BATCH_MAX];
    int n;
};
static_assert(sizeof(V3Batch) <= 4096, "V3Batch travels by value: kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(V3_THREADS) void v3_vote_batch_kernel(V3Batch B)
{
    int i = 0;
    while (i + 1 < B.n && (int)blockIdx.x >= B.wg_begin[i + 1]) ++i;
    // (workgroup-uniform: a member whose grid needs >= 4 tiles was binned by its own v3_bin_kernel launch and takes the consumer form)
    if (B.item[i].fused) v3_vote_body<true, false>(B.item[i], (int)blockIdx.x - B.wg_begin[i]);
    else v3_vote_body<false, false>(B.item[i], (int)blockIdx.x - B.wg_begin[i]);
}

// ---------------------------------------------------------------------------- v3_reduce_kernel
// grid[cell] (+)= the exact sum of every partial tile's quanta for that cell -- the owner tile's word and the halo words of
// its x / y / diagonal neighbours, over all chunks, plus 2^32 per logged wrap-around -- converted to fp32 ONCE; arg-max as in
// reduce_tiles_kernel.  A block = 16 waves = one run of 256 words of one tile's slot (LDS order); wave g adds chunks g, g + 16, ...
// The block that draws the last ticket re-zeroes the queue header for the next launch on this workspace.
__device__ __forceinline__ void v3_rezero(V3Hdr* h)
{
    for (int k = 0; k < VOTE_MAX_TILES; ++k) h->tile_count[k * V3_CNT_STRIDE] = 0u;
    for (int k = 0; k < V3_RED_FANIN; ++k)   // the arg-max groups of v3_reduce_kernel
        for (int w = 2; w < 6; ++w) h->tile_count[k * V3_CNT_STRIDE + w] = 0u;
    h->flags = 0u; h->any_extra = 0u; h->done = 0u; h->magic = V3_MAGIC;
    __threadfence();
}
// `bid` / `nblocks`: this block's index within, and the size of, ITS launch (see v3_vote_body)
__device__ __forceinline__ void v3_reduce_body(const V3Args& A, const int bps, const int bid, const int nblocks)
{
    __shared__ unsigned long long part[RED_GROUPS][RED_CELLS];
    __shared__ unsigned long long wkey[RED_GROUPS];
    __shared__ int sp[200];
    const int tid = threadIdx.x, lane = tid & 63, cg = tid >> 6;
    if (bid == 0 && tid == 0) {   // rotation table left by the vote kernel of this call: valid from the next launch on
        const unsigned long long st = A.packed[31];
        if ((st & ~0xfffull) == VOTE_TAB_PENDING) A.packed[31] = st ^ (VOTE_TAB_PENDING ^ VOTE_TAB_STAMP);
    }
    int gx, gy, gz;
    int64_t n_points;
    V3Tiling pt;
    const bool fits = v3_resolve(A, gx, gy, gz, n_points, pt);
    // everything this block needs from the header in ONE round trip (a block's life is a chain of dependent HBM round trips -- header,
    // queue counters, partial tiles, extra plane, the two reports -- of ~1.5 us each; the first two and the middle two now overlap)
    const unsigned hflags = A.hdr->flags, hextra = A.hdr->any_extra, hfmt = A.hdr->fmt, hmagic = A.hdr->magic;
    const float hquantum = A.hdr->quantum;
    if (fits) v3_split(A, pt.T, sp);
    if (!fits || (hflags & 1u)) {   // nothing valid was voted: say so (index -1, NaN)
        // Two ways here.  (a) The workspace was never initialised (garbage magic): its extra plane is garbage too, so the header
        // stays POISONED -- every call on it reports the error until the caller zeroes cppf_vote_workspace_init_bytes() bytes.
        // (b) A valid workspace whose launch gave up (carry log overflow, a shape record beyond the capacities): workgroups may have
        // added to the plane before the flag went up, so all of it is cleared here -- the "zero between launches" invariant the next
        // call relies on -- and only then the header is reset.
        const bool poisoned = hmagic != 0u && hmagic != V3_MAGIC;
        if (!poisoned) {
            uint4* p4 = reinterpret_cast<uint4*>(A.plane);
            const int64_t n16 = (int64_t)(V3_PLANE_BYTES / 16);
            for (int64_t i = (int64_t)bid * blockDim.x + tid; i < n16; i += (int64_t)nblocks * blockDim.x) p4[i] = make_uint4(0u, 0u, 0u, 0u);
            __threadfence();
            __syncthreads();
        }
        if (tid == 0) {   // (the ticket lives with the arg-max keys, which the vote / bin kernel zeroed before looking at the header)
            const unsigned tk = atomicAdd(reinterpret_cast<unsigned*>(A.packed + 18), 1u);
            if (tk == (unsigned)nblocks - 1u) {
                if (A.out_idx) *A.out_idx = -1;
                if (A.out_val) *A.out_val = __uint_as_float(0x7fc00000u);
                if (A.quantum_out) *A.quantum_out = 0.f;
                // n_rots > 72 runs one pass per window of 72 rotations: a pass that gave up leaves the grid without its window, so the
                // failure is STICKY for the rest of the call -- later passes vote on but report -1 / NaN / quantum 0 again
                A.packed[20] = 1ull;
                if (!poisoned) v3_rezero(A.hdr);
            }
        }
        return;
    }
    const bool earlier_pass_failed = A.win_base > 0 && A.packed[20] != 0ull;   // (written a kernel boundary ago, see the error path)
    if (A.quantum_out && bid == 0 && tid == 0) *A.quantum_out = (hfmt == 0u && !earlier_pass_failed) ? hquantum : 0.f;
    const int T = pt.T;
    const int slot = v3_slot_words(pt, gz);
    __syncthreads();   // (sp)
    const bool raw = hfmt == 0u;
    const bool any_extra = hextra != 0u;
    unsigned long long best = 0ull;   // this thread's arg-max key over the block's items
    // Narrow launches over many tiles (a posed object as one of a batch's members: 64-128 workgroups over ~10 tiles, i.e. 6-12 partial
    // tiles per tile where the block-wide form below is laid out for 128: ten of its sixteen waves had nothing to add, and a block's
    // ~5 items were five chains of dependent round trips one after the other): every WAVE takes an item of its own -- the few
    // chunks in flight together, then the lane's four cells finished in registers -- sixteen items per block and trip, no barrier.
    const bool wave_items = !A.fused && A.wgs <= 128;   // (workgroup-uniform: a block serves one object)
    if (wave_items) {
        for (int item = bid * RED_GROUPS + cg; item < T * bps; item += nblocks * RED_GROUPS) {
            const int t = item / bps, j = item - t * bps;
            const int tix = t / pt.nty, tiy = t - tix * pt.nty;
            const int x0 = tix * pt.tx, y0 = tiy * pt.ty;
            const int tx = min(pt.tx, gx - x0), ty = min(pt.ty, gy - y0);
            const int tyh = pt.ty + pt.hy, ltyz = tyh * gz;
            const int nwords = (tx + pt.hx) * ltyz;
            const int k0 = j * RED_CELLS + lane * 4;
            if (k0 >= nwords) continue;   // (k0 + 3 < slot: both are multiples of 4)
            const int C = sp[t], base_b = sp[64 + t];
            const unsigned n_own = (unsigned)sp[128 + t];
            unsigned long long acc[4] = {0ull, 0ull, 0ull, 0ull};
            float facc[4] = {0.f, 0.f, 0.f, 0.f};
            const uint32_t* base = A.partials + (int64_t)base_b * slot + k0;
            auto add = [&](const uint4 v) {
                if (raw) { acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
                else { facc[0] += __uint_as_float(v.x); facc[1] += __uint_as_float(v.y); facc[2] += __uint_as_float(v.z); facc[3] += __uint_as_float(v.w); }
            };
            const bool full = n_own >= 64u * (unsigned)C;
            int c = 0;
            if (full) {
                for (; c + 3 < C; c += 4) {
                    uint4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const uint4*>(base + (int64_t)(c + q) * slot);
#pragma unroll
                    for (int q = 0; q < 4; ++q) add(v[q]);
                }
            }
            for (; c < C; ++c)
                if (full || v3_chunk_live(n_own, c, C, base_b + c)) add(*reinterpret_cast<const uint4*>(base + (int64_t)c * slot));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                if (k >= nwords) continue;
                const int lx = k / ltyz, rem = k - lx * ltyz, ly = rem / gz, z = rem - ly * gz;
                if (!(lx < tx && ly < ty)) continue;   // a halo word: the owner's cell got it through the extra plane
                const int64_t cell = ((int64_t)(x0 + lx) * gy + (y0 + ly)) * gz + z;
                const unsigned long long ext = any_extra ? A.plane[cell] : 0ull;
                if (ext) A.plane[cell] = 0ull;
                float v;
                if (raw) {
                    const unsigned long long s_ = ext + acc[u];
                    if (A.grid_raw) A.grid_raw[cell] = (long long)s_ + (A.accumulate ? A.grid_raw[cell] : 0ll);
                    v = (float)((double)s_ * (double)hquantum);
                } else {
                    v = __uint_as_float((uint32_t)ext) + facc[u];
                }
                if (A.accumulate && A.grid) v = A.grid[cell] + v;
                if (A.grid) A.grid[cell] = v;
                const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)cell);
                best = key > best ? key : best;
            }
        }
    } else
    // a block takes the items (tile t, run j of RED_CELLS words) blockIdx, blockIdx + gridDim, ...: header, split and the two reports
    // once per block, not once per item (a grid of 16 tiles has 1 760 items)
    for (int item = bid; item < T * bps; item += nblocks) {
    const int t = item / bps, j = item - t * bps;
    const int tix = t / pt.nty, tiy = t - tix * pt.nty;
    const int x0 = tix * pt.tx, y0 = tiy * pt.ty;
    const int tx = min(pt.tx, gx - x0), ty = min(pt.ty, gy - y0);
    const int tyh = pt.ty + pt.hy, ltyz = tyh * gz;
    const int nwords = (tx + pt.hx) * ltyz;
    if (j * RED_CELLS >= nwords) continue;
    const int C = sp[t], base_b = sp[64 + t];
    const unsigned n_own = (unsigned)sp[128 + t];

    // ---- the tile's own words: wave g adds chunks g, g + 16, ... (16-byte loads, 8 in flight when no chunk can be empty)
    const int k0 = j * RED_CELLS + lane * 4;
    unsigned long long acc[4] = {0ull, 0ull, 0ull, 0ull};
    float facc[4] = {0.f, 0.f, 0.f, 0.f};
    if (k0 < nwords) {   // (k0 + 3 < slot: both are multiples of 4)
        const uint32_t* base = A.partials + (int64_t)base_b * slot + k0;
        auto add = [&](const uint4 v) {
            if (raw) { acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
            else { facc[0] += __uint_as_float(v.x); facc[1] += __uint_as_float(v.y); facc[2] += __uint_as_float(v.z); facc[3] += __uint_as_float(v.w); }
        };
        // binned: >= 64 records per chunk, none is empty; fused: workgroup c owns the blocks c, c + C, ...: empty from ceil(P / 64) on
        const int64_t cp = 64;
        const bool full = A.fused ? (int64_t)(C - 1) * cp < A.n_ppfs : n_own >= 64u * (unsigned)C;
        int c = cg;
        if (full) {
            for (; c + 7 * RED_GROUPS < C; c += 8 * RED_GROUPS) {
                uint4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const uint4*>(base + (int64_t)(c + q * RED_GROUPS) * slot);
#pragma unroll
                for (int q = 0; q < 8; ++q) add(v[q]);
            }
        }
        for (; c < C; c += RED_GROUPS)
            if (full || (A.fused ? (base_b + c == 0 || (int64_t)c * cp < A.n_ppfs) : v3_chunk_live(n_own, c, C, base_b + c)))
                add(*reinterpret_cast<const uint4*>(base + (int64_t)c * slot));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) part[cg][lane * 4 + u] = raw ? acc[u] : (unsigned long long)__float_as_uint(facc[u]);

    const int k = j * RED_CELLS + tid;
    int lx = 0, ly = 0, z = 0;
    bool owned = false;
    if (tid < RED_CELLS && k < nwords) {
        lx = k / ltyz;
        const int rem = k - lx * ltyz;
        ly = rem / gz; z = rem - ly * gz;
        owned = lx < tx && ly < ty;
    }
    // what the neighbours' halos and the wrap-arounds added for this cell (requested before the barrier: it travels with the partial
    // tiles); the plane is left clean for the next launch
    const int64_t cell = ((int64_t)(x0 + lx) * gy + (y0 + ly)) * gz + z;
    unsigned long long ext = 0ull;
    float prev = 0.f;
    if (owned && any_extra) ext = A.plane[cell];
    if (owned && A.accumulate && A.grid) prev = A.grid[cell];
    __syncthreads();
    if (owned) {
        float v;
        if (ext) A.plane[cell] = 0ull;
        if (raw) {
            unsigned long long s_ = ext;
#pragma unroll
            for (int g = 0; g < RED_GROUPS; ++g) s_ += part[g][tid];
            if (A.grid_raw) A.grid_raw[cell] = (long long)s_ + (A.accumulate ? A.grid_raw[cell] : 0ll);
            v = (float)((double)s_ * (double)hquantum);   // s < 2^53, the quantum a power of two: one rounding
        } else {
            v = __uint_as_float((uint32_t)ext);
#pragma unroll
            for (int g = 0; g < RED_GROUPS; ++g) v = v + __uint_as_float((uint32_t)part[g][tid]);
        }
        if (A.accumulate) v = prev + v;
        if (A.grid) A.grid[cell] = v;
        const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)cell);
        best = key > best ? key : best;
    }
    __syncthreads();   // (part: the next item's sums)
    }
    unsigned long long key = best;
    const int n_wkeys = wave_items ? RED_GROUPS : RED_CELLS / 64;   // (block-wide form: only the first RED_CELLS threads own cells)
    if (cg < n_wkeys) {
        key = wave_max_u64(key);
        if (lane == 0) wkey[cg] = key;
    }
    // two-level arg-max (see reduce_tiles_kernel): block b reports to group b mod V3_RED_FANIN, a group's last reporter reports for it.
    // A group's {key, count} sits in the spare words 2..5 of queue-counter line g of the header -- one 128-byte line, i.e. one L2 channel,
    // per group: in the caller's `packed` array the eight groups of round 2 shared one line and their atomics queued behind each other
    // (the reports were 4.7 of the kernel's 12.6 us at C2: profiles/r3_vote_phases.txt section 8)
    const unsigned gsel = (unsigned)bid & (V3_RED_FANIN - 1);
    const unsigned n_group = ((unsigned)nblocks - gsel + V3_RED_FANIN - 1) / V3_RED_FANIN;
    const unsigned n_groups_used = (unsigned)nblocks < V3_RED_FANIN ? (unsigned)nblocks : V3_RED_FANIN;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < n_wkeys; ++w) key = wkey[w] > key ? wkey[w] : key;
        auto report = [](unsigned long long* slot_, unsigned long long k_) -> unsigned {
            const unsigned long long old = atomicMax(slot_, k_);
            unsigned d1 = (unsigned)old, d2;
            asm volatile("v_mov_b32 %0, %1" : "=v"(d2) : "v"(d1));
            return atomicAdd(reinterpret_cast<unsigned*>(slot_ + 1), 1u + (d1 ^ d2));
        };
        unsigned long long* gslot = reinterpret_cast<unsigned long long*>(&A.hdr->tile_count[gsel * V3_CNT_STRIDE + 2]);
        if (report(gslot, key) == n_group - 1) {
            const unsigned long long gbest = atomicMax(gslot, 0ull);
            if (report(A.packed, gbest) == n_groups_used - 1) {
                const unsigned long long best_all = atomicMax(A.packed, 0ull);
                if (A.out_idx) *A.out_idx = earlier_pass_failed ? -1ll : (long long)(0xffffffffu - (uint32_t)(best_all & 0xffffffffull));
                if (A.out_val) *A.out_val = earlier_pass_failed ? __uint_as_float(0x7fc00000u) : ord2f((uint32_t)(best_all >> 32));
                v3_rezero(A.hdr);   // every block has finished reading the header: its report came after its last read
            }
        }
    }
}

__global__ __launch_bounds__(64 * RED_GROUPS) void v3_reduce_kernel(V3Args A, int bps)
{
    v3_reduce_body(A, bps, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(64 * RED_GROUPS) void v3_reduce_batch_kernel(V3Batch B)
{
    int i = 0;
    while (i + 1 < B.n && (int)blockIdx.x >= B.red_begin[i + 1]) ++i;
    v3_reduce_body(B.item[i], B.bps[i], (int)blockIdx.x - B.red_begin[i], B.red_begin[i + 1] - B.red_begin[i]);
}

// header + the largest carry plane a tiled grid can have (64 tiles): what must be zero before the first call on a fresh workspace
extern "C" size_t cppf_vote_workspace_init_bytes(void)
{
    return VOTE_WS_PART + V3_HDR_BYTES + V3_PLANE_BYTES;
}

// host side of the binned path; returns a negative CPPF_E* / positive hipError_t, or 0
static bool v3_eligible(int64_t n_ppfs, int n_rots, int gx, int gy, int gz)
{
    (void)n_rots;                                                           // (any 1..360: more than 72 takes several passes)
    if (n_ppfs < 1 || n_ppfs > 0xffffffffll) return false;                  // (pair numbers travel as u32)
    return v3_tiling(gx, gy, gz).T <= VOTE_MAX_TILES;
}
static int v3_fixed_bits_bound(int64_t n_ppfs, int n_rots, int gx, int gy, int gz)
{
    const V3Tiling t = v3_tiling(gx, gy, gz);
    const int wgs = v3_wgs(n_ppfs, t.T);
    if (t.T < v3_fused_tiles()) return v3_bits((unsigned)v3_fused_bits_pairs(n_ppfs, wgs / t.T, t.T), n_rots);   // (what v3_prepare passes)
    // binned: the scale follows the chunks of a launch of min(wgs, V3_BITS_WGS) workgroups (v3_split): chunk <= W / E <= P T / (that - T)
    // records (C_t = 1 + floor(n_t E / W) >= n_t E / W), and never more than a tile's queue (<= P: what E = 0 leaves, one chunk per tile)
    const int wref = wgs < V3_BITS_WGS ? wgs : V3_BITS_WGS;
    const int64_t E = wref - t.T;
    const int64_t worst = E > 0 ? min(n_ppfs, (n_ppfs * t.T + E - 1) / E + 1) : n_ppfs;
    return v3_bits((unsigned)(worst > 0x7fffffff ? 0x7fffffff : worst), n_rots);
}
static size_t v3_workspace_bytes(int64_t n_ppfs, int gx, int gy, int gz)
{
    const V3Tiling t = v3_tiling(gx, gy, gz);
    return v3_plan(n_ppfs, t, gz, v3_wgs(n_ppfs, t.T), t.T < 4 ? 0 : n_ppfs, (int64_t)gx * gy * gz).total;   // (< 4 tiles: no queues)
}
// tiles a *_dyn launch of class `many_tiles` serves: 0 -> 3 (the fused kernel), 1 -> 64 (any tiled grid), 4..64 -> that many (queues and
// the reduce launch sized for them: a posed NOCS object needs 9-12 tiles, the C5 grid 16 -- a quarter of the 64-tile footprint)
static inline int v3_dyn_tcap(int many_tiles)
{
    return many_tiles == 0 ? 3 : (many_tiles < 4 || many_tiles > VOTE_MAX_TILES ? VOTE_MAX_TILES : many_tiles);
}
static size_t v3_workspace_bytes_dyn(int many_tiles, int64_t n_ppfs)
{
    const int t_cap = v3_dyn_tcap(many_tiles), wgs = V3_WGS;
    return VOTE_WS_PART + V3_HDR_BYTES + V3_PLANE_BYTES +
           (many_tiles ? align_up((size_t)t_cap * (size_t)n_ppfs * 12, 256) + align_up((size_t)n_ppfs * V3_FRAME_BYTES, 256) : 0) +
           (size_t)wgs * V3_TILE_FLOATS * sizeof(uint32_t);
}
// What a by-value launch will do for this problem (tests, tools): out = {path, T, tx, ty, ntx, nty, hx, hy, workgroups, bits};
// path 0: global fp32 atomics (> 64 tiles), 2: fused kernel (< 4 tiles), 3: binned; (1 was round 2's kernels, gone); for n_rots > 72
// the tiled paths run ceil(n_rots / 72) passes of the same launches
extern "C" int cppf_vote_plan_query(int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int32_t* out)
{
    if (!out || n_rots < 1 || n_rots > CPPF_MAX_ROTS || gx < 1 || gy < 1 || gz < 1 || n_ppfs < 0) return CPPF_EINVAL;
    for (int k = 0; k < 10; ++k) out[k] = 0;
    if (v3_eligible(n_ppfs, n_rots, gx, gy, gz)) {
        const V3Tiling t = v3_tiling(gx, gy, gz);
        int wgs = v3_wgs(n_ppfs, t.T);
        if (t.T < v3_fused_tiles()) wgs = (wgs / t.T) * t.T;
        const int o[10] = {t.T < v3_fused_tiles() ? 2 : 3, t.T, t.tx, t.ty, t.ntx, t.nty, t.hx, t.hy, wgs, v3_fixed_bits_bound(n_ppfs, n_rots, gx, gy, gz)};
        for (int k = 0; k < 10; ++k) out[k] = o[k];
        return 0;
    }
    return 0;   // path 0: everything else stays 0
}

// what cppf_vote_grid_raw adds to a vote: the exact integer image of the grid, its quantum, and bits fixed by the caller
struct VoteExtras { long long* grid_raw; float* quantum_out; int fixed_bits; };

// Grids of fewer than this many tiles take the fused kernel (every tile's workgroups cull and screen the pair list themselves),
// the others bin first.  CPPF_FUSED_MAX_TILES (development knob, read once) moves the boundary for by-value launches; the queues'
// room in the workspace is always sized for the default boundary, so the knob can only be RAISED.
#define V3_FUSED_TILES 4
static int v3_fused_tiles()
{
    static int v = 0;
    if (v == 0) {
        const char* e = getenv("CPPF_FUSED_MAX_TILES");
        const int w = e ? atoi(e) : V3_FUSED_TILES;
        v = w < V3_FUSED_TILES ? V3_FUSED_TILES : (w > VOTE_MAX_TILES + 1 ? VOTE_MAX_TILES + 1 : w);
    }
    return v;
}

struct V3Launch { V3Args A; int red_blocks, bps; };

// the by-value arguments of the three kernels for one object: plan, workspace layout, fixed-point bits (see the comment on widths below)
static int v3_prepare(V3Launch& Lc, const float* points, const float* outputs, const float* probs, const void* point_idxs, int idx_is_i64,
                      float* grid_obj, const float* corner, float res, int64_t n_points, int64_t n_ppfs, int n_rots, int gx, int gy,
                      int gz, int adaptive, int accumulate, bool want_argmax, long long* out_idx, float* out_val, void* workspace,
                      const int32_t* shape_dev, int64_t grid_cap, int many_tiles, const VoteExtras* ex, int wg_cap, int wg_floor = V3_BITS_WGS)
{
    // Workgroups of the vote launch.  One per CU is the fastest launch on an idle chip, but every workgroup pays for its tile whatever
    // it deposits -- zeroed, dumped (113 KB) and read back by the reduce kernel: 27 of the 72 MB a C2 call moves -- so a caller that
    // keeps several instances in flight does better with fewer, longer-lived workgroups and the rest of the chip left to its other
    // streams (profiles/r4_vote_workgroups.txt: 128 instead of 256 at C2, three instances in flight: +5 % pairs/s, the instance alone
    // 7 % slower).  The hint never goes below 64 or the number of tiles; chunking follows it, the grid does NOT change with it: the
    // fixed-point scale of the fused vote is that of a 64-wide launch whatever the width (v3_fused_bits_pairs).
    const int wgs_max = wg_cap > 0 ? (wg_cap < wg_floor ? wg_floor : (wg_cap > V3_WGS ? V3_WGS : wg_cap)) : V3_WGS;
    char* ws = static_cast<char*>(workspace);
    V3Args& A = Lc.A;
    A = V3Args{};
    if (ex) { A.grid_raw = ex->grid_raw; A.quantum_out = ex->quantum_out; A.kk_force = ex->fixed_bits; }
    A.points = points; A.outputs = outputs; A.probs = probs; A.point_idxs = point_idxs; A.idx64 = idx_is_i64;
    A.corner = corner; A.res = res; A.n_ppfs = n_ppfs; A.n_points = n_points; A.n_rots = n_rots; A.adaptive = adaptive;
    A.gx = gx; A.gy = gy; A.gz = gz; A.shape = shape_dev; A.grid_cap = grid_cap;
    A.hdr = reinterpret_cast<V3Hdr*>(ws + VOTE_WS_PART);
    A.packed = reinterpret_cast<unsigned long long*>(ws);
    A.grid = grid_obj; A.accumulate = accumulate;
    A.out_idx = want_argmax ? out_idx : nullptr; A.out_val = want_argmax ? out_val : nullptr;
    const bool wide = n_rots > VOTE_WIN;
    A.tab_entries = wide ? 0 : tri(n_rots);
    A.pool_cap = n_ppfs;
    A.plane = reinterpret_cast<unsigned long long*>(ws + VOTE_WS_PART + V3_HDR_BYTES);
    if (shape_dev) {
        A.t_cap = v3_dyn_tcap(many_tiles);
        A.wgs = wgs_max;
        A.fused = many_tiles ? 0 : 1;
        if (ex && ex->grid_raw) return CPPF_EINVAL;
        if (grid_cap > (int64_t)A.t_cap * V3_TILE_FLOATS) return CPPF_EINVAL;   // (a grid of the class has at most that many cells)
        A.pool = reinterpret_cast<uint32_t*>(ws + VOTE_WS_PART + V3_HDR_BYTES + V3_PLANE_BYTES);
        A.frames = reinterpret_cast<float4*>(A.pool + (many_tiles ? align_up((size_t)A.t_cap * (size_t)n_ppfs * 12, 256) / 4 : 0));
        A.partials = reinterpret_cast<uint32_t*>(A.frames) + (many_tiles ? align_up((size_t)n_ppfs * V3_FRAME_BYTES, 256) / 4 : 0);
        const int bps = ((V3_TILE_FLOATS + RED_CELLS - 1) / RED_CELLS + RED_FANIN - 1) / RED_FANIN * RED_FANIN;
        Lc.red_blocks = A.t_cap * bps;
    } else {
        A.t = v3_tiling(gx, gy, gz);
        A.wgs = v3_wgs(n_ppfs, A.t.T);
        if (A.wgs > wgs_max) A.wgs = wgs_max > A.t.T ? wgs_max : A.t.T;
        A.fused = A.t.T < v3_fused_tiles() ? 1 : 0;
        if (A.fused) {   // static chunks: the same number for every tile
            A.wgs = (A.wgs / A.t.T) * A.t.T;
            A.kk = v3_bits((unsigned)v3_fused_bits_pairs(n_ppfs, A.wgs / A.t.T, A.t.T), n_rots);
        }
        A.t_cap = A.t.T;
        const V3Plan pl = v3_plan(n_ppfs, A.t, gz, v3_wgs(n_ppfs, A.t.T), A.fused ? 0 : n_ppfs, (int64_t)gx * gy * gz);
        A.pool = reinterpret_cast<uint32_t*>(ws + pl.pool_off);
        A.frames = reinterpret_cast<float4*>(ws + pl.frames_off);
        A.partials = reinterpret_cast<uint32_t*>(ws + pl.part_off);
        const int bps = ((pl.slot + RED_CELLS - 1) / RED_CELLS + RED_FANIN - 1) / RED_FANIN * RED_FANIN;
        Lc.red_blocks = A.t.T * bps;
    }
    Lc.bps = Lc.red_blocks / A.t_cap;
    return 0;
}

static size_t v3_vote_lds(const V3Args& A) { return V3_LDS_HEAD + (size_t)(A.tab_entries + 2) * sizeof(float2) + (size_t)V3_TILE_FLOATS * sizeof(float); }
static void v3_set_attrs()
{
    static bool attr_done = false;
    if (!attr_done) {
        const void* ks[] = {reinterpret_cast<const void*>(&v3_bin_kernel<false>), reinterpret_cast<const void*>(&v3_bin_kernel<true>),
                            reinterpret_cast<const void*>(&v3_vote_kernel<false, false>), reinterpret_cast<const void*>(&v3_vote_kernel<true, false>),
                            reinterpret_cast<const void*>(&v3_vote_kernel<false, true>), reinterpret_cast<const void*>(&v3_vote_kernel<true, true>),
                            reinterpret_cast<const void*>(&v3_vote_batch_kernel)};
        for (const void* k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
}

// the bin launch of a many-tile object: super-rounds of bin_sr x 512 pairs, as long as possible, but two workgroups for every CU first
static size_t v3_bin_geometry(V3Args& A, dim3& bin_grid)
{
    int64_t srb = A.n_ppfs / ((int64_t)V3_BIN_THREADS * 512);
    srb = srb < 1 ? 1 : (srb > V3_BIN_SR ? V3_BIN_SR : srb);
    A.bin_sr = (int)srb;
    const int64_t rounds = (A.n_ppfs + V3_BIN_THREADS * srb - 1) / (V3_BIN_THREADS * srb);
    bin_grid = dim3((unsigned)(rounds < 512 ? rounds : 512));
    return (size_t)V3_STAGE * 16 + VOTE_BELOW_N * 16 + 2 * VOTE_MAX_TILES * 4 + 64 + (V3_BIN_THREADS / 64) * (V3_IRING * 2 + (64 * V3_BIN_SR + 64) * 4);
}

static int v3_launch(const float* points, const float* outputs, const float* probs, const void* point_idxs, int idx_is_i64,
                     float* grid_obj, const float* corner, float res, int64_t n_points, int64_t n_ppfs, int n_rots, int gx, int gy,
                     int gz, int adaptive, int accumulate, bool want_argmax, long long* out_idx, float* out_val, void* workspace,
                     hipStream_t st, const int32_t* shape_dev, int64_t grid_cap, int many_tiles, const VoteExtras* ex = nullptr,
                     int wg_cap = 0)
{
    V3Launch Lc;
    const int rc = v3_prepare(Lc, points, outputs, probs, point_idxs, idx_is_i64, grid_obj, corner, res, n_points, n_ppfs, n_rots, gx, gy, gz,
                              adaptive, accumulate, want_argmax, out_idx, out_val, workspace, shape_dev, grid_cap, many_tiles, ex, wg_cap);
    if (rc != 0) return rc;
    V3Args& A = Lc.A;
    const int red_blocks = Lc.red_blocks;
    const bool wide = n_rots > VOTE_WIN;
    v3_set_attrs();
    const size_t lds_vote = v3_vote_lds(A);
    dim3 bin_grid;
    const size_t lds_bin = v3_bin_geometry(A, bin_grid);
    const int bps = Lc.bps;
    // (two workgroups of 16 waves fill a CU: at most one round of blocks, each looping over its items)
    const dim3 red_grid((unsigned)(red_blocks < 2 * V3_WGS ? red_blocks : 2 * V3_WGS));
    // n_rots <= 72: one pass.  More (a reference knob, nocs/inference.py:39 --num_rots): pass w votes rotations [72 w, 72 w + 72) of
    // every pair with the WIDE kernels and ADDS to the grid of the passes before it; the arg-max the last pass reports is the
    // arg-max of the whole vote (a pass that gives up makes the call's report -1 / NaN / quantum 0: sticky, see v3_reduce_body).
    // Integer images (grid_raw) need one scale for all passes: fixed here unless the caller fixed it.
    if (wide && A.grid_raw && !A.kk_force) A.kk_force = v3_fixed_bits_bound(n_ppfs, n_rots, gx, gy, gz);
    for (int wb = 0; wb < n_rots; wb += VOTE_WIN) {
        A.win_base = wb;
        if (wb > 0) A.accumulate = 1;
        if (A.fused) {
            if (wide) hipLaunchKernelGGL((v3_vote_kernel<true, true>), dim3((unsigned)A.wgs), dim3(V3_THREADS), lds_vote, st, A);
            else hipLaunchKernelGGL((v3_vote_kernel<true, false>), dim3((unsigned)A.wgs), dim3(V3_THREADS), lds_vote, st, A);
            CPPF_CHECK_LAUNCH();
        } else {
            if (wide) hipLaunchKernelGGL(v3_bin_kernel<true>, bin_grid, dim3(V3_BIN_THREADS), lds_bin, st, A);
            else hipLaunchKernelGGL(v3_bin_kernel<false>, bin_grid, dim3(V3_BIN_THREADS), lds_bin, st, A);
            CPPF_CHECK_LAUNCH();
            if (wide) hipLaunchKernelGGL((v3_vote_kernel<false, true>), dim3((unsigned)A.wgs), dim3(V3_THREADS), lds_vote, st, A);
            else hipLaunchKernelGGL((v3_vote_kernel<false, false>), dim3((unsigned)A.wgs), dim3(V3_THREADS), lds_vote, st, A);
            CPPF_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(v3_reduce_kernel, red_grid, dim3(64 * RED_GROUPS), 0, st, A, bps);
        CPPF_CHECK_LAUNCH();
    }
    return 0;
}

__global__ void zero_u64x2_kernel(unsigned long long* p) { p[0] = 0ull; p[1] = 0ull; }
__global__ void set_f32_kernel(float* p, float v) { *p = v; }

// Grids that would need more than 64 LDS tiles (> 1.9 M cells; none of the reference's categories comes close): the reference's own
// formulation, models/voting.py:8-66 line by line -- one thread per pair, a loop over its rotations, global fp32 atomicAdd.  Same
// arithmetic as the tiled path (exact frame, correctly rounded division, fp64 bound tests), so the same votes land in the same
// cells; the sums differ from the tiled path's by fp32 atomic order, like the reference's own.
__global__ __launch_bounds__(256) void vote_global_kernel(const float* __restrict__ points, const float* __restrict__ outputs,
                                                          const float* __restrict__ probs, const void* __restrict__ point_idxs, int idx64,
                                                          float* __restrict__ grid, const float* __restrict__ corner, float res,
                                                          int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive)
{
    const f3 cr = {corner[0], corner[1], corner[2]};
    const int64_t syz = (int64_t)gy * gz;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_ppfs; p += (int64_t)gridDim.x * blockDim.x) {
        const float2 o = reinterpret_cast<const float2*>(outputs)[p];
        int2 ij;
        if (idx64) { const longlong2 v = reinterpret_cast<const longlong2*>(point_idxs)[p]; ij = make_int2((int)v.x, (int)v.y); }
        else ij = reinterpret_cast<const int2*>(point_idxs)[p];
        f3 a, ab, xd;
        if (!pair_frame(points, ij.x, ij.y, a, ab, xd)) continue;                                  // :21
        const f3 cc = sub3(a, scl3(ab, o.x));                                                      // :23
        const float prob = probs ? fmaxf(probs[ij.x], probs[ij.y]) : 1.f;                          // :25
        const f3 x = scl3(xd, o.y), y = cross3(x, ab);                                             // :28-29
        int n = n_rots;
        if (adaptive) n = min((int)((double)(o.y / res) * (2 * CPPF_PI)), n_rots);                 // :31
        for (int i = 0; i < n; ++i) {
            const float2 cs = rot_cs(i, n);                                                        // :33
            const f3 v = sub3(add3(cc, add3(scl3(x, cs.x), scl3(y, cs.y))), cr);                   // :34, numerator of :35
            const f3 g = {v.x / res, v.y / res, v.z / res};                                        // :35 (IEEE division)
            if ((double)g.x < 0.01 || (double)g.y < 0.01 || (double)g.z < 0.01 || (double)g.x >= (double)gx - 1.01 ||
                (double)g.y >= (double)gy - 1.01 || (double)g.z >= (double)gz - 1.01) continue;    // :36-39
            const int fx = (int)g.x, fy = (int)g.y, fz = (int)g.z;                                 // :40
            const float rx = g.x - floorf(g.x), ry = g.y - floorf(g.y), rz = g.z - floorf(g.z);    // :42
            const float w0x = 1.f - rx, w0y = 1.f - ry, w0z = 1.f - rz;
            float* b = grid + ((int64_t)fx * syz + (int64_t)fy * gz + fz);
            atomicAdd(b, w0x * w0y * w0z * prob);                 atomicAdd(b + 1, w0x * w0y * rz * prob);          // :47-63
            atomicAdd(b + gz, w0x * ry * w0z * prob);             atomicAdd(b + gz + 1, w0x * ry * rz * prob);
            atomicAdd(b + syz, rx * w0y * w0z * prob);            atomicAdd(b + syz + 1, rx * w0y * rz * prob);
            atomicAdd(b + syz + gz, rx * ry * w0z * prob);        atomicAdd(b + syz + gz + 1, rx * ry * rz * prob);
        }
    }
}

// shape_dev != null: gx, gy, gz, n_points are CAPACITIES (gx*gy*gz = cells of grid_obj) and the real values come from the
// device record; only the tiled path exists in that mode.
static int vote_impl(const float* points, const float* outputs, const float* probs, const void* point_idxs, int idx_is_i64,
                     float* grid_obj, const float* corner, float res, int64_t n_points, int64_t n_ppfs, int n_rots,
                     int gx, int gy, int gz, int adaptive, int accumulate, bool want_argmax, long long* out_idx, float* out_val,
                     void* workspace, size_t workspace_bytes, hipStream_t st, const int32_t* shape_dev = nullptr,
                     int64_t grid_cap = 0, int many_tiles = 0, const VoteExtras* ex = nullptr)
{
    // `accumulate` is a flags word (include/cppf.h): bit 0 = add to the grid, CPPF_VOTE_WORKGROUPS(n) in bits 8..16 = launch at most n
    // vote workgroups (a scheduling hint for callers that keep several instances in flight; 0 = one per CU)
    const int wg_cap = (accumulate >> 8) & 0x1ff;
    if ((accumulate & ~(1 | (0x1ff << 8))) != 0) return CPPF_EINVAL;
    accumulate &= 1;
    if (wg_cap != 0 && ex) return CPPF_EINVAL;   // (integer images fix their bits from the default plan: no hint there)
    if (!points || (!grid_obj && !(ex && ex->grid_raw)) || !corner) return CPPF_EINVAL;   // (probs may be null: all ones)
    if (n_ppfs > 0 && (!outputs || !point_idxs)) return CPPF_EINVAL;
    if (n_rots < 1 || n_rots > CPPF_MAX_ROTS || gx < 1 || gy < 1 || gz < 1 || n_ppfs < 0 || n_points < 1) return CPPF_EINVAL;
    if ((int64_t)gx * gy * gz > 0x7fffffffll) return CPPF_EINVAL;
    if (shape_dev) {
        if (n_ppfs < 1 || n_ppfs > 0xffffffffll || grid_cap < 1 || grid_cap > 0x7fffffffll) return CPPF_EINVAL;
        if (!workspace || workspace_bytes < v3_workspace_bytes_dyn(many_tiles, n_ppfs)) return CPPF_EWORKSPACE;
        return v3_launch(points, outputs, probs, point_idxs, idx_is_i64, grid_obj, corner, res, n_points, n_ppfs, n_rots, gx, gy, gz,
                         adaptive, accumulate, want_argmax, out_idx, out_val, workspace, st, shape_dev, grid_cap, many_tiles, ex, wg_cap);
    }
    if (v3_eligible(n_ppfs, n_rots, gx, gy, gz)) {
        if (!workspace || workspace_bytes < v3_workspace_bytes(n_ppfs, gx, gy, gz)) return CPPF_EWORKSPACE;
        return v3_launch(points, outputs, probs, point_idxs, idx_is_i64, grid_obj, corner, res, n_points, n_ppfs, n_rots, gx, gy, gz,
                         adaptive, accumulate, want_argmax, out_idx, out_val, workspace, st, nullptr, 0, 0, ex, wg_cap);
    }
    if (ex) {
        if (n_ppfs > 0) return CPPF_EUNSUPPORTED;   // the integer image exists on the tiled integer path only (<= 64 tiles)
        // an empty pair list (a rank's slice of a short list): nothing was voted -- an all-zero image whose quantum is +inf, which
        // the MIN over the ranks' quanta ignores and cppf_grid_from_raw converts to an all-zero grid
        if (!accumulate) {
            hipError_t e = hipMemsetAsync(ex->grid_raw, 0, (size_t)gx * gy * gz * sizeof(long long), st);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(1), 0, st, ex->quantum_out, HUGE_VALF);
        CPPF_CHECK_LAUNCH();
        return 0;
    }
    // ---- no pairs, or a grid beyond 64 tiles: global atomics (+ the plain arg-max kernel)
    if (!workspace || workspace_bytes < 256) return CPPF_EWORKSPACE;
    const int64_t G = (int64_t)gx * gy * gz;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(grid_obj, 0, (size_t)G * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if (n_ppfs > 0) {
        int64_t nb = (n_ppfs + 255) / 256;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(vote_global_kernel, dim3((unsigned)nb), dim3(256), 0, st, points, outputs, probs, point_idxs, idx_is_i64, grid_obj,
                           corner, res, n_ppfs, n_rots, gx, gy, gz, adaptive);
        CPPF_CHECK_LAUNCH();
    }
    if (want_argmax) {
        unsigned long long* packed = static_cast<unsigned long long*>(workspace);
        hipLaunchKernelGGL(zero_u64x2_kernel, dim3(1), dim3(1), 0, st, packed);
        int64_t nb = (G + 63) / 64;
        if (nb > RED_MAX_BLOCKS) nb = RED_MAX_BLOCKS;
        hipLaunchKernelGGL(reduce_argmax_kernel, dim3((unsigned)nb), dim3(64 * RED_GROUPS), 0, st, grid_obj, (const float*)nullptr, 0, G,
                           packed, 1, 0, out_idx, out_val);
        CPPF_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int cppf_ppf_voting(const float* points, const float* outputs, const float* probs,
                               const int32_t* point_idxs, float* grid_obj, const float* corner, float res,
                               int64_t n_points, int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive,
                               void* workspace, size_t workspace_bytes, void* stream)
{
    return vote_impl(points, outputs, probs, point_idxs, 0, grid_obj, corner, res, n_points, n_ppfs, n_rots, gx, gy, gz,
                     adaptive, 1, false, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int cppf_vote_argmax(const float* points, const float* outputs, const float* probs,
                                const void* point_idxs, int idx_is_i64, float* grid_obj, const float* corner, float res,
                                int64_t n_points, int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive,
                                int accumulate, long long* out_idx, float* out_val, void* workspace,
                                size_t workspace_bytes, void* stream)
{
    return vote_impl(points, outputs, probs, point_idxs, idx_is_i64, grid_obj, corner, res, n_points, n_ppfs, n_rots, gx, gy, gz,
                     adaptive, accumulate, true, out_idx, out_val, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int cppf_vote_argmax_dyn(const float* points, const float* outputs, const float* probs, const void* point_idxs,
                                    int idx_is_i64, float* grid_obj, int64_t grid_capacity, const float* corner, float res,
                                    int64_t n_points_cap, int64_t n_ppfs, int n_rots, const int32_t* shape_dev, int many_tiles,
                                    int adaptive, int accumulate, long long* out_idx, float* out_val, void* workspace,
                                    size_t workspace_bytes, void* stream)
{
    if (!shape_dev) return CPPF_EINVAL;
    return vote_impl(points, outputs, probs, point_idxs, idx_is_i64, grid_obj, corner, res, n_points_cap, n_ppfs, n_rots, 1, 1, 1,
                     adaptive, accumulate, true, out_idx, out_val, workspace, workspace_bytes, (hipStream_t)stream, shape_dev,
                     grid_capacity, many_tiles);
}

// ---- the votes of several objects in one launch ----
// Items whose vote takes the fused kernel (< 4 tiles; dims by value or, shape_dev != NULL, the few-tiles class from a device
// record) and n_rots <= 72 share ONE vote launch and ONE reduce launch; every other item (>= 4 tiles, a grid beyond the tiled path,
// an empty pair list) gets its own launches exactly as cppf_vote_argmax* would issue them.  Results per item are those of its own
// cppf_vote_argmax call at the same width, bit for bit.
extern "C" int cppf_vote_batch_workgroups(int n_items, int flags)
{
    if (n_items < 1 || n_items > V3_BATCH_MAX) return CPPF_EINVAL;
    const int wg_cap = (flags >> 8) & 0x1ff;
    int w = wg_cap > 0 ? wg_cap : V3_WGS / n_items;     // default: the batch as ONE round of workgroups, one per CU (up to four objects)
    if (w < V3_BITS_WGS) w = V3_BITS_WGS;               // (never narrower than the width the fixed-point scale is chosen for)
    if (w > V3_WGS) w = V3_WGS;
    return w;
}

extern "C" int cppf_vote_argmax_batch(int n_items, const CppfVoteItem* items, int n_rots, int adaptive, int flags, void* stream)
{
    if (n_items < 1 || n_items > V3_BATCH_MAX || !items) return CPPF_EINVAL;
    if ((flags & ~(1 | (0x1ff << 8))) != 0) return CPPF_EINVAL;
    if (n_rots < 1 || n_rots > CPPF_MAX_ROTS) return CPPF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int accumulate = flags & 1;
    auto shares_the_launch = [&](const CppfVoteItem& it) -> bool {
        const bool dyn = it.shape_dev != nullptr;
        bool batched = n_rots <= VOTE_WIN && it.n_ppfs >= 1 && it.points && it.grid && it.corner && it.outputs && it.point_idxs &&
                       it.out_idx && it.out_val && it.workspace && it.n_points >= 1;
        if (batched && dyn) batched = it.n_ppfs <= 0xffffffffll && it.grid_capacity >= 1 &&
                                      it.grid_capacity <= (int64_t)v3_dyn_tcap(it.many_tiles) * V3_TILE_FLOATS &&
                                      it.workspace_bytes >= v3_workspace_bytes_dyn(it.many_tiles, it.n_ppfs);
        if (batched && !dyn) batched = it.gx >= 1 && it.gy >= 1 && it.gz >= 1 && v3_eligible(it.n_ppfs, n_rots, it.gx, it.gy, it.gz) &&
                                       it.workspace_bytes >= v3_workspace_bytes(it.n_ppfs, it.gx, it.gy, it.gz);
        return batched;
    };
    // the default width divides the chip between the objects that SHARE the launch (a chain of four with one many-tile member, which
    // takes its own launches, used to leave a quarter of the CUs idle during the other three's vote)
    int n_shared = 0;
    for (int i = 0; i < n_items; ++i) n_shared += shares_the_launch(items[i]) ? 1 : 0;
    const int width = cppf_vote_batch_workgroups(n_shared > 0 ? n_shared : n_items, flags);
    V3Batch B = {};
    int vote_wgs = 0, red_blocks = 0;
    size_t lds_vote = 0;
    for (int i = 0; i < n_items; ++i) {
        const CppfVoteItem& it = items[i];
        const bool dyn = it.shape_dev != nullptr;
        const bool batched = shares_the_launch(it);
        if (!batched) {   // its own launches (and its own argument checks)
            const int rc = dyn ? cppf_vote_argmax_dyn(it.points, it.outputs, it.probs, it.point_idxs, it.idx_is_i64, it.grid, it.grid_capacity,
                                                      it.corner, it.res, it.n_points, it.n_ppfs, n_rots, it.shape_dev, it.many_tiles, adaptive,
                                                      flags, it.out_idx, it.out_val, it.workspace, it.workspace_bytes, stream)
                               : cppf_vote_argmax(it.points, it.outputs, it.probs, it.point_idxs, it.idx_is_i64, it.grid, it.corner, it.res,
                                                  it.n_points, it.n_ppfs, n_rots, it.gx, it.gy, it.gz, adaptive, flags, it.out_idx, it.out_val,
                                                  it.workspace, it.workspace_bytes, stream);
            if (rc != 0) return rc;
            continue;
        }
        V3Launch Lc;
        const int rc = v3_prepare(Lc, it.points, it.outputs, it.probs, it.point_idxs, it.idx_is_i64, it.grid, it.corner, it.res, it.n_points,
                                  it.n_ppfs, n_rots, dyn ? 1 : it.gx, dyn ? 1 : it.gy, dyn ? 1 : it.gz, adaptive, accumulate, true, it.out_idx,
                                  it.out_val, it.workspace, it.shape_dev, dyn ? it.grid_capacity : 0, dyn ? it.many_tiles : 0, nullptr, width);
        if (rc != 0) return rc;
        if (!Lc.A.fused) {   // a grid of >= 4 tiles: its own bin launch first; its consumer and its reduce share the launches below
            v3_set_attrs();
            dim3 bin_grid;
            const size_t lds_bin = v3_bin_geometry(Lc.A, bin_grid);
            hipLaunchKernelGGL(v3_bin_kernel<false>, bin_grid, dim3(V3_BIN_THREADS), lds_bin, st, Lc.A);
            CPPF_CHECK_LAUNCH();
        }
        const int k = B.n++;
        B.item[k] = Lc.A;
        B.bps[k] = Lc.bps;
        B.wg_begin[k] = vote_wgs;
        B.red_begin[k] = red_blocks;
        vote_wgs += Lc.A.wgs;
        // reduce blocks: the batch as about one round of the chip (two 16-wave blocks per CU), never fewer than 64 per object
        int rb = 2 * V3_WGS / (n_shared > 0 ? n_shared : n_items);
        rb = rb < 64 ? 64 : rb;
        if (!Lc.A.fused && rb < V3_WGS) rb = V3_WGS;   // (many tiles: ~120 runs of cells per tile, each a chain of dependent round trips)
        red_blocks += Lc.red_blocks < rb ? Lc.red_blocks : rb;
        lds_vote = v3_vote_lds(Lc.A);
    }
    if (B.n == 0) return 0;
    B.wg_begin[B.n] = vote_wgs;
    B.red_begin[B.n] = red_blocks;
    v3_set_attrs();
    hipLaunchKernelGGL(v3_vote_batch_kernel, dim3((unsigned)vote_wgs), dim3(V3_THREADS), lds_vote, st, B);
    CPPF_CHECK_LAUNCH();
    hipLaunchKernelGGL(v3_reduce_batch_kernel, dim3((unsigned)red_blocks), dim3(64 * RED_GROUPS), 0, st, B);
    CPPF_CHECK_LAUNCH();
    return 0;
}

// ---- the vote as exact integers (pair-sharded votes: cppf_amd/sharding.py) ----
extern "C" int cppf_vote_grid_raw(const float* points, const float* outputs, const float* probs, const void* point_idxs,
                                  int idx_is_i64, long long* grid_raw, float* quantum_out, const float* corner, float res,
                                  int64_t n_points, int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive, int accumulate,
                                  int fixed_bits, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!grid_raw || !quantum_out || fixed_bits < 0 || (fixed_bits != 0 && (fixed_bits < 8 || fixed_bits > 24))) return CPPF_EINVAL;
    const VoteExtras ex = {grid_raw, quantum_out, fixed_bits};
    return vote_impl(points, outputs, probs, point_idxs, idx_is_i64, nullptr, corner, res, n_points, n_ppfs, n_rots, gx, gy, gz,
                     adaptive, accumulate, false, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, nullptr, 0, 0, &ex);
}

// grid = raw * quantum (one rounding per cell: the conversion v3_reduce_kernel applies), then the arg-max
__global__ __launch_bounds__(256) void grid_from_raw_kernel(const long long* __restrict__ raw, int64_t n, const float* __restrict__ quantum,
                                                            float* __restrict__ grid)
{
    const float q = *quantum;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        grid[i] = (q > 0.f && q < INFINITY) ? (float)((double)raw[i] * (double)q)      // (q = +inf: "nothing was voted", an all-zero image)
                                            : (q > 0.f && raw[i] == 0ll ? 0.f : __uint_as_float(0x7fc00000u));
}
extern "C" int cppf_grid_from_raw(const long long* grid_raw, int64_t n, const float* quantum, float* grid, long long* out_idx,
                                  float* out_val, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!grid_raw || !quantum || !grid || n < 1 || n > 0x7fffffffll) return CPPF_EINVAL;
    int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(grid_from_raw_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, grid_raw, n, quantum, grid);
    CPPF_CHECK_LAUNCH();
    return out_idx ? cppf_grid_argmax(grid, n, out_idx, out_val, workspace, workspace_bytes, stream) : 0;
}

extern "C" int cppf_vote_tile_cells(void) { return V3_TILE_FLOATS; }

// tiles of the grid in the tiled vote (120 KiB LDS tiles with a one-cell halo on their cut sides); 0: global-atomics path
extern "C" int cppf_vote_tiles(int gx, int gy, int gz)
{
    if (gx < 1 || gy < 1 || gz < 1) return CPPF_EINVAL;
    const int T = v3_tiling(gx, gy, gz).T;
    return T <= VOTE_MAX_TILES ? T : 0;
}

extern "C" int cppf_grid_argmax(const float* grid, int64_t n, long long* out_idx, float* out_val, void* workspace,
                                size_t workspace_bytes, void* stream)
{
    if (!grid || n < 1 || n > 0x7fffffffll || !out_idx) return CPPF_EINVAL;
    if (!workspace || workspace_bytes < 16) return CPPF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* packed = static_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(zero_u64x2_kernel, dim3(1), dim3(1), 0, st, packed);
    int64_t nb = (n + 63) / 64;
    if (nb > RED_MAX_BLOCKS) nb = RED_MAX_BLOCKS;
    hipLaunchKernelGGL(reduce_argmax_kernel, dim3((unsigned)nb), dim3(64 * RED_GROUPS), 0, st, const_cast<float*>(grid),
                       (const float*)nullptr, 0, n, packed, 1, 0, out_idx, out_val);
    CPPF_CHECK_LAUNCH();
    return 0;
}

extern "C" int cppf_abi_version(void) { return CPPF_ABI_VERSION; }
extern "C" const char* cppf_error_string(int code)
{
    switch (code) {
    case 0: return "success";
    case CPPF_EINVAL: return "cppf: invalid argument";
    case CPPF_EWORKSPACE: return "cppf: workspace missing or too small";
    case CPPF_EUNSUPPORTED: return "cppf: unsupported layer shape";
    case CPPF_ENONFINITE: return "cppf: the cloud holds non-finite coordinates";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "cppf: unknown error";
    }
}
