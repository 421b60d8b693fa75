/*
 * cppf.h -- C ABI of libcppf_hip.so: the MI355X (gfx950) implementation of CPPF's
 * point-pair-feature -> pair-MLP -> vote -> argmax hot path.
 *
 * This is the drop-in boundary.  The reference (qq456cvb/CPPF) has no FFI of its own: its
 * scripts call three CuPy RawKernels and one torch module directly.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference root); the Python
 * binding that mirrors the reference's call convention lives in cppf_amd/models/{voting,model}.py
 * and the stub a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a BORROWED pointer; `const float* / int* ...` arguments named below as
 *     "device" are HIP device pointers valid on the current device, nothing is copied or kept.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises.
 *   - return value: 0 on success, a hipError_t value (>0) when a HIP call failed, or a negative
 *     CPPF_E* code for argument errors.  No exceptions cross the ABI.
 *   - outputs follow the reference: caller-allocated, (zero-)initialised by the caller, updated
 *     in place.
 *   - workspaces are caller-provided device scratch; size them with the *_workspace_bytes()
 *     queries.  The library allocates nothing and keeps no global state.
 *   - ONE exception to "plain scratch": the workspace of the centre vote (cppf_vote_argmax*, cppf_ppf_voting, cppf_vote_grid_raw,
 *     and the back-vote entry points that take `vote_workspace`) carries STATE between calls in its first
 *     cppf_vote_workspace_init_bytes() bytes (~15.7 MB): the (cos, sin) rotation table of its last launch, the queue counters
 *     of the binned path, and the "extra plane" (one u64 per grid cell for halo words and fixed-point wrap-arounds) that every
 *     call leaves zero for the next one.  The contract (ABI version 2):
 *       * give the vote a DEDICATED allocation: never hand the same bytes to another entry point between two votes;
 *       * ZERO its first min(cppf_vote_workspace_init_bytes(), size) bytes once, before the first call (one hipMemsetAsync);
 *       * to recycle a block of a shared arena, zero that many bytes again -- zeroing less (the header alone) makes the header
 *         look fresh while foreign bytes sit in the plane, which the next vote would add to the grid.
 *     What the kernels check for themselves: a header that was never initialised (neither zero nor left by a previous call)
 *     makes EVERY call report arg-max -1 / peak NaN until the caller zeroes the block; the rotation-table cache is re-validated
 *     on every launch (stamp + the 72 entries whose value is known) and rebuilt when the check fails.  A launch that gives up
 *     on a valid workspace (a wrap-around log that overflowed under cppf_vote_grid_raw's fixed_bits, or a *_dyn shape record beyond the capacities) reports
 *     -1 / NaN once, clears the plane itself and leaves the workspace ready for the next call.
 */
#ifndef CPPF_H
#define CPPF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPPF_ABI_VERSION 4   /* 2: vote workspace contract (state in its first cppf_vote_workspace_init_bytes() bytes), cppf_vote_grid_raw; 3: batched votes (CppfVoteItem), cppf_pair_mlp_batch_plan;
                              * 4: CppfPoseTailItem assembles the finished pose record on the device (record_out ...), cppf_stage_batch */

#define CPPF_EINVAL (-1)     /* bad argument (null pointer, negative size, n_rots out of range) */
#define CPPF_EWORKSPACE (-2) /* workspace too small / missing */
#define CPPF_EUNSUPPORTED (-3) /* layer shape not supported by any device kernel */
#define CPPF_ENONFINITE (-4) /* a host cloud holds NaN / inf coordinates (cppf_host_grid_shape) */

int cppf_abi_version(void);
const char* cppf_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Centre vote.  Replaces the CuPy RawKernel `ppf_kernel` = CUDA `ppf_voting`
 * (models/voting.py:4-67), launched at nocs/inference.py:192-205.
 *   points      device f32[N,3]        outputs  device f32[n_ppfs,2] = (mu, nu)
 *   probs       device f32[N], or NULL for all ones (every caller of the reference passes ones, nocs/inference.py:201:
 *               then nothing is read)              point_idxs device i32[n_ppfs,2]
 *   grid_obj    device f32[gx,gy,gz]   in/out, accumulated (+=) like the reference's atomicAdd
 *   corner      device f32[3]          res, n_rots (1..CPPF_MAX_ROTS), adaptive: as the reference
 *   n_points    N (the reference kernel never needs it; here it bounds the scan for max(probs))
 * The reference's launch geometry (grid, block) is not part of the ABI.
 * Strategy: the grid is cut into tiles that fit LDS (each owning the samples whose floor cell it holds, plus a one-cell halo);
 * every workgroup accumulates one tile for one chunk of pairs -- or, for grids of 4 tiles and more, for one chunk of the
 * tile's queue of (pair, candidate runs) records that a binning kernel filled -- with 32-bit fixed-point LDS atomics
 * (fp32-exact quantum, see csrc/vote.hip); the raw tiles are written to `workspace` and summed AS INTEGERS into grid_obj by a
 * last kernel that also yields the arg-max (cppf_vote_argmax): the grid is the exact sum of the quantised deposits, the same
 * bits on every run.  n_rots > 72 (nocs/inference.py:39 --num_rots) runs ceil(n_rots / 72) passes of the same kernels, pass w voting
 * rotations [72 w, 72 w + 72) of every pair and adding to the grid.  Grids that would need more than 64 tiles fall back to the
 * reference's own formulation with global fp32 atomics.
 * `workspace`: cppf_vote_workspace_bytes() bytes, DEDICATED to the vote and zero on first use (see the note on workspaces at
 * the top and cppf_vote_workspace_init_bytes below).
 * ------------------------------------------------------------------------------------------- */
#define CPPF_MAX_ROTS 360
size_t cppf_vote_workspace_bytes(int64_t n_ppfs, int n_rots, int gx, int gy, int gz);
/* Fixed-point resolution of the LDS accumulation for this problem size: each deposited weight is
 * rounded to a multiple of p2 * 2^-bits, p2 = max(probs) rounded up to a power of two (24 = fp32
 * precision for every problem of practical size; 0 = global fp32 atomics path, no quantisation). */
int cppf_vote_fixed_point_bits(int64_t n_ppfs, int n_rots, int gx, int gy, int gz);
int cppf_ppf_voting(const float* points, const float* outputs, const float* probs, const int32_t* point_idxs,
                    float* grid_obj, const float* corner, float res, int64_t n_points, int64_t n_ppfs, int n_rots,
                    int gx, int gy, int gz, int adaptive, void* workspace, size_t workspace_bytes, void* stream);

/* Same vote, plus the arg-max that the reference takes on the host (nocs/inference.py:207-208:
 * grid_obj.get(); np.argmax -> first maximum in C order).  out_idx: device i64[1] flat index,
 * out_val: device f32[1] peak value (either may be NULL).
 * accumulate is a flags word: bit 0 (CPPF_VOTE_ACCUMULATE) set: grid_obj += votes (the reference's semantics, caller
 * zero-initialises, :196); clear: grid_obj = votes (spares the caller's memset).  CPPF_VOTE_WORKGROUPS(n), n in 64..256, or'ed in:
 * launch at most n vote workgroups instead of one per CU -- a scheduling hint for callers that keep several instances in flight
 * on different streams (every workgroup pays for a 113 KB tile it zeroes, dumps and the reduce kernel reads back, whatever it
 * deposits: at N = 4096, K = 128 with three instances in flight 128 workgroups give +5 % pairs/s, the instance alone runs 7 %
 * longer).  The hint changes the schedule, not the result: the grid is the exact sum of the quantised deposits
 * and, for grids of < 4 LDS tiles, the quantum is that of a 64-wide launch at EVERY width (cppf_vote_fixed_point_bits), so a call at
 * 256, at 128 and as one object of cppf_vote_argmax_batch return the same bits.  Other bits must be zero (CPPF_EINVAL).
 * point_idxs: device i32[n_ppfs,2] (idx_is_i64 == 0, what the reference passes after `.astype(cp.int32)`,
 * nocs/inference.py:202) or the original i64[n_ppfs,2] of np.random.randint (idx_is_i64 != 0; spares the copy). */
#define CPPF_VOTE_ACCUMULATE 1
#define CPPF_VOTE_WORKGROUPS(n) (((n) & 0x1ff) << 8)
int cppf_vote_argmax(const float* points, const float* outputs, const float* probs, const void* point_idxs,
                     int idx_is_i64, float* grid_obj, const float* corner, float res, int64_t n_points, int64_t n_ppfs, int n_rots,
                     int gx, int gy, int gz, int adaptive, int accumulate, long long* out_idx, float* out_val,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The same vote as EXACT INTEGERS, for a pair list that is split over several GPUs (BASELINE.json
 * configs[4] "8-GPU shard"; no counterpart in the reference, whose atomicAdd order is whatever the hardware makes it):
 *   grid_raw     device i64[gx,gy,gz]: every cell's sum of deposited quanta (accumulate != 0: added to what is there)
 *   quantum_out  device f32[1]: value of one quantum, p2 * 2^-bits (0: probs held negative / non-finite weights, the
 *                launch accumulated in fp32 and grid_raw is NOT valid)
 *   fixed_bits   0: the launch chooses (as cppf_vote_argmax does); 8..24: every deposit is rounded to p2 * 2^-fixed_bits.
 *                Ranks that vote slices of ONE pair list pass the same value -- cppf_vote_fixed_point_bits() of the WHOLE
 *                list is always safe -- so that the integer sum of their grids is the single-GPU grid bit for bit,
 *                whatever the order of the all-reduce.  More bits than the launch would choose can overflow the
 *                wrap-around log of a workgroup: reported as quantum 0.
 * Needs the tiled integer path (a grid of <= 64 tiles): CPPF_EUNSUPPORTED otherwise; by-value launches only (no `_dyn` form).
 * n_ppfs == 0 (a rank's slice of a short list) is legal: the image is zero (unchanged with accumulate) and the quantum +infinity,
 * which a MIN over the ranks' quanta ignores and cppf_grid_from_raw turns into an all-zero grid.
 * cppf_grid_from_raw: grid[i] = (float)(raw[i] * quantum) -- the one rounding cppf_vote_argmax applies -- then the arg-max
 * (out_idx / out_val may be NULL; workspace as cppf_grid_argmax). */
int cppf_vote_grid_raw(const float* points, const float* outputs, const float* probs, const void* point_idxs, int idx_is_i64,
                       long long* grid_raw, float* quantum_out, const float* corner, float res, int64_t n_points,
                       int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive, int accumulate, int fixed_bits,
                       void* workspace, size_t workspace_bytes, void* stream);
int cppf_grid_from_raw(const long long* grid_raw, int64_t n, const float* quantum, float* grid, long long* out_idx,
                       float* out_val, void* workspace, size_t workspace_bytes, void* stream);

/* np.argmax(grid, axis=None) on device (nocs/inference.py:208).  n cells; ties -> lowest index.
 * workspace: >= 16 bytes of device scratch. */
int cppf_grid_argmax(const float* grid, int64_t n, long long* out_idx, float* out_val, void* workspace,
                     size_t workspace_bytes, void* stream);

/* nocs/inference.py:209-210: T = corners[0] + unravel_index(argmax) * res (fp64); idx device i64[1];
 * T64 device f64[3] and/or T32 device f32[3] (the copy the reference hands to backvote, :225).
 * Optional: peak (device f32[1], the arg-max value) and idx_peak_f64 (device f64[2]) -> {(double)idx, (double)peak},
 * so a host can read index, value and T back as one block of doubles. */
int cppf_center_from_argmax(const long long* idx, const float* corner, double res, int gy, int gz, double* T64,
                            float* T32, const float* peak, double* idx_peak_f64, void* stream);

/* np.argmax(counts) (first maximum) and best_dir = sphere_pts[argmax] (nocs/inference.py:283-284) in one launch:
 * counts device i32[n], sphere64 device f64[n,3], best_idx device i64[1] (optional), best_dir device f64[3]. */
int cppf_counts_argmax_select(const int32_t* counts, int n, const double* sphere64, long long* best_idx,
                              double* best_dir, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The pose tail of up to 8 objects -- everything between the centre vote's arg-max and the pose record, nocs/inference.py:209-303,335,
 * for the instances of a frame (:120 loops over them) -- in SIX launches instead of six per object:
 *   1. T = corner + unravel(arg-max) * res (:209-210)  [cppf_pose_tail_begin]     4. second MLP pass on the survivors (:236-256)
 *   2. back-vote filter (:216-231)                     [cppf_backvote_count64]       [cppf_pair_mlp_decode_sel], items with second_pass
 *   3. survivor compaction (:231)                      [cppf_compact_scatter]     5. orientation vote + sphere-bin count (:259-284)
 *                                                                                     [cppf_rot_sphere_count_dirs]
 *                                                                                 6. best_dir, sign and scale sums (:283-301,335) [cppf_pose_sums]
 * Every launch runs the single-object kernel's body on item blockIdx.y of a by-value item array: per object the results are those
 * of the six single calls in brackets, bit for bit.  Buffers are the caller's (cppf_amd.inference.PoseWorkspace lays them out):
 *   rec      device f64[21]: T[3] | best_dir[2][3] | sign sums[2][3] | scale sums[4] | arg-max index, peak -- the pose record
 *   tail0    the region zeroed by launch 1 (16-byte multiple): record, sphere-bin counts, ticket, per-chunk survivor counts
 *   counts   device i32[2][n_sphere] (inside tail0); chunk_counts i32[ceil(n_pairs / 1024)] (inside tail0); ticket u32 (inside tail0)
 *   heads    device f32[n_pairs,8]: rows of the survivors are written by launch 4 (second_pass = 1), or hold every pair's heads
 *            from an all-heads first pass (second_pass = 0)
 *   mlp_workspace  the per-point table the first pass (cppf_pair_mlp_decode / _batch) left for this object
 * Common to the batch: the pair-encoder architecture (standard fused one only), n_rots, the sphere bins (unit vectors with a monotone y
 * column: fibonacci_sphere; sphere_sorted_by_y = +1 descending / -1 ascending), thr = cos(angle_tol), max_rot_pairs (:277-280). */
typedef struct CppfPoseTailItem {
    const float* pc; const float* nrm; const float* feat;          /* device f32[n_points,3], f32[n_points,3], f32[n_points,F] */
    const long long* idx64; int32_t* idx32;                         /* pair list i64[n_pairs,2]; its i32 copy (OUTPUT of launch 2).  idx64 NULL:
                                                                     * the list IS idx32 (an INPUT: drawn as int32, cppf_stage_batch) */
    const float* outputs; const float* u_rot; float* heads;         /* (mu, nu) f32[n_pairs,2]; uniforms f32[n_pairs,2]; see above */
    const float* corner; const int32_t* shape_dev;                  /* f32[3]; NULL (dims by value) or device i32[4] {n_points, gx, gy, gz} */
    const long long* argmax_idx; const float* peak;                 /* the vote's outputs */
    const float* packed; void* mlp_workspace; size_t mlp_workspace_bytes;
    const void* vote_workspace;                                     /* the vote's workspace (its cached rotation table) or NULL */
    double* rec; float* T32;
    void* tail0; size_t tail0_bytes;
    uint8_t* mask; int32_t* chunk_counts; int32_t* surv; int32_t* count; int32_t* counts;
    long long* best_idx; unsigned* ticket; void* sums_workspace; size_t sums_workspace_bytes;
    int64_t n_points, n_pairs;
    double res64;                                                   /* res as the caller holds it (fp64: T = corner + cand * res, :210) */
    float res, tol;                                                 /* fp32 res of the kernels; back-vote tolerance (3 * res) */
    int gx, gy, gz, n_dirs, second_pass;
    /* Optional (record_out != NULL): the FINISHED pose record, assembled by launch 6's last workgroup -- the host end of
     * nocs/inference.py:299-339 (axis flips from the sign sums, right axis orthogonalised or derived, scale = exp(mean) * scale_mean * 2) --
     * so that a batch's records enter the end-of-batch gather without a host round trip.
     *   record_out  device f64[16]: T[3] | up[3] | right[3] | scale[3] | arg-max index | peak | survivors | object id
     *   object id = *object_id_dev (device i64, e.g. &CppfStageDesc.object_id) when that is not NULL, else object_id_host
     * A degenerate right axis (|right| < 1e-7, :325-328: the reference draws a random vector) takes the fixed vector
     * numpy.random.default_rng(0).standard_normal(3), as cppf_amd.inference._assemble does. */
    double* record_out;
    const long long* object_id_dev;
    long long object_id_host;
    double scale_mean[3];                                           /* config/category/<cat>.yaml: scale_mean */
    int regress_right;                                              /* the category regresses the right axis (:305-312) */
} CppfPoseTailItem;
int cppf_pose_tail_batch(int n_items, const CppfPoseTailItem* items_host, int F, const int* dims, int n_res, int out_dim, int tr_bins,
                         int rot_bins, int n_rots, const float* sphere32, const double* sphere64, int n_sphere, int sphere_sorted_by_y,
                         float thr, int64_t max_rot_pairs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Back-vote filter.  Replaces `backvote_kernel` = CUDA `backvote` (models/voting.py:70-113),
 * launched at nocs/inference.py:216-228; always adaptive.
 *   out_offsets device f32[n_ppfs,3] zero-initialised by the caller (degenerate pairs are not
 *               written, as in the reference); may be NULL when only `mask` is wanted (degenerate
 *               pairs then get mask 0, as with a zero-initialised buffer)
 *   gt_center   device f32[3], tol = 3*res in the reference
 *   mask        device u8[n_ppfs] or NULL: any(out_offsets != 0) (nocs/inference.py:230)
 * ------------------------------------------------------------------------------------------- */
int cppf_backvote(const float* points, const float* outputs, float* out_offsets, const int32_t* point_idxs,
                  const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz,
                  const float* gt_center, float tol, uint8_t* mask, void* stream);
/* The same with (a) the dims optionally in a device record (shape_dev != NULL: gx, gy, gz ignored, see the `_dyn` section)
 * and (b) the workspace of the cppf_vote_argmax* call that produced gt_center (may be NULL): the vote leaves its (cos, sin)
 * rotation table there and the back-vote loads it instead of rebuilding it per workgroup.  Same results. */
int cppf_backvote_ws(const float* points, const float* outputs, float* out_offsets, const int32_t* point_idxs,
                     const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz,
                     const int32_t* shape_dev, const float* gt_center, float tol, uint8_t* mask,
                     const void* vote_workspace, void* stream);

/* Order-preserving compaction of the surviving pairs (point_idxs[mask], nocs/inference.py:231),
 * done on device.  surv: device i32[n] receives the indices i with mask[i] != 0 in increasing
 * order; count: device i32[1].  workspace >= cppf_compact_workspace_bytes(n). */
size_t cppf_compact_workspace_bytes(int64_t n);
int cppf_compact_mask(const uint8_t* mask, int64_t n, int32_t* surv, int32_t* count, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Shape-polymorphic (`_dyn`) variants: the instance shape lives in DEVICE memory.
 *
 * The reference runs one instance at a time and re-derives every size on the host (N after voxel de-duplication,
 * nocs/inference.py:140-142; grid dims, :194-195), so every instance has its own launch geometry.  A captured hipGraph
 * bakes by-value arguments in; these variants read `shape_dev` = device i32[4] {n_points, gx, gy, gz} instead, so ONE
 * captured chain serves every instance whose shape fits the launch's capacities.  Layout and results are those of the
 * by-value entry points for the same real shape, bit for bit: the kernels evaluate the same plan function on the device;
 * capacities only size allocations and launch geometry.
 *   cppf_vote_tiles           LDS tiles the vote needs for a grid (0: more than the tiled path serves) -- lets the caller
 *                             pick `many_tiles` (needed when any instance can need >= 4 tiles; costs the binning launch and the
 *                             room for the pair -> tile queues, see cppf_vote_workspace_bytes_dyn_pairs).  `many_tiles` is a tile
 *                             CAPACITY CLASS: 0 = up to 3 tiles (the fused kernel), 1 = up to 64 (any tiled grid), 4..64 = up to that
 *                             many (ABI 4: queues and reduce launch sized for them -- a posed NOCS object needs 9-12 tiles, the C5
 *                             grid 16: class 16 asks for a quarter of class 1's queues)
 *   cppf_vote_argmax_dyn      cppf_vote_argmax; grid_obj holds grid_capacity cells, the real grid f32[gx,gy,gz] occupies its
 *                             first gx*gy*gz cells; probs/points rows beyond n_points are never read.  A record that exceeds
 *                             a capacity (cells, n_points_cap, tiles) writes out_idx = -1, out_val = NaN and votes nothing.
 *   cppf_center_from_argmax_dyn, cppf_backvote_dyn   as their by-value forms, dims from the record
 *   cppf_knn_dyn, cppf_point_encoder_forward_dyn     n_points from n_dev (device i32[1], e.g. shape_dev); launches sized
 *                             for n_cap; rows >= n_points of nbrs / out are left untouched.  k <= n_points is the caller's duty.
 * ------------------------------------------------------------------------------------------- */
int cppf_vote_tiles(int gx, int gy, int gz);
int cppf_vote_tile_cells(void);   /* cells of one LDS tile: a launch serves grids of up to 3 (many_tiles = 1: 64; = n: n) times that */
/* Workspace of a *_dyn vote launch: the state block, the pair -> tile queues of the many-tile class (n_ppfs records of 12 B for
 * every tile of the class: sized for the worst case, every pair in every tile) and one partial tile per workgroup.
 * (ABI 1 had a second, smaller size that selected round 2's kernels; those are gone.) */
size_t cppf_vote_workspace_bytes_dyn_pairs(int many_tiles, int64_t n_ppfs);
/* The vote workspace keeps state between calls (queue counters and the plane of fixed-point wrap-arounds, see the note on
 * workspaces above): its first min(cppf_vote_workspace_init_bytes(), size) bytes must be ZERO before the first call on a
 * fresh allocation (one hipMemsetAsync); every call leaves them ready for the next one.  Calls on a workspace whose header
 * was never initialised report arg-max -1 / peak NaN -- every one of them, until the caller zeroes those bytes. */
size_t cppf_vote_workspace_init_bytes(void);
/* What a by-value vote launch will do for this problem (tests, tools): out int32[10] = {path, tiles, tx, ty, ntx, nty, halo_x,
 * halo_y, workgroups, fixed-point bits}; path 0: global fp32 atomics (> 64 tiles), 2: fused kernel (< 4 tiles), 3: binning +
 * queue-consuming kernels (>= 4 tiles); (1 was round 2's kernels: gone).  Host only, no device needed. */
int cppf_vote_plan_query(int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int32_t* out);
int cppf_vote_argmax_dyn(const float* points, const float* outputs, const float* probs, const void* point_idxs,
                         int idx_is_i64, float* grid_obj, int64_t grid_capacity, const float* corner, float res,
                         int64_t n_points_cap, int64_t n_ppfs, int n_rots, const int32_t* shape_dev, int many_tiles,
                         int adaptive, int accumulate, long long* out_idx, float* out_val, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The votes of up to 8 objects -- the instances of a frame (nocs/inference.py:120 loops over them), each with its own cloud, pair
 * list, grid, results and vote workspace -- enqueued together.  Items on the tiled path (a grid of <= 64 LDS tiles; dims by value,
 * or from a device record when shape_dev != NULL) with n_rots <= 72 share ONE vote launch and ONE reduce launch: workgroups
 * [w_i, w_{i+1}) are item i's launch; an item whose grid needs >= 4 tiles (a posed object, a fine grid) first gets its own binning
 * launch.  Each object then runs on fewer, longer-lived workgroups -- cppf_vote_batch_workgroups(n, flags) each, n = the items that
 * share the launch: 256 / n (at least 64) unless CPPF_VOTE_WORKGROUPS(n) in `flags` says otherwise -- which divides the partial-tile
 * traffic per object (a workgroup zeroes, dumps and has read back its 113 KB tile whatever it deposits) without idling the rest of
 * the chip, and the launch prologue is paid once.  Every other item (n_rots > 72, an empty pair list, a grid beyond the tiled path)
 * gets the launches cppf_vote_argmax / cppf_vote_argmax_dyn would issue for it.  Per item the grid, arg-max and peak are those of
 * its own cppf_vote_argmax* call, bit for bit, whatever either call's width: the grid is the exact integer sum of the quantised
 * deposits and the fixed-point scale is that of a 64-wide launch at every width (both forms).  flags: as `accumulate` of
 * cppf_vote_argmax.  Replaces n launches of `ppf_kernel` (models/voting.py:8-66, nocs/inference.py:192-205) + np.argmax (:207-208). */
typedef struct CppfVoteItem {
    const float* points;      /* device f32[n_points,3] */
    const float* outputs;     /* device f32[n_ppfs,2] (mu, nu) */
    const float* probs;       /* device f32[n_points] or NULL (all ones) */
    const void* point_idxs;   /* device i32 / i64 [n_ppfs,2] */
    float* grid;              /* device f32[gx,gy,gz] (by value) or f32[grid_capacity] (shape_dev) */
    const float* corner;      /* device f32[3] */
    long long* out_idx;       /* device i64[1] */
    float* out_val;           /* device f32[1] */
    void* workspace;          /* cppf_vote_workspace_bytes(...) / cppf_vote_workspace_bytes_dyn_pairs(...), initialised as for the single calls */
    size_t workspace_bytes;
    const int32_t* shape_dev; /* NULL: gx, gy, gz, n_points by value; else device i32[4] {n_points, gx, gy, gz}, n_points = capacity */
    int64_t grid_capacity;    /* shape_dev only */
    int64_t n_points, n_ppfs;
    float res;
    int gx, gy, gz;
    int idx_is_i64;
    int many_tiles;           /* shape_dev only */
} CppfVoteItem;
int cppf_vote_batch_workgroups(int n_items, int flags);
int cppf_vote_argmax_batch(int n_items, const CppfVoteItem* items_host, int n_rots, int adaptive, int flags, void* stream);
int cppf_center_from_argmax_dyn(const long long* idx, const float* corner, double res, const int32_t* shape_dev, double* T64,
                                float* T32, const float* peak, double* idx_peak_f64, void* stream);
int cppf_backvote_dyn(const float* points, const float* outputs, float* out_offsets, const int32_t* point_idxs,
                      const float* corner, float res, int64_t n_ppfs, int n_rots, const int32_t* shape_dev,
                      const float* gt_center, float tol, uint8_t* mask, void* stream);
int cppf_knn_dyn(const float* pc, int n_cap, const int32_t* n_dev, int k, int32_t* nbrs, void* stream);
int cppf_point_encoder_forward_dyn(const float* pc, const float* nrm, const int32_t* nbrs, int n_cap, const int32_t* n_dev,
                                   int k, const float* packed, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                                   int n_out, int n_glob, int num_layers, float* out, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* The point encoders of up to 8 instances -- models/model.py:47-57 per instance of the loop nocs/inference.py:120,180-181 -- in THREE
 * launches (search, convolution, GlobalInfoProp) instead of three per instance: what cppf_knn_dyn + cppf_point_encoder_forward_dyn
 * compute per cloud, bit for bit (ABI 4).  A cloud of 700-2000 points is that many wavefronts, a fraction of the chip; the members of
 * a chain fill it.  One-layer standard encoder only (train.py:34; anything else: CPPF_EUNSUPPORTED, call the per-cloud entry points).
 * Members may carry different weights (instances of different categories).  n_dev NULL: n_cap is the point count.
 * nbrs_ready != 0: `nbrs` already holds the member's neighbour sets (cppf_frame_cloud_dyn leaves them), no search for it. */
typedef struct CppfPointEncItem {
    const float* pc;          /* device f32[n_cap,3] */
    const float* nrm;         /* device f32[n_cap,3] */
    int32_t* nbrs;            /* device i32[n_cap,k] */
    const int32_t* n_dev;     /* device i32[1]: points in use, or NULL */
    const float* packed;      /* this member's encoder image (cppf_point_encoder_pack / _pack_device) */
    float* out;               /* device f32[n_cap, n_out + n_glob]; rows >= the point count are left untouched */
    void* workspace;          /* cppf_point_encoder_workspace_bytes(n_cap, n_out, n_glob, 1), the member's own */
    size_t workspace_bytes;
    int32_t n_cap;
    int32_t nbrs_ready;
} CppfPointEncItem;
int cppf_point_encoder_forward_batch(int n_items, const CppfPointEncItem* items_host, int k, const int32_t* hidden, int n_hidden,
                                     int rank, int n_nbr_feats, int n_out, int n_glob, int num_layers, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Orientation candidates.  Replaces `rot_voting_kernel` = CUDA `rot_voting`
 * (models/voting.py:115-148), launched at nocs/inference.py:265-275.
 *   preds_rot  device f32[n_ppfs]          outputs_up device f32[n_ppfs,n_rots,3], zero-initialised
 * ------------------------------------------------------------------------------------------- */
int cppf_rot_voting(const float* points, const float* preds_rot, float* outputs_up, const int32_t* point_idxs,
                    int64_t n_ppfs, int n_rots, void* stream);

/* Fused rot_voting + sphere-bin count (nocs/inference.py:265-284) without materialising the
 * [n,n_rots,3] candidates: for the pairs sel[0..min(*n_sel_dev, max_pairs)) (indices into the
 * pair arrays; sel == NULL -> pairs 0..), counts[j] += #(cand . sphere[j] > thr).
 *   preds_rot device f32, element i at preds_rot[i*rot_stride]
 *   sphere device f32[S,3]; counts device i32[S] zero-initialised by the caller
 *   n_sel_dev device i32[1] (number of valid entries of sel) or NULL -> n_sel_host
 *   sphere_sorted_by_y: +1 / -1 when the bins are unit vectors whose y column is sorted descending /
 *     ascending (the Fibonacci sphere of utils/util.py:102-118 is descending): enables the banded
 *     search (identical counts, ~30x less work); 0 = no assumption. */
int cppf_rot_sphere_count(const float* points, const float* preds_rot, int rot_stride, const int32_t* point_idxs,
                          const int32_t* sel, const int32_t* n_sel_dev, int64_t n_sel_host, int64_t max_pairs,
                          int n_rots, const float* sphere, int n_sphere, float thr, int sphere_sorted_by_y,
                          int32_t* counts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pair encoder.  Replaces PPFEncoder.forward_with_idx (models/model.py:117-137) incl. the PPF
 * construction (:118-129), the feature gather/concat (:132) and the ResLayer chain (:27-31).
 *   pc, nrm    device f32[N,3]    feat device f32[N,F]
 *   idxs       device i64[P,2] (idx_is_i64 != 0) or i32[P,2]
 *   packed     device f32[cppf_pair_mlp_packed_floats(...)], built by cppf_pair_mlp_pack()
 *   out        device f32[P,out_dim] logits
 * Supported on the MFMA path: dims = {2F+4, 32, 32, 16} with F = 40 and out_dim <= 144 (the only
 * architecture the reference trains, train.py:35); any other ResLayer stack runs on the generic
 * kernel (one pair per lane, at most 128 units per layer).  The MFMA path first projects every point
 * through the feat columns of layer 0 (N*128 floats in `workspace`), see csrc/pair_mlp.hip; it addresses with
 * 32-bit byte offsets and serves N < 2^23 points and P < 2^27 pairs per call (CPPF_EUNSUPPORTED beyond:
 * split the pair list).
 * ------------------------------------------------------------------------------------------- */
size_t cppf_pair_mlp_packed_floats(int F, const int* dims, int n_res, int out_dim);
/* device scratch for one call: the per-point layer-0 projection table of the MFMA path (N*128 floats),
 * 0 for the generic kernel */
size_t cppf_pair_mlp_workspace_bytes(int64_t N, int F, const int* dims, int n_res, int out_dim);
/* host-side packing: params/offs use the layout documented in oracle/cppf_oracle.c:orc_pair_mlp
 * (flat torch tensors + offset table: 6 per res layer {fc1.w, fc1.b, fc2.w, fc2.b, fc0.w|-1,
 * fc0.b|-1} then {final.w, final.b}); packed_host receives cppf_pair_mlp_packed_floats() floats. */
int cppf_pair_mlp_pack(const float* params, const int64_t* offs, int F, const int* dims, int n_res, int out_dim,
                       float* packed_host);
/* the same image built on the device from DEVICE parameters (`params` device f32, `offs` HOST): what a training
 * loop uses, where the weights change every step (train.py:92 optimizer.step) and a host pack would cost a device ->
 * host -> device round trip with a synchronisation.  MFMA path only (else CPPF_EUNSUPPORTED). */
int cppf_pair_mlp_pack_device(const float* params, const int64_t* offs, int F, const int* dims, int n_res, int out_dim,
                              float* packed_device, void* stream);
int cppf_pair_mlp_forward(const float* pc, const float* nrm, const float* feat, const void* idxs, int idx_is_i64,
                          const float* packed, int64_t N, int F, const int* dims, int n_res, int64_t P,
                          int out_dim, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Pair encoder fused with the decode of nocs/inference.py:185-188 (+ :245-256 when heads != NULL):
 * softmax over the bins + inverse-CDF draw with caller-supplied uniforms (stand-in for
 * torch.multinomial; u < 0 selects the arg-max bin) + bin -> value.  Logits never leave the CU.
 *   u_tr  device f32[P,2]   -> outputs device f32[P,2] = (mu, nu)
 *   u_rot device f32[P,2]   -> heads   device f32[P,8] = {theta_up, theta_right, aux_up, aux_right,
 *                                                         sx, sy, sz, 0}    (both NULL to skip)
 * Requires the MFMA architecture above with tr_bins = 32, rot_bins = 36, out_dim = 141.
 * Arithmetic of the draw (restated in oracle/cppf_oracle.c:orc_sample_bin / orc_exp2w, bit for bit): weights
 * 2^(l log2e - max log2e) from a fixed degree-4 core (2.7e-6 relative, through the subnormals to zero for spreads beyond ~87),
 * summed in four consecutive segments of the bins, first bin whose running sum exceeds u * total.
 * Inputs of the pair-encoder entry points (clouds, normals, features, weights) must be finite: the file is built with
 * -fno-honor-nans, a NaN or infinity gives unspecified outputs (never a fault). */
int cppf_pair_mlp_decode(const float* pc, const float* nrm, const float* feat, const void* idxs, int idx_is_i64,
                         const float* packed, int64_t N, int F, const int* dims, int n_res, int64_t P,
                         int out_dim, int tr_bins, int rot_bins, float vr0, float vr1, const float* u_tr,
                         const float* u_rot, float* outputs, float* heads, void* workspace, size_t workspace_bytes,
                         void* stream);

/* The same call for up to 8 pair lists in ONE launch -- the instances of a frame (nocs/inference.py:120 loops over them), each with
 * its own cloud, pair list, outputs, per-point workspace and weight image (the reference keeps one network per category, :79-90).
 * The lists get workgroups in proportion to their lengths; the ~9 us a launch spends before its first MFMA are paid once.
 * Results are those of n_items cppf_pair_mlp_decode calls, bit for bit.  All items with heads / u_rot or none. */
typedef struct CppfPairMlpItem {
    const float* pc;        /* device f32[n_points,3] */
    const float* nrm;       /* device f32[n_points,3] */
    const float* feat;      /* device f32[n_points,F] */
    const void* idxs;       /* device i32 / i64 [n_pairs,2] */
    const float* packed;    /* device weight image (cppf_pair_mlp_pack) */
    const float* u_tr;      /* device f32[n_pairs,2] */
    const float* u_rot;     /* device f32[n_pairs,2] or NULL */
    float* outputs;         /* device f32[n_pairs,2] */
    float* heads;           /* device f32[n_pairs,8] or NULL */
    void* workspace;        /* device, cppf_pair_mlp_workspace_bytes(n_points, ...) */
    size_t workspace_bytes;
    int64_t n_points, n_pairs;
    float vr0, vr1;
    int idx_is_i64;
    /* cppf_pair_mlp_decode_sel_batch only (the first-pass entry ignores them): */
    const int32_t* sel;       /* device i32[>= max_sel]: the surviving pairs (cppf_compact_*'s output) */
    const int32_t* n_sel_dev; /* device i32[1]: how many */
    int64_t max_sel;          /* slots of the launch for this list (capacity; 0: skip the list) */
} CppfPairMlpItem;
int cppf_pair_mlp_decode_batch(int n_items, const CppfPairMlpItem* items_host, int F, const int* dims, int n_res, int out_dim,
                               int tr_bins, int rot_bins, void* stream);
/* The geometry that launch takes for lists of n_pairs[i] pairs (host arrays; no device work): *grid workgroups; *per_xcd > 0 = the
 * XCD-pinned mapping (1, 2, 4 or 8 lists within 10 % of each other: list i runs on XCDs [i per_xcd, (i + 1) per_xcd), so each
 * XCD's L2 holds ONE list's per-point table), 0 = contiguous workgroup ranges wg_begin[0..n_items] (optional output, may be NULL).
 * For callers that size batches and for tests that must know which mapping a launch exercised. */
int cppf_pair_mlp_batch_plan(int n_items, const int64_t* n_pairs, int* per_xcd, int* grid, int* wg_begin);
/* The second pass (cppf_pair_mlp_decode_sel, below) for up to 8 lists in one launch: list i's surviving pairs sel[0 .. min(*n_sel_dev,
 * max_sel)) get their heads rows; item.workspace must hold the per-point table its first pass left.  u_tr / outputs are not used. */
int cppf_pair_mlp_decode_sel_batch(int n_items, const CppfPairMlpItem* items_host, int F, const int* dims, int n_res, int out_dim,
                                   int tr_bins, int rot_bins, void* stream);

/* The second MLP pass of nocs/inference.py:236-256 -- ppf_encoder(..., idxs=point_idxs[mask]) followed by the decode of the
 * rotation bins, the sign logits and the log-scales -- on the surviving pairs only, without materialising point_idxs[mask]:
 * slot i < min(*n_sel_dev, max_sel) works on pair sel[i] (cppf_compact_mask's output) and writes heads[sel[i]]; u_rot is
 * indexed by ORIGINAL pair like heads.  Rows of pairs that are not selected are left untouched.  `workspace` must be the one
 * the preceding cppf_pair_mlp_decode call on the same (feat, packed) used: its per-point table is reused, not rebuilt.
 * The launch is sized for max_sel slots (a captured graph cannot know the count); surplus workgroups exit at once. */
int cppf_pair_mlp_decode_sel(const float* pc, const float* nrm, const float* feat, const void* idxs, int idx_is_i64,
                             const float* packed, int64_t N, int F, const int* dims, int n_res, int64_t P, int out_dim,
                             int tr_bins, int rot_bins, const float* u_rot, const int32_t* sel, const int32_t* n_sel_dev,
                             int64_t max_sel, float* heads, void* workspace, size_t workspace_bytes, void* stream);

#ifdef CPPF_DEBUG_ENTRY
/* Profiling aid, compiled only with -DCPPF_DEBUG_ENTRY (not in the shipped library): the PPF + gather + MFMA chain of the
 * standard architecture with no epilogue (isolates the matrix pipeline when reading rocprof counters).
 * scratch: >= 4 bytes of device memory (never written in practice). */
int cppf_debug_mlp_chain_only(const float* pc, const float* nrm, const float* feat, const void* idxs, int idx_is_i64,
                              const float* packed, int64_t N, int64_t P, float* scratch, void* workspace,
                              size_t workspace_bytes, void* stream);
#endif

/* Decode from logits already in memory (generic architectures / bin counts). */
int cppf_decode_center(const float* logits, int64_t P, int ld, int tr_bins, float vr0, float vr1,
                       const float* u_tr, float* outputs, void* stream);
int cppf_decode_rot(const float* logits, int64_t P, int ld, int out_dim, int tr_bins, int rot_bins,
                    const float* u_rot, float* heads, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reductions of the pose tail.
 *  cppf_axis_sign: nocs/inference.py:287-301.  For pairs sel[0..*n_sel_dev): flip n_a to agree with
 *    ab, target = (n . best_dir > 0), sums of BCEWithLogits(aux, target) and (aux, 1-target).
 *    aux element i at aux[i*aux_stride]; best_dir device f64[3]; out device f64[3] =
 *    {up_loss_sum, down_loss_sum, n}.  The caller compares the two means.
 *  cppf_scale_sum: nocs/inference.py:335.  out device f64[4] = {sum sx, sum sy, sum sz, n} over the
 *    selected pairs; the caller finishes exp(mean)*scale_mean*2.
 *  Both are two-kernel deterministic reductions (fixed order), workspace >= cppf_reduce_workspace_bytes().
 * ------------------------------------------------------------------------------------------- */
size_t cppf_reduce_workspace_bytes(void);
int cppf_axis_sign(const float* pc, const float* nrm, const int32_t* point_idxs, const int32_t* sel,
                   const int32_t* n_sel_dev, int64_t n_sel_host, const float* aux, int aux_stride,
                   const double* best_dir, double* out, void* workspace, size_t workspace_bytes, void* stream);
int cppf_scale_sum(const float* scale_logits, int stride, const int32_t* sel, const int32_t* n_sel_dev,
                   int64_t n_sel_host, double* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The pose tail in six launches (nocs/inference.py:209-303,335 after the centre vote).  The entry points above map one
 * to one onto the reference's steps; these fuse neighbours whose only separation was a launch boundary, with identical
 * results (the integer outputs bit for bit; the fp64 sums of cppf_pose_sums within 1e-12 relative of the two-kernel forms:
 * other order of the final partial sum).
 *  cppf_pose_tail_begin: cppf_center_from_argmax(_dyn) (shape_dev non-null: dims from the record) that first zeroes
 *    zero_bytes (multiple of 16, 16-byte aligned) at zero_ptr: the bin counts, chunk counts, ticket and result record of
 *    the launches below.  T64 / idx_peak_f64 may lie inside the region.
 *  cppf_backvote_count: cppf_backvote_ws writing the mask only, plus chunk_counts[i] += survivors among pairs
 *    [1024 i, 1024 i + 1024) (int32[(n_ppfs + 1023) / 1024], zero on entry).
 *  cppf_compact_scatter: cppf_compact_mask given those counts (no counting pass, no scan launch).  n <= 8192 * 1024
 *    pairs (CPPF_EUNSUPPORTED beyond: use cppf_compact_mask).
 *  cppf_rot_sphere_count_dirs: cppf_rot_sphere_count for n_dirs angle columns in one launch: direction j reads
 *    preds_rot[p * rot_stride + j * rot_dir_step] and adds to counts[j * counts_dir_step + bin].
 *  cppf_pose_sums: np.argmax of each direction's counts and best_dir = sphere64[argmax] (cppf_counts_argmax_select),
 *    the sign sums of each direction (cppf_axis_sign; aux of direction j at aux[p * aux_stride + j]) and the scale sums
 *    (cppf_scale_sum; scale_logits may be null) in one launch.  best_idx i64[n_dirs] (may be null), best_dir f64[n_dirs][3],
 *    sign f64[n_dirs][3], scale_out f64[4]; n_dirs <= 2; workspace >= cppf_pose_sums_workspace_bytes(); ticket: one
 *    device uint32, zero before the first call and left at zero.
 * ------------------------------------------------------------------------------------------- */
int cppf_pose_tail_begin(const long long* idx, const float* corner, double res, int gy, int gz, const int32_t* shape_dev,
                         double* T64, float* T32, const float* peak, double* idx_peak_f64, void* zero_ptr,
                         size_t zero_bytes, void* stream);
int cppf_backvote_count(const float* points, const float* outputs, const int32_t* point_idxs, const float* corner,
                        float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz, const int32_t* shape_dev,
                        const float* gt_center, float tol, uint8_t* mask, int32_t* chunk_counts,
                        const void* vote_workspace, void* stream);
/* cppf_backvote_count reading the pair list as int64 (the caller's np.random.randint array, nocs/inference.py:177) and leaving
 * its int32 copy in idx32_out (int32[n_ppfs][2], may be null) for the launches that follow. */
int cppf_backvote_count64(const float* points, const float* outputs, const long long* point_idxs64, int32_t* idx32_out,
                          const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz,
                          const int32_t* shape_dev, const float* gt_center, float tol, uint8_t* mask, int32_t* chunk_counts,
                          const void* vote_workspace, void* stream);
int cppf_compact_scatter(const uint8_t* mask, int64_t n, const int32_t* chunk_counts, int32_t* surv, int32_t* count,
                         void* stream);
int cppf_rot_sphere_count_dirs(const float* points, const float* preds_rot, int rot_stride, int rot_dir_step, int n_dirs,
                               const int32_t* point_idxs, const int32_t* sel, const int32_t* n_sel_dev,
                               int64_t n_sel_host, int64_t max_pairs, int n_rots, const float* sphere, int n_sphere,
                               float thr, int sphere_sorted_by_y, int32_t* counts, int counts_dir_step, void* stream);
/* cppf_rot_sphere_count_dirs on a caller-chosen subsample: slot k < min(n_order, max_pairs) takes the survivor at position
 * order[k] of `sel` (positions outside [0, *n_sel_dev) contribute nothing).  The reference draws its 10 000 pairs by shuffling
 * the survivors (nocs/inference.py:277-280: idxs = arange(P'); np.random.shuffle(idxs); idxs[:10000]); passing that shuffled
 * prefix as `order` (device int32[n_order]) reproduces its subsample exactly.  Without it the first max_pairs survivors in
 * pair order are taken -- the same distribution, since pairs are i.i.d. */
int cppf_rot_sphere_count_dirs_order(const float* points, const float* preds_rot, int rot_stride, int rot_dir_step, int n_dirs,
                                     const int32_t* point_idxs, const int32_t* sel, const int32_t* n_sel_dev,
                                     int64_t n_sel_host, const int32_t* order, int64_t n_order, int64_t max_pairs, int n_rots,
                                     const float* sphere, int n_sphere, float thr, int sphere_sorted_by_y, int32_t* counts,
                                     int counts_dir_step, void* stream);
size_t cppf_pose_sums_workspace_bytes(void);
int cppf_pose_sums(const float* pc, const float* nrm, const int32_t* point_idxs, const int32_t* sel,
                   const int32_t* n_sel_dev, int64_t n_sel_host, const float* aux, int aux_stride, int n_dirs,
                   const int32_t* counts, int n_sphere, int counts_dir_step, const double* sphere64,
                   const float* scale_logits, int scale_stride, long long* best_idx, double* best_dir, double* sign,
                   double* scale_out, void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream);

/* nocs/inference.py:194-195 on device: corner = min(pc), dims = int32((max-min)/res)+1.
 * corner device f32[3], dims device i32[3]. */
int cppf_grid_setup(const float* pc, int64_t N, float res, float* corner, int32_t* dims, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SPRIN point encoder: produces the per-point `feat` the pair MLP gathers.
 * Replaces models/model.py:36-78 `PointEncoder.forward(pc, pc_normal, dist)` / `forward_nbrs(...)` with
 * models/sprin.py:40-107 (rifeat, conv_kernel, SparseSO3Conv, GlobalInfoProp), called at
 * nocs/inference.py:180-181 and train.py:62-64.
 *
 * cppf_knn: the neighbour sets of `torch.topk(dist, k, largest=False)` (models/model.py:47).
 *   dist != NULL: device f32[N,N], keys are dist[i][j] (the matrix the reference passes in);
 *   dist == NULL: keys are exact squared distances from pc (device f32[N,3]) -- no N x N matrix.
 *   nbrs: device i32[N,k], each row the k smallest keys (ties -> lower index) in ascending index order.
 *   One wavefront per query, keys streamed twice: the k-th smallest of the 64 per-lane minima bounds the
 *   answer, the ~3k keys below that bound are compacted into LDS and an exact 32-step bisection runs on them.
 *
 * cppf_point_encoder_forward: forward_nbrs.  pc, nrm device f32[N,3]; nbrs device i32[N,k], k <= 64;
 *   out device f32[N, n_out+n_glob].  `packed` device f32 = the image cppf_point_encoder_pack() builds on the host
 *   from the parameters in NATURAL order, which is per layer
 *     for each hidden width h_i (in_i = 6, then h_{i-1}):  W[h_i][in_i], b[h_i], ln_weight[h_i], ln_bias[h_i]
 *     Wk[rank][h_last], bk[rank]                      (spconvs.l.kernel.*)
 *     Wo_t[rank*n_in][n_out] = outnet.weight TRANSPOSED, bo[n_out], ln_weight[n_out], ln_bias[n_out]
 *     Wa[n_glob][n_out], ba[n_glob]                   (aggrs.l.linear)
 *   with n_in = n_nbr_feats (must be 2) for layer 0 and n_out+n_glob after it.  The image is that block followed,
 *   for the supported shape, by a lane-ordered copy of each layer's kernel-MLP weights for the MFMA kernel.
 *   Device kernel exists for hidden = {32,64,32,32}, rank = 32, n_out = 32, n_glob <= 32 (the
 *   configuration of train.py:34 / nocs/inference.py:82); anything else returns CPPF_EUNSUPPORTED.
 * ------------------------------------------------------------------------------------------- */
int cppf_knn(const float* pc, const float* dist, int n_points, int k, int32_t* nbrs, void* stream);
size_t cppf_point_encoder_packed_floats(const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out,
                                        int n_glob, int num_layers);
int cppf_point_encoder_pack(const float* natural /*host*/, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                            int n_out, int n_glob, int num_layers, float* packed_out /*host, packed_floats() long*/);
size_t cppf_point_encoder_workspace_bytes(int n_points, int n_out, int n_glob, int num_layers);
int cppf_point_encoder_forward(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k,
                               const float* packed, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                               int n_out, int n_glob, int num_layers, float* out, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the pair encoder.  The reference has no backward code: train.py:91
 * calls loss.backward() and torch autograd differentiates models/model.py:117-137; this entry point returns
 * the same gradients for ppffcs = {84,32,32,16} (train.py:35), F = 40, any out_dim (else CPPF_EUNSUPPORTED).
 *   params       device f32, the parameters in torch layout, flat: per res layer fc1.weight, fc1.bias,
 *                fc2.weight, fc2.bias, [fc0.weight, fc0.bias], then final.weight, final.bias
 *   offs         HOST i64[6*n_res+2]: offset of each of those tensors in `params` (-1 for an absent fc0)
 *   grad_out     device f32[n_pairs, out_dim]  (dL/dlogits)
 *   grad_params  device f32, same layout as `params`, OVERWRITTEN
 *   grad_feat    device f32[n_points, F], ACCUMULATED (+=; zero it first for a plain gradient)
 * Formulation (cppf_amd/csrc/pair_mlp_bwd.hip): one workgroup of four wavefronts per tile of 64 pairs recomputes the
 * forward and back-propagates on the fp32 matrix cores; weight-gradient tiles accumulate in registers over all the
 * workgroup's pairs and are written once to one of at most CPPF_BWD_MAX_PARTS partial gradients in the workspace
 * (tiles dealt evenly), which are then added in a fixed two-level order.  Everything that touches the 2 x 40 feature
 * columns of layer 0 is done per POINT: each pair emits one 64-float row, the 2*n_pairs (point, entry) keys are radix
 * sorted (stable), every point adds its rows in pair order per role, and d(feat) and the feature columns of the two
 * layer-0 weight gradients are products of those sums.  No atomics anywhere: every result is deterministic
 * (oracle/backward_oracle.c restates the order).
 * ------------------------------------------------------------------------------------------- */
#define CPPF_BWD_MAX_PARTS 512
size_t cppf_pair_mlp_backward_workspace_bytes(int64_t n_pairs, int64_t n_points, int F, const int* dims, int n_res,
                                              int out_dim);
int cppf_pair_mlp_backward(const float* pc, const float* nrm, const float* feat, const void* idxs, int idx_is_i64,
                           const float* params, const int64_t* offs, int64_t n_points, int F, const int* dims, int n_res,
                           int64_t n_pairs, int out_dim, const float* grad_out, float* grad_params, float* grad_feat,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the point encoder: gradients of the parameters of the one-layer standard encoder (train.py:34: hidden
 * {32,64,32,32}, rank 32, 2 neighbour features, n_out 32, n_glob 8, k <= 64; anything else CPPF_EUNSUPPORTED).  The
 * reference has no backward code -- train.py:91 differentiates models/model.py:46-61 + models/sprin.py:40-107 with
 * autograd; points and normals carry no gradient there (train.py:58-60).
 *   packed       device f32: natural parameters + forward image, as cppf_point_encoder_forward takes them
 *   out_fwd      device f32[n_points, 40]: the forward output for the same inputs (the pooled maxima are read from it)
 *   contraction  device f32[n_points, 64] kept by cppf_point_encoder_forward_train, or NULL: the backward then recomputes
 *                it (one more forward pass of the kernel-MLP per point); the result is the same bit for bit
 *   grad_out     device f32[n_points, 40]
 *   grad_packed  device f32[9 256]: d/d(natural parameters) in the natural layout (outnet weight transposed), OVERWRITTEN
 * cppf_point_encoder_pack_device builds `packed` from natural parameters that already live on the device (training).
 * One wavefront per point, at most CPPF_SPRIN_BWD_MAX_PARTS partial gradients added in a fixed two-level order: the
 * result is deterministic (oracle/sprin_bwd_oracle.c restates the order).
 * ------------------------------------------------------------------------------------------- */
#define CPPF_SPRIN_BWD_MAX_PARTS 1024
size_t cppf_point_encoder_backward_workspace_bytes(int n_points);
int cppf_point_encoder_pack_device(const float* natural, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out,
                                   int n_glob, int num_layers, float* packed, void* stream);
int cppf_point_encoder_backward(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k, const float* packed,
                                const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats, int n_out, int n_glob,
                                int num_layers, const float* out_fwd, const float* contraction, const float* grad_out,
                                float* grad_packed, void* workspace, size_t workspace_bytes, void* stream);
/* cppf_point_encoder_forward that also keeps the per-point contraction (einsum of models/sprin.py:99) for the backward */
int cppf_point_encoder_forward_train(const float* pc, const float* nrm, const int32_t* nbrs, int n_points, int k,
                                     const float* packed, const int32_t* hidden, int n_hidden, int rank, int n_nbr_feats,
                                     int n_out, int n_glob, int num_layers, float* out, float* contraction_out, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pre-processing in front of the path.  Both replace third-party host calls whose
 * results are not fully specified, so parity with the reference is unpinned; the definitions are in
 * oracle/preproc_oracle.c.
 *  cppf_voxel_dedupe     the role of ME.utils.sparse_quantize(pc, return_index=True, quantization_size=res)[1]
 *                        (nocs/inference.py:140): keep_idx device i32[n_points] receives, in ascending order, the lowest
 *                        index of every occupied voxel floor(p / res) (fp64 divide); count device i32[1] = how many.
 *  cppf_estimate_normals open3d estimate_normals(KDTreeSearchParamKNN(knn)) (utils/util.py:61-65): nbrs device
 *                        i32[n_points, k] (cppf_knn output: the point itself is a neighbour); normals device
 *                        f32[n_points, 3] = unit eigenvector of the smallest eigenvalue of the neighbours' covariance
 *                        (fp64 cumulants, 8 Jacobi sweeps), sign: component of largest magnitude positive.
 *  cppf_backproject      utils/util.py:598-631 `backproject(depth, intrinsics, instance_mask)` (nocs/inference.py:131): the
 *                        pixels with mask != 0 and depth > 0, in row-major order, back-projected through inv(intrinsics) in
 *                        fp64: pts device f64[H*W,3] (x and y negated like the reference's return value), pix device
 *                        i32[H*W] = v*W + u of each point (the reference's `idxs`), count device i32[1].  depth: device
 *                        u16[H,W] (depth_is_u16 != 0; NOCS depth PNGs) or f32[H,W]; mask device u8[H,W]; kinv_host: HOST
 *                        f64[9], row-major inverse of the 3x3 intrinsics.
 * ------------------------------------------------------------------------------------------- */
/* The whole per-instance pre-processing of nocs/inference.py:131-142 + the grid set-up of :194-195 as ONE count-driven stage for
 * captured, shape-polymorphic chains: no size visits the host.  A frame's depth image and a LABEL image (bit `label_bit` of pixel p
 * set <=> p belongs to the instance: up to 8 / 16 / 32 possibly overlapping instance masks in one upload) stay on the device;
 * the stage back-projects the instance's valid pixels (cppf_backproject's arithmetic), divides by `divisor` (1000: millimetres,
 * :132), flips x and y (:136-137), de-duplicates per voxel of edge `res` (cppf_voxel_dedupe's definition, :140-141), writes the cloud
 * to pc_out f32[n_cap,3], its PCA normals over knn_k neighbours (cppf_knn + cppf_estimate_normals, :142) to nrm_out f32[n_cap,3], the
 * grid corner to corner_out f32[3] and the instance's shape record to shape_out i32[4] = {N, gx, gy, gz} -- what the *_dyn entry
 * points read.  N = 0 (and a 1x1x1 grid) when fewer than k_min points are left: the reference skips such instances (:121-123);
 * the *_dyn vote then reports arg-max -1.  n_cap >= the number of set label pixels (the caller counts them on the host) bounds
 * every buffer; rows >= N of pc_out / nrm_out are left untouched.  nbrs_out (may be NULL): device i32[n_cap, knn_k] that receives the
 * neighbour sets the normals were fitted on -- cppf_knn's output, which a point encoder with the same k can reuse instead of
 * searching again (nocs/inference.py:180 computes the same cdist + topk).  Results equal the four single calls on the same inputs, bit for
 * bit.  cppf_mod_pairs_dyn: idx[i] <- idx[i] mod N for pairs drawn as full-range non-negative integers before N was known (:177). */
size_t cppf_frame_cloud_workspace_bytes(int H, int W, int n_cap, int knn_k);
int cppf_frame_cloud_dyn(const void* depth, int depth_is_u16, const void* labels, int label_bytes, int label_bit, int H, int W,
                         const double* kinv_host, double divisor, double res, int knn_k, int k_min, int n_cap, float* pc_out,
                         float* nrm_out, float* corner_out, int32_t* shape_out, int32_t* nbrs_out, void* workspace, size_t workspace_bytes,
                         void* stream);
/* The same with the label bit read from device memory at run time (*label_bit_dev, taken modulo the label width): ONE captured launch
 * serves whichever instance of a frame is handed to it -- a video whose instances change order, number or mask size replays the
 * graphs it has (cppf_amd.frames.FrameRunner keeps them per (category, capacity) and writes {bit, seed} per frame). */
int cppf_frame_cloud_dyn_bit(const void* depth, int depth_is_u16, const void* labels, int label_bytes, const int32_t* label_bit_dev, int H,
                             int W, const double* kinv_host, double divisor, double res, int knn_k, int k_min, int n_cap, float* pc_out,
                             float* nrm_out, float* corner_out, int32_t* shape_out, int32_t* nbrs_out, void* workspace,
                             size_t workspace_bytes, void* stream);
/* The frame stage of up to 8 instances of ONE frame (nocs/inference.py:120 loops over them) in EIGHT launches instead of sixteen
 * per instance, their pair lists and bin uniforms included (cppf_sample_pairs, the key read from *seed_dev): per instance the results
 * of cppf_frame_cloud_dyn_bit + cppf_sample_pairs, bit for bit (ABI 4).  A frame's chain is launch-bound -- ~15 kernels of ~5 us per
 * instance, and a hipGraph launch costs the host per kernel node -- so a launch serves every member (blockIdx.y), the mask kernels
 * count their own chunks (no count + scan launches), the first one clears the voxel table, the normals launch also sets up the grid
 * and draws the pairs.  Every member has its own capacity, resolution, k, workspace (cppf_frame_cloud_workspace_bytes) and outputs. */
typedef struct CppfFrameCloudItem {
    const int32_t* label_bit_dev;          /* device i32: the member's bit of the label image (taken modulo the label width) */
    const unsigned long long* seed_dev;    /* device u64: Philox key of its pair / uniform draws (n_pairs > 0) */
    float* pc_out;                         /* device f32[n_cap,3] */
    float* nrm_out;                        /* device f32[n_cap,3] */
    float* corner_out;                     /* device f32[3] */
    int32_t* shape_out;                    /* device i32[4] {N (0: fewer than k_min points), gx, gy, gz} */
    int32_t* nbrs_out;                     /* device i32[n_cap,knn_k] or NULL (kept in the workspace) */
    void* idx;                             /* device i64[n_pairs,2] (idx_is_i64) or i32[n_pairs,2] */
    float* u_tr;                           /* device f32[n_pairs,2] or NULL */
    float* u_rot;                          /* device f32[n_pairs,2] or NULL */
    void* workspace;
    size_t workspace_bytes;
    double res;
    int64_t n_pairs;                       /* 0: no draws */
    int32_t knn_k, k_min, n_cap, idx_is_i64;
} CppfFrameCloudItem;
int cppf_frame_cloud_dyn_batch(int n_items, const CppfFrameCloudItem* items_host, const void* depth, int depth_is_u16, const void* labels,
                               int label_bytes, int H, int W, const double* kinv_host, double divisor, void* stream);
/* Pair list and bin uniforms drawn on the device -- the reference draws the pairs with np.random.randint(0, N, (P, 2)) on the host
 * (nocs/inference.py:177: 8 MB per instance at C2 over PCIe) and the bins with torch.multinomial (:186,250,254).  idx device
 * i64[n_pairs,2] uniform over [0, N); u_tr / u_rot device f32[n_pairs,2] uniform over [0, 1) (either may be NULL).  N = n_points, or
 * *n_dev (device i32) when n_dev != NULL.  Philox-4x32-10 keyed by `seed`, counter = pair index: the draw of a pair is a function of
 * (seed, pair index) alone; seed_dev != NULL: the seed is read from device memory instead (a captured launch replayed with a new seed
 * per frame).  With N = 0 every index is 0.  Same distributions as the reference's generators, not the same numbers (parity tests pass arrays). */
int cppf_sample_pairs(long long* idx, float* u_tr, float* u_rot, int64_t n_pairs, int64_t n_points, const int32_t* n_dev,
                      unsigned long long seed, const unsigned long long* seed_dev, void* stream);
/* Host helper (no device): the grid of nocs/inference.py:194-195 for a HOST cloud f32[n_points,3]: corners_host f32[6] = {min xyz, max xyz},
 * dims_host i32[3] = int32((max - min) / res) + 1 with the quotient in fp32.  A cloud with a NaN / inf coordinate anywhere returns
 * CPPF_ENONFINITE (numpy's min / max would propagate it into the dims: the reference fails at np.zeros(grid_res), :196). */
int cppf_host_grid_shape(const float* pc_host, int64_t n_points, float res, float* corners_host, int32_t* dims_host);
int cppf_mod_pairs_dyn(long long* idx, int64_t n_pairs, const int32_t* n_dev, void* stream);
/* The head of a captured chain for objects that are ALREADY on the device ("inputs already resident on device"): for up
 * to 8 objects, ONE launch copies each object's cloud, normals and (optional) per-point features from wherever the caller keeps them
 * into the chain's own buffers, sets up its vote grid (nocs/inference.py:194-195: corner = min(pc), dims = int32((max - min) / res) + 1,
 * as cppf_grid_setup) and draws its pair list and bin uniforms (:177,186,250: as cppf_sample_pairs, bit for bit, with seed = desc.seed).
 * What changes from one replay to the next -- where the object lives, how many points it has, its seed, its id -- is read from a
 * DESCRIPTOR in device memory (the caller rewrites the descriptors with one small copy before each replay); what does not -- the
 * chain's buffers -- travels by value, so the launch is capturable in a hipGraph.  The reference's counterpart is the host side of the
 * instance loop, nocs/inference.py:120-142,177,194-196 (numpy arrays uploaded per instance). */
typedef struct CppfStageDesc {          /* DEVICE memory, 48 bytes, one per object of the chain */
    const float* pc_src;                /* device f32[n_points,3] */
    const float* nrm_src;               /* device f32[n_points,3] */
    const float* feat_src;              /* device f32[n_points,F] or NULL (features come from a point encoder in the chain) */
    long long n_points;                 /* <= the item's n_cap (the caller checks) */
    unsigned long long seed;            /* Philox key of the pair / uniform draws */
    long long object_id;                /* copied into the pose record (CppfPoseTailItem.object_id_dev may point here) */
} CppfStageDesc;
typedef struct CppfStageItem {          /* host memory, by value at launch: the chain's buffers */
    const CppfStageDesc* desc;          /* device */
    float* pc; float* nrm; float* feat; /* device f32[n_cap,3], f32[n_cap,3], f32[n_cap,F] (feat NULL: not copied) */
    float* corner;                      /* device f32[3] */
    int32_t* shape;                     /* device i32[4] <- {n_points, gx, gy, gz}, or NULL (a static-shape pipeline: only `corner` is written) */
    void* idx;                          /* device i64[n_pairs,2] (idx_is_i64) or i32[n_pairs,2], or NULL (no draw) */
    float* u_tr; float* u_rot;          /* device f32[n_pairs,2] each; either may be NULL */
    int64_t n_pairs, n_cap;
    int F;
    float res;
    int idx_is_i64;                     /* 1: the reference's int64 pair list (nocs/inference.py:177); 0: int32 -- half the index bytes for
                                         * every kernel of the chain that streams the list (the values are < n_points < 2^31 either way) */
} CppfStageItem;
int cppf_stage_batch(int n_items, const CppfStageItem* items_host, void* stream);
/* n_words 64-bit words from src to dst by a KERNEL of `stream` (8-byte aligned; either side may be pinned host memory, which the device
 * reads / writes in place).  For the few hundred bytes a batch driver moves per chain -- descriptors in, records out: a copy engine's
 * queue is shared between streams and in order, so such copies can wait behind another stream's unrelated transfer. */
int cppf_copy_words(void* dst, const void* src, int64_t n_words, void* stream);
/* row r of dst (n_words 64-bit words each) <- the words at src_host[r] (a HOST array of up to 32 device pointers, 8-byte aligned), one
 * launch: the result records of a chain's members into one array instead of one small copy per member (ABI 4) */
int cppf_gather_words(int n_rows, const void* const* src_host, int64_t n_words, void* dst, void* stream);
size_t cppf_backproject_workspace_bytes(int H, int W);
int cppf_backproject(const void* depth, int depth_is_u16, const uint8_t* mask, int H, int W, const double* kinv_host,
                     double* pts, int32_t* pix, int32_t* count, void* workspace, size_t workspace_bytes, void* stream);
size_t cppf_voxel_dedupe_workspace_bytes(int64_t n_points);
int cppf_voxel_dedupe(const float* pc, int64_t n_points, double res, int32_t* keep_idx, int32_t* count, void* workspace,
                      size_t workspace_bytes, void* stream);
int cppf_estimate_normals(const float* pc, const int32_t* nbrs, int64_t n_points, int k, float* normals, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CPPF_H */
