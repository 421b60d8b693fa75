/* A host in plain C over the C ABI (include/cppf.h) -- no Python, no torch: what a maintainer of a non-Python caller would write.
 *
 *   vote_host            host-only entry points (no device): ABI version, error strings, launch plans, argument checks
 *   vote_host gpu        the centre vote + arg-max of one synthetic object on the device (cppf_vote_argmax, the drop-in for
 *                        models/voting.py:8-66 + nocs/inference.py:207-210), checked against the oracle's restatement of the same
 *                        lines (oracle/cppf_oracle.c: orc_ppf_voting / orc_grid_argmax / orc_center_from_argmax)
 *   vote_host chain      the whole hot path from C: PPF + pair MLP + centre decode (cppf_pair_mlp_pack / cppf_pair_mlp_decode:
 *                        models/model.py:117-137, nocs/inference.py:185-188) on the reference's architecture with seeded random
 *                        weights, int64 pair list, then the vote of what it emitted; (mu, nu) of every pair compared BIT FOR BIT with
 *                        orc_pair_mlp(order 1) + orc_decode_center, the arg-max with the oracle's vote of the same (mu, nu)
 *
 * TEST code: it links the oracle as the checker (tests/test_abi_and_host.py, tests/test_gpu_parity.py build and run it). */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cppf.h"

void orc_ppf_voting(const float* points, const float* outputs, const float* probs, const int32_t* point_idxs, float* grid_obj,
                    const float* corner, float res, int64_t n_ppfs, int n_rots, int gx, int gy, int gz, int adaptive, int64_t* n_atomics);
int64_t orc_grid_argmax(const float* grid, int64_t n, float* val);
void orc_center_from_argmax(int64_t flat, int gy, int gz, const float* corner, double res, double* T);
int orc_pair_mlp(const float* pc, const float* nrm, const float* feat, const int64_t* idxs, int64_t N, int F, int64_t P,
                 const float* params, const int64_t* offs, const int* dims, int n_res, int out_dim, int order, float* out);
void orc_decode_center(const float* logits, int64_t P, int ld, int nb, const float* u, float vr0, float vr1, float* outputs, int32_t* bins);

#define REQUIRE(cond)                                                                  \
    do {                                                                               \
        if (!(cond)) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)
#define HIP(call)                                                                                        \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } \
    } while (0)

static uint32_t lcg_state = 12345u;
static uint32_t lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state >> 8; }
static float unif(void) { return (float)lcg() * (1.0f / 16777216.0f); }

enum { N = 1024, K = 24, P = N * K, ROTS = 72 };
static const float RES = 4e-3f;

/* a bottle-like surface of revolution around the y axis, centre of the NOCS box at `centre`; pairs; the (mu, nu) a perfect
 * network would emit: c = a - ab * mu is the foot of the perpendicular from the centre onto the line through the pair, nu its
 * distance from the centre, so that every vote circle passes through the centre (models/voting.py:19-33) */
static void make_object(float* pc, int32_t* idx, float* outputs, float* centre)
{
    centre[0] = 0.05f; centre[1] = 0.02f; centre[2] = 0.72f;
    for (int i = 0; i < N; ++i) {
        const float h = unif() - 0.5f, t = 6.2831853f * unif();
        const float r = h > 0.3f ? 0.012f : 0.035f;
        pc[3 * i] = centre[0] + r * cosf(t);
        pc[3 * i + 1] = centre[1] + 0.15f * h;
        pc[3 * i + 2] = centre[2] + r * sinf(t);
    }
    for (int p = 0; p < P; ++p) {
        const int a = p / K, b = (int)(lcg() % N);
        idx[2 * p] = a; idx[2 * p + 1] = b;
        float ab[3], n = 0.f, mu = 0.f, d2 = 0.f;
        for (int c = 0; c < 3; ++c) { ab[c] = pc[3 * a + c] - pc[3 * b + c]; n += ab[c] * ab[c]; }
        n = sqrtf(n) + 1e-7f;
        for (int c = 0; c < 3; ++c) { ab[c] /= n; mu += (pc[3 * a + c] - centre[c]) * ab[c]; }
        for (int c = 0; c < 3; ++c) { const float f = pc[3 * a + c] - ab[c] * mu - centre[c]; d2 += f * f; }
        outputs[2 * p] = mu; outputs[2 * p + 1] = sqrtf(d2);
    }
}

static int host_only(void)
{
    REQUIRE(cppf_abi_version() == CPPF_ABI_VERSION);
    REQUIRE(cppf_error_string(CPPF_EINVAL) && strlen(cppf_error_string(CPPF_EINVAL)) > 0);
    REQUIRE(cppf_error_string(CPPF_EWORKSPACE) && cppf_error_string(CPPF_EUNSUPPORTED) && cppf_error_string(0));
    /* the grid of nocs/inference.py:194-195 from a host cloud */
    const float cloud[12] = {0.f, 0.f, 0.f, 0.1f, 0.3f, 0.1f, 0.05f, 0.1f, 0.02f, 0.1f, 0.f, 0.f};
    float corners[6]; int32_t dims[3];
    REQUIRE(cppf_host_grid_shape(cloud, 4, RES, corners, dims) == 0);
    REQUIRE(dims[0] == (int32_t)(0.1f / RES) + 1 && dims[1] == (int32_t)(0.3f / RES) + 1 && dims[2] == dims[0]);
    REQUIRE(corners[0] == 0.f && corners[4] == 0.3f);
    REQUIRE(cppf_host_grid_shape(NULL, 4, RES, corners, dims) == CPPF_EINVAL);
    {   /* a NaN in the middle of the cloud is reported, not skipped by the comparisons */
        float holed[12];
        for (int i = 0; i < 12; ++i) holed[i] = cloud[i];
        holed[7] = 0.f / 0.f;
        REQUIRE(cppf_host_grid_shape(holed, 4, RES, corners, dims) == CPPF_ENONFINITE);
        holed[7] = 1.f / 0.f;
        REQUIRE(cppf_host_grid_shape(holed, 4, RES, corners, dims) == CPPF_ENONFINITE);
    }
    /* launch plans: BASELINE.json configs[1] (26 x 76 x 26: fused, < 4 tiles) and configs[4]-style (52 x 152 x 52: binned) */
    int32_t plan[10];
    REQUIRE(cppf_vote_plan_query(524288, ROTS, 26, 76, 26, plan) == 0 && plan[0] == 2 && plan[1] >= 1 && plan[1] < 4);
    REQUIRE(cppf_vote_plan_query(2097152, ROTS, 52, 152, 52, plan) == 0 && plan[0] == 3 && plan[1] == cppf_vote_tiles(52, 152, 52));
    REQUIRE(cppf_vote_workspace_bytes(524288, ROTS, 26, 76, 26) >= cppf_vote_workspace_init_bytes());
    REQUIRE(cppf_vote_batch_workgroups(4, 0) == 64 && cppf_vote_batch_workgroups(2, 0) == 128 && cppf_vote_batch_workgroups(8, 0) == 64);
    const int64_t lens[4] = {524288, 524288, 524288, 524288};
    int per_xcd = -1, grid = 0, wg_begin[5];
    REQUIRE(cppf_pair_mlp_batch_plan(4, lens, &per_xcd, &grid, wg_begin) == 0 && per_xcd == 2 && grid % 8 == 0);
    /* argument checks come before anything touches a device */
    REQUIRE(cppf_vote_argmax(NULL, NULL, NULL, NULL, 0, NULL, NULL, RES, 1, 1, ROTS, 4, 4, 4, 1, 0, NULL, NULL, NULL, 0, NULL) == CPPF_EINVAL);
    REQUIRE(cppf_sample_pairs(NULL, NULL, NULL, 16, 8, NULL, 1ull, NULL, NULL) == CPPF_EINVAL);
    printf("host ok: ABI %d\n", cppf_abi_version());
    return 0;
}

static int on_device(void)
{
    static float pc[3 * N], outputs[2 * P], centre[3];
    static int32_t idx[2 * P];
    make_object(pc, idx, outputs, centre);
    float corners[6]; int32_t dims[3];
    REQUIRE(cppf_host_grid_shape(pc, N, RES, corners, dims) == 0);
    const int gx = dims[0], gy = dims[1], gz = dims[2];
    const size_t cells = (size_t)gx * gy * gz;

    float *d_pc, *d_out, *d_grid, *d_corner, *d_val;
    int32_t* d_idx;
    long long* d_arg;
    void* d_ws;
    const size_t need = cppf_vote_workspace_bytes(P, ROTS, gx, gy, gz), init = cppf_vote_workspace_init_bytes();
    REQUIRE(need > 0);
    HIP(hipSetDevice(0));
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    HIP(hipMalloc((void**)&d_pc, sizeof pc)); HIP(hipMalloc((void**)&d_out, sizeof outputs)); HIP(hipMalloc((void**)&d_idx, sizeof idx));
    HIP(hipMalloc((void**)&d_grid, cells * 4)); HIP(hipMalloc((void**)&d_corner, 12)); HIP(hipMalloc((void**)&d_val, 4));
    HIP(hipMalloc((void**)&d_arg, 8)); HIP(hipMalloc(&d_ws, need));
    HIP(hipMemcpyAsync(d_pc, pc, sizeof pc, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_out, outputs, sizeof outputs, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_idx, idx, sizeof idx, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_corner, corners, 12, hipMemcpyHostToDevice, st));
    HIP(hipMemsetAsync(d_ws, 0, init < need ? init : need, st));         /* once per allocation (include/cppf.h, the workspace contract) */

    float* grid = (float*)malloc(cells * 4);
    float* ref = (float*)calloc(cells, 4);
    REQUIRE(grid && ref);
    long long arg = -2; float val = 0.f;
    for (int rep = 0; rep < 3; ++rep) {                                  /* the workspace is left ready for the next call */
        const int rc = cppf_vote_argmax(d_pc, d_out, NULL, d_idx, /*idx_is_i64*/ 0, d_grid, d_corner, RES, N, P, ROTS, gx, gy, gz,
                                        /*adaptive*/ 1, /*accumulate*/ 0, d_arg, d_val, d_ws, need, st);
        if (rc) { fprintf(stderr, "cppf_vote_argmax: %s\n", cppf_error_string(rc)); return 1; }
    }
    HIP(hipMemcpyAsync(grid, d_grid, cells * 4, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(&arg, d_arg, 8, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(&val, d_val, 4, hipMemcpyDeviceToHost, st));
    HIP(hipStreamSynchronize(st));

    int64_t n_atomics = 0;
    static float ones[N];                                                 /* probs: all ones, what every caller in the reference passes */
    for (int i = 0; i < N; ++i) ones[i] = 1.f;                            /* (nocs/inference.py:201; the product takes NULL for that) */
    orc_ppf_voting(pc, outputs, ones, idx, ref, corners, RES, P, ROTS, gx, gy, gz, 1, &n_atomics);
    float ref_val = 0.f;
    const int64_t ref_arg = orc_grid_argmax(ref, (int64_t)cells, &ref_val);
    double worst = 0., mass = 0., ref_mass = 0.;
    for (size_t i = 0; i < cells; ++i) {
        const double d = fabs((double)grid[i] - (double)ref[i]);
        worst = d > worst ? d : worst;
        mass += grid[i]; ref_mass += ref[i];
    }
    double T[3];
    orc_center_from_argmax(arg, gy, gz, corners, (double)RES, T);
    printf("grid %dx%dx%d  P=%d  arg-max %lld (oracle %lld)  peak %.4f (oracle %.4f)  max |cell diff| %.3g  mass %.3f (oracle %.3f)  "
           "T = (%.4f %.4f %.4f)  centre (%.4f %.4f %.4f)\n", gx, gy, gz, (int)P, arg, (long long)ref_arg, val, ref_val, worst, mass,
           ref_mass, T[0], T[1], T[2], centre[0], centre[1], centre[2]);
    REQUIRE(arg == (long long)ref_arg);                                   /* nocs/inference.py:208 */
    REQUIRE(fabs(val - ref_val) <= 2e-5 * ref_val);                       /* the fp32 sums' own rounding */
    REQUIRE(val == grid[arg]);
    REQUIRE(worst <= 1e-4 * ref_val);
    REQUIRE(fabs(mass - ref_mass) <= 1e-5 * ref_mass);
    for (int c = 0; c < 3; ++c) REQUIRE(fabs(T[c] - centre[c]) <= 1.5 * RES);   /* the vote found the object */
    printf("device ok\n");
    free(grid); free(ref);
    hipFree(d_pc); hipFree(d_out); hipFree(d_idx); hipFree(d_grid); hipFree(d_corner); hipFree(d_val); hipFree(d_arg); hipFree(d_ws);
    hipStreamDestroy(st);
    return 0;
}


/* train.py:35 / config: ppffcs [84, 32, 32, 16], 141 logits = 2 x 32 centre bins + 2 x 36 orientation bins + 2 + 3 */
enum { F = 40, N_RES = 3, OUT_DIM = 141, TR_BINS = 32, ROT_BINS = 36 };
static const int DIMS[N_RES + 1] = {2 * F + 4, 32, 32, 16};
static const float VR0 = 0.25f, VR1 = 0.25f;                /* config/category/bottle.yaml: vote_range */

static int chain_on_device(void)
{
    static float pc[3 * N], nrm[3 * N], feat[N * F], unused_outputs[2 * P], centre[3], u_tr[2 * P], outputs[2 * P], ref_out[2 * P];
    static int32_t idx32[2 * P];
    static int64_t idx64[2 * P];
    make_object(pc, idx32, unused_outputs, centre);
    for (int i = 0; i < N; ++i) {                           /* normals: radial (the surface of revolution's), unit length */
        const float x = pc[3 * i] - centre[0], z = pc[3 * i + 2] - centre[2], r = sqrtf(x * x + z * z) + 1e-12f;
        nrm[3 * i] = x / r; nrm[3 * i + 1] = 0.f; nrm[3 * i + 2] = z / r;
    }
    for (int i = 0; i < N * F; ++i) feat[i] = 2.f * unif() - 1.f;
    for (int i = 0; i < 2 * P; ++i) { idx64[i] = idx32[i]; u_tr[i] = unif(); }
    /* parameters in the oracle's / cppf_pair_mlp_pack's layout: flat torch tensors + offset table (include/cppf.h) */
    static float params[65536];
    int64_t offs[6 * N_RES + 2], at = 0;
    for (int l = 0; l < N_RES; ++l) {
        const int Kd = DIMS[l], Nn = DIMS[l + 1];
        const int sizes[6] = {Nn * Kd, Nn, Nn * Nn, Nn, Kd != Nn ? Nn * Kd : 0, Kd != Nn ? Nn : 0};
        for (int t = 0; t < 6; ++t) {
            offs[6 * l + t] = sizes[t] ? at : -1;
            const float bound = 1.f / sqrtf((float)(t == 2 || t == 3 ? Nn : Kd));       /* torch.nn.Linear's default init */
            for (int i = 0; i < sizes[t]; ++i) params[at++] = bound * (2.f * unif() - 1.f);
        }
    }
    offs[6 * N_RES] = at;
    for (int i = 0; i < OUT_DIM * DIMS[N_RES]; ++i) params[at++] = 0.25f * (2.f * unif() - 1.f);
    offs[6 * N_RES + 1] = at;
    for (int i = 0; i < OUT_DIM; ++i) params[at++] = 0.25f * (2.f * unif() - 1.f);
    REQUIRE(at <= (int64_t)(sizeof params / sizeof params[0]));

    const size_t n_packed = cppf_pair_mlp_packed_floats(F, DIMS, N_RES, OUT_DIM);
    const size_t mlp_ws = cppf_pair_mlp_workspace_bytes(N, F, DIMS, N_RES, OUT_DIM);
    REQUIRE(n_packed > 0 && mlp_ws > 0);
    float* packed = (float*)malloc(n_packed * 4);
    REQUIRE(packed && cppf_pair_mlp_pack(params, offs, F, DIMS, N_RES, OUT_DIM, packed) == 0);

    float corners[6]; int32_t dims[3];
    REQUIRE(cppf_host_grid_shape(pc, N, RES, corners, dims) == 0);
    const int gx = dims[0], gy = dims[1], gz = dims[2];
    const size_t cells = (size_t)gx * gy * gz;
    const size_t need = cppf_vote_workspace_bytes(P, ROTS, gx, gy, gz), init = cppf_vote_workspace_init_bytes();

    float *d_pc, *d_nrm, *d_feat, *d_packed, *d_u, *d_out, *d_grid, *d_corner, *d_val;
    int64_t* d_idx;
    long long* d_arg;
    void *d_ws, *d_mlp_ws;
    HIP(hipSetDevice(0));
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    HIP(hipMalloc((void**)&d_pc, sizeof pc)); HIP(hipMalloc((void**)&d_nrm, sizeof nrm)); HIP(hipMalloc((void**)&d_feat, sizeof feat));
    HIP(hipMalloc((void**)&d_packed, n_packed * 4)); HIP(hipMalloc((void**)&d_u, sizeof u_tr)); HIP(hipMalloc((void**)&d_out, sizeof outputs));
    HIP(hipMalloc((void**)&d_idx, sizeof idx64)); HIP(hipMalloc((void**)&d_grid, cells * 4)); HIP(hipMalloc((void**)&d_corner, 12));
    HIP(hipMalloc((void**)&d_val, 4)); HIP(hipMalloc((void**)&d_arg, 8)); HIP(hipMalloc(&d_ws, need)); HIP(hipMalloc(&d_mlp_ws, mlp_ws));
    HIP(hipMemcpyAsync(d_pc, pc, sizeof pc, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_nrm, nrm, sizeof nrm, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_feat, feat, sizeof feat, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_packed, packed, n_packed * 4, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_u, u_tr, sizeof u_tr, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_idx, idx64, sizeof idx64, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_corner, corners, 12, hipMemcpyHostToDevice, st));
    HIP(hipMemsetAsync(d_ws, 0, init < need ? init : need, st));
    long long arg = -2; float val = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        int rc = cppf_pair_mlp_decode(d_pc, d_nrm, d_feat, d_idx, /*idx_is_i64*/ 1, d_packed, N, F, DIMS, N_RES, P, OUT_DIM, TR_BINS,
                                      ROT_BINS, VR0, VR1, d_u, NULL, d_out, NULL, d_mlp_ws, mlp_ws, st);
        if (rc) { fprintf(stderr, "cppf_pair_mlp_decode: %s\n", cppf_error_string(rc)); return 1; }
        rc = cppf_vote_argmax(d_pc, d_out, NULL, d_idx, /*idx_is_i64*/ 1, d_grid, d_corner, RES, N, P, ROTS, gx, gy, gz, 1, 0, d_arg, d_val,
                              d_ws, need, st);
        if (rc) { fprintf(stderr, "cppf_vote_argmax: %s\n", cppf_error_string(rc)); return 1; }
    }
    float* grid = (float*)malloc(cells * 4);
    float* ref = (float*)calloc(cells, 4);
    float* logits = (float*)malloc((size_t)P * OUT_DIM * 4);
    REQUIRE(grid && ref && logits);
    HIP(hipMemcpyAsync(outputs, d_out, sizeof outputs, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(grid, d_grid, cells * 4, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(&arg, d_arg, 8, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(&val, d_val, 4, hipMemcpyDeviceToHost, st));
    HIP(hipStreamSynchronize(st));

    REQUIRE(orc_pair_mlp(pc, nrm, feat, idx64, N, F, P, params, offs, DIMS, N_RES, OUT_DIM, /*order*/ 1, logits) == 0);
    orc_decode_center(logits, P, OUT_DIM, TR_BINS, u_tr, VR0, VR1, ref_out, NULL);
    int64_t differing = 0;
    for (int i = 0; i < 2 * P; ++i) differing += memcmp(&outputs[i], &ref_out[i], 4) != 0;
    static float ones[N];
    for (int i = 0; i < N; ++i) ones[i] = 1.f;
    int64_t n_atomics = 0;
    orc_ppf_voting(pc, ref_out, ones, idx32, ref, corners, RES, P, ROTS, gx, gy, gz, 1, &n_atomics);
    float ref_val = 0.f;
    const int64_t ref_arg = orc_grid_argmax(ref, (int64_t)cells, &ref_val);
    double worst = 0.;
    for (size_t i = 0; i < cells; ++i) { const double d = fabs((double)grid[i] - (double)ref[i]); worst = d > worst ? d : worst; }
    printf("chain: P=%d pairs, %lld of %d (mu, nu) values differ from the oracle's; grid %dx%dx%d, %lld samples landed; arg-max %lld "
           "(oracle %lld) peak %.5f (oracle %.5f) max |cell diff| %.3g\n", (int)P, (long long)differing, 2 * (int)P, gx, gy, gz,
           (long long)(n_atomics / 8), arg, (long long)ref_arg, val, ref_val, worst);
    REQUIRE(differing == 0);                                              /* PPF + MLP + decode: bit for bit */
    REQUIRE(arg >= 0 && (size_t)arg < cells && val == grid[arg]);
    REQUIRE(worst <= 1e-4 * ref_val + 1e-5);
    /* nocs/inference.py:208: the same cell -- or, when two cells tie within the fp32 sums' own rounding, one of the tied ones */
    REQUIRE(arg == (long long)ref_arg || ref[arg] >= ref_val - 2e-5f * ref_val);
    printf("chain ok\n");
    free(grid); free(ref); free(logits); free(packed);
    hipFree(d_pc); hipFree(d_nrm); hipFree(d_feat); hipFree(d_packed); hipFree(d_u); hipFree(d_out); hipFree(d_idx); hipFree(d_grid);
    hipFree(d_corner); hipFree(d_val); hipFree(d_arg); hipFree(d_ws); hipFree(d_mlp_ws);
    hipStreamDestroy(st);
    return 0;
}

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) return on_device();
    if (argc > 1 && strcmp(argv[1], "chain") == 0) return chain_on_device();
    return host_only();
}
