/*
 * CPU restatement of the per-instance pre-processing in front of the CPPF path.
 *
 * TEST INFRASTRUCTURE ONLY (see cppf_oracle.c).
 *
 * PARITY UNPINNED against the reference: both steps are third-party calls whose libraries are absent here and
 * whose results are not fully specified --
 *   voxel de-duplication   ME.utils.sparse_quantize(pc, return_index=True, quantization_size=res)
 *                          (nocs/inference.py:140; MinkowskiEngine, README.md:74): one representative index per
 *                          occupied voxel floor(p / res); WHICH point represents a voxel is hash-order dependent.
 *                          Here: the lowest original index, output sorted by index.
 *   normals                open3d estimate_normals(KDTreeSearchParamKNN(knn)) (utils/util.py:61-65): the
 *                          eigenvector of the smallest eigenvalue of the covariance of the knn nearest neighbours
 *                          (the point itself included); open3d leaves the SIGN unspecified.
 *                          Here: covariance from fp64 cumulants like open3d (E[xx] - E[x]E[x]), cyclic Jacobi (8 sweeps,
 *                          fp64), sign chosen so that the component of largest magnitude is positive.
 * What IS pinned: these definitions against numpy (np.unique on the voxel keys, np.linalg.eigh up to sign) in
 * tests/test_oracle_golden.py, and the HIP kernels against this file bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t key; int32_t idx; } vox_ent;
static int vox_cmp(const void* a, const void* b)
{
    const vox_ent *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static int i32_cmp(const void* a, const void* b) { return (*(const int32_t*)a > *(const int32_t*)b) - (*(const int32_t*)a < *(const int32_t*)b); }

/* voxel coordinate: floor((double)p / (double)res) (numpy float64 semantics of the scripts: pc is float64 there,
 * nocs/inference.py:131-140); packed 21 bits per axis with a 2^20 bias */
static int64_t vox_key(const float* p, double res)
{
    int64_t k = 0;
    for (int c = 0; c < 3; ++c) {
        const int64_t v = (int64_t)floor((double)p[c] / res) + (1 << 20);
        k = (k << 21) | (v & ((1 << 21) - 1));
    }
    return k;
}

/* keep[] = lowest index of every occupied voxel, ascending; returns the count */
int64_t orc_voxel_dedupe(const float* pc, int64_t N, double res, int32_t* keep)
{
    vox_ent* e = malloc(sizeof(vox_ent) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < N; ++i) { e[i].key = vox_key(pc + 3 * i, res); e[i].idx = (int32_t)i; }
    qsort(e, (size_t)N, sizeof(vox_ent), vox_cmp);
    int64_t n = 0;
    for (int64_t i = 0; i < N; ++i)
        if (i == 0 || e[i].key != e[i - 1].key) keep[n++] = e[i].idx;
    qsort(keep, (size_t)n, sizeof(int32_t), i32_cmp);
    free(e);
    return n;
}

/* smallest-eigenvalue eigenvector of the symmetric 3x3 matrix (a00 a01 a02 / a11 a12 / a22): cyclic Jacobi */
static void smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double* out)
{
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 8; ++sweep)
        for (int r = 0; r < 3; ++r) {
            const int p = PQ[r][0], q = PQ[r][1];
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) {   /* A <- A J */
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {   /* A <- J^T A */
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {   /* V <- V J */
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    double v[3] = {V[0][m], V[1][m], V[2][m]};
    const double n = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (!(n > 0.0)) { out[0] = 0; out[1] = 0; out[2] = 1; return; }
    int big = 0;
    if (fabs(v[1]) > fabs(v[big])) big = 1;
    if (fabs(v[2]) > fabs(v[big])) big = 2;
    const double sg = v[big] < 0.0 ? -1.0 : 1.0;
    for (int c = 0; c < 3; ++c) out[c] = sg * (v[c] / n);
}

/* normals[n] (fp32) from the k neighbours nbrs[n][0..k) (the point itself is one of them) */
void orc_estimate_normals(const float* pc, const int32_t* nbrs, int64_t N, int k, float* normals)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < k; ++j) {
            const float* p = pc + 3 * (int64_t)nbrs[n * k + j];
            const double x = p[0], y = p[1], z = p[2];
            c[0] += x; c[1] += y; c[2] += z;
            c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
        }
        for (int i = 0; i < 9; ++i) c[i] /= (double)k;
        double v[3];
        smallest_eigvec(c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2], c[6] - c[1] * c[1], c[7] - c[1] * c[2],
                        c[8] - c[2] * c[2], v);
        for (int i = 0; i < 3; ++i) normals[3 * n + i] = (float)v[i];
    }
}

/* utils/util.py:598-631 backproject: pixels with mask != 0 and depth > 0 in row-major order (np.where); xyz = inv(K) @ (u, v, 1)
 * with each 3-term product as k0*u, fma(k1, v, .), + k2; pts = xyz * z / xyz.z; x and y negated.  depth as double (exact for
 * u16 and f32 inputs).  Returns the number of points; pix[i] = v*W + u. */
int64_t orc_backproject(const double* depth, const uint8_t* mask, int H, int W, const double* kinv, double* pts, int32_t* pix)
{
    int64_t n = 0;
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const int64_t p = (int64_t)v * W + u;
            if (!mask[p] || !(depth[p] > 0.0)) continue;
            double xyz[3];
            for (int c = 0; c < 3; ++c) xyz[c] = fma(kinv[3 * c + 1], (double)v, kinv[3 * c] * (double)u) + kinv[3 * c + 2];
            const double z = depth[p];
            pts[3 * n] = -(xyz[0] * z / xyz[2]);
            pts[3 * n + 1] = -(xyz[1] * z / xyz[2]);
            pts[3 * n + 2] = xyz[2] * z / xyz[2];
            pix[n] = (int32_t)p;
            ++n;
        }
    return n;
}
