// Stream compaction pieces shared by pose_tail.hip (survivor lists) and preproc.hip (the frame stage): chunks of CMP_BLOCK mask bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CMP_BLOCK 1024
#define CMP_SELF_MAX 8192

// The scatter step alone, for chunk counts that already exist (the kernel that wrote the mask counted its chunks): every block sums
// the counts of the chunks before its own (<= CMP_SELF_MAX of them, from L2) instead of waiting for a scan kernel.  surv = the indices
// of the non-zero mask bytes in increasing order, *total = their number.  Launch: >= ceil(n / CMP_BLOCK) blocks of CMP_BLOCK threads.
__device__ __forceinline__ void compact_scatter_self_body(const uint8_t* __restrict__ mask, int64_t n,
                                                          const int32_t* __restrict__ chunk_counts,
                                                          int32_t* __restrict__ surv, int32_t* __restrict__ total)
{
    const unsigned nblocks = (unsigned)((n + CMP_BLOCK - 1) / CMP_BLOCK);   // (a batched launch is as wide as its longest list)
    if (blockIdx.x >= nblocks) return;
    __shared__ int wsum[CMP_BLOCK / 64];
    __shared__ int wpre[CMP_BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int before = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += CMP_BLOCK) before += chunk_counts[k];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    const int64_t i = (int64_t)blockIdx.x * CMP_BLOCK + threadIdx.x;
    const bool f = i < n && mask[i] != 0;
    const unsigned long long b = __ballot(f);
    if (lane == 0) { wsum[w] = __popcll(b); wpre[w] = before; }
    __syncthreads();
    int woff = 0, base = 0;
    for (int k = 0; k < CMP_BLOCK / 64; ++k) { base += wpre[k]; woff += k < w ? wsum[k] : 0; }
    if (f) surv[base + woff + __popcll(b & ((1ull << lane) - 1ull))] = (int32_t)i;
    if (blockIdx.x == nblocks - 1 && threadIdx.x == 0) {
        int own = 0;
        for (int k = 0; k < CMP_BLOCK / 64; ++k) own += wsum[k];
        *total = base + own;
    }
}

// a block of CMP_BLOCK threads that has just decided its mask byte: the chunk's count (thread 0 returns it; one barrier)
__device__ __forceinline__ int compact_chunk_count(bool f)
{
    __shared__ int csum[CMP_BLOCK / 64];
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) csum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    int s = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < CMP_BLOCK / 64; ++w) s += csum[w];
    return s;
}

// kNN of several clouds in one launch (sprin.hip), for the other translation units
struct CppfKnnBatchItem { const float* pc; int32_t* nbrs; const int32_t* n_dev; int n_cap; int k; };
int cppf_internal_knn_batch(int n_items, const CppfKnnBatchItem* items, void* stream);
