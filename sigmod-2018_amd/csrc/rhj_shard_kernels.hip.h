// rhj_shard_kernels.hip.h — the two device steps of bucket-range sharding: the bucket
// histogram of a relation (what HistJob computes per thread, preprocess.c:181-195, here for the ranks'
// load balance) and the stable selection of the tuples whose bucket lies in a rank's range.  Bucket b of
// R only ever meets bucket b of S (rhjoin.c:42-57), and a stable selection keeps the input order, so the
// join of the selected tuples is the canonical result restricted to those buckets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rhj.h"

namespace rhj {

constexpr int SH_BLOCK = 256;
constexpr int SH_TILE = 4096;                     // tuples per selection tile (16 per lane)

// hist[b] += tuples of bucket b; per-workgroup LDS histogram, one global atomic per non-empty bin
__global__ __launch_bounds__(SH_BLOCK) void k_bucket_hist(const rhj_tuple *in, uint64_t n, int bits, unsigned long long *hist)
{
    extern __shared__ uint32_t sh_hist[];
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    for (uint32_t b = threadIdx.x; b < bins; b += SH_BLOCK) sh_hist[b] = 0;
    __syncthreads();
    const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
    for (uint64_t i = (uint64_t)blockIdx.x * SH_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SH_BLOCK)
        atomicAdd(&sh_hist[in4[i].x & mask], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += SH_BLOCK)
        if (sh_hist[b]) atomicAdd(&hist[b], (unsigned long long)sh_hist[b]);
}

__device__ __forceinline__ bool sh_in_range(uint32_t klo, uint32_t mask, uint32_t lo, uint32_t hi)
{
    return ((klo & mask) - lo) < (hi - lo);       // lo <= bucket < hi, unsigned
}

__global__ __launch_bounds__(SH_BLOCK) void k_select_count(const rhj_tuple *in, uint64_t n, uint32_t mask, uint32_t lo, uint32_t hi,
                                                           uint64_t *tile_count)
{
    __shared__ uint32_t wsum[SH_BLOCK / 64];
    const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
    const uint64_t beg = (uint64_t)blockIdx.x * SH_TILE;
    uint32_t c = 0;
#pragma unroll 4
    for (int k = 0; k < SH_TILE / SH_BLOCK; ++k) {
        const uint64_t i = beg + (uint64_t)k * SH_BLOCK + threadIdx.x;
        c += (i < n && sh_in_range(in4[i].x, mask, lo, hi)) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = (uint64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// wave w of the tile owns 1024 consecutive tuples, 16 rounds of 64: rank = ballots below + earlier rounds + earlier waves
__global__ __launch_bounds__(SH_BLOCK) void k_select_write(const rhj_tuple *in, uint64_t n, uint32_t mask, uint32_t lo, uint32_t hi,
                                                           const uint64_t *tile_base, rhj_tuple *out, uint64_t capacity)
{
    __shared__ uint32_t wsum[SH_BLOCK / 64];
    const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
    uint4 *out4 = reinterpret_cast<uint4 *>(out);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t beg = (uint64_t)blockIdx.x * SH_TILE + (uint64_t)w * (SH_TILE / 4);
    constexpr int R = SH_TILE / 4 / 64;            // 16 rounds per wave
    uint4 t[R];
    uint64_t m[R];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint64_t i = beg + (uint64_t)k * 64 + lane;
        t[k] = make_uint4(0, 0, 0, 0);
        if (i < n) t[k] = in4[i];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint64_t i = beg + (uint64_t)k * 64 + lane;
        m[k] = __ballot(i < n && sh_in_range(t[k].x, mask, lo, hi));
        mine += (uint32_t)__popcll(m[k]);
    }
    if (lane == 0) wsum[w] = mine;
    __syncthreads();
    uint64_t at = tile_base[blockIdx.x];
    for (uint32_t i = 0; i < w; ++i) at += wsum[i];
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint64_t dst = at + (uint64_t)__popcll(m[k] & lt);
        if (((m[k] >> lane) & 1ull) && dst < capacity) out4[dst] = t[k];
        at += (uint64_t)__popcll(m[k]);
    }
}

}  // namespace rhj
