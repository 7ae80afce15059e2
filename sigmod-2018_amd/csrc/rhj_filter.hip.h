// rhj_filter.hip.h — filter scan: predicate masks, index list
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
#pragma once
#include "rhj_common.hip.h"

namespace rhj {

// --------------------------------------------------------------------- filter

constexpr int FILTER_ROUNDS = 8;                     // rounds of 128 elements (two per lane) per wave
constexpr int FILTER_WAVE_ELEMS = FILTER_ROUNDS * 2 * WAVE;      // 1024
constexpr int FILTER_TILE = 256 / WAVE * FILTER_WAVE_ELEMS;      // 4096 elements per workgroup

__device__ __forceinline__ bool filter_pred(uint64_t v, uint64_t k, int op)
{
    return op == 0 ? v < k : op == 1 ? v > k : v == k;
}

// Pass 1: evaluate the predicate once.  A lane takes two consecutive elements per round
// (one 16-byte load); the round's result is kept as two 64-bit ballot masks (even / odd
// elements), and hits are counted per 4096-element tile.
__global__ __launch_bounds__(256) void k_filter_mask(const uint64_t *col, const uint64_t *sel, uint64_t n, int op,
                                                     uint64_t value, uint64_t *masks, uint64_t *tile_count)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t wbase = (uint64_t)blockIdx.x * FILTER_TILE + (uint64_t)w * FILTER_WAVE_ELEMS;
    uint64_t v0[FILTER_ROUNDS], v1[FILTER_ROUNDS];
    const bool fast = wbase + FILTER_WAVE_ELEMS <= n;          // whole wave range in bounds
#pragma unroll
    for (int k = 0; k < FILTER_ROUNDS; ++k) {
        const uint64_t i = wbase + (uint64_t)k * 2 * WAVE + 2 * lane;
        v0[k] = v1[k] = 0;
        if (fast) {
            const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>((sel ? sel : col) + i);
            if (sel) { v0[k] = col[x.x]; v1[k] = col[x.y]; } else { v0[k] = x.x; v1[k] = x.y; }
        } else {
            if (i < n) v0[k] = sel ? col[sel[i]] : col[i];
            if (i + 1 < n) v1[k] = sel ? col[sel[i + 1]] : col[i + 1];
        }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < FILTER_ROUNDS; ++k) {
        const uint64_t i = wbase + (uint64_t)k * 2 * WAVE + 2 * lane;
        const uint64_t me = __ballot(i < n && filter_pred(v0[k], value, op));
        const uint64_t mo = __ballot(i + 1 < n && filter_pred(v1[k], value, op));
        if (lane == 0 && wbase + (uint64_t)k * 2 * WAVE < n) {
            masks[(wbase >> 6) + 2 * k] = me;
            masks[(wbase >> 6) + 2 * k + 1] = mo;
        }
        cnt += (uint32_t)__popcll(me) + (uint32_t)__popcll(mo);
    }
    if (lane == 0) wsum[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = (uint64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Two-column equality (SelfJoin / JoinInterNode, inter_res.c:234-263, :363-389): same mask layout as
// k_filter_mask, predicate colA[selA ? selA[i] : i] == colB[selB ? selB[i] : i].
__global__ __launch_bounds__(256) void k_filter_mask_eq2(const uint64_t *colA, const uint64_t *selA, const uint64_t *colB,
                                                         const uint64_t *selB, uint64_t n, uint64_t *masks, uint64_t *tile_count)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t wbase = (uint64_t)blockIdx.x * FILTER_TILE + (uint64_t)w * FILTER_WAVE_ELEMS;
    uint64_t a0[FILTER_ROUNDS], a1[FILTER_ROUNDS], b0[FILTER_ROUNDS], b1[FILTER_ROUNDS];
#pragma unroll
    for (int k = 0; k < FILTER_ROUNDS; ++k) {
        const uint64_t i = wbase + (uint64_t)k * 2 * WAVE + 2 * lane;
        a0[k] = a1[k] = 0; b0[k] = b1[k] = 1;
        if (i < n)     { a0[k] = colA[selA ? selA[i] : i];         b0[k] = colB[selB ? selB[i] : i]; }
        if (i + 1 < n) { a1[k] = colA[selA ? selA[i + 1] : i + 1]; b1[k] = colB[selB ? selB[i + 1] : i + 1]; }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < FILTER_ROUNDS; ++k) {
        const uint64_t i = wbase + (uint64_t)k * 2 * WAVE + 2 * lane;
        const uint64_t me = __ballot(i < n && a0[k] == b0[k]);
        const uint64_t mo = __ballot(i + 1 < n && a1[k] == b1[k]);
        if (lane == 0 && wbase + (uint64_t)k * 2 * WAVE < n) {
            masks[(wbase >> 6) + 2 * k] = me;
            masks[(wbase >> 6) + 2 * k + 1] = mo;
        }
        cnt += (uint32_t)__popcll(me) + (uint32_t)__popcll(mo);
    }
    if (lane == 0) wsum[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = (uint64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Pass 2: turn the masks into the ascending index list.  One WAVE per pair of 4096-element tiles, grid-stride,
// no barrier: lane l owns the two mask words (even / odd elements) of one 128-element round, 32 rounds per tile.
//   few hits (<= FILTER_SPARSE per two tiles): every lane walks its own set bits in element order and stores them
//     behind its exclusive prefix — the work is proportional to the hits, a tile pair without any costs one load;
//   many hits: the wave goes through its 64 rounds one by one (the owning lane's masks and start are broadcast),
//     so that its stores are coalesced.
// (One workgroup per tile with the slice code alone: 98 us for 400 M rows without a hit — workgroup dispatch —
// and 216 us at 1 % selectivity — 450 vector instructions per slice whatever the number of hits.)
constexpr uint32_t FILTER_SPARSE = 1024;

// SELF (at most FILTER_SELF_TILES tiles: the filters of a query workload): no scan launches — tile_base holds the tiles' COUNTS and
// every wave sums the counts in front of its tiles itself (L2-resident, sixteen loads a lane at most); the wave of the last
// task leaves the hit total in pinned host memory: mask + write are the filter's two launches, nothing is copied back.
constexpr uint64_t FILTER_SELF_TILES = 1024;
template <bool SELF>
__global__ __launch_bounds__(256) void k_filter_write(uint64_t n, const uint64_t *masks, const uint64_t *tile_base,
                                                      uint64_t *out, unsigned long long *h_total)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t ntiles = (n + FILTER_TILE - 1) / FILTER_TILE;
    const uint64_t ntasks = (ntiles + 1) / 2;
    const uint64_t stride = (uint64_t)gridDim.x * (256 / WAVE);
    const uint64_t lt = lanemask_lt();
    for (uint64_t task = (uint64_t)blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6); task < ntasks; task += stride) {
        const uint64_t tile = 2 * task + (lane >> 5);
        const uint64_t ebase = tile * FILTER_TILE + (uint64_t)(lane & 31u) * (2 * WAVE);    // first element of this lane's round
        uint64_t me = 0, mo = 0;
        if (ebase < n) {
            const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(masks + (ebase >> 6));
            me = x.x; mo = x.y;
        }
        uint64_t tb = 0;
        if (SELF) {
            uint64_t front = 0;                       // hits of the tiles in front of this pair
            for (uint64_t t = lane; t < 2 * task; t += WAVE) front += tile_base[t];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) front += __shfl_xor(front, d, 64);
            const uint64_t c0 = 2 * task < ntiles ? tile_base[2 * task] : 0, c1 = 2 * task + 1 < ntiles ? tile_base[2 * task + 1] : 0;
            tb = front + (lane >= 32 ? c0 : 0);
            if (task == ntasks - 1 && lane == 0) __hip_atomic_store(h_total, front + c0 + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else
            tb = tile < ntiles ? tile_base[tile] : 0;
        const uint32_t pc = (uint32_t)__popcll(me) + (uint32_t)__popcll(mo);
        if (__ballot(pc != 0) == 0) continue;
        uint32_t incl = pc;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += y;
        }
        const uint32_t t0 = __shfl(incl, 31, 64), t2 = __shfl(incl, 63, 64);
        if (t2 <= FILTER_SPARSE) {
            uint64_t pos = tb + (incl - pc) - (lane >= 32 ? t0 : 0u);
            while (me | mo) {                          // ascending: element 2b of the even word, 2b + 1 of the odd one
                const uint32_t be = me ? (uint32_t)__builtin_ctzll(me) : 64u, bo = mo ? (uint32_t)__builtin_ctzll(mo) : 64u;
                const bool odd = bo < be;
                out[pos++] = ebase + 2u * (odd ? bo : be) + (odd ? 1u : 0u);
                if (odd) mo &= mo - 1; else me &= me - 1;
            }
        } else {
            // many hits: round by round (lane r's masks and start broadcast to the wave), coalesced stores
            const uint64_t start = tb + (incl - pc) - (lane >= 32 ? t0 : 0u);
            const uint32_t melo = (uint32_t)me, mehi = (uint32_t)(me >> 32), molo = (uint32_t)mo, mohi = (uint32_t)(mo >> 32);
            const uint32_t stlo = (uint32_t)start, sthi = (uint32_t)(start >> 32);
            for (uint32_t r = 0; r < WAVE; ++r) {
                const uint64_t mer = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mehi, (int)r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)melo, (int)r);
                const uint64_t mor = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mohi, (int)r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)molo, (int)r);
                if ((mer | mor) == 0) continue;
                const uint64_t st = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)sthi, (int)r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)stlo, (int)r);
                const uint32_t before = (uint32_t)__popcll(mer & lt) + (uint32_t)__popcll(mor & lt);
                const uint64_t i = task * (2 * FILTER_TILE) + (uint64_t)r * (2 * WAVE) + 2 * lane;
                const uint32_t e = (uint32_t)((mer >> lane) & 1ull);
                if (e) out[st + before] = i;
                if ((mor >> lane) & 1ull) out[st + before + e] = i + 1;
            }
        }
    }
}

}  // namespace rhj
