// rhj_common.hip.h — argument structs, wave/block scans and the small exclusive-scan kernels shared by every path
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rhj.h"

namespace rhj {

constexpr int WAVE = 64;

struct RelArgs {                 // one relation through one partition pass
    const rhj_tuple *in;
    rhj_tuple       *out;
    uint32_t        *cnt;        // [tiles][bins] counts, then (after scan) start offsets
    uint64_t         n;
    uint32_t         tiles;
    uint32_t         range_lo;   // sharded join: only buckets [range_lo, range_lo + range_span) of the join's radix are kept
    const uint8_t   *dig_in;     // pass 2 of the run form: this pass' digit per input tuple, written by pass 1
    uint8_t         *dig_out;    // pass 1 of the run form: the next pass' digit per output tuple
    // two-pass partition (run form): pass 1 partitions every tile in place and leaves a run table;
    // a pass-2 tile is `group` consecutive pass-1 tiles' runs of one pass-1 digit
    uint16_t        *runs;       // [bins1 + 1][tiles1] start of each digit's run inside its pass-1 tile; row bins1 = the tile's count
    uint32_t         tiles1;     // pass-1 tiles
    uint32_t         group;      // pass-1 tiles per pass-2 tile
    uint32_t         groups;     // pass-2 tiles per pass-1 digit = ceil(tiles1 / group); tiles = bins1 * groups in pass 2
    uint32_t         range_span; // (0: every bucket — the ordinary join)
    // pass 1 counting pass 2's digits itself (radix bits <= 12): a workgroup walks a STRIP of PT_STRIP pass-1 tiles of one
    // group and leaves its (pass-1 digit, pass-2 digit) counts, 16 bits each: part[d][strip][digit], strip = group * parts + p
    uint16_t        *part;
    uint32_t         parts;      // strips per group = ceil(group / strip)
    uint32_t         per;        // groups per scan slice = ceil(groups / FH_SLICES)
    uint32_t         strip;      // pass-1 tiles a strip (1..PT_STRIP: fewer for small relations, whose strips would not fill the chip)
    uint32_t         range_bits; // the low key bits range_lo / range_span count in (0: this partition's whole radix; the low-radix path: the caller's bits)
    // pass 2's start offsets come in two parts: cnt[tile (d, j)][digit] = the tuples of (d, digit) in groups of j's slice in
    // front of j, sbase[d][slice][digit] = where that slice starts in the output (bucket start + the slices in front of it)
    uint32_t        *sbase;
};

struct Unit {
    uint64_t off;                // offset inside the bucket's probe (or build) side
    uint32_t bucket;
    uint32_t count;
};

struct BucketMeta {
    uint64_t table_off;          // first slot in the 32-bit or 64-bit table arena
    uint32_t slots;              // 32-bit table: slot count; 64-bit table: log2(slot count)
    uint32_t mode;               // 0 inactive, 1 32-bit table (LDS-built), 2 64-bit table
};

struct PlanSummary {
    uint64_t units;              // probe units
    uint64_t build_units;        // 64-bit-table build chunks
    uint64_t hbm_slots;          // total slots of all 64-bit tables
    uint64_t lds_buckets;        // buckets with an LDS-built 32-bit table
    uint64_t tab32_slots;        // total slots of all 32-bit tables
    uint64_t max_lds_slots;      // largest 32-bit table
    uint64_t max_build;          // largest build side
    uint64_t matches;            // filled by k_offsets / the fused kernel's last workgroup out
    uint64_t fused_ok;           // every active bucket's build side <= lds_cap (fused path usable)
    uint32_t wide_row_ids;       // two-pass partition: 1 = the intermediate array keeps 16-byte tuples, 0 = 12-byte
    uint32_t row_id_overflow;    // a row id above 2^32 - 1 went through a 12-byte intermediate: run again wide
};

struct JoinArgs {
    const rhj_tuple *partR, *partS;
    const uint64_t  *histR, *histS, *psumR, *psumS;   // [bins]
    const Unit      *units;
    const BucketMeta*meta;
    const PlanSummary *summary;
    uint32_t        *tab32;          // 32-bit table arena
    uint64_t        *tab64;          // 64-bit table arena
    uint64_t        *unit_count;     // [units] matches per unit (count pass)
    uint32_t        *unit_flag;      // [units] 1 = a tag-matching candidate failed verification
    const uint64_t  *unit_base;      // [units] exclusive scan of unit_count
    rhj_result_tuple*out;
    uint64_t         out_capacity;
    uint32_t         ablate;         // timing experiments only (RHJ_ABLATE): 1 no gathers, 2 no table reads
    uint32_t         parent_mask;    // low-radix path: bucket & parent_mask = the bucket of the CALLER's radix this sub-bucket belongs to
    const uint8_t   *parent_flip;    // low-radix path: [parent buckets] 1 = S probes; null: every bucket chooses for itself (rhjoin.c:86)
    // tiled path: what the count pass learnt per probe tuple, indexed like the partitioned relations (S behind R): the emit
    // pass takes tuples with at most one match from here and goes back to the table only for the others
    uint8_t         *stash_cnt;      // matches, saturating at 255
    uint64_t        *stash_row;      // build row id of the first match
    uint64_t         stash_nR;       // S's tuples start here
};

// Which side of bucket b is streamed (probes): R when histR >= histS, rhjoin.c:86 — decided by the bucket itself, or, when a
// join on few radix bits is run on finer sub-buckets (rhj_lowradix.hip.h), by the caller's bucket the sub-bucket belongs to.
__device__ __forceinline__ bool bucket_flip(const JoinArgs &a, uint32_t b, uint64_t cR, uint64_t cS)
{
    return a.parent_flip ? a.parent_flip[b & a.parent_mask] != 0 : cR < cS;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// Inclusive scan over the 64 lanes with DPP moves (row shifts inside the 16-lane rows, then the row totals broadcast
// to the rows behind them): six register-to-register steps of a few cycles each, where the shuffle form went through
// the LDS crossbar six times in a row (~100 cycles each) — these scans sit in the latency chains of every kernel here.
// (Lanes a shift has no source for keep the 0 handed in as `old`; row_bcast:15 / :31 write rows 1,3 / 2,3 only.)
#ifndef RHJ_SHFL_SCAN
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);    // row_bcast:15
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);    // row_bcast:31
    return x;
}
#else
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}
#endif

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t x)
{
#ifndef RHJ_SHFL_SCAN
#define RHJ_DPP64(ctrl, rows)                                                                                   \
    x += ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), ctrl, rows, 0xf, false) << 32) | \
         (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, ctrl, rows, 0xf, false)
    RHJ_DPP64(0x111, 0xf); RHJ_DPP64(0x112, 0xf); RHJ_DPP64(0x114, 0xf); RHJ_DPP64(0x118, 0xf);
    RHJ_DPP64(0x142, 0xa); RHJ_DPP64(0x143, 0xc);
#undef RHJ_DPP64
#else
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
#endif
    return x;
}

__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t *total)
{
    const uint32_t x = wave_incl_scan_u32(v);
    *total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}

template <int NT>
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t *total, uint64_t *sm /*NT/64+1*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t x = wave_incl_scan_u64(v);
    __syncthreads();                       // sm reuse across calls
    if (lane == 63) sm[w] = x;
    __syncthreads();
    if (threadIdx.x < 64) {                // one wave scans the wave totals
        const uint64_t t = threadIdx.x < NT / 64 ? sm[threadIdx.x] : 0;
        const uint64_t y = wave_incl_scan_u64(t);
        if (threadIdx.x < NT / 64) sm[threadIdx.x] = y - t;
        if (threadIdx.x == NT / 64 - 1) sm[NT / 64] = y;
    }
    __syncthreads();
    if (total) *total = sm[NT / 64];
    return sm[w] + x - v;
}

// N values a thread through the same three barriers (the plan scans five: one after the other they cost it 15 barriers)
template <int NT, int N>
__device__ __forceinline__ void block_excl_scan_n(const uint64_t (&v)[N], uint64_t (&excl)[N], uint64_t (&total)[N],
                                                  uint64_t *sm /*N * (NT/64+1)*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = wave_incl_scan_u64(v[i]);
    __syncthreads();                       // sm reuse across calls
    if (lane == 63)
#pragma unroll
        for (int i = 0; i < N; ++i) sm[i * (NT / 64 + 1) + w] = x[i];
    __syncthreads();
    if (threadIdx.x < 64) {                // one wave scans the wave totals
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t *s = sm + i * (NT / 64 + 1);
            const uint64_t t = threadIdx.x < NT / 64 ? s[threadIdx.x] : 0;
            const uint64_t y = wave_incl_scan_u64(t);
            if (threadIdx.x < NT / 64) s[threadIdx.x] = y - t;
            if (threadIdx.x == NT / 64 - 1) s[NT / 64] = y;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        total[i] = sm[i * (NT / 64 + 1) + NT / 64];
        excl[i] = sm[i * (NT / 64 + 1) + w] + x[i] - v[i];
    }
}

// Exclusive scan of n u64 counts in three launches (n up to ~1M per 1024 block sums):
//   k_offsets_local  per 1024-element block: exclusive scan in place -> base, block total
//   k_offsets_blocks one workgroup: exclusive scan of the block totals, grand total
//   k_offsets_add    add the block base
__global__ __launch_bounds__(1024) void k_offsets_local(const uint64_t *cnt, uint64_t *base, const uint64_t *n_ptr,
                                                        uint64_t n_fixed, uint64_t *block_sum)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    const uint64_t n = n_ptr ? *n_ptr : n_fixed;
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    if ((uint64_t)blockIdx.x * 1024 >= n) { if (threadIdx.x == 0) block_sum[blockIdx.x] = 0; return; }
    const uint64_t v = i < n ? cnt[i] : 0;
    uint64_t tot;
    const uint64_t e = block_excl_scan<1024>(v, &tot, sm);
    if (i < n) base[i] = e;
    if (threadIdx.x == 0) block_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_offsets_blocks(uint64_t *block_sum, uint32_t nblocks, uint64_t *total_out)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    uint64_t carry = 0;
    for (uint32_t i0 = 0; i0 < nblocks; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const uint64_t v = i < nblocks ? block_sum[i] : 0;
        uint64_t tot;
        const uint64_t e = block_excl_scan<1024>(v, &tot, sm);
        if (i < nblocks) block_sum[i] = carry + e;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(1024) void k_offsets_add(uint64_t *base, const uint64_t *n_ptr, uint64_t n_fixed,
                                                      const uint64_t *block_sum)
{
    const uint64_t n = n_ptr ? *n_ptr : n_fixed;
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < n) base[i] += block_sum[blockIdx.x];
}

}  // namespace rhj
