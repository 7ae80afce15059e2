// rhj_partition.hip.h — stable radix partition: one LDS-staged pass (bits <= 8) and two passes in run form (bits 9..15)
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
#pragma once
#include "rhj_common.hip.h"

namespace rhj {

// ------------------------------------------------------------------ partition

constexpr int PT_BLOCK = 512;                     // threads per partition workgroup
constexpr int PT_V = 8;                           // tuples per thread
constexpr int PT_TILE = PT_BLOCK * PT_V;          // 4096 tuples = 64 KiB staged in LDS
constexpr int PT_WAVES = PT_BLOCK / WAVE;
constexpr int PT_MAX_BITS = 8;                    // digit bits per pass
constexpr uint32_t PT_MAX_GROUP = 256;            // pass-1 tiles per pass-2 tile (run form), at most

// The lanes of a wave that hold the same digit as this lane (match-any over `bits` ballots).
// pb is all ones when the lane's bit is set: peers keeps m where the bit is set and ~m where it is
// clear, i.e. peers &= ~(m ^ pb), one three-input bit operation per half and bit.
__device__ __forceinline__ uint64_t digit_peers(uint32_t d, bool ok, int bits)
{
    const uint64_t valid = __ballot(ok);
    uint32_t plo = (uint32_t)valid, phi = (uint32_t)(valid >> 32);
#pragma unroll
    for (int b = 0; b < PT_MAX_BITS; ++b) {
        if (b < bits) {                               // wave-uniform
            const uint32_t pb = ok ? 0u - ((d >> b) & 1u) : 0u;
            const uint64_t m = __ballot(pb != 0);
            plo &= ~((uint32_t)m ^ pb);
            phi &= ~((uint32_t)(m >> 32) ^ pb);
        }
    }
    return ((uint64_t)phi << 32) | plo;
}

// Exclusive scan of the values held by the first `items` threads (items <= 4 * 64; the others pass 0) with ONE barrier: every wave
// scans its 64 values, leaves its total in `sm4` and adds the totals of the waves in front of it — where the workgroup scan of
// rhj_common.hip.h takes three barriers (a batch of pass 2 has two such scans at more than 64 digits / runs: 100M x 1B at 14 bits,
// the low-radix path's 256 digits; in pass 1, one scan a tile, the same change measured +4 % on 100M x 1B and is not there).  `sm4`: four words nobody else touches until the caller's next barrier.  Returns the thread's
// exclusive prefix, *total = the sum (valid in every thread).
__device__ __forceinline__ uint32_t scan_upto_256(uint32_t v, uint32_t *total, uint32_t *sm4)
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t x = wave_incl_scan_u32(v);
    if (lane == 63 && w < 4) sm4[w] = x;
    __syncthreads();
    const uint32_t t0 = sm4[0], t1 = sm4[1], t2 = sm4[2], t3 = sm4[3];
    *total = t0 + t1 + t2 + t3;
    const uint32_t carry = (w > 0 ? t0 : 0u) + (w > 1 ? t1 : 0u) + (w > 2 ? t2 : 0u);
    return carry + x - v;
}

// Per-tile digit histogram of the one-pass partition: cnt[tile][digit] for digit = (key >> shift) & mask.
// (RANGED: a sharded join — tuples of another rank's buckets are not counted; compiled apart, see k_local_part)
template <bool RANGED>
__global__ __launch_bounds__(256) void k_hist_tiles(RelArgs r0, RelArgs r1, int shift, int bits)
{
    extern __shared__ uint32_t lds_u32[];
    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t *tile_h = lds_u32;                   // [bins]
    for (uint32_t tile = blockIdx.x; tile < r.tiles; tile += gridDim.x) {
        for (uint32_t b = threadIdx.x; b < bins; b += 256) tile_h[b] = 0;
        __syncthreads();
        const uint64_t beg = (uint64_t)tile * PT_TILE;
        const uint64_t end = min(beg + (uint64_t)PT_TILE, r.n);
#pragma unroll 4
        for (uint64_t i = beg + threadIdx.x; i < end; i += 256) {
            const uint32_t k = (uint32_t)(r.in[i].value >> shift);
            if (RANGED && ((k & mask) - r.range_lo) >= r.range_span) continue;   // sharded join: another rank's bucket
            atomicAdd(&tile_h[k & mask], 1u);
        }
        __syncthreads();
        uint32_t *row = r.cnt + (size_t)tile * bins;
        for (uint32_t b = threadIdx.x; b < bins; b += 256) row[b] = tile_h[b];
        __syncthreads();
    }
}

// Scan of the per-tile digit counts into per-tile start offsets, in four small kernels:
//   k_scan_chunks  column sums per (digit, chunk of tiles)         -> chunk_sum[rel][digit][chunk]
//   k_scan_bins    one wave per digit: exclusive scan over chunks  -> chunk_sum (in place), hist[rel][digit]
//   k_scan_psum    exclusive scan over digits                      -> psum[rel][digit]
//   k_scan_apply   counts -> psum[digit] + chunk prefix + tiles before this one (in place)
__global__ __launch_bounds__(256) void k_scan_chunks(RelArgs r0, RelArgs r1, int bits, uint32_t chunks,
                                                     uint64_t *chunk_sum /*[2][bins][chunks]*/)
{
    const RelArgs &r = blockIdx.z ? r1 : r0;
    const uint32_t bins = 1u << bits;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bins) return;
    const uint32_t per = (r.tiles + chunks - 1) / chunks;
    const uint32_t t0 = blockIdx.y * per, t1 = min(t0 + per, r.tiles);
    uint64_t s = 0;
    for (uint32_t t = t0; t < t1; ++t) s += r.cnt[(size_t)t * bins + b];
    chunk_sum[((size_t)blockIdx.z * bins + b) * chunks + blockIdx.y] = s;
}

__global__ __launch_bounds__(WAVE) void k_scan_bins(int bits, uint32_t chunks, uint64_t *chunk_sum, uint64_t *hist)
{
    const uint32_t bins = 1u << bits;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    uint64_t *cs = chunk_sum + ((size_t)blockIdx.y * bins + b) * chunks;
    uint64_t carry = 0;
    for (uint32_t c0 = 0; c0 < chunks; c0 += WAVE) {
        const uint32_t c = c0 + lane;
        const uint64_t v = c < chunks ? cs[c] : 0;
        uint64_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t y = __shfl_up(x, d, 64);
            if ((int)lane >= d) x += y;
        }
        if (c < chunks) cs[c] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
    if (lane == 0) hist[(size_t)blockIdx.y * bins + b] = carry;
}

__global__ __launch_bounds__(1024) void k_scan_psum(int bits, const uint64_t *hist, uint64_t *psum)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    const uint32_t bins = 1u << bits;
    const uint64_t *h = hist + (size_t)blockIdx.x * bins;
    uint64_t *p = psum + (size_t)blockIdx.x * bins;
    const uint32_t per = (bins + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per;
    uint64_t mine = 0;
    for (uint32_t b = b0; b < min(b0 + per, bins); ++b) mine += h[b];
    uint64_t base = block_excl_scan<1024>(mine, nullptr, sm);
    for (uint32_t b = b0; b < min(b0 + per, bins); ++b) {
        p[b] = base;
        base += h[b];
    }
}

__global__ __launch_bounds__(256) void k_scan_apply(RelArgs r0, RelArgs r1, int bits, uint32_t chunks,
                                                    const uint64_t *chunk_sum, const uint64_t *psum)
{
    const RelArgs &r = blockIdx.z ? r1 : r0;
    const uint32_t bins = 1u << bits;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bins) return;
    const uint32_t per = (r.tiles + chunks - 1) / chunks;
    const uint32_t t0 = blockIdx.y * per, t1 = min(t0 + per, r.tiles);
    uint64_t run = psum[(size_t)blockIdx.z * bins + b] + chunk_sum[((size_t)blockIdx.z * bins + b) * chunks + blockIdx.y];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = r.cnt[(size_t)t * bins + b];
        r.cnt[(size_t)t * bins + b] = (uint32_t)run;
        run += c;
    }
}

// One stable partition pass on digit = (key >> shift) & ((1 << bits) - 1), bits <= 8.
// Tile order in memory is (wave, round, lane); a tuple's stable rank inside its digit is
//   digit_start + (same digit in earlier waves) + (same digit in earlier rounds of this
//   wave) + (same digit in lower lanes of this round)
// computed with one match-any (bits ballots) per round and per-wave LDS counters — no
// atomics, so the placement does not depend on any hardware ordering.
template <bool RANGED>
__global__ __launch_bounds__(PT_BLOCK) void k_scatter_lds(RelArgs r0, RelArgs r1, int shift, int bits)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4    *stage = reinterpret_cast<uint4 *>(smem);                        // [PT_TILE]
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + (size_t)PT_TILE * 16); // [PT_WAVES][bins]
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t *dstart = wcnt + PT_WAVES * bins;                                 // [bins]
    uint32_t *delta = dstart + bins;                                           // [bins]
    uint64_t *sm = reinterpret_cast<uint64_t *>(delta + bins);                 // scan scratch

    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t tile = blockIdx.x;
    if (tile >= r.tiles) return;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    const uint64_t beg = (uint64_t)tile * PT_TILE;
    const uint32_t count = (uint32_t)min((uint64_t)PT_TILE, r.n - beg);
    const uint4 *in = reinterpret_cast<const uint4 *>(r.in) + beg;

    for (uint32_t i = threadIdx.x; i < PT_WAVES * bins; i += PT_BLOCK) wcnt[i] = 0;

    uint4 t[PT_V];
    bool ok[PT_V];
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint32_t i = w * (WAVE * PT_V) + k * WAVE + lane;
        ok[k] = i < count;
        if (ok[k]) t[k] = in[i];
    }
    __syncthreads();

    uint32_t lrank[PT_V], dig[PT_V];
    uint32_t *mycnt = wcnt + w * bins;
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint64_t key = ((uint64_t)t[k].y << 32) | t[k].x;   // (a 32-bit form makes hipcc spill t[] to scratch)
        const uint32_t d = (uint32_t)(key >> shift) & mask;
        if (RANGED) ok[k] = ok[k] && (d - r.range_lo) < r.range_span;         // sharded join: another rank's buckets are dropped here
        dig[k] = d;
        const uint64_t peers = digit_peers(d, ok[k], bits);
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        uint32_t old = 0;
        if (ok[k] && rank == 0) {                       // lowest lane of each digit group
            old = mycnt[d];
            mycnt[d] = old + (uint32_t)__popcll(peers);
        }
        const int leader = ok[k] ? __ffsll((unsigned long long)peers) - 1 : 0;
        old = __shfl(old, leader, 64);
        lrank[k] = old + rank;
    }
    __syncthreads();

    // per digit: exclusive prefix over waves, digit totals
    uint64_t mytotal = 0;
    if (threadIdx.x < bins) {
        uint32_t run = 0;
        for (int ww = 0; ww < PT_WAVES; ++ww) {
            const uint32_t c = wcnt[ww * bins + threadIdx.x];
            wcnt[ww * bins + threadIdx.x] = run;
            run += c;
        }
        mytotal = run;
    }
    uint64_t kept64;                                      // the tile's tuples that stay (all of them unless the join is sharded)
    const uint64_t ds = block_excl_scan<PT_BLOCK>(mytotal, &kept64, sm);
    if (threadIdx.x < bins) {
        dstart[threadIdx.x] = (uint32_t)ds;
        delta[threadIdx.x] = r.cnt[(size_t)tile * bins + threadIdx.x] - (uint32_t)ds;   // mod 2^32
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < PT_V; ++k)
        if (ok[k]) stage[dstart[dig[k]] + mycnt[dig[k]] + lrank[k]] = t[k];
    __syncthreads();

    uint4 *out = reinterpret_cast<uint4 *>(r.out);
    const uint32_t kept = RANGED ? (uint32_t)kept64 : count;
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint32_t p = k * PT_BLOCK + threadIdx.x;
        if (p < kept) {
            const uint4 v = stage[p];
            const uint32_t d = (v.x >> shift) & mask;
            const uint32_t dst = delta[d] + p;
            out[dst] = v;
        }
    }
}

// Intermediate tuple of the two-pass partition when row ids fit 32 bits: {key, u32 row id}, 12 bytes.
// Every row id the reference puts into a relation is an index below the relation's size
// (inter_res.c:202,225), so this is the normal case; the ABI does not promise it, so a sample decides
// (k_rowid_sample) and pass 1 raises row_id_overflow if a wider row id slips through (the host then
// runs the join again with 16-byte intermediates).
struct __attribute__((aligned(4))) Tuple12 { uint32_t klo, khi, rid; };

// first and last 2048 row ids of both relations -> summary->wide_row_ids, and summary->row_id_overflow cleared (one
// workgroup writes both words: no memset in front).  expect_narrow: the host launched the 12-byte kernels only — it does so
// until a join of this process needed the 16-byte ones — so a wide sample is reported as an overflow, which makes the
// caller run the join again with the 16-byte kernels.
__global__ __launch_bounds__(1024) void k_rowid_sample(RelArgs r0, RelArgs r1, int nrel, int force_wide, int expect_narrow,
                                                       PlanSummary *summary, int cols)
{
    uint32_t mine = force_wide ? 1u : 0u;
    for (int rel = 0; rel < nrel && !cols; ++rel) {       // (key columns: the row ids are positions below 2^32)
        const RelArgs &r = rel ? r1 : r0;
        for (uint32_t j = threadIdx.x; j < 2048u; j += 1024u)
            if (j < r.n) mine |= (uint32_t)(r.in[j].row_id >> 32) | (uint32_t)(r.in[r.n - 1 - j].row_id >> 32);
    }
    const int wide = __syncthreads_or(mine != 0);
    if (threadIdx.x == 0) {
        summary->wide_row_ids = wide ? 1u : 0u;
        summary->row_id_overflow = (wide && expect_narrow) ? 1u : 0u;
    }
}

// ---- two-pass partition in run form (radix bits 9..15) ------------------------------------------
// Pass 1 needs no histogram and no global offsets: every 4096-tuple tile is stably partitioned on
// the LOW digit inside LDS and written back to the same place in the intermediate array, fully
// coalesced, together with its run table (where each digit's run starts inside the tile) and each
// tuple's HIGH digit as one byte.  The LSD order pass 2 must read — (low digit, tile, position) — is
// then a sequence of runs: pass-2 tile (d, j) is the concatenation of the runs of digit d of pass-1
// tiles [j * group, (j + 1) * group) — about 15/16 of 4096 tuples on uniform keys, any size on
// skewed ones (processed 4096 at a time).  Histogram (from the digit bytes), scan and an LDS-staged
// scatter over these tiles give the final array.  Compared with two offset-driven passes this drops
// the first pass' histogram read of both relations and turns the first pass' scattered run writes
// into streaming writes; pass 2 reads 1 KiB runs instead of a contiguous tile.
// RANGED: a sharded join (rhj_join_device_range) — only the tuples whose bucket, the low shift + bits + next_bits key bits, lies
// in the rank's range go on; the tile shrinks in place, its run table says by how much, and everything downstream reads runs.
// Compiled apart: the test in the ordinary kernel cost it 27 % on 100M x 1B (r03: 5.7 -> 7.3 ms) although it never fires there.
// H2: the workgroup walks a strip of PT_STRIP tiles and counts (this pass' digit, the next pass' digit) in LDS — 2^(bits +
// next_bits) 16-bit cells, at most 8 KiB — one LDS atomic a tuple, which this kernel has room for (+0.5 %); the strip's
// counts go out once (RelArgs::part) and k_cnt_from_parts sums a group's strips into pass 2's count table.  Without it
// (radix bits 13..15: the cells would cost the second resident workgroup) k_hist_runs counts from the digit bytes, which
// cost 0.15 ms on 100M + 100M tuples — as many LDS atomics, but in a kernel that does nothing else.
// DIG: the next pass' digit of every tuple as one byte (k_hist_runs and the low-radix emit read them).
#ifndef PT_STRIP_N
#define PT_STRIP_N 4
#endif
constexpr uint32_t PT_STRIP = PT_STRIP_N;           // tiles a strip, at most (16-bit cells: 15 is the limit; RelArgs::strip: fewer for small inputs)
// COL (rhj_join_keys_device): the input is a KEY COLUMN — 8 bytes a tuple, the row id is the position (what GetRelation makes of
// a base relation: {col[i], i}, inter_res.c:199-204) — instead of the ABI's 16-byte tuples: pass 1 reads half the bytes.
template <bool RANGED, bool H2, bool DIG, bool COL = false>
__global__ __launch_bounds__(PT_BLOCK) void k_local_part(RelArgs r0, RelArgs r1, int shift, int bits, int next_shift, int next_bits,
                                                         PlanSummary *summary, uint32_t h2_off)
{
    const bool T12 = summary->wide_row_ids == 0;          // 12-byte intermediates (workgroup-uniform)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4    *stage = reinterpret_cast<uint4 *>(smem);                        // [PT_TILE]
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + (size_t)PT_TILE * 16); // [PT_WAVES][bins]
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t *dstart = wcnt + PT_WAVES * bins;                                 // [bins]
    uint64_t *sm = reinterpret_cast<uint64_t *>(dstart + 2 * bins);            // scan scratch
    uint32_t *h2 = reinterpret_cast<uint32_t *>(smem + h2_off);                // H2: [2^(bits + next_bits) / 2], two cells a word

    const RelArgs &r = blockIdx.y ? r1 : r0;
    uint32_t tile = blockIdx.x, tile_end = tile + 1;
    if (H2) {
        const uint32_t strip = blockIdx.x, j = strip / r.parts, p = strip - j * r.parts;
        if (j >= r.groups) return;
        tile = j * r.group + p * r.strip;
        tile_end = min(min(tile + r.strip, (j + 1) * r.group), r.tiles);
        for (uint32_t i = threadIdx.x; i < (1u << (bits + next_bits - 1)); i += PT_BLOCK) h2[i] = 0;
    } else if (tile >= r.tiles) return;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    bool wide = false;

    // (one tile unless H2; the barriers of a round keep the rounds apart.  Counting costs this kernel 5 %: 4096 LDS atomics a
    // tile at about two lanes a clock, wherever they are issued — rank loop or write-out; neither reading the next tile before
    // this tile's write-out nor barriers that wait for LDS traffic only, s_waitcnt lgkmcnt(0) + s_barrier, changed anything.)
    for (; tile < tile_end; ++tile) {
    const uint64_t beg = (uint64_t)tile * PT_TILE;
    const uint32_t count = (uint32_t)min((uint64_t)PT_TILE, r.n - beg);
    const uint4 *in = reinterpret_cast<const uint4 *>(r.in) + beg;
    const uint2 *col = reinterpret_cast<const uint2 *>(r.in) + beg;           // COL: r.in is the key column

    for (uint32_t i = threadIdx.x; i < PT_WAVES * bins; i += PT_BLOCK) wcnt[i] = 0;

    uint4 t[PT_V];
    bool ok[PT_V];
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint32_t i = w * (WAVE * PT_V) + k * WAVE + lane;
        ok[k] = i < count;
        if (COL) { if (ok[k]) { const uint2 kv = col[i]; t[k] = make_uint4(kv.x, kv.y, (uint32_t)(beg + i), (uint32_t)((beg + i) >> 32)); } }
        else if (ok[k]) t[k] = in[i];                 // (with the nt policy: C3 +1 %, at 14 bits +25 % — gpurun_out/r04t/nt2_*.txt)
    }
    __syncthreads();

    uint32_t lrank[PT_V], dig[PT_V];
    uint32_t *mycnt = wcnt + w * bins;
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint64_t key = ((uint64_t)t[k].y << 32) | t[k].x;   // (a 32-bit form makes hipcc spill t[] to scratch:
        const uint32_t d = (uint32_t)(key >> shift) & mask;       //  shift is a run-time 0 for that reason)
        if (RANGED) ok[k] = ok[k] && (((uint32_t)key & ((1u << (r.range_bits ? (int)r.range_bits : max(shift + bits, next_shift + next_bits))) - 1u)) - r.range_lo) < r.range_span;
        dig[k] = d;
        uint64_t peers = __ballot(ok[k]);               // rolled form: digit_peers() measured 6 % faster in the scatter
        for (int b = 0; b < bits; ++b) {                 // kernels but 3 % slower in this one, which sits on the HBM limit
            const uint64_t m = __ballot(ok[k] && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        uint32_t old = 0;
        if (ok[k] && rank == 0) {                       // lowest lane of each digit group
            old = mycnt[d];
            mycnt[d] = old + (uint32_t)__popcll(peers);
        }
        const int leader = ok[k] ? __ffsll((unsigned long long)peers) - 1 : 0;
        old = __shfl(old, leader, 64);
        lrank[k] = old + rank;
    }
    __syncthreads();

    uint64_t mytotal = 0;
    if (threadIdx.x < bins) {
        uint32_t run = 0;
        for (int ww = 0; ww < PT_WAVES; ++ww) {
            const uint32_t c = wcnt[ww * bins + threadIdx.x];
            wcnt[ww * bins + threadIdx.x] = run;
            run += c;
        }
        mytotal = run;
    }
    uint32_t kept = count;                                // the tile's tuples that stay (all of them unless the join is sharded)
    if (bins <= (uint32_t)WAVE) {                         // the digits are wave 0's lanes: a wave scan, not a workgroup scan (three barriers)
        if (w == 0) {
            uint32_t tot;
            const uint32_t ds = wave_excl_scan_u32((uint32_t)mytotal, &tot);
            if (lane < bins) { dstart[lane] = ds; r.runs[(size_t)lane * r.tiles + tile] = (uint16_t)ds; }
            if (lane == 0) {
                r.runs[(size_t)bins * r.tiles + tile] = (uint16_t)(RANGED ? tot : count);
                if (RANGED) dstart[bins] = tot;
            }
        }
        __syncthreads();
        if (RANGED) kept = dstart[bins];
    } else {
        uint64_t kept64;
        const uint64_t ds = block_excl_scan<PT_BLOCK>(mytotal, &kept64, sm);
        if (RANGED) kept = (uint32_t)kept64;
        if (threadIdx.x < bins) dstart[threadIdx.x] = (uint32_t)ds;
        if (threadIdx.x <= bins) r.runs[(size_t)threadIdx.x * r.tiles + tile] = (uint16_t)(threadIdx.x < bins ? (uint32_t)ds : kept);
        __syncthreads();
    }

    // (staging 12-byte tuples as three word arrays, the big win of pass 2, measured +-0 here; the digit bytes as one 8-byte
    // store per thread +15 %: this kernel sits on the HBM limit)
#pragma unroll
    for (int k = 0; k < PT_V; ++k)
        if (ok[k]) stage[dstart[dig[k]] + mycnt[dig[k]] + lrank[k]] = t[k];
    __syncthreads();

    uint4 *out = reinterpret_cast<uint4 *>(r.out) + beg;
    Tuple12 *out12 = reinterpret_cast<Tuple12 *>(r.out) + beg;
    uint8_t *dg = r.dig_out + beg;
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint32_t p = k * PT_BLOCK + threadIdx.x;
        if (p < kept) {
            const uint4 v = stage[p];
            if (T12) { out12[p] = Tuple12{v.x, v.y, v.z}; wide = wide || v.w != 0; }
            else out[p] = v;
            if (DIG) dg[p] = (uint8_t)((v.x >> next_shift) & ((1u << next_bits) - 1u));
            if (H2) {
                const uint32_t cell = (((v.x >> shift) & mask) << next_bits) | ((v.x >> next_shift) & ((1u << next_bits) - 1u));
                atomicAdd(&h2[cell >> 1], 1u << ((cell & 1u) * 16u));
            }
        }
    }
    }
    if (T12 && __ballot(wide) != 0 && (threadIdx.x & 63) == 0) atomicOr(&summary->row_id_overflow, 1u);
    if (H2) {
        __syncthreads();
        const uint32_t strips = r.groups * r.parts, half = 1u << (next_bits - 1);     // words a digit row
        uint32_t *dst = reinterpret_cast<uint32_t *>(r.part);
        for (uint32_t i = threadIdx.x; i < (1u << (bits + next_bits - 1)); i += PT_BLOCK) {
            const uint32_t d = i >> (next_bits - 1), hw = i & (half - 1u);
            dst[((size_t)d * strips + blockIdx.x) * half + hw] = h2[i];
        }
    }
}
// pass-2 tile -> its runs: thread i < group describes run i (two coalesced reads of the transposed table)
__device__ __forceinline__ void pt_run_of(const RelArgs &r, uint32_t tile2, uint32_t i, uint32_t &phys, uint32_t &len)
{
    const uint32_t d = tile2 / r.groups, j = tile2 % r.groups;
    const uint32_t t = j * r.group + i;
    phys = 0; len = 0;
    if (i < r.group && t < r.tiles1) {
        const uint32_t a = r.runs[(size_t)d * r.tiles1 + t], b = r.runs[(size_t)(d + 1) * r.tiles1 + t];
        len = b - a;
        phys = t * (uint32_t)PT_TILE + a;
    }
}

// where pass-2 tile `tile2` puts its first tuple of `digit` (RelArgs::cnt + RelArgs::sbase)
constexpr uint32_t FH_SLICES = 8;
__device__ __forceinline__ uint32_t pt_start(const RelArgs &r, uint32_t tile2, uint32_t digit, uint32_t bins)
{
    const uint32_t d = tile2 / r.groups, j = tile2 - d * r.groups;
    return r.cnt[(size_t)tile2 * bins + digit] + r.sbase[((size_t)d * FH_SLICES + j / r.per) * bins + digit];
}

// cnt[tile2][digit] of pass 2 from the digit bytes pass 1 wrote.  One WAVE per pass-2 tile, no
// workgroup barrier: the lanes hold the run table, four runs share one wave load (sixteen lanes a run, a dword of
// four digit bytes a lane), eight such loads are in flight before their LDS atomics.  (A workgroup per tile was bound by its
// chain of dependent latencies: 6 us per tile, 0.32 ms for 100M + 100M tuples.)
constexpr int HR_BLOCK = 256;
__global__ __launch_bounds__(HR_BLOCK) void k_hist_runs(RelArgs r0, RelArgs r1, int bits)
{
    extern __shared__ uint32_t lds_u32[];
    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t bins = 1u << bits;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t *h = lds_u32 + w * bins;                 // this wave's histogram
    // Which tiles a workgroup takes: `chunk` consecutive pass-1 digits d of ONE group j, its four waves side by side.  The runs of
    // digits d, d + 1, .. of a pass-1 tile lie next to each other in the digit array, 32 bytes each at 14 bits, so the lines this
    // workgroup fetches are used whole; with consecutive tile numbers (same d, consecutive j) every run cost a line of its own and
    // the kernel fetched 5.1 bytes a tuple for the one it counts (100M + 100M at 14 bits: 0.20 -> 0.18 ms; 100M x 1B the same 1.1 ms).
    const uint32_t bins1 = r.tiles / r.groups;
    uint32_t chunk = 32;
    while (chunk > 4u && r.tiles / chunk < 2048u) chunk >>= 1;
    for (uint32_t qc = blockIdx.x * chunk; qc < r.tiles; qc += gridDim.x * chunk)
    for (uint32_t qt = qc + w; qt < min(qc + chunk, r.tiles); qt += HR_BLOCK / WAVE) {
        const uint32_t tj = qt / bins1, tile2 = (qt - tj * bins1) * r.groups + tj;
        for (uint32_t b = lane; b < bins; b += WAVE) h[b] = 0;
        for (uint32_t c0 = 0; c0 < r.group; c0 += WAVE) {
            uint32_t phys, len;
            pt_run_of(r, tile2, c0 + lane, phys, len);
            const uint32_t nrun = min((uint32_t)WAVE, r.group - c0);
            // four runs per wave load: sixteen lanes a run, four digit bytes a lane (the digit array is padded by 64 bytes,
            // a run's last dword may reach past its end: those bytes are not counted); eight loads in flight.
            // (The LDS atomics — about two lanes a clock and CU — are this kernel's time, 1.1 ms on 100M x 1B.  Counting by
            // match-any instead, a byte of every lane a round and the lowest lane of a digit adding the group, took 3.1 ms: runs
            // are 20 .. 45 bytes, so most rounds run for a few lanes.)
            const uint32_t g4 = lane >> 4, sub4 = (lane & 15u) * 4u;
            for (uint32_t q0 = 0; q0 < nrun; q0 += 32) {
                uint32_t dw[8], ll[8], pp[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t run = q0 + (uint32_t)q * 4u + g4;
                    pp[q] = __shfl(phys, (int)(run & 63u), 64);
                    const uint32_t l = __shfl(len, (int)(run & 63u), 64);
                    ll[q] = run < nrun ? l : 0u;
                    dw[q] = 0;
                    if (sub4 < ll[q]) dw[q] = *reinterpret_cast<const uint32_t *>(r.dig_in + pp[q] + sub4);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    for (uint32_t off = sub4;;) {
                        if (off < ll[q]) {
                            const uint32_t nb = min(ll[q] - off, 4u);
                            for (uint32_t b = 0; b < nb; ++b) atomicAdd(&h[(dw[q] >> (8u * b)) & 0xffu], 1u);
                        }
                        off += WAVE;                          // skewed keys: a run longer than 64 tuples
                        if (off >= ll[q]) break;
                        dw[q] = *reinterpret_cast<const uint32_t *>(r.dig_in + pp[q] + off);
                    }
                }
            }
        }
        uint32_t *row = r.cnt + (size_t)tile2 * bins;
        for (uint32_t b = lane; b < bins; b += WAVE) row[b] = h[b];
    }
}

// Pass 2's start offsets for the two-pass partition, in two launches (the per-tile scan kernels above took four, plus a
// bucket-histogram kernel, its memset and a psum kernel behind the scatter).  A bucket is (pass-2 digit, pass-1 digit d) and a pass-2 tile is (d, group j),
// so inside a bucket the tiles in front of (d, j) are just the groups j' < j: a scan along j for every (d, digit).
// k_group_scan — grid: pass-1 digits x relations x FH_SLICES slices of the groups, 1024 threads = digits x rows — turns its
// slice's counts into exclusive prefixes along j in place and leaves the slice's totals; FROM_PARTS: the counts are the sums
// of a group's strips (pass 1 counted, k_local_part<., true, .>), else k_hist_runs wrote them.
// Two phases a round of GS_ROWS groups: every thread sums up to eight digits of one group over the group's strips (16-byte
// loads, eight in flight: with one digit a thread and 2-byte loads the kernel took 45 us on 100M + 100M tuples), the sums meet
// in LDS, and the scan along the groups runs there.
template <bool FROM_PARTS>
__global__ __launch_bounds__(1024) void k_group_scan(RelArgs r0, RelArgs r1, int bits, uint32_t *slice_tot, int msd)
{
    __shared__ uint32_t acc[8192];                    // [rows of the round][digit]
    __shared__ uint32_t tot[1024];                    // [row chunk][digit]
    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t bins = 1u << bits, d = blockIdx.x;
    const uint32_t cpt = min(8u, bins), lanes = bins / cpt, rows = 1024u / lanes;        // digits a thread, threads a group, groups a round
    const uint32_t row = threadIdx.x / lanes, c0 = (threadIdx.x % lanes) * cpt;
    const uint32_t b = threadIdx.x & (bins - 1u), rr = threadIdx.x >> bits, nrr = 1024u >> bits;  // phase 2: digit, chunk of cpt rows
    const uint32_t j0 = min(blockIdx.z * r.per, r.groups), j1 = min(j0 + r.per, r.groups);
    uint32_t *dst = r.cnt + (size_t)d * r.groups * bins;
    uint32_t carry = 0;
    for (uint32_t jb = j0; jb < j1; jb += rows) {
        uint32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (FROM_PARTS) {
            // The round's R groups times S = rows / R shares of a group's strips keep all threads busy: with 256 pass-1 digits
            // (the low-radix path on 8 bits) a slice has 3 groups of 56 strips, and three threads loading 56 strips each in
            // 4096 such workgroups took 0.11 ms.  The shares meet in LDS by atomic adds.
            const uint32_t R = min(rows, j1 - jb);
            uint32_t S = max(1u, min(rows / R, r.parts));
            if (S < 4u) S = 1;                        // (no shares: plain stores, no clearing — 100M + 100M at 12 bits has S = 2: 35 vs 40 us)
            if (S > 1u) {
                for (uint32_t i = threadIdx.x; i < rows * bins; i += 1024u) acc[i] = 0;
                __syncthreads();
            }
            const uint32_t jr = row / S, sp = row - jr * S;
            if (jr < R) {
                const uint16_t *sj = r.part + ((size_t)d * r.groups + jb + jr) * r.parts * bins + c0;
                // a thread's digits of one strip in ONE load (16, 8 or 4 bytes), eight strips in flight
                for (uint32_t p = sp; p < r.parts; p += 8 * S) {
                    uint4 x[8];
#pragma unroll
                    for (uint32_t q = 0; q < 8; ++q) {
                        x[q] = uint4{0, 0, 0, 0};
                        if (p + q * S < r.parts) {
                            const uint16_t *at = sj + (size_t)(p + q * S) * bins;
                            if (cpt == 8) x[q] = *reinterpret_cast<const uint4 *>(at);
                            else if (cpt == 4) { const uint2 y = *reinterpret_cast<const uint2 *>(at); x[q].x = y.x; x[q].y = y.y; }
                            else x[q].x = *reinterpret_cast<const uint32_t *>(at);
                        }
                    }
#pragma unroll
                    for (uint32_t q = 0; q < 8; ++q) {
                        v[0] += x[q].x & 0xffffu; v[1] += x[q].x >> 16; v[2] += x[q].y & 0xffffu; v[3] += x[q].y >> 16;
                        v[4] += x[q].z & 0xffffu; v[5] += x[q].z >> 16; v[6] += x[q].w & 0xffffu; v[7] += x[q].w >> 16;
                    }
                }
                if (S > 1u) {
#pragma unroll
                    for (uint32_t c = 0; c < 8; ++c)
                        if (c < cpt && v[c]) atomicAdd(&acc[jr * bins + c0 + c], v[c]);
                }
            }
            if (S == 1u) {                            // (jr == row; rows behind the slice's end store zeros)
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c)
                    if (c < cpt) acc[row * bins + c0 + c] = v[c];
            }
        } else {
            const uint32_t j = jb + row;
            if (j < j1) {
                const uint32_t *sj = dst + (size_t)j * bins + c0;
                if (cpt == 8) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(sj), y = *reinterpret_cast<const uint4 *>(sj + 4);
                    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
                } else
                    for (uint32_t c = 0; c < cpt; ++c) v[c] = sj[c];
            }
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c)
                if (c < cpt) acc[row * bins + c0 + c] = v[c];
        }
        __syncthreads();
        // thread (digit b, chunk rr) owns rows rr * cpt .. + cpt - 1 of the round
        uint32_t mine = 0;
        for (uint32_t i = 0; i < cpt; ++i) mine += acc[(rr * cpt + i) * bins + b];
        tot[rr * bins + b] = mine;
        __syncthreads();
        uint32_t ex = carry, all = 0;
        const uint32_t live = min(nrr, (min(rows, j1 - jb) + cpt - 1u) / cpt);        // chunks with groups in them (the others hold zeros)
        for (uint32_t q = 0; q < live; ++q) {
            const uint32_t x = tot[q * bins + b];
            ex += q < rr ? x : 0u;
            all += x;
        }
        for (uint32_t i = 0; i < cpt; ++i) {
            const uint32_t jj = jb + rr * cpt + i;
            if (jj < j1) dst[(size_t)jj * bins + b] = ex;
            ex += acc[(rr * cpt + i) * bins + b];
        }
        carry += all;
        __syncthreads();
    }
    if (rr == 0)                                      // [relation][slice][bucket = digit << bits1 | d, or d << bits | digit (msd)]
        slice_tot[((size_t)blockIdx.y * FH_SLICES + blockIdx.z) * gridDim.x * bins + (msd ? (size_t)d * bins + b : (size_t)b * gridDim.x + d)] = carry;
}

// slice totals -> bucket histogram (u64) + exclusive psum of the full radix, and RelArgs::sbase: bucket start + the slices
// in front (bucket = digit << bits1 | d).  Grid: relations x FH_SLICES — every workgroup sums and scans all buckets (32 K
// coalesced loads at most), workgroup z writes slice z's row of sbase, workgroup 0 the histogram and the psum: sbase is read
// by pass 2 digit-wise, so its stores are scattered, and 32 K scattered 4-byte stores from ONE compute unit took 20 us.
// (PER > 0: buckets a thread, known at compile time — all slice totals of a thread are loaded at once and stay in registers.)
template <int PER>
__global__ __launch_bounds__(1024) void k_bucket_psum(int bits1, int bits, const uint32_t *slice_tot, uint32_t *sbase, uint64_t *hist,
                                                      uint64_t *psum, int staged, int msd)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    const uint32_t all_bins = 1u << (bits1 + bits), bins = 1u << bits, bins1 = 1u << bits1, zme = blockIdx.y;
    const uint32_t *st = slice_tot + (size_t)blockIdx.x * FH_SLICES * all_bins;          // [slice][bucket]
    uint32_t *sb = sbase + (size_t)blockIdx.x * bins1 * FH_SLICES * bins;                // [d][slice][digit]
    uint64_t *h = hist + (size_t)blockIdx.x * all_bins, *p = psum + (size_t)blockIdx.x * all_bins;
    const uint32_t per = PER ? (uint32_t)PER : (all_bins + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, all_bins);
    // bucket -> its cell of sbase: (pass-1 digit d, slice, pass-2 digit); msd: pass 1 took the HIGH bits, so d = bucket >> bits
    auto sb_at = [&](uint32_t bk) -> size_t {
        const uint32_t d = msd ? bk >> bits : bk & (bins1 - 1u), dg = msd ? bk & (bins - 1u) : bk >> bits1;
        return ((size_t)d * FH_SLICES + zme) * bins + dg;
    };
    uint64_t mine = 0;
    if (PER) {
        uint32_t c[PER ? PER : 1][FH_SLICES];
#pragma unroll
        for (int i = 0; i < PER; ++i)
#pragma unroll
            for (uint32_t z = 0; z < FH_SLICES; ++z) c[i][z] = b0 + i < all_bins ? st[(size_t)z * all_bins + b0 + i] : 0u;
        uint32_t t[PER ? PER : 1], front[PER ? PER : 1];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            t[i] = 0; front[i] = 0;
#pragma unroll
            for (uint32_t z = 0; z < FH_SLICES; ++z) { front[i] += z < zme ? c[i][z] : 0u; t[i] += c[i][z]; }
            mine += t[i];
        }
        uint64_t base = block_excl_scan<1024>(mine, nullptr, sm);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t bk = b0 + i;
            if (bk < all_bins) {
                if (zme == 0) { h[bk] = t[i]; p[bk] = base; }
                sb[sb_at(bk)] = (uint32_t)base + front[i];   // (mod 2^32,
                base += t[i];                                                                                            //  like the tile counts)
            }
        }
        return;
    }
    // Any count.  A thread's buckets are consecutive (the scan wants them so), which makes its loads of the slice totals
    // 64-byte-strided over the wave: 0.13 ms for 16 K buckets.  `staged`: the workgroup first reads all totals coalesced and
    // leaves every bucket's total and the sum of the slices in front of its own in LDS (2 x 4 bytes a bucket).
    extern __shared__ uint32_t lds_bp[];
    uint32_t *tl = lds_bp;
    if (staged) {                                     // 16 K buckets a round (2 x 64 KiB of LDS): 15 bits take two
        const uint32_t chunk = min(all_bins, 16384u), cper = chunk / 1024u;
        uint32_t *fl2 = lds_bp + chunk;
        uint64_t carry = 0;
        for (uint32_t c0 = 0; c0 < all_bins; c0 += chunk) {
            for (uint32_t i = threadIdx.x; i < chunk; i += 1024u) {
                const uint32_t bk = c0 + i;
                uint32_t c[FH_SLICES], t = 0, front = 0;
#pragma unroll
                for (uint32_t z = 0; z < FH_SLICES; ++z) c[z] = st[(size_t)z * all_bins + bk];
#pragma unroll
                for (uint32_t z = 0; z < FH_SLICES; ++z) { front += z < zme ? c[z] : 0u; t += c[z]; }
                tl[i] = t; fl2[i] = front;
            }
            __syncthreads();
            const uint32_t i0 = threadIdx.x * cper;
            uint64_t sum = 0, all = 0;
            for (uint32_t i = i0; i < i0 + cper; ++i) sum += tl[i];
            uint64_t base = carry + block_excl_scan<1024>(sum, &all, sm);
            for (uint32_t i = i0; i < i0 + cper; ++i) {
                const uint32_t bk = c0 + i, t = tl[i];
                if (zme == 0) { h[bk] = t; p[bk] = base; }
                sb[sb_at(bk)] = (uint32_t)base + fl2[i];
                base += t;
            }
            carry += all;
            __syncthreads();
        }
        return;
    }
    constexpr uint32_t CH = 4;                        // four buckets at a time, their 32 slice totals in flight together
    for (uint32_t c0 = b0; c0 < b1; c0 += CH) {
        uint32_t c[CH][FH_SLICES];
#pragma unroll
        for (uint32_t i = 0; i < CH; ++i)
#pragma unroll
            for (uint32_t z = 0; z < FH_SLICES; ++z) c[i][z] = c0 + i < b1 ? st[(size_t)z * all_bins + c0 + i] : 0u;
#pragma unroll
        for (uint32_t i = 0; i < CH; ++i)
#pragma unroll
            for (uint32_t z = 0; z < FH_SLICES; ++z) mine += c[i][z];
    }
    uint64_t base = block_excl_scan<1024>(mine, nullptr, sm);
    for (uint32_t c0 = b0; c0 < b1; c0 += CH) {
        uint32_t c[CH][FH_SLICES];
#pragma unroll
        for (uint32_t i = 0; i < CH; ++i)
#pragma unroll
            for (uint32_t z = 0; z < FH_SLICES; ++z) c[i][z] = c0 + i < b1 ? st[(size_t)z * all_bins + c0 + i] : 0u;
#pragma unroll
        for (uint32_t i = 0; i < CH; ++i) {
            const uint32_t bk = c0 + i;
            if (bk < b1) {
                uint32_t t = 0, front = 0;
#pragma unroll
                for (uint32_t z = 0; z < FH_SLICES; ++z) { front += z < zme ? c[i][z] : 0u; t += c[i][z]; }
                if (zme == 0) { h[bk] = t; p[bk] = base; }
                sb[sb_at(bk)] = (uint32_t)base + front;
                base += t;
            }
        }
    }
}

// T12: 12-byte intermediates in; O12: 12-byte tuples out too (the join's own partition when the row ids fit 32 bits:
// the fused kernel then streams and gathers 12 instead of 16 bytes per tuple; rhj_partition_device() hands out
// rhj_tuple and keeps 16-byte output).  (A run-time switch here cost 30 %: compiled apart, launched side by side.)
#ifndef SR_VN
#define SR_VN 8         // tuples per thread and batch of pass 2 (the batch is independent of pass 1's 4096-tuple tiles)
#endif
#ifndef SR_MINW
#define SR_MINW 4
#endif
constexpr int SR_V = SR_VN;
constexpr int SR_TILE = PT_BLOCK * SR_V;
constexpr uint32_t SR_RUNOFF = PT_MAX_GROUP + 8;  // run-start table of a pass-2 tile: entries behind the last run hold the tile's total
#ifdef RHJ_INSTRUMENT
__device__ uint64_t g_sr_dbg[256 * 16];           // diagnostics build: phase stamps (100 MHz) of the first 256 workgroups' FOURTH batch
#define SR_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 256 && iter == 3) g_sr_dbg[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SR_STAMP(slot) do { } while (0)
#endif
#ifndef SR_AOS
#define SR_AOS 0        // A/B only: 1 = 12-byte tuples staged in 16-byte slots
#endif
#ifndef SR_NOSHFL
#define SR_NOSHFL 1
#endif
#ifndef SR_PIPE
#define SR_PIPE 1       // 1: the next batch's tuple loads are issued before the current batch is written out (0: the round-2 loop, for A/B)
#endif
#if SR_PIPE
// A batch of a pass-2 tile costs a workgroup a chain of dependent steps; in-kernel stamps of the round-2 loop (r03, 10.6 us
// per batch, two workgroups per CU): ranks 2.4 us, staging 1.2, run table of the next tile 1.3, run search + load issue 1.8,
// write-out 2.1, barriers 1.5 — with the loads and stores switched off the kernel still took 0.52 of its 0.70 ms per 100 M
// tuples: the chain, not the bandwidth, sets the pace.  So:
//   * software pipeline: a batch's tuples are dead in registers once they are staged, so the NEXT batch's loads are issued
//     right there — before the write-out — and fly behind the write-out, the barriers and the zeroing of the counters; the
//     next TILE's run table is built in the other half of a double buffer at the same point;
//   * run search by the wave: lanes 0..7 find the run of their round's first element (one binary search for the wave), a lane's
//     own run is at most three further on (three independent reads) — it was a chain of ~30 dependent LDS reads per lane;
//   * 12-byte tuples are staged as three word arrays (the 16-byte slots' scattered stores were 20 % of the kernel's CU cycles
//     in bank conflicts; a quarter of their bytes was padding);
// (Measured and not kept: the per-wave digit counters bumped by LDS atomics that return the old value instead of read +
// write by the group's lowest lane: +4 % on C3, +3 % on C4; 16-byte staging slots for 12-byte tuples: +27 % / +10 %.)
template <bool T12, bool O12>
__global__ __launch_bounds__(PT_BLOCK, SR_MINW) void k_scatter_runs(RelArgs r0, RelArgs r1, int shift, int bits, uint32_t search0,
                                                           const PlanSummary *summary)
{
    if ((summary->wide_row_ids == 0) != T12) return;      // the other instantiation's launch moves the data
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4    *stage = reinterpret_cast<uint4 *>(smem);                        // [SR_TILE] (16-byte tuples)
    uint32_t *s_klo = reinterpret_cast<uint32_t *>(smem);                      // [SR_TILE] x 3 (12-byte tuples)
    uint32_t *s_khi = s_klo + SR_TILE, *s_rid = s_khi + SR_TILE;
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + (size_t)SR_TILE * 16); // [PT_WAVES][bins]
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t *dstart = wcnt + PT_WAVES * bins;                                 // [bins]
    uint32_t *delta = dstart + bins;                                           // [bins]
    uint64_t *sm = reinterpret_cast<uint64_t *>(delta + bins);                 // scan scratch [PT_BLOCK / 64 + 1]
    uint32_t *gbase = reinterpret_cast<uint32_t *>(sm + PT_BLOCK / 64 + 2);    // [bins] next output position per digit
    uint32_t *runoff0 = gbase + bins;                                          // [2][SR_RUNOFF] first element of run i (double-buffered per tile)
    uint32_t *rbase0 = runoff0 + 2 * SR_RUNOFF;                                // [2][PT_MAX_GROUP] physical index of element e of run i = rbase[i] + e

    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    const uint4 *in = reinterpret_cast<const uint4 *>(r.in);
    uint4 *out = reinterpret_cast<uint4 *>(r.out);

    // Workgroups are dispatched round-robin over the 8 XCDs (blockIdx.x % 8), each with its own L2.  Consecutive pass-2 tiles
    // write ADJACENT pieces of every digit's output, so the cache line at the seam is completed by the neighbour tile: the
    // tiles are dealt to the XCDs in blocks of as many consecutive tiles as an XCD has workgroups, which walk the block
    // together — both halves of a seam line meet in the same L2 and leave as one full-line write (-6 % against a plain grid
    // stride) — and the blocks go round-robin over the XCDs, so that the oversized tiles of a hot digit (Zipf keys) are
    // shared by all of them (-5 % on 100M x 1B against one contiguous eighth per XCD; same on uniform keys).
    const uint32_t xcd = blockIdx.x & 7u, per_xcd = gridDim.x >> 3;
    const uint32_t slot = blockIdx.x >> 3;
    const uint32_t tstep = 8u * per_xcd;
    const uint32_t t_end = r.tiles;
    uint32_t tile2 = xcd * per_xcd + slot;
    if (tile2 >= t_end) return;

    // run table of a tile: scan of the run lengths the threads hold -> runoff / rbase of buffer `buf` (entries from `group` on
    // hold the tile's total, also in runoff[PT_MAX_GROUP]); the tile's output positions per digit -> gbase (the previous
    // tile's last batch has consumed gbase)
    // (At most 64 runs — 64 pass-1 digits or fewer — are wave 0's alone: one wave scan instead of a workgroup scan and its three
    // barriers, which cost every batch 1.5 of its 10 us; the caller reads the tile's total from the table behind its next barrier.
    // With more runs, up to four a lane of wave 0, that wave became the straggler: 100M x 1B at 14 bits +12 %.)
    const bool wave_runs = r.group <= (uint32_t)WAVE;  // (workgroup-uniform)
    auto build_runs = [&](uint32_t buf, uint32_t phys, uint32_t len, uint32_t gb) {
        uint32_t *runoff = runoff0 + buf * SR_RUNOFF, *rbase = rbase0 + buf * PT_MAX_GROUP;
        if (wave_runs) {
            if (w == 0) {
                uint32_t tot;
                const uint32_t off = wave_excl_scan_u32(len, &tot);
                runoff[lane] = lane < r.group ? off : tot;
                for (uint32_t i = WAVE + lane; i < SR_RUNOFF; i += WAVE) runoff[i] = tot;
                rbase[lane] = phys - off;
            }
            if (threadIdx.x < bins) gbase[threadIdx.x] = gb;
            return 0u;
        }
        uint32_t tot;                                 // (at most PT_MAX_GROUP = 256 runs: thread i holds run i's length, the others 0)
        const uint32_t off = scan_upto_256(len, &tot, reinterpret_cast<uint32_t *>(sm));
        if (threadIdx.x < SR_RUNOFF) runoff[threadIdx.x] = threadIdx.x < r.group ? off : tot;
        if (threadIdx.x < PT_MAX_GROUP) rbase[threadIdx.x] = phys - off;
        if (threadIdx.x < bins) gbase[threadIdx.x] = gb;
        return tot;
    };
    uint32_t tk[SR_V], th[SR_V], tr[SR_V], tw[SR_V];  // the batch's tuples (tw: upper row-id word of 16-byte tuples)
    // run search + loads of batch [sb, sb + count) of the tile whose run table is in buffer `buf`
    auto issue_loads = [&](uint32_t buf, uint32_t sb, uint32_t count) {
        const uint32_t *runoff = runoff0 + buf * SR_RUNOFF, *rbase = rbase0 + buf * PT_MAX_GROUP;
        // lanes 0..7: the last run that starts at or before the first element of this wave's round `lane`
        uint32_t pos = 0;
        {
            const uint32_t e0 = sb + w * (WAVE * SR_V) + (lane & 7u) * WAVE;
            for (uint32_t s2 = search0; s2 >= 1; s2 >>= 1)
                if (runoff[pos + s2] <= e0) pos += s2;
        }
#pragma unroll
        for (int k = 0; k < SR_V; ++k) {
            const uint32_t i = w * (WAVE * SR_V) + k * WAVE + lane;
            const uint32_t e = sb + i;
            const uint32_t j0 = (uint32_t)__builtin_amdgcn_readlane((int)pos, k);
            const uint32_t a1 = runoff[j0 + 1], a2 = runoff[j0 + 2], a3 = runoff[j0 + 3], a4 = runoff[j0 + 4];   // (wave-uniform addresses)
            uint32_t j = j0 + (a1 <= e ? 1u : 0u) + (a2 <= e ? 1u : 0u) + (a3 <= e ? 1u : 0u);
            if (a4 <= sb + w * (WAVE * SR_V) + (uint32_t)k * WAVE + (WAVE - 1)) {            // short or empty runs: more than four runs under these 64 elements (wave-uniform)
                j = j0;
                while (runoff[j + 1] <= e && j + 1 < r.group) ++j;
            }
            tk[k] = th[k] = tr[k] = tw[k] = 0;
            if (i < count) {
                if (T12) { const Tuple12 x = reinterpret_cast<const Tuple12 *>(r.in)[rbase[j] + e]; tk[k] = x.klo; th[k] = x.khi; tr[k] = x.rid; }
                else { const uint4 x = in[rbase[j] + e]; tk[k] = x.x; th[k] = x.y; tr[k] = x.z; tw[k] = x.w; }
            }
        }
    };

    uint32_t nphys = 0, nlen = 0, ngb = 0;            // the NEXT tile's run (thread i: run i) and output positions, prefetched
    pt_run_of(r, tile2, threadIdx.x, nphys, nlen);
    if (threadIdx.x < bins) ngb = pt_start(r, tile2, threadIdx.x, bins);
    uint32_t buf = 0;
    uint32_t total = build_runs(buf, nphys, nlen, ngb);
    {
        const uint32_t nt = tile2 + tstep;
        nphys = 0; nlen = 0;
        if (nt < t_end) {
            pt_run_of(r, nt, threadIdx.x, nphys, nlen);
            if (threadIdx.x < bins) ngb = pt_start(r, nt, threadIdx.x, bins);
        }
    }
    for (uint32_t i = threadIdx.x; i < PT_WAVES * bins; i += PT_BLOCK) wcnt[i] = 0;
    __syncthreads();
    if (wave_runs) total = runoff0[buf * SR_RUNOFF + SR_RUNOFF - 1];
    uint32_t sb = 0;
    uint32_t count = min((uint32_t)SR_TILE, total);
    issue_loads(buf, sb, count);
    bool more = true;
    for (uint32_t iter = 0; more; ++iter) {
        (void)iter;                                   // (stamps of the diagnostics build)
        SR_STAMP(0);
        __syncthreads();                              // counters are zero, run tables visible
        SR_STAMP(1);
#ifdef RHJ_INSTRUMENT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // diagnostics: how long the batch's loads are still out at this point
        SR_STAMP(10);
#endif
        uint32_t rk[SR_V];                            // rank among the wave's tuples of the digit so far | digit << 16
        uint32_t *mycnt = wcnt + w * bins;
#pragma unroll
        for (int k = 0; k < SR_V; ++k) {
            const bool ok = w * (WAVE * SR_V) + k * WAVE + lane < count;
            const uint64_t key = ((uint64_t)th[k] << 32) | tk[k];
            const uint32_t d = (uint32_t)(key >> shift) & mask;
            const uint64_t peers = digit_peers(d, ok, bits);
            const uint32_t rank = (uint32_t)__popcll(peers & lt);
#if SR_NOSHFL
            // the whole digit group reads its counter, then its lowest lane adds the group (a wave's LDS operations complete in
            // order; one lane per counter writes: no ordering between lanes is relied on) — no shuffle from the leader
            const uint32_t old = mycnt[d];
            if (ok && rank == 0) mycnt[d] = old + (uint32_t)__popcll(peers);
#else
            uint32_t old = 0;
            if (ok && rank == 0) { old = mycnt[d]; mycnt[d] = old + (uint32_t)__popcll(peers); }   // lowest lane of each digit group
            const int leader = ok ? __ffsll((unsigned long long)peers) - 1 : 0;
            old = __shfl(old, leader, 64);
#endif
            rk[k] = (old + rank) | (d << 16);
        }
        SR_STAMP(2);
        __syncthreads();
        SR_STAMP(3);

        uint64_t mytotal = 0;
        if (threadIdx.x < bins) {
            uint32_t run = 0;
            for (int ww = 0; ww < PT_WAVES; ++ww) {
                const uint32_t c = wcnt[ww * bins + threadIdx.x];
                wcnt[ww * bins + threadIdx.x] = run;
                run += c;
            }
            mytotal = run;
        }
        uint64_t ds;
        if (bins <= (uint32_t)WAVE) {                 // the digits are wave 0's lanes: a wave scan, no workgroup scan (workgroup-uniform)
            uint32_t unused;
            ds = w == 0 ? wave_excl_scan_u32((uint32_t)mytotal, &unused) : 0u;
        } else {                                      // at most 256 digits
            uint32_t unused;
            ds = scan_upto_256((uint32_t)mytotal, &unused, reinterpret_cast<uint32_t *>(sm) + 4);
        }
        if (threadIdx.x < bins) {
            dstart[threadIdx.x] = (uint32_t)ds;
            const uint32_t gb = gbase[threadIdx.x];
            delta[threadIdx.x] = gb - (uint32_t)ds;                 // mod 2^32
            gbase[threadIdx.x] = gb + (uint32_t)mytotal;
        }
        __syncthreads();
        SR_STAMP(4);

#pragma unroll
        for (int k = 0; k < SR_V; ++k) {
            if (w * (WAVE * SR_V) + k * WAVE + lane < count) {
                const uint32_t d = rk[k] >> 16;
                const uint32_t p = dstart[d] + mycnt[d] + (rk[k] & 0xffffu);
                if (T12 && !SR_AOS) { s_klo[p] = tk[k]; s_khi[p] = th[k]; s_rid[p] = tr[k]; }
                else stage[p] = make_uint4(tk[k], th[k], tr[k], tw[k]);
            }
        }

        SR_STAMP(5);
        // ---- the next batch: same tile, or the next tile (its run table goes to the other buffer)
        const uint32_t wcount = count;                // the batch being written out
        uint32_t nsb = sb + SR_TILE;
        bool next_tile = false;
        if (nsb >= total) {
            nsb = 0;
            tile2 += tstep;
            more = tile2 < t_end;
            next_tile = more;
        }
        if (next_tile) {                              // (workgroup-uniform)
            buf ^= 1u;
            total = build_runs(buf, nphys, nlen, ngb);
            const uint32_t nt = tile2 + tstep;
            nphys = 0; nlen = 0;
            if (nt < t_end) {
                pt_run_of(r, nt, threadIdx.x, nphys, nlen);
                if (threadIdx.x < bins) ngb = pt_start(r, nt, threadIdx.x, bins);
            }
        }
        SR_STAMP(6);
        __syncthreads();                              // staged tile, run table and gbase are visible
        SR_STAMP(7);
        if (next_tile && wave_runs) total = runoff0[buf * SR_RUNOFF + SR_RUNOFF - 1];
        sb = nsb;
        if (more) {
            count = min((uint32_t)SR_TILE, total - sb);
            issue_loads(buf, sb, count);              // in flight during the write-out below
        }
        SR_STAMP(8);

#pragma unroll
        for (int k = 0; k < SR_V; ++k) {
            const uint32_t p = k * PT_BLOCK + threadIdx.x;
            if (p < wcount) {
                uint4 v;
                if (T12 && !SR_AOS) v = make_uint4(s_klo[p], s_khi[p], s_rid[p], 0u);
                else v = stage[p];
                const uint32_t d = (uint32_t)((((uint64_t)v.y << 32) | v.x) >> shift) & mask;
                if (O12) reinterpret_cast<Tuple12 *>(r.out)[delta[d] + p] = Tuple12{v.x, v.y, v.z};
                else out[delta[d] + p] = v;
            }
        }
        for (uint32_t i = threadIdx.x; i < PT_WAVES * bins; i += PT_BLOCK) wcnt[i] = 0;
        SR_STAMP(9);
    }
}
#else
template <bool T12, bool O12>
__global__ __launch_bounds__(PT_BLOCK, SR_MINW) void k_scatter_runs(RelArgs r0, RelArgs r1, int shift, int bits, uint32_t search0,
                                                           const PlanSummary *summary)
{
    if ((summary->wide_row_ids == 0) != T12) return;      // the other instantiation's launch moves the data
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4    *stage = reinterpret_cast<uint4 *>(smem);                        // [SR_TILE]
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + (size_t)SR_TILE * 16); // [PT_WAVES][bins]
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t *dstart = wcnt + PT_WAVES * bins;                                 // [bins]
    uint32_t *delta = dstart + bins;                                           // [bins]
    uint64_t *sm = reinterpret_cast<uint64_t *>(delta + bins);                 // scan scratch [PT_BLOCK / 64 + 1]
    uint32_t *gbase = reinterpret_cast<uint32_t *>(sm + PT_BLOCK / 64 + 2);    // [bins] next output position per digit
    uint32_t *runoff = gbase + bins;                                           // [PT_MAX_GROUP + 1] first element of run i
    uint32_t *rbase = runoff + PT_MAX_GROUP + 1;                               // [PT_MAX_GROUP] physical index of element e of run i = rbase[i] + e

    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    const uint4 *in = reinterpret_cast<const uint4 *>(r.in);
    uint4 *out = reinterpret_cast<uint4 *>(r.out);

    // Workgroups are dispatched round-robin over the 8 XCDs (blockIdx.x % 8), each with its own L2.
    // Consecutive pass-2 tiles write ADJACENT pieces of every digit's output, so the cache line at the
    // seam is completed by the neighbour tile: the tiles are dealt to the XCDs in blocks of as many
    // consecutive tiles as an XCD has workgroups, which walk the block together — both halves of a seam
    // line meet in the same L2 and leave as one full-line write (-6 % against a plain grid stride) — and
    // the blocks go round-robin over the XCDs, so that the oversized tiles of a hot digit (Zipf keys) are
    // shared by all of them (-5 % on 100M x 1B against one contiguous eighth per XCD; same on uniform
    // keys).  The next tile's run table and output offsets are fetched while the current tile is moved.
    const uint32_t xcd = blockIdx.x & 7u, per_xcd = gridDim.x >> 3;           // gridDim.x is a multiple of 8
    const uint32_t slot = blockIdx.x >> 3;
    const uint32_t tstep = 8u * per_xcd;                                        // the next block of this XCD
    const uint32_t t_end = r.tiles;
    const uint32_t t_first = xcd * per_xcd + slot;
    uint32_t nphys = 0, nlen = 0, ngb = 0;
    if (t_first < t_end) {
        pt_run_of(r, t_first, threadIdx.x, nphys, nlen);
        if (threadIdx.x < bins) ngb = pt_start(r, t_first, threadIdx.x, bins);
    }
    for (uint32_t tile2 = t_first; tile2 < t_end; tile2 += tstep) {
    uint32_t total;
    {
        const uint32_t phys = nphys, len = nlen;
        uint64_t tot64;
        const uint32_t off = (uint32_t)block_excl_scan<PT_BLOCK>(len, &tot64, sm);
        total = (uint32_t)tot64;
        if (threadIdx.x < PT_MAX_GROUP) { runoff[threadIdx.x] = threadIdx.x < r.group ? off : total; rbase[threadIdx.x] = phys - off; }
        if (threadIdx.x == 0) runoff[PT_MAX_GROUP] = total;
        if (threadIdx.x < bins) gbase[threadIdx.x] = ngb;
        const uint32_t nt = tile2 + tstep;
        nphys = 0; nlen = 0;
        if (nt < t_end) {
            pt_run_of(r, nt, threadIdx.x, nphys, nlen);
            if (threadIdx.x < bins) ngb = pt_start(r, nt, threadIdx.x, bins);
        }
    }
    __syncthreads();

    for (uint32_t sb = 0; sb < total; sb += SR_TILE) {
        const uint32_t count = min((uint32_t)SR_TILE, total - sb);
        for (uint32_t i = threadIdx.x; i < PT_WAVES * bins; i += PT_BLOCK) wcnt[i] = 0;

        uint4 t[SR_V];
        bool ok[SR_V];
        uint32_t pos = 0;                             // last run that starts at or before the element
#pragma unroll
        for (int k = 0; k < SR_V; ++k) {
            const uint32_t i = w * (WAVE * SR_V) + k * WAVE + lane;
            ok[k] = i < count;
            const uint32_t e = sb + i;
            if (k == 0) {
                for (uint32_t s2 = search0; s2 >= 1; s2 >>= 1)
                    if (runoff[pos + s2] <= e) pos += s2;
            } else {
                // 64 elements further on: usually the next run or the one after it
                if (runoff[pos + 1] <= e) ++pos;
                if (runoff[pos + 1] <= e) ++pos;
                if (runoff[pos + 1] <= e) {           // short or empty runs in between: search again
                    pos = 0;
                    for (uint32_t s2 = search0; s2 >= 1; s2 >>= 1)
                        if (runoff[pos + s2] <= e) pos += s2;
                }
            }
            if (ok[k]) {
                if (T12) { const Tuple12 x = reinterpret_cast<const Tuple12 *>(r.in)[rbase[pos] + e]; t[k] = make_uint4(x.klo, x.khi, x.rid, 0u); }
                else t[k] = in[rbase[pos] + e];
            }
        }
        __syncthreads();

        uint32_t lrank[SR_V], dig[SR_V];
        uint32_t *mycnt = wcnt + w * bins;
#pragma unroll
        for (int k = 0; k < SR_V; ++k) {
            const uint64_t key = ((uint64_t)t[k].y << 32) | t[k].x;
            const uint32_t d = (uint32_t)(key >> shift) & mask;
            dig[k] = d;
            const uint64_t peers = digit_peers(d, ok[k], bits);
            const uint32_t rank = (uint32_t)__popcll(peers & lt);
            uint32_t old = 0;
            if (ok[k] && rank == 0) {
                old = mycnt[d];
                mycnt[d] = old + (uint32_t)__popcll(peers);
            }
            const int leader = ok[k] ? __ffsll((unsigned long long)peers) - 1 : 0;
            old = __shfl(old, leader, 64);
            lrank[k] = old + rank;
        }
        __syncthreads();

        uint64_t mytotal = 0;
        if (threadIdx.x < bins) {
            uint32_t run = 0;
            for (int ww = 0; ww < PT_WAVES; ++ww) {
                const uint32_t c = wcnt[ww * bins + threadIdx.x];
                wcnt[ww * bins + threadIdx.x] = run;
                run += c;
            }
            mytotal = run;
        }
        const uint64_t ds = block_excl_scan<PT_BLOCK>(mytotal, nullptr, sm);
        if (threadIdx.x < bins) {
            dstart[threadIdx.x] = (uint32_t)ds;
            const uint32_t gb = gbase[threadIdx.x];
            delta[threadIdx.x] = gb - (uint32_t)ds;                 // mod 2^32
            gbase[threadIdx.x] = gb + (uint32_t)mytotal;
        }
        __syncthreads();

#pragma unroll
        for (int k = 0; k < SR_V; ++k)
            if (ok[k]) stage[dstart[dig[k]] + mycnt[dig[k]] + lrank[k]] = t[k];
        __syncthreads();

#pragma unroll
        for (int k = 0; k < SR_V; ++k) {
            const uint32_t p = k * PT_BLOCK + threadIdx.x;
            if (p < count) {
                const uint4 v = stage[p];
                const uint32_t d = (uint32_t)((((uint64_t)v.y << 32) | v.x) >> shift) & mask;
                if (O12) reinterpret_cast<Tuple12 *>(r.out)[delta[d] + p] = Tuple12{v.x, v.y, v.z};
                else out[delta[d] + p] = v;
            }
        }
        __syncthreads();
    }
    }   // grid-stride loop
}

#endif

}  // namespace rhj
