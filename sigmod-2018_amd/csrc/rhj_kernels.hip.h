// rhj_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the radix hash
// join and the filter scan.  Integer / indexing work only: the roofline is HBM.
//
// Reference loops these kernels replace (file:line in VagelisN/Sigmod-2018):
//   k_hist_tiles     HistJob                      preprocess.c:181-195
//   k_scan_*         hist merge + psum            preprocess.c:83-102 / :328-340
//   k_scatter_lds    SerialReorderArray scatter   preprocess.c:349-359 (stable)
//   k_plan           bucket loop / side choice    rhjoin.c:79-102 (>= picks the probe side)
//   k_build_lds / k_build_hbm   InitIndex + CreateIndex   rhjoin.c:253-273, :219-250
//   k_probe          GetResults                   rhjoin.c:141-217
//   k_filter_*       Filter                       filter.c:110-183
//
// Partition.  A pass handles at most 8 digit bits: a workgroup ranks a 4096-tuple
// tile stably (wave match-any ballots + per-wave LDS counters), stages it in LDS in
// digit order and writes every digit run as full 16-byte-per-lane coalesced stores
// (a direct 4096-way scatter of 16-byte tuples measured 2.6x write amplification and
// 834 GB/s on MI355X: profiles/r01a).  Wider radixes (9..12 bits) run two stable LSD
// passes (low half, then high half), which yields the same stable order as one pass.
//
// Hash index.  The reference chains bucket positions in DESCENDING order behind a
// prime-modulus slot (CreateIndex walks last->first and appends at the tail), which
// is what fixes the order of duplicate matches.  Here each bucket's index is an
// ORDERED linear-probing table (Amble & Knuth): an entry is (tag | position+1),
// inserted with atomic max so that along every probe run entries are in descending
// (tag, position) order.  The final table is the same for every insertion
// interleaving (deterministic), a walk from a key's home slot meets that key's
// duplicates in descending position — the reference's chain order — and can stop at
// the first entry whose tag is smaller.  Tags only pre-filter: every candidate is
// verified against the build tuple's full 64-bit key, so results are exact.
// Tables live in HBM (L2-resident while their bucket is being probed):
//   * 32-bit entries (16-bit tag, 16-bit position), built in LDS by one workgroup per
//     bucket and dumped, when the bucket's build side has <= lds_cap tuples;
//   * 64-bit entries (32-bit tag, 32-bit position), built with global atomics, else.
// Probe units are tiles of 2048 probe tuples in canonical order; any number of
// workgroups share a bucket's table, so a hot bucket costs no extra build work.
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
    uint32_t         pad;
    const uint8_t   *dig_in;     // pass 2 of the run form: this pass' digit per input tuple, written by pass 1
    uint8_t         *dig_out;    // pass 1 of the run form: the next pass' digit per output tuple
    // two-pass partition (run form): pass 1 partitions every tile in place and leaves a run table;
    // a pass-2 tile is `group` consecutive pass-1 tiles' runs of one pass-1 digit
    uint16_t        *runs;       // [bins1 + 1][tiles1] start of each digit's run inside its pass-1 tile; row bins1 = the tile's count
    uint32_t         tiles1;     // pass-1 tiles
    uint32_t         group;      // pass-1 tiles per pass-2 tile
    uint32_t         groups;     // pass-2 tiles per pass-1 digit = ceil(tiles1 / group); tiles = bins1 * groups in pass 2
    uint32_t         pad2;
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
    uint64_t matches;            // filled by k_offsets / k_fused_total
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
    uint32_t         pad;
};

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

__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t *total)
{
    const int lane = threadIdx.x & 63;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

template <int NT>
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t *total, uint64_t *sm /*NT/64+1*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();                       // sm reuse across calls
    if (lane == 63) sm[w] = x;
    __syncthreads();
    if (threadIdx.x < 64) {                // one wave scans the wave totals
        const uint64_t t = threadIdx.x < NT / 64 ? sm[threadIdx.x] : 0;
        uint64_t y = t;
#pragma unroll
        for (int d = 1; d < NT / 64; d <<= 1) {
            const uint64_t z = __shfl_up(y, d, 64);
            if ((int)threadIdx.x >= d) y += z;
        }
        if (threadIdx.x < NT / 64) sm[threadIdx.x] = y - t;
        if (threadIdx.x == NT / 64 - 1) sm[NT / 64] = y;
    }
    __syncthreads();
    if (total) *total = sm[NT / 64];
    return sm[w] + x - v;
}

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

// Per-tile digit histogram of the one-pass partition: cnt[tile][digit] for digit = (key >> shift) & mask.
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

// Bucket histogram (u32, from k_hist_tiles' atomics) -> u64 hist + exclusive psum.
__global__ __launch_bounds__(1024) void k_full_psum(int bits, const uint32_t *full_hist, uint64_t *hist, uint64_t *psum)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    const uint32_t bins = 1u << bits;
    const uint32_t *f = full_hist + (size_t)blockIdx.x * bins;
    uint64_t *h = hist + (size_t)blockIdx.x * bins, *p = psum + (size_t)blockIdx.x * bins;
    const uint32_t per = (bins + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per;
    uint64_t mine = 0;
    for (uint32_t b = b0; b < min(b0 + per, bins); ++b) mine += f[b];
    uint64_t base = block_excl_scan<1024>(mine, nullptr, sm);
    for (uint32_t b = b0; b < min(b0 + per, bins); ++b) {
        h[b] = f[b];
        p[b] = base;
        base += f[b];
    }
}

// One stable partition pass on digit = (key >> shift) & ((1 << bits) - 1), bits <= 8.
// Tile order in memory is (wave, round, lane); a tuple's stable rank inside its digit is
//   digit_start + (same digit in earlier waves) + (same digit in earlier rounds of this
//   wave) + (same digit in lower lanes of this round)
// computed with one match-any (bits ballots) per round and per-wave LDS counters — no
// atomics, so the placement does not depend on any hardware ordering.
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
    const uint64_t ds = block_excl_scan<PT_BLOCK>(mytotal, nullptr, sm);
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
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint32_t p = k * PT_BLOCK + threadIdx.x;
        if (p < count) {
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

// first and last 2048 row ids of both relations -> summary->wide_row_ids (the host cleared both words)
__global__ __launch_bounds__(256) void k_rowid_sample(RelArgs r0, RelArgs r1, int nrel, int force_wide, PlanSummary *summary)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;       // 8 workgroups: 2048 positions from each end
    uint32_t mine = force_wide ? 1u : 0u;
    for (int rel = 0; rel < nrel; ++rel) {
        const RelArgs &r = rel ? r1 : r0;
        if (j < r.n) mine |= (uint32_t)(r.in[j].row_id >> 32) | (uint32_t)(r.in[r.n - 1 - j].row_id >> 32);
    }
    if (__ballot(mine != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(&summary->wide_row_ids, 1u);
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
__global__ __launch_bounds__(PT_BLOCK) void k_local_part(RelArgs r0, RelArgs r1, int shift, int bits, int next_shift, int next_bits,
                                                         PlanSummary *summary)
{
    const bool T12 = summary->wide_row_ids == 0;          // 12-byte intermediates (workgroup-uniform)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4    *stage = reinterpret_cast<uint4 *>(smem);                        // [PT_TILE]
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + (size_t)PT_TILE * 16); // [PT_WAVES][bins]
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t *dstart = wcnt + PT_WAVES * bins;                                 // [bins]
    uint64_t *sm = reinterpret_cast<uint64_t *>(dstart + 2 * bins);            // scan scratch

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
        const uint64_t key = ((uint64_t)t[k].y << 32) | t[k].x;   // (a 32-bit form makes hipcc spill t[] to scratch:
        const uint32_t d = (uint32_t)(key >> shift) & mask;       //  shift is a run-time 0 for that reason)
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
    const uint64_t ds = block_excl_scan<PT_BLOCK>(mytotal, nullptr, sm);
    if (threadIdx.x < bins) dstart[threadIdx.x] = (uint32_t)ds;
    if (threadIdx.x <= bins) r.runs[(size_t)threadIdx.x * r.tiles + tile] = (uint16_t)(threadIdx.x < bins ? (uint32_t)ds : count);
    __syncthreads();

#pragma unroll
    for (int k = 0; k < PT_V; ++k)
        if (ok[k]) stage[dstart[dig[k]] + mycnt[dig[k]] + lrank[k]] = t[k];
    __syncthreads();

    uint4 *out = reinterpret_cast<uint4 *>(r.out) + beg;
    Tuple12 *out12 = reinterpret_cast<Tuple12 *>(r.out) + beg;
    uint8_t *dg = r.dig_out + beg;
    bool wide = false;
#pragma unroll
    for (int k = 0; k < PT_V; ++k) {
        const uint32_t p = k * PT_BLOCK + threadIdx.x;
        if (p < count) {
            const uint4 v = stage[p];
            if (T12) { out12[p] = Tuple12{v.x, v.y, v.z}; wide = wide || v.w != 0; }
            else out[p] = v;
            dg[p] = (uint8_t)((v.x >> next_shift) & ((1u << next_bits) - 1u));
        }
    }
    if (T12 && __ballot(wide) != 0 && (threadIdx.x & 63) == 0) atomicOr(&summary->row_id_overflow, 1u);
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

// cnt[tile2][digit] of pass 2 from the digit bytes pass 1 wrote.  One WAVE per pass-2 tile, no
// workgroup barrier: the lanes hold the run table, every run is one 64-byte load of the whole wave,
// sixteen runs' loads are in flight before their LDS atomics.  (A workgroup per tile was bound by its
// chain of dependent latencies: 6 us per tile, 0.32 ms for 100M + 100M tuples.)
constexpr int HR_BLOCK = 256;
__global__ __launch_bounds__(HR_BLOCK) void k_hist_runs(RelArgs r0, RelArgs r1, int bits)
{
    extern __shared__ uint32_t lds_u32[];
    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t bins = 1u << bits;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t *h = lds_u32 + w * bins;                 // this wave's histogram
    const uint32_t stride = gridDim.x * (HR_BLOCK / WAVE);
    for (uint32_t tile2 = blockIdx.x * (HR_BLOCK / WAVE) + w; tile2 < r.tiles; tile2 += stride) {
        for (uint32_t b = lane; b < bins; b += WAVE) h[b] = 0;
        for (uint32_t c0 = 0; c0 < r.group; c0 += WAVE) {
            uint32_t phys, len;
            pt_run_of(r, tile2, c0 + lane, phys, len);
            const uint32_t nrun = min((uint32_t)WAVE, r.group - c0);
            for (uint32_t q0 = 0; q0 < nrun; q0 += 16) {
                uint32_t dg[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)phys, (int)((q0 + q) & 63u));
                    const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)((q0 + q) & 63u));
                    dg[q] = (q0 + q < nrun && lane < l) ? r.dig_in[p + lane] : 0xffffffffu;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (dg[q] != 0xffffffffu) atomicAdd(&h[dg[q]], 1u);
                    const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)((q0 + q) & 63u));
                    if (q0 + q < nrun && l > WAVE) {          // skewed keys: a run longer than one load
                        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)phys, (int)((q0 + q) & 63u));
                        for (uint32_t e = lane + WAVE; e < l; e += WAVE) atomicAdd(&h[r.dig_in[p + e]], 1u);
                    }
                }
            }
        }
        uint32_t *row = r.cnt + (size_t)tile2 * bins;
        for (uint32_t b = lane; b < bins; b += WAVE) row[b] = h[b];
    }
}

// bucket histogram of the full radix = column sums of pass 2's counts per pass-1 digit
// (grid: pass-1 digits x relations; 1024 threads = digits x slices of the tile groups)
__global__ __launch_bounds__(1024) void k_full_from_cnt(RelArgs r0, RelArgs r1, int bits1, int bits, uint32_t *full_hist)
{
    __shared__ uint32_t part[1024];
    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t bins = 1u << bits, d = blockIdx.x;
    const uint32_t b = threadIdx.x & (bins - 1u), slice = threadIdx.x >> bits, slices = 1024u >> bits;
    uint32_t s = 0;
    const uint32_t *base = r.cnt + (size_t)d * r.groups * bins + b;
#pragma unroll 4
    for (uint32_t j = slice; j < r.groups; j += slices) s += base[(size_t)j * bins];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < bins) {
        uint32_t t = 0;
        for (uint32_t q = 0; q < slices; ++q) t += part[q * bins + threadIdx.x];
        full_hist[((size_t)blockIdx.y << (bits1 + bits)) + ((threadIdx.x << bits1) | d)] = t;
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
        if (threadIdx.x < bins) ngb = r.cnt[(size_t)t_first * bins + threadIdx.x];
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
            if (threadIdx.x < bins) ngb = r.cnt[(size_t)nt * bins + threadIdx.x];
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

// ----------------------------------------------------------------------- plan

constexpr int PR_BLOCK = 256;                     // probe workgroup
constexpr int PR_V = 4;                           // probe tuples per thread
constexpr int PR_UNIT = PR_BLOCK * PR_V;          // 2048 probe tuples per unit

struct PlanArgs {
    const uint64_t *histR, *histS;
    Unit           *units, *build_units;
    uint32_t       *lds_buckets;    // list of buckets whose table is built in LDS
    BucketMeta     *meta;
    PlanSummary    *summary;
    uint32_t        lds_cap;        // largest build side served by an LDS-built table
    uint32_t        lds_max_slots;  // LDS slot budget
    uint32_t        build_chunk;    // build tuples per 64-bit-table build unit
    uint32_t        span_lds;       // probe tuples per unit in LDS-table buckets (PR_UNIT on the tiled path)
};

constexpr uint32_t T32_PAD = 8;         // replica of the first 8 entries behind every 32-bit table

__device__ __forceinline__ uint32_t lds_slots_for(uint64_t bc, uint32_t max_slots)
{
    uint32_t s = (uint32_t)(bc + (bc >> 1)) + 4u;          // load factor <= 2/3 when it fits
    s = (s + 3u) & ~3u;                                    // 16-byte dump granule
    return min(max(s, 64u), max_slots);
}

__device__ __forceinline__ void plan_body(const PlanArgs &a, int bits, uint64_t *sm /*1024 / 64 + 1*/, unsigned long long *red /*2*/)
{
    const uint32_t bins = 1u << bits;
    const uint32_t per = (bins + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, bins);
    if (threadIdx.x < 2) red[threadIdx.x] = 0;

    uint64_t nu = 0, nbu = 0, slots64 = 0, nlds = 0, slots32 = 0;
    uint32_t max_build = 0, max_slots = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const uint64_t cR = a.histR[b], cS = a.histS[b];
        if (cR == 0 || cS == 0) continue;
        const uint64_t pc = cR >= cS ? cR : cS, bc = cR >= cS ? cS : cR;   // rhjoin.c:86 (>=)
        const uint64_t span = bc <= a.lds_cap ? a.span_lds : PR_UNIT;
        nu += (pc + span - 1) / span;
        max_build = max(max_build, (uint32_t)min(bc, (uint64_t)0xffffffffu));
        if (bc <= a.lds_cap) {
            const uint32_t s = lds_slots_for(bc, a.lds_max_slots);
            nlds += 1; slots32 += s + T32_PAD;
            max_slots = max(max_slots, s);
        } else {
            nbu += (bc + a.build_chunk - 1) / a.build_chunk;
            slots64 += 1ull << (64 - __clzll((unsigned long long)(2 * bc - 1)));   // pow2 >= 2*bc
        }
    }
    uint64_t tot_u, tot_b, tot_s64, tot_l, tot_s32;
    uint64_t ubase = block_excl_scan<1024>(nu, &tot_u, sm);
    uint64_t bbase = block_excl_scan<1024>(nbu, &tot_b, sm);
    uint64_t s64base = block_excl_scan<1024>(slots64, &tot_s64, sm);
    uint64_t lbase = block_excl_scan<1024>(nlds, &tot_l, sm);
    uint64_t s32base = block_excl_scan<1024>(slots32, &tot_s32, sm);
    atomicMax(&red[0], (unsigned long long)max_build);
    atomicMax(&red[1], (unsigned long long)max_slots);

    for (uint32_t b = b0; b < b1; ++b) {
        const uint64_t cR = a.histR[b], cS = a.histS[b];
        BucketMeta m = {0, 0, 0};
        if (cR != 0 && cS != 0) {
            const uint64_t pc = cR >= cS ? cR : cS, bc = cR >= cS ? cS : cR;
            if (bc <= a.lds_cap) {
                m.slots = lds_slots_for(bc, a.lds_max_slots);
                m.mode = 1;
                m.table_off = s32base;
                s32base += m.slots + T32_PAD;
                a.lds_buckets[lbase++] = b;
            } else {
                const uint32_t lg = 64 - __clzll((unsigned long long)(2 * bc - 1));
                m.slots = lg;
                m.mode = 2;
                m.table_off = s64base;
                s64base += 1ull << lg;
                for (uint64_t o = 0; o < bc; o += a.build_chunk) {
                    Unit u; u.off = o; u.bucket = b; u.count = (uint32_t)min((uint64_t)a.build_chunk, bc - o);
                    a.build_units[bbase++] = u;
                }
            }
            const uint64_t span = bc <= a.lds_cap ? a.span_lds : PR_UNIT;
            for (uint64_t o = 0; o < pc; o += span) {
                Unit u; u.off = o; u.bucket = b; u.count = (uint32_t)min(span, pc - o);
                a.units[ubase++] = u;
            }
        }
        a.meta[b] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        PlanSummary s;
        s.units = tot_u; s.build_units = tot_b; s.hbm_slots = tot_s64; s.lds_buckets = tot_l;
        s.tab32_slots = tot_s32; s.max_lds_slots = red[1]; s.max_build = red[0]; s.matches = 0;
        s.fused_ok = tot_b == 0;
        s.wide_row_ids = a.summary->wide_row_ids; s.row_id_overflow = a.summary->row_id_overflow;     // the partition's words
        *a.summary = s;
    }
}

__global__ __launch_bounds__(1024) void k_plan(PlanArgs a, int bits)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    __shared__ unsigned long long red[2];
    plan_body(a, bits, sm, red);
}

// Small joins (one-pass partition, at most SMALL_TILES tiles per relation: 1M x 1M has 245): the four scan kernels of
// the partition and the plan in ONE single-workgroup launch — such a join is bound by its launches, not by its bytes.
// Per relation: thread (slice, digit) sums its slice of the tiles' counts, the digits' totals are scanned, and the
// same thread turns its counts into start offsets in place; then the plan over the two histograms.
constexpr uint32_t SMALL_TILES = 1024;
__global__ __launch_bounds__(1024) void k_small_scan_plan(RelArgs r0, RelArgs r1, int bits, uint64_t *hist, uint64_t *psum, PlanArgs a)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    __shared__ unsigned long long red[2];
    __shared__ uint32_t part[1024];
    __shared__ uint32_t base_sh[2][256];
    const uint32_t bins = 1u << bits;
    // threads 0..511 take R, 512..1023 S: thread (slice, digit) of its half
    const uint32_t rel = threadIdx.x >> 9, t = threadIdx.x & 511u;
    const RelArgs &r = rel ? r1 : r0;
    const uint32_t d = t & (bins - 1u), slice = t >> bits, slices = 512u >> bits;
    const uint32_t per = (r.tiles + slices - 1u) / slices;
    const uint32_t t0 = min(slice * per, r.tiles), t1 = min(t0 + per, r.tiles);
    uint32_t *col = r.cnt + d;
    uint32_t acc = 0;
#pragma unroll 16
    for (uint32_t i = t0; i < t1; ++i) acc += col[(size_t)i * bins];
    part[threadIdx.x] = acc;
    __syncthreads();
    uint64_t tot = 0;                                 // threads 0..bins-1: R's digits, 512..512+bins-1: S's
    if (t < bins)
        for (uint32_t q = 0; q < slices; ++q) tot += part[rel * 512u + q * bins + t];
    // one scan over both halves: S's digits sit behind R's, so take R's total off again
    uint64_t all;
    const uint64_t ex = block_excl_scan<1024>(tot, &all, sm);
    if (t < bins) {
        const uint64_t e = rel ? ex - r0.n : ex;      // exclusive prefix inside S = prefix over both - all of R
        hist[(size_t)rel * bins + t] = tot;
        psum[(size_t)rel * bins + t] = e;
        base_sh[rel][t] = (uint32_t)e;
    }
    __syncthreads();
    uint32_t run = base_sh[rel][d];
    for (uint32_t q = 0; q < slice; ++q) run += part[rel * 512u + q * bins + d];
    uint32_t c[16];
    for (uint32_t i0 = t0; i0 < t1; i0 += 16) {       // sixteen loads in flight, then their stores
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = i0 + j < t1 ? col[(size_t)(i0 + j) * bins] : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (i0 + j < t1) col[(size_t)(i0 + j) * bins] = run;
            run += c[j];
        }
    }
    __syncthreads();
    plan_body(a, bits, sm, red);
}

// ---------------------------------------------------------------- hash tables

// 64-bit table insert (build side too large for LDS): one thread per build tuple.
__global__ __launch_bounds__(256) void k_build_hbm(JoinArgs a, const Unit *build_units)
{
    if (blockIdx.x >= a.summary->build_units) return;
    const Unit un = build_units[blockIdx.x];
    const uint32_t b = un.bucket;
    const bool flip = a.histR[b] < a.histS[b];
    const rhj_tuple *bd = flip ? a.partR + a.psumR[b] : a.partS + a.psumS[b];
    const BucketMeta m = a.meta[b];
    unsigned long long *tbl = (unsigned long long *)(a.tab64 + m.table_off);
    const uint32_t lg = m.slots;
    const uint64_t smask = (1ull << lg) - 1ull;
    for (uint32_t i = threadIdx.x; i < un.count; i += 256) {
        const uint64_t pos = un.off + i;
        const uint64_t h = mix64(bd[pos].value);
        uint64_t s = h >> (64 - lg);
        unsigned long long v = ((unsigned long long)(uint32_t)h << 32) | (unsigned long long)(pos + 1);
        for (;;) {
            const unsigned long long old = atomicMax(&tbl[s], v);
            if (old == 0) break;
            if (old < v) v = old;            // displaced entry carries on
            s = (s + 1) & smask;
        }
    }
}

__device__ __forceinline__ uint32_t t32_home(uint64_t h, uint32_t slots) { return __umulhi((uint32_t)(h >> 32), slots); }
__device__ __forceinline__ uint32_t t32_tag(uint64_t h) { return (uint32_t)(h >> 16) & 0xffffu; }

// 32-bit table: one workgroup per bucket builds it in LDS and dumps it to the arena.
constexpr int BL_BLOCK = 1024;
constexpr int BL_V = 4;
__global__ __launch_bounds__(BL_BLOCK) void k_build_lds(JoinArgs a, const uint32_t *lds_buckets)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];
    if (blockIdx.x >= a.summary->lds_buckets) return;
    const uint32_t b = lds_buckets[blockIdx.x];
    const uint64_t cR = a.histR[b], cS = a.histS[b];
    const bool flip = cR < cS;
    const rhj_tuple *bd = flip ? a.partR + a.psumR[b] : a.partS + a.psumS[b];
    const uint32_t bc = (uint32_t)(flip ? cR : cS);
    const BucketMeta m = a.meta[b];
    const uint32_t slots = m.slots;
    for (uint32_t s = threadIdx.x; s < slots; s += BL_BLOCK) tbl[s] = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < bc; i0 += BL_BLOCK * BL_V) {
        uint64_t key[BL_V];
#pragma unroll
        for (int k = 0; k < BL_V; ++k) {
            const uint32_t i = i0 + k * BL_BLOCK + threadIdx.x;
            key[k] = i < bc ? bd[i].value : 0;
        }
#pragma unroll
        for (int k = 0; k < BL_V; ++k) {
            const uint32_t i = i0 + k * BL_BLOCK + threadIdx.x;
            if (i >= bc) continue;
            const uint64_t h = mix64(key[k]);
            uint32_t s = t32_home(h, slots);
            uint32_t v = (t32_tag(h) << 16) | (i + 1u);
            for (;;) {
                const uint32_t old = atomicMax(&tbl[s], v);
                if (old == 0) break;
                if (old < v) v = old;
                s = s + 1 == slots ? 0 : s + 1;
            }
        }
    }
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(a.tab32 + m.table_off);     // table_off and slots are multiples of 4
    const uint4 *src = reinterpret_cast<const uint4 *>(tbl);
    for (uint32_t s = threadIdx.x; s < slots / 4; s += BL_BLOCK) dst[s] = src[s];
    if (threadIdx.x < T32_PAD / 4) dst[slots / 4 + threadIdx.x] = src[threadIdx.x];   // wrap-free chunk reads
}

// 32-bit table in HBM.  A probe reads the two 16-byte-aligned groups of four entries from
// the one that holds its home slot (unaligned 16-byte loads are split by the texture
// addresser and measured ~4x its cycles); entries in front of the home slot are skipped.
// The dump carries 8 padding entries (a replica of the first 8) so no read wraps.
struct Tab32 {
    typedef uint32_t slot_t;
    typedef uint32_t entry_t;
    static constexpr int CH = 8;
    const uint32_t *t;
    uint32_t  slots;
    __device__ __forceinline__ slot_t home(uint64_t h) const { return t32_home(h, slots); }
    __device__ __forceinline__ uint32_t tag(uint64_t h) const { return t32_tag(h); }
    __device__ __forceinline__ slot_t advance(slot_t s, uint32_t by) const { s += by; return s >= slots ? s - slots : s; }
    __device__ __forceinline__ entry_t load(slot_t s) const { return t[s]; }
    __device__ __forceinline__ uint32_t skip(slot_t s) const { return s & 3u; }
    __device__ __forceinline__ void load_chunk(slot_t s, entry_t (&e)[CH]) const
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(t + (s & ~3u));
        const uint4 a = g[0], b = g[1];
        e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
    }
    __device__ __forceinline__ bool live(entry_t e, uint32_t tg) const { return e != 0 && (e >> 16) >= tg; }
    __device__ __forceinline__ bool hit(entry_t e, uint32_t tg) const { return (e >> 16) == tg; }
    __device__ __forceinline__ uint32_t pos(entry_t e) const { return (e & 0xffffu) - 1u; }
};

struct Tab64 {
    typedef uint64_t slot_t;
    typedef uint64_t entry_t;
    static constexpr int CH = 4;
    const uint64_t *t;
    uint32_t  lg;
    __device__ __forceinline__ slot_t home(uint64_t h) const { return h >> (64 - lg); }
    __device__ __forceinline__ uint32_t tag(uint64_t h) const { return (uint32_t)h; }
    __device__ __forceinline__ slot_t advance(slot_t s, uint32_t by) const { return (s + by) & ((1ull << lg) - 1ull); }
    __device__ __forceinline__ entry_t load(slot_t s) const { return t[s]; }
    __device__ __forceinline__ uint32_t skip(slot_t) const { return 0; }
    __device__ __forceinline__ void load_chunk(slot_t s, entry_t (&e)[CH]) const
    {
#pragma unroll
        for (int j = 0; j < CH; ++j) e[j] = t[advance(s, j)];
    }
    __device__ __forceinline__ bool live(entry_t e, uint32_t tg) const { return e != 0 && (uint32_t)(e >> 32) >= tg; }
    __device__ __forceinline__ bool hit(entry_t e, uint32_t tg) const { return (uint32_t)(e >> 32) == tg; }
    __device__ __forceinline__ uint32_t pos(entry_t e) const { return (uint32_t)e - 1u; }
};

__device__ __forceinline__ uint4 make_pair(bool flip, uint32_t prl, uint32_t prh, uint32_t bl, uint32_t bh)
{
    return flip ? make_uint4(bl, bh, prl, prh) : make_uint4(prl, prh, bl, bh);
}

// Probe one unit (<= PR_UNIT probe tuples, memory order (wave, round, lane)).
//
// Count pass (WRITE = false): every tag-matching candidate is verified against the build
// tuple's 64-bit key; the unit's verified match count goes to unit_count[u], and
// unit_flag[u] records whether ANY candidate failed verification (a 16/32-bit tag
// collision between different keys: rare).
// Emit pass (WRITE = true): in a unit without such a collision every candidate is a
// match, so the offsets follow from the candidate counts alone and the pairs are
// written in one sweep (probe order; per probe tuple in table-walk order = descending
// build position, rhjoin.c:227,240-246).  A flagged unit re-verifies while it emits.
//
// Loads are issued phase by phase before the first is consumed: probe tuples, one
// 8-slot table chunk per tuple, then the candidates' build tuples.
template <bool WRITE, class Table>
__device__ __forceinline__ void probe_unit(const JoinArgs &a, const Table &T, const rhj_tuple *pr,
                                           const rhj_tuple *bd, uint32_t count, bool flip, uint32_t u, uint32_t *wsum)
{
    constexpr int CH = Table::CH;
    typedef typename Table::slot_t slot_t;
    typedef typename Table::entry_t entry_t;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint4 *pr4 = reinterpret_cast<const uint4 *>(pr);
    const uint4 *bd4 = reinterpret_cast<const uint4 *>(bd);
    const uint2 *bd2 = reinterpret_cast<const uint2 *>(bd);
    const bool exact = WRITE ? a.unit_flag[u] != 0 : a.ablate == 0;      // verify candidates?

    uint4 q[PR_V];
    bool ok[PR_V];
#pragma unroll
    for (int k = 0; k < PR_V; ++k) {
        const uint32_t i = w * (WAVE * PR_V) + k * WAVE + lane;
        ok[k] = i < count;
        q[k] = ok[k] ? pr4[i] : make_uint4(0, 0, 0, 0);
    }
    slot_t s0[PR_V];
    uint32_t tg[PR_V];
    entry_t e[PR_V][CH];
#pragma unroll
    for (int k = 0; k < PR_V; ++k) {
        const uint64_t h = mix64(((uint64_t)q[k].y << 32) | q[k].x);
        s0[k] = T.home(h);
        tg[k] = T.tag(h);
        if (ok[k] && a.ablate != 2) T.load_chunk(s0[k], e[k]);
        else {
#pragma unroll
            for (int j = 0; j < CH; ++j) e[k][j] = 0;
        }
    }
    uint32_t hm[PR_V];                  // chunk entries that carry this key's tag
    uint32_t p0[PR_V], p1[PR_V];        // build positions of the first two of them
    bool more[PR_V];                    // the run continues past the chunk
#pragma unroll
    for (int k = 0; k < PR_V; ++k) {
        uint32_t mask = 0, a0 = 0, a1 = 0;
        bool live = true;
        const uint32_t sk = T.skip(s0[k]);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const bool in = (uint32_t)j >= sk;               // at or behind the home slot
            live = live && (!in || T.live(e[k][j], tg[k]));
            if (in && live && T.hit(e[k][j], tg[k])) {
                if (mask == 0) a0 = T.pos(e[k][j]);
                else if ((mask & (mask - 1)) == 0) a1 = T.pos(e[k][j]);
                mask |= 1u << j;
            }
        }
        hm[k] = mask; p0[k] = a0; p1[k] = a1;
        more[k] = live;
    }

    // ---- matches per probe tuple
    uint32_t m[PR_V];
    bool fp = false;                    // a candidate failed verification (count pass)
    if (!exact) {
        // emit pass of a collision-free unit: candidates == matches
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            uint32_t c = (uint32_t)__popc(hm[k]);
            if (more[k]) {
                slot_t s = T.advance(s0[k], CH - T.skip(s0[k]));
                entry_t x = T.load(s);
                while (T.live(x, tg[k])) { c += T.hit(x, tg[k]); s = T.advance(s, 1); x = T.load(s); }
            }
            m[k] = c;
        }
    } else {
        // first two candidates of every tuple: gather all, then compare
        uint2 g0[PR_V], g1[PR_V];
        uint32_t rest[PR_V];
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            uint32_t r = hm[k];
            g0[k] = make_uint2(0, 0); g1[k] = make_uint2(0, 0);
            if (r) { r &= r - 1; g0[k] = bd2[2 * (size_t)p0[k]]; }
            if (r) { r &= r - 1; g1[k] = bd2[2 * (size_t)p1[k]]; }
            rest[k] = r;
        }
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            const uint32_t nc = (uint32_t)__popc(hm[k]);
            const bool eq0 = nc >= 1 && g0[k].x == q[k].x && g0[k].y == q[k].y;
            const bool eq1 = nc >= 2 && g1[k].x == q[k].x && g1[k].y == q[k].y;
            uint32_t c = (uint32_t)eq0 + (uint32_t)eq1;
            fp = fp || (nc >= 1 && !eq0) || (nc >= 2 && !eq1);
            if (rest[k]) {                                // third and later candidates of the chunk
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if ((rest[k] >> j) & 1u) {
                        const uint2 v = bd2[2 * (size_t)T.pos(e[k][j])];
                        const bool eq = v.x == q[k].x && v.y == q[k].y;
                        c += eq; fp = fp || !eq;
                    }
                }
            }
            if (more[k]) {                                // run longer than the chunk
                slot_t s = T.advance(s0[k], CH - T.skip(s0[k]));
                entry_t x = T.load(s);
                while (T.live(x, tg[k])) {
                    if (T.hit(x, tg[k])) {
                        const uint2 v = bd2[2 * (size_t)T.pos(x)];
                        const bool eq = v.x == q[k].x && v.y == q[k].y;
                        c += eq; fp = fp || !eq;
                    }
                    s = T.advance(s, 1);
                    x = T.load(s);
                }
            }
            m[k] = c;
        }
    }

    // ---- offsets in (wave, round, lane) order
    uint32_t off[PR_V], run = 0;
#pragma unroll
    for (int k = 0; k < PR_V; ++k) {
        uint32_t tot;
        off[k] = run + wave_excl_scan_u32(m[k], &tot);
        run += tot;
    }
    if (lane == 0) wsum[w] = run;
    if (!WRITE) {
        const uint64_t any_fp = __ballot(fp);
        if (lane == 0) wsum[PR_BLOCK / WAVE + w] = any_fp != 0;
    }
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int i = 0; i < PR_BLOCK / WAVE; ++i) {
        const uint32_t v = wsum[i];
        if ((uint32_t)i < w) wbase += v;
        total += v;
    }
    if (!WRITE) {
        if (threadIdx.x == 0) {
            uint32_t f = 0;
#pragma unroll
            for (int i = 0; i < PR_BLOCK / WAVE; ++i) f |= wsum[PR_BLOCK / WAVE + i];
            a.unit_count[u] = total;
            a.unit_flag[u] = f;
        }
        return;
    }

    // ---- emit
    const uint64_t base = a.unit_base[u] + wbase;
    const uint64_t cap = a.out_capacity;
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    if (!exact) {
        // first candidate of every tuple in one batch of gathers, the rest in a loop
        uint2 r0[PR_V];
        uint32_t rest[PR_V];
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            uint32_t r = hm[k];
            r0[k] = make_uint2(0, 0);
            if (r) { r &= r - 1; r0[k] = bd2[2 * (size_t)p0[k] + 1]; }
            rest[k] = r;
        }
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            if (m[k] == 0) continue;
            uint64_t at = base + off[k];
            const uint32_t prl = q[k].z, prh = q[k].w;
            if (hm[k]) { if (at < cap) out[at] = make_pair(flip, prl, prh, r0[k].x, r0[k].y); ++at; }
            if (rest[k]) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if ((rest[k] >> j) & 1u) {
                        const uint2 v = bd2[2 * (size_t)T.pos(e[k][j]) + 1];
                        if (at < cap) out[at] = make_pair(flip, prl, prh, v.x, v.y);
                        ++at;
                    }
                }
            }
            if (more[k]) {
                slot_t s = T.advance(s0[k], CH - T.skip(s0[k]));
                entry_t x = T.load(s);
                while (T.live(x, tg[k])) {
                    if (T.hit(x, tg[k])) {
                        const uint2 v = bd2[2 * (size_t)T.pos(x) + 1];
                        if (at < cap) out[at] = make_pair(flip, prl, prh, v.x, v.y);
                        ++at;
                    }
                    s = T.advance(s, 1);
                    x = T.load(s);
                }
            }
        }
    } else {
        // flagged unit: verify every candidate again while emitting
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            if (m[k] == 0) continue;
            uint64_t at = base + off[k];
            const uint32_t prl = q[k].z, prh = q[k].w;
            slot_t s = s0[k];
            entry_t x = T.load(s);
            while (T.live(x, tg[k])) {
                if (T.hit(x, tg[k])) {
                    const uint4 v = bd4[T.pos(x)];
                    if (v.x == q[k].x && v.y == q[k].y) {
                        if (at < cap) out[at] = make_pair(flip, prl, prh, v.z, v.w);
                        ++at;
                    }
                }
                s = T.advance(s, 1);
                x = T.load(s);
            }
        }
    }
}

template <bool WRITE>
__global__ __launch_bounds__(PR_BLOCK) void k_probe(JoinArgs a)
{
    __shared__ uint32_t wsum[2 * PR_BLOCK / WAVE];
    // XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs, so
    // give XCD x the x-th contiguous eighth of the canonical unit list; the tables and
    // build sides an XCD's L2 has to hold are then those of a handful of adjacent buckets.
    const uint32_t nu = (uint32_t)a.summary->units;
    const uint32_t per = (nu + 7u) / 8u;
    const uint32_t u = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || u >= nu) return;
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const uint64_t cR = a.histR[b], cS = a.histS[b];
    const bool flip = cR < cS;                                         // S is streamed (r_s == 1)
    const rhj_tuple *pr = (flip ? a.partS + a.psumS[b] : a.partR + a.psumR[b]) + un.off;
    const rhj_tuple *bd = flip ? a.partR + a.psumR[b] : a.partS + a.psumS[b];
    const BucketMeta m = a.meta[b];
    if (m.mode == 1) {
        Tab32 T{a.tab32 + m.table_off, m.slots};
        probe_unit<WRITE>(a, T, pr, bd, un.count, flip, u, wsum);
    } else {
        Tab64 T{a.tab64 + m.table_off, m.slots};
        probe_unit<WRITE>(a, T, pr, bd, un.count, flip, u, wsum);
    }
}

// ------------------------------------------------------------- fused LDS join
//
// One persistent workgroup per CU takes units = (bucket, up to FJ_SPAN probe tuples) in canonical
// order through a ticket, so that a unit's predecessors are always running or done.
//   build    CSR slot index of the bucket's build side in LDS (fj_build)
//   phase 1  stream the unit's probe keys; one 8-entry tag window per tuple in LDS; ONE global
//            gather per candidate (key to verify + row id) — from LDS instead when the build
//            tuples fit there too (RES, build side <= ~7 K tuples); per probe tuple stash the
//            match count (u8) and the first match's build row id (u64), further matches go to
//            the overflow stash (fj_count_batch)
//   chain    publish the unit's match total right away (8-byte {flag,value} word per unit,
//            agent-scope relaxed atomics)
//   emit     deferred behind the NEXT unit's build and phase 1: decoupled look-back over the
//            predecessors (never waits by then), then stream probe row ids + stash + overflow
//            stash and write the pairs at their final canonical positions (fj_emit_stream).
//            Units the overflow stash cannot describe emit immediately by walking the index again.
// Random global accesses per probe tuple: one 128-byte line (the gather); everything
// else is streaming or LDS.
constexpr int FJ_BLOCK = 1024;
constexpr int FJ_WAVES = FJ_BLOCK / WAVE;
constexpr int FJ_V = 4;
constexpr int FJ_BATCH = FJ_BLOCK * FJ_V;       // 4096 probe tuples per batch
constexpr uint32_t FJ_SPAN = 65536;             // probe tuples per unit
constexpr uint32_t FJ_LDS_EXTRA = 1024;         // bytes of LDS behind the table
constexpr uint32_t FJ_OVF_CAP = 32768;          // overflow entries per unit before it falls back to the index walk
constexpr uint32_t FJ_OVF_J = 15;               // match ordinals 1..15 (2nd..16th match) have an overflow slot
constexpr uint32_t FJ_GROUPS = FJ_SPAN / 256;   // a group = the 256 tuples one wave counts in one batch

struct FusedArgs {
    JoinArgs  j;
    uint8_t  *stash_cnt;      // [nR + nS] matches per probe tuple, saturating at 255
    uint64_t *stash_row;      // [nR + nS] build row id of the first match
    uint64_t *status;         // [units] (flag << 62) | value ; flag 1 = unit total, 2 = inclusive prefix
    uint32_t *ticket;
    uint64_t  nR;
    uint32_t  allow_resident;
    uint32_t  pad;
    uint64_t *dbg;            // diagnostic builds only: [units][8] phase stamps (100 MHz), else null
    uint64_t *ovf;            // [grid][2][FJ_OVF_CAP] build row ids of second and later matches (per workgroup, double-buffered)
    uint32_t *ovf_base;       // [grid][2][FJ_SPAN / 256][16] first overflow slot of (256-tuple group, match ordinal)
};


// LDS index of the fused kernel: the build positions of a bucket grouped by hash slot (CSR).
//   ent[p]   tag16 << 16 | build position, the entries of one slot contiguous and in DESCENDING value
//            order.  Equal keys have equal tags, so the positions of one key come out descending —
//            the order in which the reference's bucket/chain index hands out the matches of a key
//            (CreateIndex walks last->first and appends at the tail, rhjoin.c:219-250).  Tags are
//            1..0xfffe: 0 is the empty cell during the build, 0xffff the 8 pad entries behind the array.
//   H[s + 1] 16-bit start of slot s in ent[], H[s + 2] its end (two per 32-bit word)
// Built by a counting sort in LDS: count per slot, exclusive scan, then every tuple enters its slot's
// range by ordered insertion (atomicMax on the cell, carry the smaller value to the next cell: the
// range ends up sorted for every interleaving, like the chains this replaces).  A probe reads the
// slot's start and end and a window of 8 entries, compares the 8 tags at once and keeps a bit mask of
// the hits: no pointer chasing and no loop whose trip count is the longest chain of the wave (the
// linked chains spent 2/3 of the probe's vector instructions there).  Slots longer than 8 continue
// window by window.
struct FjIndex {
    uint32_t *ent;       // [bc + 8]
    uint32_t *dirw;      // [(hs + 3) / 2]
    uint32_t  hs;
    __device__ __forceinline__ uint32_t slot(uint64_t h) const { return __umulhi((uint32_t)(h >> 32), hs); }
    __device__ __forceinline__ uint32_t H(uint32_t j) const { return reinterpret_cast<const uint16_t *>(dirw)[j]; }
};
__device__ __forceinline__ uint32_t fj_tag(uint64_t h) { return min((uint32_t)(h >> 16) & 0xffffu, 0xfffdu) + 1u; }

// Partitioned relations as the fused kernel sees them: rhj_tuple (16 B), or — N32: the partition found every row id
// below 2^32 and wrote Tuple12 — 12 bytes per tuple.  Whole tuple as {key lo, key hi, row id lo, row id hi}.
template <bool N32> __device__ __forceinline__ uint4 pt_load(const rhj_tuple *base, uint64_t i)
{
    if (N32) { const Tuple12 x = reinterpret_cast<const Tuple12 *>(base)[i]; return make_uint4(x.klo, x.khi, x.rid, 0u); }
    return reinterpret_cast<const uint4 *>(base)[i];
}
template <bool N32> __device__ __forceinline__ uint2 pt_load_key(const rhj_tuple *base, uint64_t i)
{
    if (N32) { const Tuple12 *x = reinterpret_cast<const Tuple12 *>(base) + i; return make_uint2(x->klo, x->khi); }
    return reinterpret_cast<const uint2 *>(base)[2 * i];
}

// hit mask of the first min(n, 8) entries of the window at `start`
__device__ __forceinline__ uint32_t fj_window(const FjIndex &X, uint32_t start, uint32_t n, uint32_t tgs)
{
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = X.ent[start + j];
    const uint32_t tg = tgs >> 16;
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) m |= ((e[j] >> 16) == tg) ? (1u << j) : 0u;
    return m & ((1u << min(n, 8u)) - 1u);
}

// Per probe tuple: sn = window start | remaining slot length << 16, tm = tag << 16 | hit mask of the
// current window.
__device__ __forceinline__ void fj_lookup(const FjIndex &X, uint64_t key, bool ok, uint32_t &sn, uint32_t &tm)
{
    const uint64_t h = mix64(key);
    const uint32_t s = X.slot(h);
    const uint32_t d0 = X.H(s + 1u), n = ok ? X.H(s + 2u) - d0 : 0u;
    const uint32_t tgs = fj_tag(h) << 16;
    sn = d0 | (n << 16);
    tm = tgs | fj_window(X, d0, n, tgs);
}

// One round of the probe: every tuple that still has a candidate hands out its next one (pos[k], a
// build position) — first from the window's hit mask, and when that is used up and the slot is longer
// than the window, from the next window.  Returns whether any lane of the wave got a candidate.
__device__ __forceinline__ bool fj_round(const FjIndex &X, uint32_t (&sn)[FJ_V], uint32_t (&tm)[FJ_V], uint32_t (&pos)[FJ_V],
                                         bool &last)
{
    bool more = false;
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) more = more || ((tm[k] & 0xffu) == 0 && (sn[k] >> 16) > 8u);
    while (__ballot(more) != 0) {                     // rare: a slot with more than 8 entries
        more = false;
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            if ((tm[k] & 0xffu) == 0 && (sn[k] >> 16) > 8u) {
                sn[k] += 8u - (8u << 16);             // start += 8, length -= 8
                tm[k] |= fj_window(X, sn[k] & 0xffffu, sn[k] >> 16, tm[k] & 0xffff0000u);
                more = more || ((tm[k] & 0xffu) == 0 && (sn[k] >> 16) > 8u);
            }
        }
    }
    bool found = false;
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) {
        pos[k] = 0xffffffffu;
        const uint32_t m = tm[k] & 0xffu;
        if (m != 0) {
            const uint32_t j = (uint32_t)__builtin_ctz(m);
            pos[k] = X.ent[(sn[k] & 0xffffu) + j] & 0xffffu;
            tm[k] &= tm[k] - 1u;                      // the mask sits in the low bits
            found = true;
        }
    }
    bool rest = false;                                // spares the caller a round that finds nothing
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) rest = rest || (tm[k] & 0xffu) != 0 || (sn[k] >> 16) > 8u;
    last = __ballot(rest) == 0;
    return __ballot(found) != 0;
}

// Build the index of one bucket's build side (whole workgroup).  RES: the tuples are copied to LDS
// on the way and the second pass reads them there.  `tmp` is global scratch of at least 4 * bc bytes
// for the cooperative sort of long slots.
constexpr uint32_t FJ_LONG = 16;                      // slots above this are filled by fetch-add and ranked afterwards
constexpr int FJ_SMALL = 4;                           // batches of 4096 build tuples whose (slot, tag) words are kept for the fill pass
template <bool RES, bool N32>
__device__ __forceinline__ void fj_build(const FjIndex &X, const rhj_tuple *part, uint64_t boff, uint32_t bc, uint4 *ltup,
                                         uint32_t *tmp, uint32_t *wsum, uint32_t *sh_pick)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t ndw = (X.hs + 3u) / 2u;
    for (uint32_t i = tid; i < ndw; i += FJ_BLOCK) X.dirw[i] = 0;
    for (uint32_t i = tid; i < bc; i += FJ_BLOCK) X.ent[i] = 0;
    if (tid < 8) X.ent[bc + tid] = 0xffff0000u;
    __syncthreads();
    // ---- count: H[s + 1] += 1.  Build sides of up to 4 batches (16 K tuples) are hashed only once: the
    // (slot, tag) word of tuple i is parked in ent[i], picked up into registers before the fill pass
    // clears the array, and the fill pass needs neither the key nor a second hash.
    const bool small = bc <= FJ_SMALL * FJ_BATCH;
    {
        uint4 t[FJ_V], tn[FJ_V];                       // current and prefetched batch of build tuples
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const uint32_t i = k * FJ_BLOCK + tid;
            t[k] = make_uint4(0, 0, 0, 0);
            if (i < bc) { if (RES) t[k] = pt_load<N32>(part, boff + i); else { const uint2 kv = pt_load_key<N32>(part, boff + i); t[k].x = kv.x; t[k].y = kv.y; } }
        }
        for (uint32_t i0 = 0; i0 < bc; i0 += FJ_BATCH) {
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {               // next batch's loads fly while this batch is counted
                const uint32_t i = i0 + FJ_BATCH + k * FJ_BLOCK + tid;
                tn[k] = make_uint4(0, 0, 0, 0);
                if (i < bc) { if (RES) tn[k] = pt_load<N32>(part, boff + i); else { const uint2 kv = pt_load_key<N32>(part, boff + i); tn[k].x = kv.x; tn[k].y = kv.y; } }
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = i0 + k * FJ_BLOCK + tid;
                if (i < bc) {
                    if (RES) ltup[i] = t[k];
                    const uint64_t h = mix64(((uint64_t)t[k].y << 32) | t[k].x);
                    const uint32_t sl = X.slot(h);
                    if (small) X.ent[i] = (sl << 16) | fj_tag(h);       // parked in the still unused entry array
                    const uint32_t j = sl + 1u;
                    atomicAdd(&X.dirw[j >> 1], (j & 1u) ? 0x10000u : 1u);
                }
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) t[k] = tn[k];
        }
    }
    __syncthreads();
    // ---- exclusive scan over the halfwords: H[s + 1] = start of slot s, H[hs + 1] = bc
    {
        const uint32_t chunk = (ndw + FJ_BLOCK - 1u) / FJ_BLOCK;
        const uint32_t lo = min(tid * chunk, ndw), hi = min(lo + chunk, ndw);
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = X.dirw[i]; sum += (v & 0xffffu) + (v >> 16); }
        uint32_t tot;
        uint32_t run = wave_excl_scan_u32(sum, &tot);
        if (lane == 0) wsum[w] = tot;
        __syncthreads();
        for (uint32_t i = 0; i < w; ++i) run += wsum[i];
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t v = X.dirw[i];
            const uint32_t a0 = run; run += v & 0xffffu;
            const uint32_t a1 = run; run += v >> 16;
            X.dirw[i] = a0 | (a1 << 16);
        }
    }
    __syncthreads();
    // ---- fill.  Ordered insertion into the slot's range: atomicMax on the cell, go on with the
    // smaller of the two values; exactly n values enter n cells, so a carry always finds an empty cell
    // inside the range.  Long slots (many duplicates of one key, where that would be quadratic) take
    // places in arrival order — the range's last cell counts the arrivals until the last arrival
    // overwrites it — and are ranked afterwards.
    bool has_long = false;
    auto insert = [&](uint32_t sl, uint32_t v) {
        const uint32_t a = X.H(sl + 1u), n = X.H(sl + 2u) - a;
        if (n <= FJ_LONG) {
            for (uint32_t p = a;; ++p) {
                const uint32_t old = atomicMax(&X.ent[p], v);
                if (old == 0) break;
                v = min(old, v);
            }
        } else {
            has_long = true;
            const uint32_t arrival = atomicAdd(&X.ent[a + n - 1u], 1u);
            X.ent[a + arrival] = v;                  // arrival n - 1: everybody has counted, the counter cell is free
        }
    };
    if (small) {
        uint32_t hw[FJ_SMALL][FJ_V];
#pragma unroll
        for (int b = 0; b < FJ_SMALL; ++b)
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = (uint32_t)b * FJ_BATCH + k * FJ_BLOCK + tid;
                hw[b][k] = i < bc ? X.ent[i] : 0;
            }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < FJ_SMALL; ++b)
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = (uint32_t)b * FJ_BATCH + k * FJ_BLOCK + tid;
                if (i < bc) X.ent[i] = 0;
            }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < FJ_SMALL; ++b)
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = (uint32_t)b * FJ_BATCH + k * FJ_BLOCK + tid;
                if (i < bc) insert(hw[b][k] >> 16, (hw[b][k] << 16) | i);
            }
    } else {
        uint4 t[FJ_V], tn[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const uint32_t i = k * FJ_BLOCK + tid;
            t[k] = make_uint4(0, 0, 0, 0);
            if (i < bc) { if (RES) t[k] = ltup[i]; else { const uint2 kv = pt_load_key<N32>(part, boff + i); t[k].x = kv.x; t[k].y = kv.y; } }
        }
        for (uint32_t i0 = 0; i0 < bc; i0 += FJ_BATCH) {
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = i0 + FJ_BATCH + k * FJ_BLOCK + tid;
                tn[k] = make_uint4(0, 0, 0, 0);
                if (i < bc) { if (RES) tn[k] = ltup[i]; else { const uint2 kv = pt_load_key<N32>(part, boff + i); tn[k].x = kv.x; tn[k].y = kv.y; } }
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = i0 + k * FJ_BLOCK + tid;
                if (i < bc) {
                    const uint64_t h = mix64(((uint64_t)t[k].y << 32) | t[k].x);
                    insert(X.slot(h), (fj_tag(h) << 16) | i);
                }
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) t[k] = tn[k];
        }
    }
    if (__syncthreads_or(has_long)) {
        // long slots: one at a time, ranked by the whole workgroup (descending value)
        for (uint32_t next = 0;;) {
            if (tid == 0) *sh_pick = 0xffffffffu;
            __syncthreads();
            for (uint32_t sl = tid; sl < X.hs; sl += FJ_BLOCK)
                if (sl >= next && X.H(sl + 2u) - X.H(sl + 1u) > FJ_LONG) { atomicMin(sh_pick, sl); break; }
            __syncthreads();
            const uint32_t pick = *sh_pick;
            if (pick == 0xffffffffu) break;
            const uint32_t a = X.H(pick + 1u), n = X.H(pick + 2u) - a;
            for (uint32_t i = tid; i < n; i += FJ_BLOCK) {
                const uint32_t v = X.ent[a + i];
                uint32_t r = 0;
                for (uint32_t j = 0; j < n; ++j) r += X.ent[a + j] > v;
                tmp[r] = v;
            }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += FJ_BLOCK) X.ent[a + i] = tmp[i];
            next = pick + 1u;
            __syncthreads();
        }
    }
}

// Count the matches of one batch (FJ_V probe tuples per lane): lockstep rounds of
// {walk every live chain to its next tag hit (LDS), fetch those candidates' build tuples
// together, verify the 64-bit keys}.  RES: the build tuples are resident in LDS (no global
// access at all); otherwise each candidate is one 16-byte gather from the bucket's build side.
// Gathers of build tuples go through a buffer descriptor of the bucket's build side and carry
// sc1 (L1 bypass): a gathered line is used once per candidate, so allocating it in the 32 KiB
// vector L1 only evicts the streamed probe data.  A/B on MI355X (tools/ab.py, fused kernel on
// 100Mx100M@12): plain 4.64 ms, nt 4.06 ms, sc1 3.8 ms.
template <bool N32>
struct FjGather {
    __amdgpu_buffer_rsrc_t rsrc;
    static constexpr uint32_t STRIDE = N32 ? 12u : 16u;
    __device__ __forceinline__ void init(const rhj_tuple *part, uint64_t boff, uint32_t bc)
    {
        const uint64_t addr = (uint64_t)part + boff * STRIDE;     // wave-uniform by construction: make it provable
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        const uint32_t bytes = __builtin_amdgcn_readfirstlane(bc * STRIDE);
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
    }
    __device__ __forceinline__ uint4 load(uint32_t pos) const
    {
        if (N32) {
            typedef uint32_t v3 __attribute__((ext_vector_type(3)));
            const v3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(pos * 12u), 0, 16 /* sc1 */);
            return make_uint4(v.x, v.y, v.z, 0u);
        }
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(pos * 16u), 0, 16 /* sc1 */);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};

// Overflow stash of the gather path.  Phase 1 has every match's build row id in registers at the
// moment it verifies it, so besides the first one (stash_row) it keeps the others too: a wave that
// finds second-or-later matches in round r (r = 1, 2, ...; a tuple whose tag hits are all matches
// finds its (r+1)-th match exactly there) takes a contiguous run of slots with one LDS atomic, records
// the run's start for (its 256-tuple group, r) and stores the row ids in (k, lane) order.  The emit
// pass recomputes the same ranks from the stashed counts, so it needs neither the index nor a gather
// and can run any time later.  Tuples that break the rule (a tag hit with a foreign key next to two or
// more matches, more than 16 matches, capacity) make the unit fall back to the index walk.
struct FjOvf {
    uint64_t *buf;        // this unit's overflow entries
    uint32_t *table;      // this unit's [group][16] run starts
    uint32_t *counter;    // LDS bump counter
    uint32_t  gid;        // group of this wave in this batch
};

template <bool RES, bool OVF, bool N32>
__device__ __forceinline__ void fj_count_batch(const FjIndex &X, const FjGather<N32> &G, const uint4 *ltup,
                                               const uint4 (&q)[FJ_V], const bool (&okk)[FJ_V], uint32_t (&c)[FJ_V],
                                               uint32_t (&flo)[FJ_V], uint32_t (&fhi)[FJ_V], bool (&fp)[FJ_V],
                                               const FjOvf &O)
{
    uint32_t sn[FJ_V], tm[FJ_V];
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) {
        fj_lookup(X, ((uint64_t)q[k].y << 32) | q[k].x, okk[k], sn[k], tm[k]);
        c[k] = 0; flo[k] = 0; fhi[k] = 0; fp[k] = false;
    }
    bool last = false;
    for (uint32_t round = 0; !last; ++round) {
        uint32_t pos[FJ_V];
        if (!fj_round(X, sn, tm, pos, last)) break;
        uint4 g[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            g[k] = make_uint4(0, 0, 0, 0);
            if (pos[k] != 0xffffffffu) g[k] = RES ? ltup[pos[k]] : G.load(pos[k]);
        }
        bool ex[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const bool eq = pos[k] != 0xffffffffu && g[k].x == q[k].x && g[k].y == q[k].y;
            if (eq && c[k] == 0) { flo[k] = g[k].z; fhi[k] = g[k].w; }
            ex[k] = OVF && eq && c[k] != 0;
            c[k] += eq;
            fp[k] = fp[k] || (pos[k] != 0xffffffffu && !eq);      // a tag hit with a different key
        }
        if (OVF && round != 0) {
            uint64_t mk[FJ_V];
            uint32_t tot = 0;
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) { mk[k] = __ballot(ex[k]); tot += (uint32_t)__popcll(mk[k]); }
            if (tot != 0) {
                const uint32_t lane = threadIdx.x & 63;
                const uint64_t lt = lanemask_lt();
                uint32_t base = 0;
                if (lane == 0) {
                    base = atomicAdd(O.counter, tot);
                    if (round <= FJ_OVF_J) O.table[O.gid * 16u + round] = base;
                }
                base = __builtin_amdgcn_readfirstlane(base);
                uint32_t pre = base;
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    const uint32_t slot = pre + (uint32_t)__popcll(mk[k] & lt);
                    if (ex[k] && slot < FJ_OVF_CAP) reinterpret_cast<uint2 *>(O.buf)[slot] = make_uint2(g[k].z, g[k].w);
                    pre += (uint32_t)__popcll(mk[k]);
                }
            }
        }
    }
}

// Exclusive prefix of unit u > 0 in the chained scan (called by ONE wave): sums the predecessors'
// words 64 at a time until it meets an inclusive prefix; waits only for aggregates, which every unit
// publishes right after its phase 1.
__device__ __forceinline__ uint64_t fj_lookback(unsigned long long *st, uint32_t u, uint32_t lane)
{
    uint64_t excl = 0;
    int64_t j = (int64_t)u - 1;
    for (;;) {
        const int64_t idx = j - lane;
        unsigned long long v = 2ull << 62;    // virtual "prefix 0" in front of unit 0
        if (idx >= 0) {
            do {
                v = __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 62) == 0) __builtin_amdgcn_s_sleep(2);
            } while ((v >> 62) == 0);
        }
        const uint64_t full = __ballot((v >> 62) == 2);
        const int stop = full ? __ffsll((unsigned long long)full) - 1 : 64;   // nearest inclusive prefix
        uint64_t part = lane <= (uint32_t)stop ? (v & ((1ull << 62) - 1)) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        excl += part;
        if (full) break;
        j -= 64;
    }
    return excl;
}

// The first-match stash is 8 bytes per probe tuple: the build row id — or, when the partition found every row
// id below 2^32 (N32: summary->wide_row_ids == 0; pass 1 checks every tuple and the host runs again wide
// otherwise), the low words of the build AND the probe row id, so that the deferred emit pass does not read
// the probe tuples again.
template <bool N32> __device__ __forceinline__ void fj_stash_put(uint2 *srow, uint32_t i, uint32_t lo, uint32_t hi, uint32_t probe_lo)
{
    srow[i] = N32 ? make_uint2(lo, probe_lo) : make_uint2(lo, hi);
}

// Deferred emit pass of a gather-path unit: pure streaming of the probe row ids, the stash and (DUP)
// the overflow stash, 8 tuples per lane; the index is not needed.  DUP = false: every probe tuple has
// zero or one match (the foreign-key case), offsets come from ballots.
template <bool DUP, bool N32>
__device__ __forceinline__ void fj_emit_stream(const FusedArgs &f, uint32_t u, uint64_t base, uint32_t *wsum,
                                               const uint64_t *ovf, uint32_t *table, uint32_t *grab)
{
    constexpr int V = FJ_V;                           // a wave's step is one 256-tuple group of phase 1
    const JoinArgs &a = f.j;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const bool flip = a.histR[b] < a.histS[b];
    const uint64_t ppos = (flip ? a.psumS[b] : a.psumR[b]) + un.off;
    const uint2 *pr2 = reinterpret_cast<const uint2 *>((flip ? a.partS : a.partR) + ppos);
    const uint8_t *scnt = f.stash_cnt + (flip ? f.nR : 0) + ppos;
    const uint2 *srow = reinterpret_cast<const uint2 *>(f.stash_row + (flip ? f.nR : 0) + ppos);
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const uint64_t cap = a.out_capacity;
    const uint64_t lt = lanemask_lt();
    const uint32_t ngroups = (un.count + 255u) >> 8;

    // group totals (phase 1 left them in column 0 of the table) -> exclusive starts, once per unit;
    // after that the waves run without any barrier
    {
        const uint32_t t = threadIdx.x;
        uint32_t v = 0;
        if (t < ngroups) v = __hip_atomic_load(&table[t * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t tot;
        uint32_t ex = wave_excl_scan_u32(v, &tot);
        __syncthreads();                              // wsum reuse
        if (t == 0) *grab = 0;
        if (lane == 0) wsum[w] = tot;
        __syncthreads();
        for (uint32_t i = 0; i < w && i < FJ_GROUPS / WAVE; ++i) ex += wsum[i];
        if (t < ngroups) __hip_atomic_store(&table[t * 16u], ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
    }

    for (;;) {                                        // groups are handed out as in phase 1
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(grab, 1u);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= ngroups) break;
        uint32_t c[V];
        uint2 first[V], prow[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const uint32_t i = g * 256u + k * WAVE + lane;
            const bool ok = i < un.count;
            c[k] = ok ? (scnt[i] & 0x7fu) : 0;
            first[k] = ok ? srow[i] : make_uint2(0, 0);
            if (N32) { prow[k] = make_uint2(first[k].y, 0u); first[k].y = 0u; }
            else     prow[k] = ok ? pr2[2 * (size_t)i + 1] : make_uint2(0, 0);
        }
        // the group's table row: lane 0 its start in the unit's output, lane j the run start of ordinal j
        uint32_t tbl_v = 0;
        if (lane < 16) tbl_v = __hip_atomic_load(&table[g * 16u + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t off[V], wrun = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (DUP) {
                uint32_t tot;
                off[k] = wrun + wave_excl_scan_u32(c[k], &tot);
                wrun += tot;
            } else {
                const uint64_t m = __ballot(c[k] != 0);
                off[k] = wrun + (uint32_t)__popcll(m & lt);
                wrun += (uint32_t)__popcll(m);
            }
        }
        const uint64_t wbase = base + (uint32_t)__builtin_amdgcn_readlane((int)tbl_v, 0);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const uint64_t at = wbase + off[k];
            if (c[k] != 0 && at < cap) out[at] = make_pair(flip, prow[k].x, prow[k].y, first[k].x, first[k].y);
        }
        if (DUP) {
            // second and later matches: ordinal j of the group's tuples sits in one run of the overflow
            // stash, in (k, lane) order (fj_count_batch).  Four ordinals per step, loads before stores.
            for (uint32_t j0 = 1;; j0 += 4) {
                uint2 r[4][V];
                bool any = false;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const uint32_t j = j0 + jj;
                    uint64_t mk[V];
                    uint32_t tot = 0;
#pragma unroll
                    for (int k = 0; k < V; ++k) { mk[k] = __ballot(c[k] > j); tot += (uint32_t)__popcll(mk[k]); }
                    any = any || tot != 0;
                    uint32_t pre = (uint32_t)__builtin_amdgcn_readlane((int)tbl_v, (int)(j & 15u));
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const uint32_t slot = pre + (uint32_t)__popcll(mk[k] & lt);
                        r[jj][k] = make_uint2(0, 0);
                        if (c[k] > j) r[jj][k] = reinterpret_cast<const uint2 *>(ovf)[slot];
                        pre += (uint32_t)__popcll(mk[k]);
                    }
                }
                if (!any) break;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const uint64_t at = wbase + off[k] + j0 + jj;
                        if (c[k] > j0 + jj && at < cap) out[at] = make_pair(flip, prow[k].x, prow[k].y, r[jj][k].x, r[jj][k].y);
                    }
                }
                if (j0 + 4 > FJ_OVF_J) break;
            }
        }
    }
}

// MAYRES = false compiles the gather path only (the host picks it when the average bucket
// cannot fit LDS anyway); MAYRES = true decides per unit.
// A workgroup takes units through the ticket until none is left.  The emit pass of a unit that does
// not need its index any more (no probe tuple with two or more matches: the foreign-key case) is
// DEFERRED behind the next unit's build + phase 1: by then its output base has long been published,
// so such units never wait on the chain (the wait was 18 % of a unit in the in-kernel stamps).
// Diagnostics of the fused kernel (in-kernel phase stamps, RHJ_STAMPS; parts switched off, RHJ_ABLATE)
// are compiled in only with -DRHJ_INSTRUMENT (tools/): the production kernel carries no trace of them.
#ifdef RHJ_INSTRUMENT
#define FJ_DBG (f.dbg)
#define FJ_ABLATE (a.ablate)
#else
#define FJ_DBG ((uint64_t *)nullptr)
#define FJ_ABLATE 0u
#endif
template <bool MAYRES, bool N32>
__global__ __launch_bounds__(FJ_BLOCK) void k_join_fused(FusedArgs f, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];
    __shared__ uint32_t sh_u;
    __shared__ uint32_t sh_ovf;
    __shared__ uint32_t sh_grab;
    __shared__ uint32_t sh_pick;
    __shared__ uint64_t sh_base;
    __shared__ uint32_t wsum[FJ_WAVES];
    const JoinArgs &a = f.j;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long *st = (unsigned long long *)f.status;
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const uint64_t cap = a.out_capacity;
    const bool emitting = out != nullptr && FJ_ABLATE != 3;
    if ((a.summary->wide_row_ids == 0) != N32) return;                 // the other instantiation's launch does the join
    uint32_t pend = 0xffffffffu;                      // unit whose emit pass is deferred
    uint64_t pend_total = 0;
    bool pend_dup = false;
    const uint64_t *pend_ovf = nullptr;
    uint32_t *pend_table = nullptr;

    for (uint32_t iter = 0;; ++iter) {
    __syncthreads();
    if (threadIdx.x == 0) { sh_u = atomicAdd(f.ticket, 1u); sh_ovf = 0; sh_grab = 0; }
    __syncthreads();
    const uint32_t u = sh_u;
    if (u >= a.summary->units || !a.summary->fused_ok) break;         // grid is an upper bound; tiled path takes over
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const uint64_t cR = a.histR[b], cS = a.histS[b];
    const bool flip = cR < cS;                                         // S is streamed (r_s == 1)
    const uint64_t ppos = (flip ? a.psumS[b] : a.psumR[b]) + un.off;   // position in the probe relation
    const rhj_tuple *prp = flip ? a.partS : a.partR;                   // probe tuple i of the unit: pt_load<N32>(prp, ppos + i)
    const rhj_tuple *bdp = flip ? a.partR : a.partS;                   // build tuple i of the bucket: pt_load<N32>(bdp, bpos + i)
    const uint64_t bpos = flip ? a.psumR[b] : a.psumS[b];
    const uint32_t bc = (uint32_t)(flip ? cR : cS);
    // LDS: [resident build tuples 16 B x bc] [entries 4 B x (bc + 8)] [slot starts 2 B x (hs + 1)]
    const uint32_t bcp = (bc + 3u) & ~3u;
    // slots: one per build tuple when that fits behind the entries, fewer (longer slots) for the
    // largest build sides, never below a quarter (host-side cap: 4.5 B per build tuple)
    uint32_t hs0 = bc < 64u ? 64u : bc;
    {
        const uint32_t room = (lds_bytes - 64u - 4u * bcp) / 2u - 2u; // 16-bit slot starts that still fit
        if (hs0 > room) hs0 = room & ~1u;
    }
    // build tuples go to LDS too when they fit beside the index (wave-uniform per unit)
    const bool RES = MAYRES && f.allow_resident && (size_t)bcp * 20 + (size_t)(hs0 + 3) / 2 * 4 + 64 <= lds_bytes;
    uint4 *ltup = reinterpret_cast<uint4 *>(tbl);
    FjIndex X;
    X.ent = tbl + (RES ? 4u * bcp : 0u);
    X.hs = hs0;
    X.dirw = X.ent + bcp + 8u;
    uint8_t *scnt = f.stash_cnt + (flip ? f.nR : 0) + ppos;
    uint2 *srow = reinterpret_cast<uint2 *>(f.stash_row + (flip ? f.nR : 0) + ppos);
    FjGather<N32> G;
    G.init(bdp, bpos, bc);
    FjOvf O;                                          // double-buffered: the previous unit's emit may still be pending
    O.buf = f.ovf + ((size_t)blockIdx.x * 2 + (iter & 1u)) * FJ_OVF_CAP;
    O.table = f.ovf_base + ((size_t)blockIdx.x * 2 + (iter & 1u)) * (FJ_GROUPS * 16u);
    O.counter = &sh_ovf;
    O.gid = 0;

    if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 0] = __builtin_amdgcn_s_memrealtime();
    // ---- build
    if (RES) fj_build<true, N32>(X, bdp, bpos, bc, ltup, reinterpret_cast<uint32_t *>(O.buf), wsum, &sh_pick);
    else     fj_build<false, N32>(X, bdp, bpos, bc, ltup, reinterpret_cast<uint32_t *>(O.buf), wsum, &sh_pick);
    if (FJ_ABLATE == 1) continue;                      // timing experiment: build only
    if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 1] = __builtin_amdgcn_s_memrealtime();

    // ---- phase 1: count (+ stash of the first match when the build tuples are not resident)
    uint32_t mine = 0;
    bool needs_index = false;                         // the emit pass must walk the index again
    // The waves take 256-tuple groups from a workgroup counter: with a fixed share per wave the barrier
    // behind this loop waited 20 us of a 156 us unit for the slowest wave's gathers.
    const uint32_t ngroups1 = (un.count + 255u) >> 8;
    for (;;) {
        uint32_t grp = 0;
        if (lane == 0) grp = atomicAdd(&sh_grab, 1u);
        grp = __builtin_amdgcn_readfirstlane(grp);
        if (grp >= ngroups1) break;
        const uint32_t t0 = grp << 8;
        uint4 q[FJ_V];
        bool okk[FJ_V];
        uint32_t c[FJ_V], flo[FJ_V], fhi[FJ_V];
        bool fp[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const uint32_t i = t0 + k * WAVE + lane;
            okk[k] = i < un.count;
            q[k] = okk[k] ? pt_load<N32>(prp, ppos + i) : make_uint4(0, 0, 0, 0);
        }
        O.gid = grp;
        if (RES) fj_count_batch<true, true, N32>(X, G, ltup, q, okk, c, flo, fhi, fp, O);
        else     fj_count_batch<false, true, N32>(X, G, ltup, q, okk, c, flo, fhi, fp, O);
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const uint32_t i = t0 + k * WAVE + lane;
            if (i < un.count) {
                // count byte: 0..126 exact, 127 = saturated (recounted in phase 2); bit 7 = some tag hit of
                // this tuple was a different key, so phase 2 must verify its candidates again
                scnt[i] = (uint8_t)(min(c[k], 127u) | (fp[k] ? 0x80u : 0u));
                fj_stash_put<N32>(srow, i, flo[k], fhi[k], q[k].z);
            }
            mine += c[k];
            needs_index = needs_index || (fp[k] && c[k] >= 2u) || c[k] > FJ_OVF_J + 1u;   // its overflow entries are not where the emit pass expects them
        }
        {                                             // group total for the barrier-free emit pass
            uint32_t gt;
            wave_excl_scan_u32(c[0] + c[1] + c[2] + c[3], &gt);
            if (lane == 0) O.table[O.gid * 16u] = gt;
        }
    }

    // ---- unit total -> chained scan
    if (FJ_DBG && lane == 0) { if (w == 0) FJ_DBG[(size_t)u * 8 + 2] = __builtin_amdgcn_s_memrealtime(); }
    {
        uint32_t tot;
        wave_excl_scan_u32(mine, &tot);
        if (lane == 0) wsum[w] = tot;
    }
    bool unit_needs_index = __syncthreads_or(needs_index) != 0;   // also publishes wsum
    const uint32_t ovf_total = sh_ovf;
    unit_needs_index = unit_needs_index || ovf_total > FJ_OVF_CAP;
    uint64_t total = 0;
#pragma unroll
    for (int i = 0; i < FJ_WAVES; ++i) total += wsum[i];
    if (threadIdx.x == 0) {
        // aggregate first: successors only ever wait for this word
        __hip_atomic_store(&st[u], ((u == 0 ? 2ull : 1ull) << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.unit_count[u] = total;
    }

    // ---- the deferred emit pass of the previous unit, whose base is certainly known by now
    if (pend != 0xffffffffu) {
        if (w == 0) {
            const uint64_t excl = pend == 0 ? 0 : fj_lookback(st, pend, lane);
            if (lane == 0) {
                if (pend != 0) __hip_atomic_store(&st[pend], (2ull << 62) | (excl + pend_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh_base = excl;
            }
        }
        __syncthreads();
        if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)pend * 8 + 5] = __builtin_amdgcn_s_memrealtime();
        if (pend_dup) fj_emit_stream<true, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab);
        else          fj_emit_stream<false, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab);
        if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)pend * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        pend = 0xffffffffu;
        __syncthreads();
        if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 7] = __builtin_amdgcn_s_memrealtime();
    }
    if (emitting && !unit_needs_index) {              // this unit's emit pass needs no index: defer it
        pend = u;
        pend_total = total;
        pend_dup = ovf_total != 0;
        pend_ovf = O.buf;
        pend_table = O.table;
        if (FJ_DBG && threadIdx.x == 0) { FJ_DBG[(size_t)u * 8 + 3] = FJ_DBG[(size_t)u * 8 + 4] = __builtin_amdgcn_s_memrealtime(); }
        continue;
    }

    if (w == 0) {
        const uint64_t excl = u == 0 ? 0 : fj_lookback(st, u, lane);
        if (lane == 0) {
            if (u != 0) __hip_atomic_store(&st[u], (2ull << 62) | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_base = excl;
        }
    }
    __syncthreads();

    if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    // ---- phase 2: emit (general form: duplicates and tag collisions walk the index again)
    uint64_t run = sh_base;
    if (!emitting) continue;
    // FJ_H batches per iteration; a wave's slice of the iteration is contiguous: order (wave, half,
    // round, lane).  FJ_H = 2 (more loads in flight, half the barriers) measured +11 % on the kernel:
    // it spills at the 128-VGPR limit of a 1024-thread workgroup.
    constexpr int FJ_H = 1;
    for (uint32_t t0 = 0; t0 < un.count; t0 += FJ_H * FJ_BATCH) {
        uint32_t c[FJ_H][FJ_V], flo[FJ_H][FJ_V], fhi[FJ_H][FJ_V];
        uint4 q[FJ_H][FJ_V];
        bool okk[FJ_H][FJ_V], fpt[FJ_H][FJ_V];
#pragma unroll
        for (int h = 0; h < FJ_H; ++h) {
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = t0 + w * (WAVE * FJ_V * FJ_H) + h * (WAVE * FJ_V) + k * WAVE + lane;
                okk[h][k] = i < un.count;
                fpt[h][k] = false;
                q[h][k] = okk[h][k] ? pt_load<N32>(prp, ppos + i) : make_uint4(0, 0, 0, 0);
                const uint32_t sb = okk[h][k] ? scnt[i] : 0;
                c[h][k] = sb & 0x7fu;
                fpt[h][k] = (sb & 0x80u) != 0;
                const uint2 fr = okk[h][k] ? srow[i] : make_uint2(0, 0);
                flo[h][k] = fr.x; fhi[h][k] = N32 ? 0u : fr.y;
            }
        }
#pragma unroll
        for (int h = 0; h < FJ_H; ++h) {
            {
                // saturated counts: recount from the index (also yields the exact number to emit)
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    if (c[h][k] == 127u) {
                        const uint64_t hh = mix64(((uint64_t)q[h][k].y << 32) | q[h][k].x);
                        const uint32_t t = fj_tag(hh), sl = X.slot(hh);
                        uint32_t n = 0;
                        for (uint32_t at = X.H(sl + 1u), end = X.H(sl + 2u); at < end; ++at) {
                            const uint32_t nd = X.ent[at];
                            if ((nd >> 16) == t) { const uint4 v = RES ? ltup[nd & 0xffffu] : G.load(nd & 0xffffu); n += (v.x == q[h][k].x && v.y == q[h][k].y); }
                        }
                        c[h][k] = n;
                    }
                }
            }
        }
        uint32_t off[FJ_H][FJ_V], wrun = 0;
#pragma unroll
        for (int h = 0; h < FJ_H; ++h) {
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                uint32_t tot;
                off[h][k] = wrun + wave_excl_scan_u32(c[h][k], &tot);
                wrun += tot;
            }
        }
        __syncthreads();                              // wsum reuse
        if (lane == 0) wsum[w] = wrun;
        __syncthreads();
        uint64_t wbase = run;
        uint32_t batch_total = 0;
#pragma unroll
        for (int i = 0; i < FJ_WAVES; ++i) {
            const uint32_t v = wsum[i];
            if ((uint32_t)i < w) wbase += v;
            batch_total += v;
        }
        run += batch_total;
        // The first match comes from the stash (phase 1 kept its row id).  Further matches of a tuple
        // are fetched in lockstep rounds over its chain; when no tag hit of the tuple was a foreign key
        // (the rule: stash bit 7 clear) its first candidate IS that first match and is skipped unfetched.
#pragma unroll
        for (int h = 0; h < FJ_H; ++h) {
            uint64_t at[FJ_V];
            uint32_t sn[FJ_V], tm[FJ_V];
            bool skip[FJ_V];
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                at[k] = wbase + off[h][k];
                const bool direct = c[h][k] >= 1 && !fpt[h][k];             // stash holds its first emitted pair
                if (direct) { if (at[k] < cap) out[at[k]] = make_pair(flip, q[h][k].z, q[h][k].w, flo[h][k], fhi[h][k]); ++at[k]; }
                const bool walk = direct ? c[h][k] >= 2 : c[h][k] >= 1;
                fj_lookup(X, ((uint64_t)q[h][k].y << 32) | q[h][k].x, walk, sn[k], tm[k]);
                skip[k] = direct;
            }
            bool last = false;
            for (bool first_round = true; !last; first_round = false) {
                uint32_t pos[FJ_V];
                if (!fj_round(X, sn, tm, pos, last)) break;
                if (first_round) {
#pragma unroll
                    for (int k = 0; k < FJ_V; ++k)
                        if (skip[k]) pos[k] = 0xffffffffu;            // already emitted from the stash
                }
                uint4 g[FJ_V];
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    g[k] = make_uint4(0, 0, 0, 0);
                    if (pos[k] != 0xffffffffu) g[k] = RES ? ltup[pos[k]] : G.load(pos[k]);
                }
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    if (pos[k] != 0xffffffffu && g[k].x == q[h][k].x && g[k].y == q[h][k].y) {
                        if (at[k] < cap) out[at[k]] = make_pair(flip, q[h][k].z, q[h][k].w, g[k].z, g[k].w);
                        ++at[k];
                    }
                }
            }
        }
    }
    if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    }   // ticket loop

    if (pend != 0xffffffffu) {                        // last deferred emit of this workgroup
        __syncthreads();
        if (w == 0) {
            const uint64_t excl = pend == 0 ? 0 : fj_lookback(st, pend, lane);
            if (lane == 0) {
                if (pend != 0) __hip_atomic_store(&st[pend], (2ull << 62) | (excl + pend_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh_base = excl;
            }
        }
        __syncthreads();
        if (pend_dup) fj_emit_stream<true, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab);
        else          fj_emit_stream<false, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab);
    }
}

// total matches of the fused path = inclusive prefix of the last unit
__global__ void k_fused_total(const uint64_t *status, const PlanSummary *summary, uint64_t unit_bound, uint64_t *total_out)
{
    // when the plan rejected the fused path its unit list is the tiled one and can be longer than the
    // status array: nothing was published, nothing to read
    const uint64_t n = summary->units;
    *total_out = (summary->fused_ok && n && n <= unit_bound) ? (status[n - 1] & ((1ull << 62) - 1)) : 0;
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

__global__ __launch_bounds__(256) void k_filter_write(uint64_t n, const uint64_t *masks, const uint64_t *tile_base,
                                                      uint64_t *out)
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
        const uint64_t tb = tile < ntiles ? tile_base[tile] : 0;
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

#ifdef RHJ_INSTRUMENT
// Diagnostics build only (tools/gather_bench.py): the access pattern of phase 1 of the fused join without the
// join — every workgroup gathers random 16-byte tuples from its own contiguous region (a bucket's build side)
// through the same sc1 buffer loads, FJ_V in flight per lane; optionally it also streams `stream_elems` tuples per
// round like the probe side.  Gives the chip's rate for this pattern: the ceiling phase 1 can be compared with.
__global__ __launch_bounds__(FJ_BLOCK) void k_gather_bench(const rhj_tuple *base, uint32_t region_elems, uint32_t rounds,
                                                           const uint4 *stream, uint32_t stream_per_round, uint4 *sink)
{
    FjGather<false> G;
    G.init(base, (uint64_t)blockIdx.x * region_elems, region_elems);
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4 *sp = stream + ((size_t)blockIdx.x * rounds) * stream_per_round * FJ_BLOCK + threadIdx.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint4 v[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            x = x * 1664525u + 1013904223u;
            v[k] = G.load(__umulhi(x, region_elems));
        }
        for (uint32_t q = 0; q < stream_per_round; ++q) { const uint4 t = sp[((size_t)r * stream_per_round + q) * FJ_BLOCK]; acc.y ^= t.x; }
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) acc.x ^= v[k].x ^ v[k].z;
    }
    if (acc.x == 0x12345u && acc.y == 0x54321u) sink[threadIdx.x] = acc;
}
#endif

}  // namespace rhj
