// rhj_join_tiled.hip.h — plan (bucket loop / side choice), HBM hash tables and the tiled count / emit probe
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
#pragma once
#include "rhj_common.hip.h"
#include "rhj_partition.hip.h"

namespace rhj {

// ----------------------------------------------------------------------- plan

constexpr int PR_BLOCK = 256;                     // probe workgroup
constexpr int PR_V = 4;                           // probe tuples per thread
constexpr int PR_UNIT = PR_BLOCK * PR_V;          // 2048 probe tuples per unit
#ifndef PR_MINW
#define PR_MINW 5                                 // waves per SIMD the probe kernels are compiled for (96 VGPRs)
#endif

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
    uint32_t        parent_mask;    // low-radix path (see JoinArgs): the side choice of the caller's bucket
    const uint8_t  *parent_flip;
    uint32_t       *zero;           // words the plan clears for the kernels behind it (the fused join's ticket and status words), or null
    uint32_t        zero_words;
    // rhj_join_device_slice (a hot bucket shared by several devices): of bucket slice_b0 only the probe tuples from slice_o0 on
    // get units, of bucket slice_b1 only those before slice_o1 (0: all of them); 0xffffffff: no such bucket
    uint32_t        slice_b0 = 0xffffffffu, slice_b1 = 0xffffffffu;
    uint64_t        slice_o0 = 0, slice_o1 = 0;
};

// the probe tuples [lo, hi) of bucket b that this join takes (all of them unless the join is a slice; lo >= hi: none)
__device__ __forceinline__ void plan_probe_span(const PlanArgs &a, uint32_t b, uint64_t pc, uint64_t &lo, uint64_t &hi)
{
    lo = b == a.slice_b0 ? a.slice_o0 : 0u;
    hi = b == a.slice_b1 && a.slice_o1 != 0u ? min(a.slice_o1, pc) : pc;
}

constexpr uint32_t T32_PAD = 8;         // replica of the first 8 entries behind every 32-bit table

__device__ __forceinline__ uint32_t lds_slots_for(uint64_t bc, uint32_t max_slots)
{
    uint32_t s = (uint32_t)(bc + (bc >> 1)) + 4u;          // load factor <= 2/3 when it fits
    s = (s + 3u) & ~3u;                                    // 16-byte dump granule
    return min(max(s, 64u), max_slots);
}

__device__ __forceinline__ void plan_body(const PlanArgs &a, int bits, uint64_t *sm /*5 * (1024 / 64 + 1)*/, unsigned long long *red /*2*/)
{
    const uint32_t bins = 1u << bits;
    const uint32_t per = (bins + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, bins);
    if (threadIdx.x < 2) red[threadIdx.x] = 0;
    // Launched as several workgroups (k_plan on many buckets), every one computes and scans everything and the WAVES are dealt
    // round-robin to the workgroups for the stores: the meta records and unit lists of 16 K buckets from one compute unit took
    // 0.1 ms (100M x 1B at 14 bits).
    if (a.zero)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.zero_words; i += gridDim.x * blockDim.x) a.zero[i] = 0;

    uint64_t nu = 0, nbu = 0, slots64 = 0, nlds = 0, slots32 = 0;
    uint32_t max_build = 0, max_slots = 0;
    // (a thread's bucket sizes are read eight buckets at a time: one bucket after the other the loop waited for every pair)
    constexpr uint32_t PB = 8;
    for (uint32_t bb = b0; bb < b1; bb += PB) {
      uint64_t cRv[PB], cSv[PB];
#pragma unroll
      for (uint32_t i = 0; i < PB; ++i) {
          const bool in = bb + i < b1;
          cRv[i] = in ? a.histR[bb + i] : 0;
          cSv[i] = in ? a.histS[bb + i] : 0;
      }
#pragma unroll
      for (uint32_t i = 0; i < PB; ++i) {
        const uint32_t b = bb + i;
        const uint64_t cR = cRv[i], cS = cSv[i];
        if (cR == 0 || cS == 0) continue;
        const bool flip = a.parent_flip ? a.parent_flip[b & a.parent_mask] != 0 : cR < cS;   // rhjoin.c:86 (>=)
        const uint64_t pc = flip ? cS : cR, bc = flip ? cR : cS;
        uint64_t o_lo, o_hi;
        plan_probe_span(a, b, pc, o_lo, o_hi);
        if (o_lo >= o_hi) continue;                                        // (a slice that leaves nothing of this bucket)
        const uint64_t span = bc <= a.lds_cap ? a.span_lds : PR_UNIT;
        nu += ((uint32_t)(o_hi - o_lo) - 1u) / (uint32_t)span + 1u;        // (bucket sizes are in [1, 2^32): 32-bit divisions)
        max_build = max(max_build, (uint32_t)min(bc, (uint64_t)0xffffffffu));
        if (bc <= a.lds_cap) {
            const uint32_t s = lds_slots_for(bc, a.lds_max_slots);
            nlds += 1; slots32 += s + T32_PAD;
            max_slots = max(max_slots, s);
        } else {
            nbu += ((uint32_t)bc - 1u) / a.build_chunk + 1u;
            slots64 += 1ull << (64 - __clzll((unsigned long long)(2 * bc - 1)));   // pow2 >= 2*bc
        }
      }
    }
    const uint64_t mine5[5] = {nu, nbu, slots64, nlds, slots32};
    uint64_t ex5[5], tot5[5];
    block_excl_scan_n<1024, 5>(mine5, ex5, tot5, sm);
    uint64_t ubase = ex5[0], bbase = ex5[1], s64base = ex5[2], lbase = ex5[3], s32base = ex5[4];
    const uint64_t tot_u = tot5[0], tot_b = tot5[1], tot_s64 = tot5[2], tot_l = tot5[3], tot_s32 = tot5[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) {                    // (one LDS atomic a wave: 1024 on one address took 10 us)
        max_build = max(max_build, (uint32_t)__shfl_xor((int)max_build, o, 64));
        max_slots = max(max_slots, (uint32_t)__shfl_xor((int)max_slots, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&red[0], (unsigned long long)max_build);
        atomicMax(&red[1], (unsigned long long)max_slots);
    }

    const bool my_stores = ((threadIdx.x >> 6) % gridDim.x) == blockIdx.x;
    for (uint32_t bb = b0; my_stores && bb < b1; bb += PB) {
      uint64_t cRv[PB], cSv[PB];
#pragma unroll
      for (uint32_t i = 0; i < PB; ++i) {
          const bool in = bb + i < b1;
          cRv[i] = in ? a.histR[bb + i] : 0;
          cSv[i] = in ? a.histS[bb + i] : 0;
      }
#pragma unroll
      for (uint32_t i = 0; i < PB; ++i) {
        const uint32_t b = bb + i;
        if (b >= b1) break;
        const uint64_t cR = cRv[i], cS = cSv[i];
        BucketMeta m = {0, 0, 0};
        uint64_t o_lo = 0, o_hi = 0;
        if (cR != 0 && cS != 0) {
            const bool flip0 = a.parent_flip ? a.parent_flip[b & a.parent_mask] != 0 : cR < cS;
            plan_probe_span(a, b, flip0 ? cS : cR, o_lo, o_hi);
        }
        if (o_lo < o_hi) {
            const bool flip = a.parent_flip ? a.parent_flip[b & a.parent_mask] != 0 : cR < cS;
            const uint64_t pc = flip ? cS : cR, bc = flip ? cR : cS;
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
            (void)pc;
            for (uint64_t o = o_lo; o < o_hi; o += span) {
                Unit u; u.off = o; u.bucket = b; u.count = (uint32_t)min(span, o_hi - o);
                a.units[ubase++] = u;
            }
        }
        a.meta[b] = m;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
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
    __shared__ uint64_t sm[5 * (1024 / 64 + 1)];
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
    __shared__ uint64_t sm[5 * (1024 / 64 + 1)];
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
    __shared__ uint64_t r_kept;                       // R's tuples that take part: all of R, unless the join is sharded by bucket range
    if (threadIdx.x == 512) r_kept = ex;              // (everything in front of S's first digit)
    __syncthreads();
    if (t < bins) {
        const uint64_t e = rel ? ex - r_kept : ex;    // exclusive prefix inside S = prefix over both - R's part
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
                                           const rhj_tuple *bd, uint32_t count, bool flip, uint32_t u, uint32_t *wsum,
                                           uint8_t *scnt, uint2 *srow)
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
    // emit pass: a tuple the count pass found at most one match for is served from the stash (count + first match's row id)
    // and never goes back to the table — on a foreign-key join the emit pass is a stream, not a second probe
    bool stashed[PR_V];
    uint32_t sc[PR_V];
    uint2 sr[PR_V];
#pragma unroll
    for (int k = 0; k < PR_V; ++k) {
        const uint32_t i = w * (WAVE * PR_V) + k * WAVE + lane;
        ok[k] = i < count;
        q[k] = ok[k] ? pr4[i] : make_uint4(0, 0, 0, 0);
        stashed[k] = false; sc[k] = 0; sr[k] = make_uint2(0, 0);
        if (WRITE && ok[k]) { sc[k] = scnt[i]; sr[k] = srow[i]; stashed[k] = sc[k] <= 1u; }
    }
    slot_t s0[PR_V];
    uint32_t tg[PR_V];
    entry_t e[PR_V][CH];
#pragma unroll
    for (int k = 0; k < PR_V; ++k) {
        const uint64_t h = mix64(((uint64_t)q[k].y << 32) | q[k].x);
        s0[k] = T.home(h);
        tg[k] = T.tag(h);
        if (ok[k] && a.ablate != 2 && !stashed[k]) T.load_chunk(s0[k], e[k]);
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
        if (WRITE && (stashed[k] || !ok[k])) { hm[k] = 0; more[k] = false; }
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
        // first two candidates of every tuple: gather all (the count pass the whole tuple: the first match's row id
        // goes to the stash), then compare
        uint4 g0[PR_V], g1[PR_V];
        uint32_t rest[PR_V];
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            uint32_t r = hm[k];
            g0[k] = make_uint4(0, 0, 0, 0); g1[k] = make_uint4(0, 0, 0, 0);
            if (r) { r &= r - 1; if (WRITE) { const uint2 v = bd2[2 * (size_t)p0[k]]; g0[k].x = v.x; g0[k].y = v.y; } else g0[k] = bd4[p0[k]]; }
            if (r) { r &= r - 1; if (WRITE) { const uint2 v = bd2[2 * (size_t)p1[k]]; g1[k].x = v.x; g1[k].y = v.y; } else g1[k] = bd4[p1[k]]; }
            rest[k] = r;
        }
#pragma unroll
        for (int k = 0; k < PR_V; ++k) {
            const uint32_t nc = (uint32_t)__popc(hm[k]);
            const bool eq0 = nc >= 1 && g0[k].x == q[k].x && g0[k].y == q[k].y;
            const bool eq1 = nc >= 2 && g1[k].x == q[k].x && g1[k].y == q[k].y;
            uint32_t c = (uint32_t)eq0 + (uint32_t)eq1;
            uint2 first = eq0 ? make_uint2(g0[k].z, g0[k].w) : make_uint2(g1[k].z, g1[k].w);      // (meaningful when c != 0)
            fp = fp || (nc >= 1 && !eq0) || (nc >= 2 && !eq1);
            if (rest[k]) {                                // third and later candidates of the chunk
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if ((rest[k] >> j) & 1u) {
                        const uint4 v = bd4[T.pos(e[k][j])];
                        const bool eq = v.x == q[k].x && v.y == q[k].y;
                        if (eq && c == 0) first = make_uint2(v.z, v.w);
                        c += eq; fp = fp || !eq;
                    }
                }
            }
            if (more[k]) {                                // run longer than the chunk
                slot_t s = T.advance(s0[k], CH - T.skip(s0[k]));
                entry_t x = T.load(s);
                while (T.live(x, tg[k])) {
                    if (T.hit(x, tg[k])) {
                        const uint4 v = bd4[T.pos(x)];
                        const bool eq = v.x == q[k].x && v.y == q[k].y;
                        if (eq && c == 0) first = make_uint2(v.z, v.w);
                        c += eq; fp = fp || !eq;
                    }
                    s = T.advance(s, 1);
                    x = T.load(s);
                }
            }
            m[k] = c;
            if (!WRITE) {
                const uint32_t i = w * (WAVE * PR_V) + k * WAVE + lane;
                if (ok[k]) { scnt[i] = (uint8_t)min(c, 255u); srow[i] = first; }
            }
        }
    }
    if (WRITE) {
#pragma unroll
        for (int k = 0; k < PR_V; ++k)
            if (stashed[k]) m[k] = sc[k];
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
            if (stashed[k]) { if (at < cap) out[at] = make_pair(flip, prl, prh, sr[k].x, sr[k].y); continue; }
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
            if (stashed[k]) { if (at < cap) out[at] = make_pair(flip, prl, prh, sr[k].x, sr[k].y); continue; }
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
__global__ __launch_bounds__(PR_BLOCK, PR_MINW) void k_probe(JoinArgs a)
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
    const uint64_t spos = (flip ? a.stash_nR + a.psumS[b] : a.psumR[b]) + un.off;          // the unit's first probe tuple in the stash
    uint8_t *scnt = a.stash_cnt + spos;
    uint2 *srow = reinterpret_cast<uint2 *>(a.stash_row + spos);
    const BucketMeta m = a.meta[b];
    if (m.mode == 1) {
        Tab32 T{a.tab32 + m.table_off, m.slots};
        probe_unit<WRITE>(a, T, pr, bd, un.count, flip, u, wsum, scnt, srow);
    } else {
        Tab64 T{a.tab64 + m.table_off, m.slots};
        probe_unit<WRITE>(a, T, pr, bd, un.count, flip, u, wsum, scnt, srow);
    }
}

}  // namespace rhj
