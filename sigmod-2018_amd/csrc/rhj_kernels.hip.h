// rhj_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the radix hash
// join and the filter scan.  Integer / indexing work only: the roofline is HBM.
//
// Reference loops these kernels replace (file:line in VagelisN/Sigmod-2018):
//   k_hist           HistJob                      preprocess.c:181-195
//   k_scan_*         hist merge + psum            preprocess.c:83-102 / :328-340
//   k_scatter        SerialReorderArray scatter   preprocess.c:349-359 (stable)
//   k_plan           bucket loop / side choice    rhjoin.c:79-102 (>= picks the probe side)
//   k_build_hbm      InitIndex + CreateIndex      rhjoin.c:253-273, :219-250
//   k_probe          CreateIndex + GetResults     rhjoin.c:219-250, :141-217
//   k_filter_*       Filter                       filter.c:110-183
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
//   * table in LDS   (32-bit entries, 16-bit tag, 16-bit position) when the build
//     side of the bucket has <= lds_cap tuples: one workgroup per probe unit builds
//     it from the bucket's build side and streams the probe side through it;
//   * table in HBM   (64-bit entries, 32-bit tag, 32-bit position) otherwise, built
//     by k_build_hbm with global atomics and shared by all probe units of the bucket.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rhj.h"

namespace rhj {

constexpr int WAVE = 64;

struct RelArgs {                 // one relation through the partition kernels
    const rhj_tuple *in;
    rhj_tuple       *out;
    uint32_t        *cnt;        // [tiles][bins] counts, then (after scan) start offsets
    uint64_t         n;
    uint32_t         tile_len;   // tuples per tile (multiple of 64)
    uint32_t         tiles;
};

struct Unit {                    // one probe work unit, in canonical order
    uint64_t off;                // offset inside the bucket's probe side
    uint32_t bucket;
    uint32_t count;              // probe tuples in this unit
};

struct BucketMeta {
    uint64_t table_off;          // HBM mode: first slot in the table arena
    uint32_t slots;              // LDS mode: slot count; HBM mode: log2(slot count)
    uint32_t mode;               // 0 inactive, 1 LDS table, 2 HBM table
};

struct PlanSummary {
    uint64_t units;              // probe units
    uint64_t build_units;        // HBM build chunks
    uint64_t hbm_slots;          // total 64-bit slots of all HBM tables
    uint64_t max_build_lds;      // largest build side among LDS buckets
    uint64_t hbm_units;          // probe units that use an HBM table
    uint64_t max_lds_slots;      // largest LDS slot count
    uint64_t matches;            // filled by k_offsets
    uint64_t pad;
};

struct JoinArgs {
    const rhj_tuple *partR, *partS;
    const uint64_t  *histR, *histS, *psumR, *psumS;   // [bins]
    const Unit      *units;
    const BucketMeta*meta;
    const PlanSummary *summary;
    uint64_t        *tables;         // HBM table arena
    uint64_t        *unit_count;     // [units] matches per unit (count pass)
    const uint64_t  *unit_base;      // [units] exclusive scan of unit_count
    rhj_result_tuple*out;
    uint64_t         out_capacity;
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

// ------------------------------------------------------------------ partition

// Pass 1: per-tile digit histogram.  One wave per tile; LDS counters, one global
// row write per tile.  Reads the key half of each 16-byte AoS tuple.
__global__ __launch_bounds__(WAVE) void k_hist(RelArgs r0, RelArgs r1, int bits)
{
    extern __shared__ uint32_t lds_u32[];
    const RelArgs &r = blockIdx.y ? r1 : r0;
    if (blockIdx.x >= r.tiles) return;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    const uint32_t lane = threadIdx.x;
    for (uint32_t b = lane; b < bins; b += WAVE) lds_u32[b] = 0;
    __syncthreads();
    const uint64_t beg = (uint64_t)blockIdx.x * r.tile_len;
    const uint64_t end = min(beg + (uint64_t)r.tile_len, r.n);
    const rhj_tuple *in = r.in;
    uint64_t i = beg + lane;
    for (; i + 3 * WAVE < end; i += 4 * WAVE) {
        const uint64_t k0 = in[i].value, k1 = in[i + WAVE].value;
        const uint64_t k2 = in[i + 2 * WAVE].value, k3 = in[i + 3 * WAVE].value;
        atomicAdd(&lds_u32[(uint32_t)k0 & mask], 1u);
        atomicAdd(&lds_u32[(uint32_t)k1 & mask], 1u);
        atomicAdd(&lds_u32[(uint32_t)k2 & mask], 1u);
        atomicAdd(&lds_u32[(uint32_t)k3 & mask], 1u);
    }
    for (; i < end; i += WAVE) atomicAdd(&lds_u32[(uint32_t)in[i].value & mask], 1u);
    __syncthreads();
    uint32_t *row = r.cnt + (size_t)blockIdx.x * bins;
    for (uint32_t b = lane; b < bins; b += WAVE) row[b] = lds_u32[b];
}

// Scan step 1: per (bin, chunk of tiles) column sums.
__global__ __launch_bounds__(256) void k_scan_chunks(RelArgs r0, RelArgs r1, int bits, uint32_t chunks,
                                                     uint64_t *chunk_sum /*[2][chunks][bins]*/)
{
    const RelArgs &r = blockIdx.z ? r1 : r0;
    const uint32_t bins = 1u << bits;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bins) return;
    const uint32_t per = (r.tiles + chunks - 1) / chunks;
    const uint32_t t0 = blockIdx.y * per, t1 = min(t0 + per, r.tiles);
    uint64_t s = 0;
    for (uint32_t t = t0; t < t1; ++t) s += r.cnt[(size_t)t * bins + b];
    chunk_sum[((size_t)blockIdx.z * chunks + blockIdx.y) * bins + b] = s;
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
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < NT / 64; ++i) { const uint64_t t = sm[i]; sm[i] = run; run += t; }
        sm[NT / 64] = run;
    }
    __syncthreads();
    if (total) *total = sm[NT / 64];
    return sm[w] + x - v;
}

// Scan step 2 (one workgroup per relation): bucket totals -> hist, exclusive scan
// over buckets -> psum, chunk sums -> exclusive chunk prefixes.
__global__ __launch_bounds__(1024) void k_scan_bins(int bits, uint32_t chunks, uint64_t *chunk_sum,
                                                    uint64_t *hist /*[2][bins]*/, uint64_t *psum /*[2][bins]*/)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    const uint32_t bins = 1u << bits;
    uint64_t *cs = chunk_sum + (size_t)blockIdx.x * chunks * bins;
    uint64_t *h = hist + (size_t)blockIdx.x * bins, *p = psum + (size_t)blockIdx.x * bins;
    const uint32_t per = (bins + 1023) / 1024;       // consecutive bins per thread
    const uint32_t b0 = threadIdx.x * per;
    uint64_t mine = 0;
    for (uint32_t b = b0; b < min(b0 + per, bins); ++b) {
        uint64_t run = 0;
        for (uint32_t c = 0; c < chunks; ++c) {
            const uint64_t t = cs[(size_t)c * bins + b];
            cs[(size_t)c * bins + b] = run;
            run += t;
        }
        h[b] = run;
        mine += run;
    }
    uint64_t base = block_excl_scan<1024>(mine, nullptr, sm);
    for (uint32_t b = b0; b < min(b0 + per, bins); ++b) {
        p[b] = base;
        base += h[b];
    }
}

// Scan step 3: counts -> start offsets  psum[bin] + (tiles before this one).
__global__ __launch_bounds__(256) void k_scan_apply(RelArgs r0, RelArgs r1, int bits, uint32_t chunks,
                                                    const uint64_t *chunk_sum, const uint64_t *psum)
{
    const RelArgs &r = blockIdx.z ? r1 : r0;
    const uint32_t bins = 1u << bits;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bins) return;
    const uint32_t per = (r.tiles + chunks - 1) / chunks;
    const uint32_t t0 = blockIdx.y * per, t1 = min(t0 + per, r.tiles);
    uint64_t run = psum[(size_t)blockIdx.z * bins + b] +
                   chunk_sum[((size_t)blockIdx.z * chunks + blockIdx.y) * bins + b];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = r.cnt[(size_t)t * bins + b];
        r.cnt[(size_t)t * bins + b] = (uint32_t)run;
        run += c;
    }
}

// Pass 2: stable scatter.  One wave per tile owns the tile's running offsets in LDS.
// Per round of 64 tuples every lane takes a slot with an LDS atomic add (unique, any
// order); lanes that share a digit with another lane of the round are then re-ranked
// in lane order with ballots, which makes the placement the stable one whatever order
// the LDS served the adds in.  Cost grows with the number of digits that repeat inside
// a round, not with the number of digits.
__global__ __launch_bounds__(WAVE) void k_scatter(RelArgs r0, RelArgs r1, int bits)
{
    extern __shared__ uint32_t lds_u32[];
    const RelArgs &r = blockIdx.y ? r1 : r0;
    if (blockIdx.x >= r.tiles) return;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    const uint32_t lane = threadIdx.x;
    const uint32_t *row = r.cnt + (size_t)blockIdx.x * bins;
    for (uint32_t b = lane; b < bins; b += WAVE) lds_u32[b] = row[b];
    __syncthreads();
    const uint64_t beg = (uint64_t)blockIdx.x * r.tile_len;
    const uint64_t end = min(beg + (uint64_t)r.tile_len, r.n);
    const uint4 *in = reinterpret_cast<const uint4 *>(r.in);
    uint4 *out = reinterpret_cast<uint4 *>(r.out);
    const uint64_t lt = lanemask_lt();
    constexpr int U = 4;
    for (uint64_t base = beg; base < end; base += U * WAVE) {
        uint4 t[U];
        bool ok[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint64_t i = base + (uint64_t)k * WAVE + lane;
            ok[k] = i < end;
            if (ok[k]) t[k] = in[i];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t d = t[k].x & mask;
            uint32_t old = 0, fin = 0;
            if (ok[k]) old = atomicAdd(&lds_u32[d], 1u);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (ok[k]) fin = *reinterpret_cast<volatile uint32_t *>(&lds_u32[d]);
            uint32_t dest = old;
            uint64_t multi = __ballot(ok[k] && (fin - old) > 1u);
            while (multi) {
                const int leader = __ffsll((unsigned long long)multi) - 1;
                const uint32_t dd = __shfl(d, leader, 64);
                const bool in_group = ok[k] && d == dd;
                const uint64_t g = __ballot(in_group);
                if (in_group) dest = fin - (uint32_t)__popcll(g) + (uint32_t)__popcll(g & lt);
                multi &= ~g;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (ok[k]) out[dest] = t[k];
        }
    }
}

// ----------------------------------------------------------------------- plan

struct PlanArgs {
    const uint64_t *histR, *histS;
    Unit           *units, *build_units;
    BucketMeta     *meta;
    PlanSummary    *summary;
    uint32_t        lds_cap;        // largest build side served by an LDS table
    uint32_t        lds_max_slots;  // LDS slot budget
    uint32_t        unit_lds;       // probe tuples per unit, LDS buckets
    uint32_t        unit_hbm;       // probe tuples per unit, HBM buckets
    uint32_t        build_chunk;    // build tuples per HBM build unit
};

__global__ __launch_bounds__(1024) void k_plan(PlanArgs a, int bits)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    __shared__ uint64_t red[4];
    const uint32_t bins = 1u << bits;
    const uint32_t per = (bins + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, bins);
    if (threadIdx.x < 4) red[threadIdx.x] = 0;

    uint64_t nu = 0, nbu = 0, slots = 0, hbm_units = 0;
    uint32_t max_build = 0, max_slots = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const uint64_t cR = a.histR[b], cS = a.histS[b];
        if (cR == 0 || cS == 0) continue;
        const uint64_t pc = cR >= cS ? cR : cS, bc = cR >= cS ? cS : cR;   // rhjoin.c:86 (>=)
        if (bc <= a.lds_cap) {
            nu += (pc + a.unit_lds - 1) / a.unit_lds;
            max_build = max(max_build, (uint32_t)bc);
            uint32_t s = (uint32_t)(bc + (bc >> 1)) + 1u;
            s = min(max(s, 64u), a.lds_max_slots);
            max_slots = max(max_slots, s);
        } else {
            const uint64_t u = (pc + a.unit_hbm - 1) / a.unit_hbm;
            nu += u; hbm_units += u;
            nbu += (bc + a.build_chunk - 1) / a.build_chunk;
            uint32_t lg = 64 - __clzll((unsigned long long)(2 * bc - 1));   // pow2 >= 2*bc
            slots += 1ull << lg;
        }
    }
    uint64_t tot_u, tot_b, tot_s;
    uint64_t ubase = block_excl_scan<1024>(nu, &tot_u, sm);
    uint64_t bbase = block_excl_scan<1024>(nbu, &tot_b, sm);
    uint64_t sbase = block_excl_scan<1024>(slots, &tot_s, sm);
    atomicMax((unsigned long long *)&red[0], (unsigned long long)max_build);
    atomicMax((unsigned long long *)&red[1], (unsigned long long)max_slots);
    atomicAdd((unsigned long long *)&red[2], (unsigned long long)hbm_units);

    for (uint32_t b = b0; b < b1; ++b) {
        const uint64_t cR = a.histR[b], cS = a.histS[b];
        BucketMeta m = {0, 0, 0};
        if (cR != 0 && cS != 0) {
            const uint64_t pc = cR >= cS ? cR : cS, bc = cR >= cS ? cS : cR;
            uint32_t span;
            if (bc <= a.lds_cap) {
                uint32_t s = (uint32_t)(bc + (bc >> 1)) + 1u;
                m.slots = min(max(s, 64u), a.lds_max_slots);
                m.mode = 1;
                span = a.unit_lds;
            } else {
                const uint32_t lg = 64 - __clzll((unsigned long long)(2 * bc - 1));
                m.slots = lg;
                m.mode = 2;
                m.table_off = sbase;
                sbase += 1ull << lg;
                span = a.unit_hbm;
                for (uint64_t o = 0; o < bc; o += a.build_chunk) {
                    Unit u; u.off = o; u.bucket = b; u.count = (uint32_t)min((uint64_t)a.build_chunk, bc - o);
                    a.build_units[bbase++] = u;
                }
            }
            for (uint64_t o = 0; o < pc; o += span) {
                Unit u; u.off = o; u.bucket = b; u.count = (uint32_t)min((uint64_t)span, pc - o);
                a.units[ubase++] = u;
            }
        }
        a.meta[b] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        PlanSummary s;
        s.units = tot_u; s.build_units = tot_b; s.hbm_slots = tot_s;
        s.max_build_lds = red[0]; s.max_lds_slots = red[1]; s.hbm_units = red[2];
        s.matches = 0; s.pad = 0;
        *a.summary = s;
    }
}

// ---------------------------------------------------------------- hash tables

// HBM table insert: one thread per build tuple of an HBM bucket.
__global__ __launch_bounds__(256) void k_build_hbm(JoinArgs a, const Unit *build_units)
{
    if (blockIdx.x >= a.summary->build_units) return;
    const Unit un = build_units[blockIdx.x];
    const uint32_t b = un.bucket;
    const bool flip = a.histR[b] < a.histS[b];
    const rhj_tuple *bd = flip ? a.partR + a.psumR[b] : a.partS + a.psumS[b];
    const BucketMeta m = a.meta[b];
    unsigned long long *tbl = (unsigned long long *)(a.tables + m.table_off);
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

struct LdsTable {
    typedef uint32_t slot_t;
    uint32_t *t;
    uint32_t  slots;
    __device__ __forceinline__ uint32_t home(uint64_t h) const { return __umulhi((uint32_t)(h >> 32), slots); }
    __device__ __forceinline__ uint32_t tag(uint64_t h) const { return (uint32_t)(h >> 16) & 0xffffu; }
    __device__ __forceinline__ uint32_t next(uint32_t s) const { return s + 1 == slots ? 0 : s + 1; }
    __device__ __forceinline__ bool probe(uint32_t s, uint32_t tg, bool &hit, uint32_t &pos) const
    {
        const uint32_t e = t[s];
        const uint32_t et = e >> 16;
        if (e == 0 || et < tg) return false;       // end of this key's run
        hit = et == tg;
        pos = (e & 0xffffu) - 1u;
        return true;
    }
};

struct HbmTable {
    typedef uint64_t slot_t;
    const uint64_t *t;
    uint32_t  lg;
    __device__ __forceinline__ uint64_t home(uint64_t h) const { return h >> (64 - lg); }
    __device__ __forceinline__ uint32_t tag(uint64_t h) const { return (uint32_t)h; }
    __device__ __forceinline__ uint64_t next(uint64_t s) const { return (s + 1) & ((1ull << lg) - 1ull); }
    __device__ __forceinline__ bool probe(uint64_t s, uint32_t tg, bool &hit, uint32_t &pos) const
    {
        const uint64_t e = t[s];
        const uint32_t et = (uint32_t)(e >> 32);
        if (e == 0 || et < tg) return false;
        hit = et == tg;
        pos = (uint32_t)e - 1u;
        return true;
    }
};

template <int BLOCK>
__device__ __forceinline__ void lds_build(uint32_t *tbl, uint32_t slots, const rhj_tuple *bd, uint32_t bc)
{
    for (uint32_t s = threadIdx.x; s < slots; s += BLOCK) tbl[s] = 0;
    __syncthreads();
    LdsTable T{tbl, slots};
    for (uint32_t i = threadIdx.x; i < bc; i += BLOCK) {
        const uint64_t h = mix64(bd[i].value);
        uint32_t s = T.home(h);
        uint32_t v = (T.tag(h) << 16) | (i + 1u);
        for (;;) {
            const uint32_t old = atomicMax(&tbl[s], v);
            if (old == 0) break;
            if (old < v) v = old;
            s = T.next(s);
        }
    }
    __syncthreads();
}

// Probe one unit.  WRITE = false: count matches (first pass).  WRITE = true: emit
// (row_idR,row_idS) pairs at unit_base[u] + running offset, in probe order and, per
// probe tuple, in table-walk order = descending build position (rhjoin.c:227,240-246).
template <int BLOCK, bool WRITE, class Table>
__device__ __forceinline__ void probe_unit(const JoinArgs &a, const Table &T, const rhj_tuple *pr,
                                           const rhj_tuple *bd, uint32_t count, bool flip, uint32_t u,
                                           uint64_t *sm)
{
    uint64_t local = 0;                              // count pass: this thread's matches
    uint64_t run = WRITE ? a.unit_base[u] : 0;       // write pass: next free output slot
    for (uint32_t t0 = 0; t0 < count; t0 += BLOCK) {
        const uint32_t i = t0 + threadIdx.x;
        const bool ok = i < count;
        uint64_t key = 0, prow = 0;
        uint32_t c = 0;
        uint64_t m0 = 0, m1 = 0;                     // row ids of the first two matches
        typename Table::slot_t s0 = 0;
        uint32_t tg = 0;
        if (ok) {
            const uint4 q = reinterpret_cast<const uint4 *>(pr)[i];
            key = ((uint64_t)q.y << 32) | q.x;
            prow = ((uint64_t)q.w << 32) | q.z;
            const uint64_t h = mix64(key);
            s0 = T.home(h);
            tg = T.tag(h);
            typename Table::slot_t s = s0;
            for (;;) {
                bool hit; uint32_t pos;
                if (!T.probe(s, tg, hit, pos)) break;
                if (hit) {
                    if (WRITE) {
                        const uint4 w = reinterpret_cast<const uint4 *>(bd)[pos];
                        if (w.x == q.x && w.y == q.y) {
                            const uint64_t brow = ((uint64_t)w.w << 32) | w.z;
                            if (c == 0) m0 = brow; else if (c == 1) m1 = brow;
                            ++c;
                        }
                    } else {
                        c += bd[pos].value == key;
                    }
                }
                s = T.next(s);
            }
        }
        if (!WRITE) { local += c; continue; }

        uint64_t tile_total;
        const uint64_t off = run + block_excl_scan<BLOCK>((uint64_t)c, &tile_total, sm);
        run += tile_total;
        if (c) {
            uint4 *out = reinterpret_cast<uint4 *>(a.out);
            const uint64_t cap = a.out_capacity;
            if (c <= 2) {
                if (off < cap) {
                    const uint64_t r = flip ? m0 : prow, s = flip ? prow : m0;
                    out[off] = make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)s, (uint32_t)(s >> 32));
                }
                if (c == 2 && off + 1 < cap) {
                    const uint64_t r = flip ? m1 : prow, s = flip ? prow : m1;
                    out[off + 1] = make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)s, (uint32_t)(s >> 32));
                }
            } else {                                  // long duplicate run: walk again, emit as we go
                uint64_t at = off;
                typename Table::slot_t s = s0;
                for (;;) {
                    bool hit; uint32_t pos;
                    if (!T.probe(s, tg, hit, pos)) break;
                    if (hit) {
                        const rhj_tuple w = bd[pos];
                        if (w.value == key) {
                            if (at < cap) {
                                const uint64_t r = flip ? w.row_id : prow, sv = flip ? prow : w.row_id;
                                out[at] = make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)sv, (uint32_t)(sv >> 32));
                            }
                            ++at;
                        }
                    }
                    s = T.next(s);
                }
            }
        }
    }
    if (!WRITE) {
        uint64_t total;
        block_excl_scan<BLOCK>(local, &total, sm);
        if (threadIdx.x == 0) a.unit_count[u] = total;
    }
}

template <int BLOCK, bool WRITE>
__global__ __launch_bounds__(BLOCK) void k_probe(JoinArgs a, uint32_t lds_slots_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tbl = reinterpret_cast<uint32_t *>(smem);
    uint64_t *sm = reinterpret_cast<uint64_t *>(smem + (((size_t)lds_slots_max * 4 + 15) & ~(size_t)15));
    const uint32_t u = blockIdx.x;
    if (u >= a.summary->units) return;
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const uint64_t cR = a.histR[b], cS = a.histS[b];
    const bool flip = cR < cS;                                         // S is streamed (r_s == 1)
    const rhj_tuple *pr = (flip ? a.partS + a.psumS[b] : a.partR + a.psumR[b]) + un.off;
    const rhj_tuple *bd = flip ? a.partR + a.psumR[b] : a.partS + a.psumS[b];
    const uint32_t bc = (uint32_t)(flip ? cR : cS);
    const BucketMeta m = a.meta[b];
    if (m.mode == 1) {
        lds_build<BLOCK>(tbl, m.slots, bd, bc);
        LdsTable T{tbl, m.slots};
        probe_unit<BLOCK, WRITE>(a, T, pr, bd, un.count, flip, u, sm);
    } else {
        HbmTable T{a.tables + m.table_off, m.slots};
        probe_unit<BLOCK, WRITE>(a, T, pr, bd, un.count, flip, u, sm);
    }
}

// Exclusive scan of n u64 counts (one workgroup, chunked); total -> *total_out.
__global__ __launch_bounds__(1024) void k_offsets(const uint64_t *cnt, uint64_t *base, const uint64_t *n_ptr,
                                                  uint64_t n_fixed, uint64_t *total_out)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    const uint64_t n = n_ptr ? *n_ptr : n_fixed;
    uint64_t carry = 0;
    for (uint64_t i0 = 0; i0 < n; i0 += 1024) {
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t v = i < n ? cnt[i] : 0;
        uint64_t tot;
        const uint64_t e = block_excl_scan<1024>(v, &tot, sm);
        if (i < n) base[i] = carry + e;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// --------------------------------------------------------------------- filter

constexpr int FILTER_ROUNDS = 16;                    // 64-element rounds per wave
constexpr int FILTER_TILE = 256 / WAVE * FILTER_ROUNDS * WAVE;   // 4096 elements per workgroup

__device__ __forceinline__ bool filter_pred(uint64_t v, uint64_t k, int op)
{
    return op == 0 ? v < k : op == 1 ? v > k : v == k;
}

// Pass 1: evaluate the predicate once, keep it as one 64-bit mask per 64 elements,
// count hits per 4096-element tile.
__global__ __launch_bounds__(256) void k_filter_mask(const uint64_t *col, const uint64_t *sel, uint64_t n, int op,
                                                     uint64_t value, uint64_t *masks, uint64_t *tile_count)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t wbase = (uint64_t)blockIdx.x * FILTER_TILE + (uint64_t)w * FILTER_ROUNDS * WAVE;
    uint32_t cnt = 0;
#pragma unroll 4
    for (int k = 0; k < FILTER_ROUNDS; ++k) {
        const uint64_t i = wbase + (uint64_t)k * WAVE + lane;
        bool p = false;
        if (i < n) {
            const uint64_t v = sel ? col[sel[i]] : col[i];
            p = filter_pred(v, value, op);
        }
        const uint64_t mk = __ballot(p);
        if (lane == 0 && wbase + (uint64_t)k * WAVE < n) masks[(wbase >> 6) + k] = mk;
        cnt += (uint32_t)__popcll(mk);
    }
    if (lane == 0) wsum[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = (uint64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Pass 2: turn the masks into the ascending index list.
__global__ __launch_bounds__(256) void k_filter_write(uint64_t n, const uint64_t *masks, const uint64_t *tile_base,
                                                      uint64_t *out)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t wbase = (uint64_t)blockIdx.x * FILTER_TILE + (uint64_t)w * FILTER_ROUNDS * WAVE;
    uint64_t mine = 0;
    if (lane < FILTER_ROUNDS && wbase + (uint64_t)lane * WAVE < n) mine = masks[(wbase >> 6) + lane];
    uint32_t pc = (uint32_t)__popcll(mine);
    uint32_t incl = pc;                                // inclusive scan over the 16 round counts
#pragma unroll
    for (int d = 1; d < FILTER_ROUNDS; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += y;
    }
    const uint32_t wave_total = __shfl(incl, FILTER_ROUNDS - 1, 64);
    if (lane == 0) wsum[w] = wave_total;
    __syncthreads();
    uint64_t base = tile_base[blockIdx.x];
    for (uint32_t i = 0; i < w; ++i) base += wsum[i];
    const uint64_t lt = lanemask_lt();
    for (int k = 0; k < FILTER_ROUNDS; ++k) {
        const uint64_t mk = __shfl(mine, k, 64);
        const uint32_t before = __shfl(incl - pc, k, 64);
        if ((mk >> lane) & 1ull)
            out[base + before + (uint32_t)__popcll(mk & lt)] = wbase + (uint64_t)k * WAVE + lane;
    }
}

}  // namespace rhj
