// rhj_join_fused.hip.h — fused LDS join: CSR slot index, probe with stash, chained output offsets, deferred streaming emit
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
#pragma once
#include "rhj_common.hip.h"
#include "rhj_partition.hip.h"
#include "rhj_join_tiled.hip.h"

namespace rhj {

// ------------------------------------------------------------- fused LDS join
//
// One persistent workgroup per CU takes units = (bucket, up to FJ_SPAN probe tuples) in canonical
// order through a ticket, so that a unit's predecessors are always running or done.
//   build    CSR slot index of the bucket's build side in LDS (fj_build)
//   phase 1  stream the unit's probe keys; one 8-entry tag window per tuple in LDS; ONE global
//            gather per candidate (key to verify + row id); per probe tuple stash the match count
//            (u8) and the first match's build row id (u64), further matches go to the overflow
//            stash (fj_count_batch).  When the build tuples fit in LDS too (RES, build side <= ~7 K
//            tuples) the candidates of a key are taken as ONE run of its sorted slot and verified
//            there (fj_run_of, fj_count_res): exact counts without a round per match, whatever the
//            number of duplicates; multi-match tuples stash (run start, count)
//   chain    publish the unit's match total right away (8-byte {flag,value} word per unit,
//            agent-scope relaxed atomics)
//   emit     deferred behind the NEXT unit's build and phase 1: decoupled look-back over the
//            predecessors (never waits by then), then stream probe row ids + stash + overflow
//            stash and write the pairs at their final canonical positions (fj_emit_stream).
//            Resident units with multi-match tuples emit right behind their own phase 1, output-
//            centric from the runs (fj_emit_res); units the stash cannot describe emit immediately by
//            walking the index again.
// Random global accesses per probe tuple: one 128-byte line (the gather); everything
// else is streaming or LDS.
constexpr int FJ_BLOCK = 1024;
constexpr int FJ_WAVES = FJ_BLOCK / WAVE;
constexpr int FJ_V = 4;
constexpr int FJ_BATCH = FJ_BLOCK * FJ_V;       // 4096 probe tuples per batch
constexpr uint32_t FJ_SPAN = 65536;             // probe tuples per unit
constexpr uint32_t FJ_LDS_EXTRA = 2048;         // bytes of LDS behind the table (static: the words below, the speculative kernel's group prefixes)
constexpr uint32_t FJ_OVF_CAP = 32768;          // overflow entries per unit before it falls back to the index walk
constexpr uint32_t FJ_REC_CAP = 512;                  // records (second and later matches; 8 bytes each, room for 16) of one 256-tuple group: two a tuple on average; more: k_join_walk takes the unit
// bytes of FusedArgs::ovf for `wgs` workgroups: two overflow buffers each, and behind them all a piece of FJ_REC_CAP records per wave
// (fj_walk_group's staging)
constexpr size_t fj_ovf_bytes(size_t wgs) { return wgs * 2 * (size_t)32768 * 8 + wgs * (size_t)(1024 / 64) * FJ_REC_CAP * 16; }
constexpr uint32_t FJ_OVF_J = 15;               // match ordinals 1..15 (2nd..16th match) have an overflow slot
#ifndef FJ_WIN
#define FJ_WIN 8                                // entries of a slot compared at once (the array is padded by 8)
#endif
constexpr uint32_t FJ_GROUPS = FJ_SPAN / 256;   // a group = the 256 tuples one wave counts in one batch
constexpr uint32_t FJ_PATCH_CAP = 128;          // irregular tuples per unit whose match rounds are noted in the overflow buffer's tail
constexpr uint32_t FJ_OVF_ENT = FJ_OVF_CAP - FJ_PATCH_CAP / 2;   // overflow entries a unit may use (the tail holds the patch words)
constexpr uint64_t FJ_NO_TOTAL = ~0ull;         // PlanSummary::matches when the last unit's inclusive prefix was never published

struct FusedArgs {
    JoinArgs  j;
    uint8_t  *stash_cnt;      // [nR + nS] matches per probe tuple, saturating at 255
    uint64_t *stash_row;      // [nR + nS] build row id of the first match
    uint64_t *status;         // [units] (flag << 62) | value ; flag 1 = unit total, 2 = inclusive prefix
    uint32_t *ticket;         // word 0: next unit; word 1: workgroups that are through (the last one out leaves the match total);
                              // word 2: entries of the walk list; word 3: k_join_walk's own ticket; word 4: the foreign-key
                              // speculation failed (k_join_spec, k_join_exact); word 5: why k_join_exact gave up (1 a check of the
                              // speculation failed, 2 an input it does not take); words 6..7 (one u64): the totals it predicted, summed
    uint64_t  nR;
    uint32_t  allow_resident;
    uint32_t  radix_bits;     // the join's radix width (the bits every key of a bucket shares)
    uint32_t  lr_mode;        // low-radix path (rhj_lowradix.hip.h): a tuple with several matches leaves the place of its pairs in its stash row
    uint32_t  spec;           // foreign-key speculation (k_join_spec, below): 0 none, 1 every S tuple has exactly one match, 2 every R tuple
    uint32_t *xrows;          // k_join_exact (rhj_join_exact.hip.h): [grid][XJ_SCRATCH] a unit's build row ids in entry order
    uint64_t  unit_bound;     // status words there are
    uint64_t *host_summary;   // pinned host block that receives the plan summary (with the match total) at the end, or null
    uint64_t *dbg;            // diagnostic builds only: [units][8] phase stamps (100 MHz), else null
    uint64_t *ovf;            // [grid][2][FJ_OVF_CAP] build row ids of second and later matches (per workgroup, double-buffered)
    uint32_t *ovf_base;       // [grid][2][FJ_SPAN / 256][16] first overflow slot of (256-tuple group, match ordinal)
    struct FjWalkItem *walk;  // [unit_bound] units whose pairs k_join_walk writes (the stash cannot describe them)
};

// A unit the streaming emit cannot serve — a tag collision beside two or more matches, more than 16 matches of one
// tuple, an overflow stash that ran full — is handed to k_join_walk, which rebuilds the index and walks it again.
struct FjWalkItem {
    uint32_t unit;
    uint32_t flags;           // bit 3 (8): nothing was stashed for this unit (k_join_spec's direct path): every tuple is recounted from the index;
                              // bit 0: the unit's build tuples were LDS-resident in k_join_fused (its stash keeps runs for multi-match
                              // tuples); bit 1: k_join_fused hashed with FjHashT<true> — the stash's "a tag hit was a foreign key"
                              // bit is a statement about THAT hash's tags, so the walk must index with the same one
    uint64_t base;            // first output position of the unit (its look-back is done)
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
// Two hashes behind slot and tag.  H32 (kernels that may keep build tuples LDS-resident: C2, C4, the contest's joins): ONE
// 32-bit multiply of the key without the radix bits its bucket shares, the upper key word folded in by a rotation; slot =
// upper half scaled to hs by a 24-bit multiply (hs < 2^16), tag = lower half after one xor-shift.  mix64's two 64-bit
// multiplies are eight quarter-rate 32-bit ones, paid in the build's count pass, its fill pass and the probe: A/B (own process
// per build, r03a) c3b14 -3 %, C2 -7 %, C4 -1 % of the kernel.  The gather kernels (C3 at 12 bits) keep mix64: their time is
// the gathers', and with the short hash they measured 3 % SLOWER (2.47 -> 2.55 ms, r03c; slot by __umulhi the same).
// Results never depend on the hash: every candidate is verified against the 64-bit key.
template <bool H32> struct FjHashT;
template <> struct FjHashT<false> {
    typedef uint64_t type;
    static __device__ __forceinline__ type hash(uint64_t key, uint32_t) { return mix64(key); }
    static __device__ __forceinline__ uint32_t slot(type h, uint32_t hs) { return __umulhi((uint32_t)(h >> 32), hs); }
    static __device__ __forceinline__ uint32_t tag(type h) { return min((uint32_t)(h >> 16) & 0xffffu, 0xfffdu) + 1u; }
};
template <> struct FjHashT<true> {
    typedef uint32_t type;
    static __device__ __forceinline__ type hash(uint64_t key, uint32_t bits)
    {
        const uint32_t x = (uint32_t)(key >> bits) ^ __builtin_rotateleft32((uint32_t)(key >> 32) >> bits, 16);
        const uint32_t h = x * 0x9e3779b1u;
        return h ^ (h >> 15);
    }
    static __device__ __forceinline__ uint32_t slot(type h, uint32_t hs) { return __umul24(h >> 16, hs) >> 16; }
    static __device__ __forceinline__ uint32_t tag(type h) { return min(h & 0xffffu, 0xfffdu) + 1u; }
};
template <bool H32>
struct FjIndexT {
    typedef typename FjHashT<H32>::type hash_t;
    uint32_t *ent;       // [bc + 8]
    uint32_t *dirw;      // [(hs + 3) / 2]
    uint32_t  hs;
    uint32_t  bits;      // the join's radix bits: shared by every key of the bucket
    // Build sides of up to SMALL batches of 4096 tuples are read and hashed ONCE: the count pass parks (slot, tag) in the still
    // unused entry array and the fill pass takes the words from there through registers.  Seven batches for the gather
    // kernels — a 12-bit bucket of a 100 M relation is 24.4 K tuples: 1.2 GB of key reads and a second mix64 per build tuple
    // less on C3, -1.9 % of the kernel (r03, A/B) — four where the resident code needs the registers.
    static constexpr int SMALL = H32 ? 4 : 7;
    __device__ __forceinline__ hash_t hash(uint64_t key) const { return FjHashT<H32>::hash(key, bits); }
    __device__ __forceinline__ uint32_t slot(hash_t h) const { return FjHashT<H32>::slot(h, hs); }
    static __device__ __forceinline__ uint32_t tag(hash_t h) { return FjHashT<H32>::tag(h); }
    __device__ __forceinline__ uint32_t H(uint32_t j) const { return reinterpret_cast<const uint16_t *>(dirw)[j]; }
};

// Partitioned relations as the fused kernel sees them: rhj_tuple (16 B), or — N32: the partition found every row id
// below 2^32 and wrote Tuple12 — 12 bytes per tuple.  Whole tuple as {key lo, key hi, row id lo, row id hi}.
template <bool N32> __device__ __forceinline__ uint4 pt_load(const rhj_tuple *base, uint64_t i)
{
    if (N32) { const Tuple12 x = reinterpret_cast<const Tuple12 *>(base)[i]; return make_uint4(x.klo, x.khi, x.rid, 0u); }
    return reinterpret_cast<const uint4 *>(base)[i];
}
// A unit's probe tuples are read once: with the nt policy they leave the build sides' lines in the L2s alone — C3 probe stage
// 1.893 -> 1.869 ms (gpurun_out/r04t/aux_abn.txt; through a descriptor: nt 1.885, sc1 1.909, nt sc1 1.867, sc0 sc1 1.899, sc0 nt
// 1.879; the PAIR stores with nt: 1.908 -> 1.971, they lose the L2's write combining; the resident kernels — no gathers — gain
// nothing: C4 5.38 -> 5.42, and keep the default policy).  -DFJ_PROBE_PLAIN: the default policy everywhere (A/B).
template <bool N32> __device__ __forceinline__ uint4 pt_load_nt(const rhj_tuple *base, uint64_t i)
{
#ifndef FJ_PROBE_PLAIN
    if (N32) {
        const uint32_t *x = reinterpret_cast<const uint32_t *>(reinterpret_cast<const Tuple12 *>(base) + i);
        return make_uint4(__builtin_nontemporal_load(x), __builtin_nontemporal_load(x + 1), __builtin_nontemporal_load(x + 2), 0u);
    }
#endif
    return pt_load<N32>(base, i);
}
#ifdef FJ_OUT_NT
typedef uint32_t fj_v4u __attribute__((ext_vector_type(4)));
#define FJ_STORE_PAIR(ptr, v) do { const uint4 v_ = (v); const fj_v4u w_ = {v_.x, v_.y, v_.z, v_.w}; __builtin_nontemporal_store(w_, reinterpret_cast<fj_v4u *>(ptr)); } while (0)
#else
#define FJ_STORE_PAIR(ptr, v) (*(ptr) = (v))
#endif
template <bool N32> __device__ __forceinline__ uint2 pt_load_key(const rhj_tuple *base, uint64_t i)
{
    if (N32) { const Tuple12 *x = reinterpret_cast<const Tuple12 *>(base) + i; return make_uint2(x->klo, x->khi); }
    return reinterpret_cast<const uint2 *>(base)[2 * i];
}

// hit mask of the first min(n, 8) entries of the window at `start`
template <class IX>
__device__ __forceinline__ uint32_t fj_window(const IX &X, uint32_t start, uint32_t n, uint32_t tgs)
{
    uint32_t e[FJ_WIN];
#pragma unroll
    for (int j = 0; j < FJ_WIN; ++j) e[j] = X.ent[start + j];
    const uint32_t tg = tgs >> 16;
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < FJ_WIN; ++j) m |= ((e[j] >> 16) == tg) ? (1u << j) : 0u;
    return m & ((1u << min(n, (uint32_t)FJ_WIN)) - 1u);
}

// Per probe tuple: sn = window start | remaining slot length << 16, tm = tag << 16 | hit mask of the
// current window.
template <class IX>
__device__ __forceinline__ void fj_lookup(const IX &X, uint64_t key, bool ok, uint32_t &sn, uint32_t &tm)
{
    const auto h = X.hash(key);
    const uint32_t s = X.slot(h);
    const uint32_t d0 = X.H(s + 1u), n = ok ? X.H(s + 2u) - d0 : 0u;
    const uint32_t tgs = X.tag(h) << 16;
    sn = d0 | (n << 16);
    tm = tgs | fj_window(X, d0, n, tgs);
}

// One round of the probe: every tuple that still has a candidate hands out its next one (pos[k], a
// build position) — first from the window's hit mask, and when that is used up and the slot is longer
// than the window, from the next window.  Returns whether any lane of the wave got a candidate.
template <class IX>
__device__ __forceinline__ bool fj_round(const IX &X, uint32_t (&sn)[FJ_V], uint32_t (&tm)[FJ_V], uint32_t (&pos)[FJ_V],
                                         bool &last)
{
    bool more = false;
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) more = more || ((tm[k] & 0xffu) == 0 && (sn[k] >> 16) > (uint32_t)FJ_WIN);
    while (__ballot(more) != 0) {                     // rare: a slot with more than 8 entries
        more = false;
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            if ((tm[k] & 0xffu) == 0 && (sn[k] >> 16) > (uint32_t)FJ_WIN) {
                sn[k] += (uint32_t)FJ_WIN - ((uint32_t)FJ_WIN << 16);     // start += window, length -= window
                tm[k] |= fj_window(X, sn[k] & 0xffffu, sn[k] >> 16, tm[k] & 0xffff0000u);
                more = more || ((tm[k] & 0xffu) == 0 && (sn[k] >> 16) > (uint32_t)FJ_WIN);
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
    for (int k = 0; k < FJ_V; ++k) rest = rest || (tm[k] & 0xffu) != 0 || (sn[k] >> 16) > (uint32_t)FJ_WIN;
    last = __ballot(rest) == 0;
    return __ballot(found) != 0;
}

// Build the index of one bucket's build side (whole workgroup).  RES: the tuples are copied to LDS
// on the way and the second pass reads them there.  `tmp` is global scratch of at least 4 * bc bytes
// for the cooperative sort of long slots.
constexpr uint32_t FJ_LONG = 16;                      // slots above this are filled by fetch-add and ranked afterwards
// FJ_SMALL (FjIndexT::SMALL): batches of 4096 build tuples whose (slot, tag) words are kept in registers for the fill pass
template <bool RES, bool N32, class IX>
__device__ __forceinline__ void fj_build(const IX &X, const rhj_tuple *part, uint64_t boff, uint32_t bc, uint4 *ltup,
                                         uint32_t *tmp, uint32_t *wsum, uint32_t *sh_pick, bool any_order = false)
{
    // any_order (workgroup-uniform): the entries of a slot may stand in ANY order — every slot is filled by fetch-add and nothing
    // is ranked.  For the units of the speculative gather kernel that the hypothesis' relation probes: a probe tuple there has one
    // match or the speculation is off (and the ordinary kernel builds its own index), so the order the reference hands out a
    // tuple's matches in (descending build position, rhjoin.c:219-250) never shows.  Ordered insertion is ~1.5 LDS atomics with a
    // dependent loop per build tuple against one: C3 probe stage -0.7 % (timing-only on all units: 1.841 -> 1.814 ms).
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t ndw = (X.hs + 3u) / 2u;
    for (uint32_t i = tid; i < ndw; i += FJ_BLOCK) X.dirw[i] = 0;
    for (uint32_t i = tid; i < bc; i += FJ_BLOCK) X.ent[i] = 0;
    if (tid < 8) X.ent[bc + tid] = 0xffff0000u;
    __syncthreads();
    // ---- count: H[s + 1] += 1.  Build sides of up to 4 batches (16 K tuples) are hashed only once: the
    // (slot, tag) word of tuple i is parked in ent[i], picked up into registers before the fill pass
    // clears the array, and the fill pass needs neither the key nor a second hash.
    constexpr int FJ_SMALL = IX::SMALL;
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
                    const auto h = X.hash(((uint64_t)t[k].y << 32) | t[k].x);
                    const uint32_t sl = X.slot(h);
                    if (small) X.ent[i] = (sl << 16) | X.tag(h);       // parked in the still unused entry array
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
        if (!any_order && n <= FJ_LONG) {
            for (uint32_t p = a;; ++p) {
                const uint32_t old = atomicMax(&X.ent[p], v);
                if (old == 0) break;
                v = min(old, v);
            }
        } else {
            has_long = !any_order;
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
                    const auto h = X.hash(((uint64_t)t[k].y << 32) | t[k].x);
                    insert(X.slot(h), (X.tag(h) << 16) | i);
                }
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) t[k] = tn[k];
        }
    }
    if (__syncthreads_or(has_long)) {
        // Long slots of up to 64 entries (a key with 17..64 duplicates: the contest's relations are full of them — a
        // bucket of `small` has ~100 such slots): ranked by ONE WAVE each, in registers — lane i holds entry i, its rank
        // is the number of larger entries (64 scalar broadcasts), and the entry goes to its place; the waves find their
        // slots in the stripes of the slot array they scan, no barrier.  (Ranked one at a time by the whole workgroup,
        // with three barriers each, these slots were 190 us of a 300 us unit.)
        for (uint32_t s0 = w * WAVE; s0 < X.hs; s0 += FJ_BLOCK) {
            const uint32_t sl = s0 + lane;
            uint32_t a = 0, n = 0;
            if (sl < X.hs) { a = X.H(sl + 1u); n = X.H(sl + 2u) - a; }
            uint64_t todo = __ballot(n > FJ_LONG && n <= (uint32_t)WAVE);
            while (todo) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t sa = (uint32_t)__builtin_amdgcn_readlane((int)a, src), sn = (uint32_t)__builtin_amdgcn_readlane((int)n, src);
                const uint32_t v = lane < sn ? X.ent[sa + lane] : 0u;
                uint32_t r = 0;
                for (uint32_t j = 0; j < sn; ++j) r += (uint32_t)__builtin_amdgcn_readlane((int)v, (int)j) > v;
                if (lane < sn) X.ent[sa + r] = v;      // (a wave's LDS operations complete in order: every read above is done)
            }
        }
        __syncthreads();
        // longer slots: one at a time, ranked by the whole workgroup (descending value)
        for (uint32_t next = 0;;) {
            if (tid == 0) *sh_pick = 0xffffffffu;
            __syncthreads();
            for (uint32_t sl = tid; sl < X.hs; sl += FJ_BLOCK)
                if (sl >= next && X.H(sl + 2u) - X.H(sl + 1u) > (uint32_t)WAVE) { atomicMin(sh_pick, sl); break; }
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
#ifndef FJ_GATHER_AUX
#define FJ_GATHER_AUX 16   /* sc1 */
#endif
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
#ifdef FJ_ABL_G8          // timing experiment only (wrong results): 8-byte gathers from a third less memory
            typedef uint32_t v2 __attribute__((ext_vector_type(2)));
            const v2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(pos * 8u), 0, FJ_GATHER_AUX);
            return make_uint4(w.x, w.y, w.x, 0u);
#endif
            typedef uint32_t v3 __attribute__((ext_vector_type(3)));
            const v3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(pos * 12u), 0, FJ_GATHER_AUX);
            return make_uint4(v.x, v.y, v.z, 0u);
        }
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(pos * 16u), 0, FJ_GATHER_AUX);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    // the same descriptor for a STREAM (a unit's probe tuples), with the cache policy as a compile-time constant (1 sc0, 2 nt, 16 sc1)
    template <int AUX> __device__ __forceinline__ uint4 stream(uint32_t pos) const
    {
        if (N32) {
            typedef uint32_t v3 __attribute__((ext_vector_type(3)));
            const v3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(pos * 12u), 0, AUX);
            return make_uint4(v.x, v.y, v.z, 0u);
        }
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(pos * 16u), 0, AUX);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};

// Overflow stash of the gather path.  Phase 1 has every match's build row id in registers at the
// moment it verifies it, so besides the first one (stash_row) it keeps the others too: a wave that
// finds second-or-later matches in round r (r = 1, 2, ...; a tuple whose tag hits are all matches
// finds its (r+1)-th match exactly there) takes a contiguous run of slots with one LDS atomic, records
// the run's start for (its 256-tuple group, r) and stores the row ids in (k, lane) order.  The emit
// pass recomputes the same ranks from the stashed counts, so it needs neither the index nor a gather
// and can run any time later.  A tuple that breaks the rule because a tag hit of a FOREIGN key sits between its matches (a
// 16-bit tag collision inside one slot: ~300 of C3's 100 M probe tuples, but in 7 % of its units) leaves the rounds its
// matches were found in as a 16-bit map in the unit's patch list, from which the emit pass recomputes exactly what phase 1
// stored; more than 16 matches of one tuple, a full overflow buffer or patch list send the unit to k_join_walk.
struct FjOvf {
    uint64_t *buf;        // this unit's overflow entries
    uint32_t *table;      // this unit's [group][16] run starts (null: the wave keeps them itself, FjOvf::runs)
    uint32_t *counter;    // LDS bump counter
    uint32_t  gid;        // group of this wave in this batch
    uint32_t  runs;       // table == null: lane r holds the start of the run of match round r (what the table's row would hold)
};

template <bool RES, bool OVF, bool N32, class IX>
__device__ __forceinline__ void fj_count_batch(const IX &X, const FjGather<N32> &G, const uint4 *ltup,
                                               const uint4 (&q)[FJ_V], const bool (&okk)[FJ_V], uint32_t (&c)[FJ_V],
                                               uint32_t (&flo)[FJ_V], uint32_t (&fhi)[FJ_V], bool (&fp)[FJ_V],
                                               uint32_t (&bm)[FJ_V], FjOvf &O)
{
    uint32_t sn[FJ_V], tm[FJ_V];
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) {
        fj_lookup(X, ((uint64_t)q[k].y << 32) | q[k].x, okk[k], sn[k], tm[k]);
        c[k] = 0; flo[k] = 0; fhi[k] = 0; fp[k] = false; bm[k] = 0;
    }
    bool last = false;
    for (uint32_t round = 0; !last; ++round) {
        uint32_t pos[FJ_V];
        if (!fj_round(X, sn, tm, pos, last)) break;
        uint4 g[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            g[k] = make_uint4(0, 0, 0, 0);
#ifdef FJ_FLOOR          // timing experiment only (wrong results): every tag hit is taken for a match, its position for the row id
            if (pos[k] != 0xffffffffu) g[k] = make_uint4(q[k].x, q[k].y, pos[k], 0u);
#else
            if (pos[k] != 0xffffffffu) g[k] = RES ? ltup[pos[k]] : G.load(pos[k]);
#endif
        }
        bool ex[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
#ifdef FJ_ABL_G8
            const bool eq = pos[k] != 0xffffffffu && (g[k].x | 1u) != 0u;          // every tag hit counts (timing only)
#else
            const bool eq = pos[k] != 0xffffffffu && g[k].x == q[k].x && g[k].y == q[k].y;
#endif
            if (eq && c[k] == 0) { flo[k] = g[k].z; fhi[k] = g[k].w; }
            ex[k] = OVF && eq && c[k] != 0;
            c[k] += eq;
            if (OVF && eq) bm[k] |= 1u << min(round, 31u);        // the rounds this tuple's matches were found in
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
                    if (round <= FJ_OVF_J && O.table) O.table[O.gid * 16u + round] = base;
                }
                base = __builtin_amdgcn_readfirstlane(base);
                if (!O.table && lane == round) O.runs = base;
                uint32_t pre = base;
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    const uint32_t slot = pre + (uint32_t)__popcll(mk[k] & lt);
                    if (ex[k] && slot < FJ_OVF_ENT) reinterpret_cast<uint2 *>(O.buf)[slot] = make_uint2(g[k].z, g[k].w);
                    pre += (uint32_t)__popcll(mk[k]);
                }
            }
        }
    }
}

// Resident build sides (RES): the matches of a probe key without a round per match.  A slot's entries are sorted by
// (tag, position) descending, so the entries that carry the key's tag are ONE contiguous run: its start and length
// come from the 8-entry window's hit mask (slots of up to 8 entries: the foreign-key case) or from two binary
// searches over the slot (longer slots: duplicate-heavy build sides — the contest's own relations have keys with
// hundreds of duplicates, where a round per match made a wave spin through its longest chain).  Every entry of the
// run is verified against the resident tuple's 64-bit key.  cnt = verified matches; first = row id of the first one;
// run = start | cnt << 16 is what the emit pass needs when cnt >= 2 and the run is CLEAN (every entry a match: the
// j-th match is entry start + j); fp = some entry of the run belongs to another key (a 16-bit tag collision inside
// one slot) — with two or more matches beside it the unit takes the index-walking emit.
template <class IX>
__device__ __forceinline__ void fj_run_of(const IX &X, uint64_t key, bool ok, uint32_t &start, uint32_t &len)
{
    const auto h = X.hash(key);
    const uint32_t s = X.slot(h);
    const uint32_t d0 = X.H(s + 1u), n = ok ? X.H(s + 2u) - d0 : 0u;
    const uint32_t tg = X.tag(h);
    if (n <= (uint32_t)FJ_WIN) {
        const uint32_t m = fj_window(X, d0, n, tg << 16);
        start = d0 + (m ? (uint32_t)__builtin_ctz(m) : 0u);
        len = (uint32_t)__popc(m);
    } else {
        uint32_t lo = 0, hi = n;                       // first entry whose tag is <= tg
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((X.ent[d0 + mid] >> 16) > tg) lo = mid + 1u; else hi = mid; }
        const uint32_t lower = lo;
        hi = n;                                        // first entry whose tag is < tg
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((X.ent[d0 + mid] >> 16) >= tg) lo = mid + 1u; else hi = mid; }
        start = d0 + lower;
        len = lo - lower;
    }
}

constexpr uint32_t FJ_RUN_LOCK = 64;             // run entries verified lane by lane before the wave takes a long run together
template <class IX>
__device__ __forceinline__ void fj_count_res(const IX &X, const uint4 *ltup, const uint4 (&q)[FJ_V], const bool (&okk)[FJ_V],
                                             uint32_t (&c)[FJ_V], uint32_t (&flo)[FJ_V], uint32_t (&fhi)[FJ_V], bool (&fp)[FJ_V],
                                             uint32_t (&run)[FJ_V])
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t start[FJ_V], len[FJ_V];
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) {
        fj_run_of(X, ((uint64_t)q[k].y << 32) | q[k].x, okk[k], start[k], len[k]);
        c[k] = 0; flo[k] = 0; fhi[k] = 0;
    }
    // runs of up to FJ_RUN_LOCK entries (the foreign-key case and light duplication): lockstep over the run position,
    // the four tuples of a lane side by side (LDS reads only)
    uint32_t longest = 0;
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) longest = max(longest, min(len[k], FJ_RUN_LOCK));
    for (uint32_t j = 0; __ballot(j < longest) != 0; ++j) {
        // unconditional loads at a clamped position: the four entry reads, then the four tuple reads, are in flight
        // together (four predicated read -> read chains one after the other cost 800 cycles per step)
        uint32_t e[FJ_V];
        uint4 v[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) e[k] = X.ent[start[k] + (j < len[k] ? j : 0u)];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) v[k] = ltup[e[k] & 0xffffu];      // (an empty run reads the slot's first entry or a pad entry: position 0)
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const bool eq = j < min(len[k], FJ_RUN_LOCK) && v[k].x == q[k].x && v[k].y == q[k].y;
            if (eq && c[k] == 0) { flo[k] = v[k].z; fhi[k] = v[k].w; }
            c[k] += eq;
        }
    }
    // longer runs (a key with many duplicates on the build side), one tuple at a time by the whole wave: 64 entries per
    // step instead of one (a lane walking a 600-entry run alone held its wave for 230 us)
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) {
        uint64_t todo = __ballot(len[k] > FJ_RUN_LOCK);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)start[k], src), n = (uint32_t)__builtin_amdgcn_readlane((int)len[k], src);
            const uint32_t kx = (uint32_t)__builtin_amdgcn_readlane((int)q[k].x, src), ky = (uint32_t)__builtin_amdgcn_readlane((int)q[k].y, src);
            uint32_t found = (uint32_t)__builtin_amdgcn_readlane((int)c[k], src), f0 = 0, f1 = 0;
            const bool had = found != 0;
            for (uint32_t j0 = FJ_RUN_LOCK; j0 < n; j0 += WAVE) {
                const uint32_t j = j0 + lane;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (j < n) v = ltup[X.ent[s0 + j] & 0xffffu];
                const uint64_t eqm = __ballot(j < n && v.x == kx && v.y == ky);
                if (eqm != 0 && found == 0) {
                    const int fl = __builtin_ctzll(eqm);
                    f0 = (uint32_t)__builtin_amdgcn_readlane((int)v.z, fl);
                    f1 = (uint32_t)__builtin_amdgcn_readlane((int)v.w, fl);
                }
                found += (uint32_t)__popcll(eqm);
            }
            if ((int)lane == src) {
                c[k] = found;
                if (!had && found != 0) { flo[k] = f0; fhi[k] = f1; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < FJ_V; ++k) {
        fp[k] = c[k] != len[k];
        run[k] = start[k] | (c[k] << 16);
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
        part = wave_incl_scan_u64(part);              // the sum over the lanes = the scan's last lane
        excl += ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(part >> 32), 63) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)part, 63);
        if (full) break;
        j -= 64;
    }
    return excl;
}

// Exclusive prefix of a unit's 256-tuple group g over the groups in front of it, for the waves of ONE workgroup that take the
// groups from a counter (so the groups below g are done or in the hands of waves that never wait for g): the chained scan of
// fj_lookback in LDS.  gp[g] = (flag << 30) | value, flag 1 = the group's total, 2 = its inclusive prefix; cleared per unit.
__device__ __forceinline__ uint32_t fj_group_lookback(uint32_t *gp, uint32_t g, uint32_t total, uint32_t lane)
{
    if (lane == 0) __hip_atomic_store(&gp[g], ((g == 0 ? 2u : 1u) << 30) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (g == 0) return 0;
    uint32_t excl = 0;
    int32_t j = (int32_t)g - 1;
    for (;;) {
        const int32_t idx = j - (int32_t)lane;
        uint32_t v = 2u << 30;                        // virtual "prefix 0" in front of group 0
        if (idx >= 0) {
            do {
                v = __hip_atomic_load(&gp[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((v >> 30) == 0) __builtin_amdgcn_s_sleep(1);
            } while ((v >> 30) == 0);
        }
        const uint64_t full = __ballot((v >> 30) == 2u);
        const int stop = full ? __ffsll((unsigned long long)full) - 1 : 64;   // nearest inclusive prefix
        const uint32_t part = wave_incl_scan_u32(lane <= (uint32_t)stop ? (v & 0x3fffffffu) : 0u);
        excl += (uint32_t)__builtin_amdgcn_readlane((int)part, 63);
        if (full) break;
        j -= 64;
    }
    if (lane == 0) __hip_atomic_store(&gp[g], (2u << 30) | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// The pairs of one 256-tuple group (one wave) from registers: c = matches per tuple, B = the rounds they were found in (bit r:
// round r), first = the first match's build row id, prow = the probe tuple's row id; tbl_v: lane j = start of the overflow run
// of round j (lane 0 unused); wbase = the group's first output position.  Called by fj_emit_stream with what phase 1 stashed,
// and by the speculative kernel straight from phase 1 (fj_body, SPEC: the other relation probes).
template <bool DUP, bool N32>
__device__ __forceinline__ void fj_emit_group(bool flip, const uint32_t (&c)[FJ_V], const uint32_t (&B)[FJ_V],
                                              const uint2 (&first)[FJ_V], const uint2 (&prow)[FJ_V], uint32_t tbl_v, uint64_t wbase,
                                              const uint64_t *ovf, uint4 *out, uint64_t cap, uint2 *lr_row)
{
    constexpr int V = FJ_V;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt();
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
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const uint64_t at = wbase + off[k];
        if (c[k] != 0 && at < cap) out[at] = make_pair(flip, prow[k].x, prow[k].y, first[k].x, first[k].y);
        if (DUP && N32 && lr_row && c[k] >= 2u)        // low-radix path: k_lr_emit copies this tuple's pairs from here
            lr_row[k * WAVE + lane] = make_uint2((uint32_t)at, prow[k].x);
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
                bool later = false;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    // round j left an overflow entry of this tuple: a match found there that is not the tuple's first
                    mk[k] = __ballot(((B[k] >> j) & 1u) != 0 && (B[k] & ((1u << j) - 1u)) != 0);
                    tot += (uint32_t)__popcll(mk[k]);
                    later = later || (B[k] >> j) > 1u;
                }
                any = any || tot != 0 || __ballot(later) != 0;
                uint32_t pre = (uint32_t)__builtin_amdgcn_readlane((int)tbl_v, (int)(j & 15u));
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const uint32_t slot = pre + (uint32_t)__popcll(mk[k] & lt);
                    r[jj][k] = make_uint2(0, 0);
                    if ((mk[k] >> lane) & 1ull) r[jj][k] = reinterpret_cast<const uint2 *>(ovf)[min(slot, FJ_OVF_CAP - 1u)];   // (a run beyond the buffer: the unit is redone)
                    pre += (uint32_t)__popcll(mk[k]);
                }
            }
            if (!any) break;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint32_t j = j0 + jj;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const uint32_t before = B[k] & ((1u << j) - 1u);       // the matches of earlier rounds come first
                    const uint64_t at = wbase + off[k] + (uint32_t)__popc(before);
                    if (((B[k] >> j) & 1u) != 0 && before != 0 && at < cap) out[at] = make_pair(flip, prow[k].x, prow[k].y, r[jj][k].x, r[jj][k].y);
                }
            }
            if (j0 + 4 > FJ_OVF_J) break;
        }
    }
}

// Deferred emit pass of a gather-path unit: pure streaming of the probe row ids, the stash and (DUP)
// the overflow stash, 8 tuples per lane; the index is not needed.  DUP = false: every probe tuple has
// zero or one match (the foreign-key case), offsets come from ballots.
template <bool DUP, bool N32>
__device__ __forceinline__ void fj_emit_stream(const FusedArgs &f, uint32_t u, uint64_t base, uint32_t *wsum,
                                               const uint64_t *ovf, uint32_t *table, uint32_t *grab, uint32_t npatch)
{
    const uint32_t *patch = reinterpret_cast<const uint32_t *>(ovf + FJ_OVF_ENT);   // (tuple index << 16 | match rounds) of the unit's irregular tuples
    constexpr int V = FJ_V;                           // a wave's step is one 256-tuple group of phase 1
    const JoinArgs &a = f.j;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const bool flip = bucket_flip(a, b, a.histR[b], a.histS[b]);
    const uint64_t ppos = (flip ? a.psumS[b] : a.psumR[b]) + un.off;
    const uint2 *pr2 = reinterpret_cast<const uint2 *>((flip ? a.partS : a.partR) + ppos);
    const uint8_t *scnt = f.stash_cnt + (flip ? f.nR : 0) + ppos;
    const uint2 *srow = reinterpret_cast<const uint2 *>(f.stash_row + (flip ? f.nR : 0) + ppos);
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const uint64_t cap = a.out_capacity;
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
        uint32_t c[V], B[V];                          // matches; the rounds they were found in (bit r: round r)
        uint2 first[V], prow[V];
        bool irr = false;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const uint32_t i = g * 256u + k * WAVE + lane;
            const bool ok = i < un.count;
            const uint32_t sb = ok ? scnt[i] : 0u;
            c[k] = sb & 0x7fu;
            B[k] = (1u << min(c[k], 31u)) - 1u;       // the rule: every tag hit a match, the (r + 1)-th match found in round r
            if (DUP && (sb & 0x80u) && c[k] >= 2u) { irr = true; B[k] = 0x80000000u | i; }
            first[k] = ok ? srow[i] : make_uint2(0, 0);
            if (N32) { prow[k] = make_uint2(first[k].y, 0u); first[k].y = 0u; }
            else     prow[k] = ok ? pr2[2 * (size_t)i + 1] : make_uint2(0, 0);
        }
        if (DUP && npatch != 0 && __ballot(irr) != 0) {   // rare: a foreign key's tag hit between a tuple's matches — its rounds are on the patch list
#pragma unroll
            for (int k = 0; k < V; ++k) {
                if (B[k] & 0x80000000u) {
                    const uint32_t i = B[k] & 0xffffu;
                    uint32_t rounds = 0;
                    for (uint32_t e = 0; e < npatch; ++e) { const uint32_t w32 = patch[e]; if ((w32 >> 16) == i) rounds = w32 & 0xffffu; }
                    B[k] = rounds;
                }
            }
        }
        // the group's table row: lane 0 its start in the unit's output, lane j the run start of ordinal j
        uint32_t tbl_v = 0;
        if (lane < 16) tbl_v = __hip_atomic_load(&table[g * 16u + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t wbase = base + (uint32_t)__builtin_amdgcn_readlane((int)tbl_v, 0);
        fj_emit_group<DUP, N32>(flip, c, B, first, prow, tbl_v, wbase, ovf, out, cap,
                                f.lr_mode ? const_cast<uint2 *>(srow) + g * 256u : nullptr);
    }
}

// Emit pass of a RESIDENT unit some of whose probe tuples have two or more matches, all of them in clean runs
// (fj_count_res): runs right behind the unit's own phase 1, while its index is still in LDS.  The stash holds, per
// probe tuple, the count byte and either the first match's row id (count <= 1) or the run (start | count << 16).
// Groups of 256 tuples are handed out to the waves as in phase 1; inside a 64-tuple round without a multi-match tuple
// the offsets come from one ballot; otherwise the lanes turn to the round's OUTPUT positions, 64 at a time: position o
// belongs to the tuple whose inclusive prefix is the first above o (binary search over the lanes by shuffles) and is
// match o - (its exclusive prefix) of that tuple = entry start + that of its run — any number of matches per tuple at
// 64 coalesced pairs per store (a lane walking its own 600-match chain wrote one pair per round).
template <bool N32, class IX>
__device__ __forceinline__ void fj_emit_res(const FusedArgs &f, const IX &X, const uint4 *ltup, uint32_t u, uint64_t base,
                                            uint32_t *wsum, uint32_t *table, uint32_t *grab)
{
    constexpr int V = FJ_V;
    const JoinArgs &a = f.j;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const bool flip = bucket_flip(a, b, a.histR[b], a.histS[b]);
    const uint64_t ppos = (flip ? a.psumS[b] : a.psumR[b]) + un.off;
    const uint2 *pr2 = reinterpret_cast<const uint2 *>((flip ? a.partS : a.partR) + ppos);
    const uint8_t *scnt = f.stash_cnt + (flip ? f.nR : 0) + ppos;
    const uint2 *srow = reinterpret_cast<const uint2 *>(f.stash_row + (flip ? f.nR : 0) + ppos);
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const uint64_t cap = a.out_capacity;
    const uint64_t lt = lanemask_lt();
    const uint32_t ngroups = (un.count + 255u) >> 8;
    {                                                 // group totals (column 0 of the table) -> exclusive starts
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
    for (;;) {
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
            if (c[k] >= 2u) c[k] = first[k].x >> 16;  // the exact count sits beside the run's start
        }
        uint32_t gstart = 0;
        if (lane == 0) gstart = __hip_atomic_load(&table[g * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint64_t wat = base + (uint32_t)__builtin_amdgcn_readfirstlane((int)gstart);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (__ballot(c[k] > 1u) == 0) {           // zero or one match per tuple
                const uint64_t mm = __ballot(c[k] != 0);
                const uint64_t dst = wat + (uint32_t)__popcll(mm & lt);
                if (c[k] != 0 && dst < cap) out[dst] = make_pair(flip, prow[k].x, prow[k].y, first[k].x, first[k].y);
                wat += (uint32_t)__popcll(mm);
            } else {
                uint32_t t;
                const uint32_t excl = wave_excl_scan_u32(c[k], &t);
                const uint32_t incl = excl + c[k];
                for (uint32_t o0 = 0; o0 < t; o0 += WAVE) {
                    const uint32_t o = o0 + lane;
                    uint32_t lo_ = 0;
#pragma unroll
                    for (int stp = 32; stp >= 1; stp >>= 1) {
                        const uint32_t v = __shfl(incl, (int)(lo_ + stp - 1u), 64);
                        if (v <= o) lo_ += stp;
                    }
                    const int src = (int)min(lo_, 63u);
                    const uint32_t ci = __shfl(c[k], src, 64), ei = __shfl(excl, src, 64);
                    const uint32_t fx = __shfl(first[k].x, src, 64), fy = __shfl(first[k].y, src, 64);
                    const uint32_t px = __shfl(prow[k].x, src, 64), py = __shfl(prow[k].y, src, 64);
                    if (o < t) {
                        uint32_t bl = fx, bh = fy;
                        if (ci >= 2u) {
                            const uint4 v = ltup[X.ent[(fx & 0xffffu) + (o - ei)] & 0xffffu];
                            bl = v.z; bh = v.w;
                        }
                        const uint64_t dst = wat + o;
                        if (dst < cap) out[dst] = make_pair(flip, px, py, bl, bh);
                    }
                }
                wat += t;
            }
        }
    }
}

// SPEC, the other relation probes a gathered build side (fj_body): one 256-tuple group from its keys to its pairs, by one wave.
// Round 0 takes every tuple's first candidate, four gathers a lane as in fj_count_batch.  What is left — the second and later
// candidates of the tuples that have them: a quarter of the tuples of a foreign-key join's unique side — is walked LANE BY LANE:
// every step each lane takes the next candidate of the first of its four tuples that has one, so a step is ONE gather with most
// lanes busy (a round per match ordinal was four gathers with a quarter, a twelfth, a fiftieth of the lanes: the vector memory
// instructions a group issues, not their lanes, set this phase's pace — 70 a group with the overflow runs and the four-ordinal
// emit loop, profiles/README.md r04b).  A second or later match leaves an 8-byte record {build row id, tuple | ordinal}
// in the wave's own piece of the overflow buffer (always the same 4 KB: L1/L2-resident; the probe row id stays in the lane that
// holds the tuple and is shuffled to the record when it is read back: 16-byte records with it cost 2.5 % of the kernel); the matches of a tuple are
// found in slot order = descending build position (rhjoin.c:219-250), so the ordinal is the number found before.  Then the
// group's total goes into the chained scan of the unit's groups (fj_group_lookback), the first matches leave from the
// registers and the records are read back 64 at a time, each pair to its tuple's offset + ordinal.  Returns the lane's matches.
// The candidate walk of one group: c = matches per tuple, first = the first match's build row id, the records of the others in
// rec[0 .. returned count) (at most FJ_REC_CAP are stored; `cannot`: more than that, or an ordinal beyond eight bits).
template <bool N32, class IX>
__device__ __forceinline__ uint32_t fj_walk_group(const IX &X, const FjGather<N32> &G, const uint4 (&q)[FJ_V], const bool (&okk)[FJ_V],
                                                  uint4 *rec, uint32_t (&c)[FJ_V], uint32_t (&first)[FJ_V], bool &cannot)
{
    constexpr int V = FJ_V;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt();
    uint32_t sn[V], tm[V];
    // (The lane-by-lane loop below picks one of the lane's four tuples by a run-time index.  hipcc turns a select between
    // loads of the caller's array into a load through a selected POINTER — and the array into 64 bytes of scratch per lane;
    // values that come out of an (empty) asm statement are not loads any more.)
    uint32_t qx[V], qy[V], qz[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        qx[k] = q[k].x; qy[k] = q[k].y; qz[k] = q[k].z;
        asm volatile("" : "+v"(qx[k]), "+v"(qy[k]), "+v"(qz[k]));
    }
    {
        uint32_t pos[V];
        uint4 g[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            fj_lookup(X, ((uint64_t)q[k].y << 32) | q[k].x, okk[k], sn[k], tm[k]);
            pos[k] = 0xffffffffu;
            const uint32_t m = tm[k] & 0xffu;
            if (m != 0) { pos[k] = X.ent[(sn[k] & 0xffffu) + (uint32_t)__builtin_ctz(m)] & 0xffffu; tm[k] &= tm[k] - 1u; }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = pos[k] != 0xffffffffu ? G.load(pos[k]) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < V; ++k) {
#ifdef FJ_ABL_G8
            const bool eq = pos[k] != 0xffffffffu;
#else
            const bool eq = pos[k] != 0xffffffffu && g[k].x == q[k].x && g[k].y == q[k].y;
#endif
            first[k] = eq ? g[k].z : 0u;
            c[k] = eq ? 1u : 0u;
        }
    }
    uint32_t ne = 0;                                   // records of this group so far (wave-uniform)
    // FJ_SER candidates a lane and step: which candidates a lane walks is a matter of the index alone (LDS), so their gathers are
    // issued together and verified in order afterwards — the chain of a group is steps x one gather latency
#ifndef FJ_SER
#define FJ_SER 1                                   // (2 and 3 measured the same on C3: it is not this chain that sets the pace)
#endif
    for (;;) {
        uint32_t ks[FJ_SER], kx[FJ_SER], ky[FJ_SER], pr[FJ_SER], p[FJ_SER];
#pragma unroll
        for (int j = 0; j < FJ_SER; ++j) {
            ks[j] = V;                                 // the first of the lane's tuples with a candidate left
#pragma unroll
            for (int k = V - 1; k >= 0; --k)
                if ((tm[k] & 0xffu) != 0 || (sn[k] >> 16) > (uint32_t)FJ_WIN) ks[j] = (uint32_t)k;
            uint32_t s_ = sn[0], t_ = tm[0];
            kx[j] = qx[0]; ky[j] = qy[0]; pr[j] = qz[0];
#pragma unroll
            for (int k = 1; k < V; ++k)
                if (ks[j] == (uint32_t)k) { s_ = sn[k]; t_ = tm[k]; kx[j] = qx[k]; ky[j] = qy[k]; pr[j] = qz[k]; }
            p[j] = 0xffffffffu;
            if (ks[j] < (uint32_t)V) {
                if ((t_ & 0xffu) == 0) {               // the window is used up and the slot goes on
                    s_ += (uint32_t)FJ_WIN - ((uint32_t)FJ_WIN << 16);
                    t_ |= fj_window(X, s_ & 0xffffu, s_ >> 16, t_ & 0xffff0000u);
                }
                const uint32_t m = t_ & 0xffu;
                if (m != 0) { p[j] = X.ent[(s_ & 0xffffu) + (uint32_t)__builtin_ctz(m)] & 0xffffu; t_ &= t_ - 1u; }
            }
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (ks[j] == (uint32_t)k) { sn[k] = s_; tm[k] = t_; }
        }
        if (__ballot(ks[0] < (uint32_t)V) == 0) break;
#ifdef FJ_ABL_NOSER          // timing experiment only (wrong results): second and later candidates are not fetched
        continue;
#endif
        uint4 g[FJ_SER];
#pragma unroll
        for (int j = 0; j < FJ_SER; ++j) g[j] = p[j] != 0xffffffffu ? G.load(p[j]) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < FJ_SER; ++j) {
            uint32_t cc = c[0];
#pragma unroll
            for (int k = 1; k < V; ++k) if (ks[j] == (uint32_t)k) cc = c[k];
#ifdef FJ_ABL_G8
            const bool eq = p[j] != 0xffffffffu && (g[j].x | kx[j] | ky[j] | 1u) != 0u;     // (timing only: every tag hit counts)
#else
            const bool eq = p[j] != 0xffffffffu && g[j].x == kx[j] && g[j].y == ky[j];
#endif
            const bool isrec = eq && cc != 0;
            const uint64_t mk = __ballot(isrec);
            const uint32_t slot = ne + (uint32_t)__popcll(mk & lt);
#ifndef FJ_ABL_NOREC      // (timing experiment only, wrong results: second and later matches are found but leave no record and no pair)
#ifndef FJ_REC16
            if (isrec && slot < FJ_REC_CAP) reinterpret_cast<uint2 *>(rec)[slot] = make_uint2(g[j].z, (ks[j] * WAVE + lane) | (cc << 8));
#else
            if (isrec && slot < FJ_REC_CAP) rec[slot] = make_uint4(g[j].z, pr[j], (ks[j] * WAVE + lane) | (cc << 8), 0u);
#endif
#endif
            ne += (uint32_t)__popcll(mk);
            cannot = cannot || (eq && cc >= 65535u);   // (the ordinal has sixteen bits)
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (ks[j] == (uint32_t)k && eq) { if (cc == 0) first[k] = g[j].z; c[k] = cc + 1u; }
        }
    }
    cannot = cannot || ne > FJ_REC_CAP;
#ifdef FJ_ABL_NOREC
    return 0;
#endif
    return min(ne, FJ_REC_CAP);
}

// the records of a group, read back 64 at a time: each pair to its tuple's offset (off[k] of the lane that holds the tuple) + ordinal
__device__ __forceinline__ void fj_emit_records(bool flip, const uint4 *rec, uint32_t nrec, const uint32_t (&off)[FJ_V], uint64_t wbase,
                                                uint4 *out, uint64_t cap, const uint4 (&q)[FJ_V])
{
    constexpr int V = FJ_V;
    const uint32_t lane = threadIdx.x & 63;
    (void)q;
    for (uint32_t e0 = 0; e0 < nrec; e0 += WAVE) {
        const uint32_t e = e0 + lane;
#ifndef FJ_REC16          // 8-byte records {build row id, tuple | ordinal}: the probe row id comes from the lane that holds the tuple (16-byte
                         // records with the probe row id in them, -DFJ_REC16: C3 probe stage 1.837 -> 1.883 ms)
        const uint2 r8 = e < nrec ? reinterpret_cast<const uint2 *>(rec)[e] : make_uint2(0, 0);
        uint4 r = make_uint4(r8.x, 0u, r8.y, 0u);
        {
            const int s8 = (int)(r.z & 63u);
            const uint32_t k8 = (r.z >> 6) & 3u;
            r.y = __shfl(q[0].z, s8, 64);
#pragma unroll
            for (int k = 1; k < V; ++k) { const uint32_t pk = __shfl(q[k].z, s8, 64); if (k8 == (uint32_t)k) r.y = pk; }
        }
#else
        const uint4 r = e < nrec ? rec[e] : make_uint4(0, 0, 0, 0);
#endif
        const int src = (int)(r.z & 63u);
        const uint32_t kk = (r.z >> 6) & 3u;
        uint32_t o = __shfl(off[0], src, 64);
#pragma unroll
        for (int k = 1; k < V; ++k) { const uint32_t ok_ = __shfl(off[k], src, 64); if (kk == (uint32_t)k) o = ok_; }
        const uint64_t at = wbase + o + (r.z >> 8);
        if (e < nrec && at < cap) FJ_STORE_PAIR(&out[at], make_pair(flip, r.y, 0u, r.x, 0u));
    }
}

// SPEC, the other relation probes: the group's pairs leave at once (see above).  Returns the lane's matches.
template <bool N32, class IX>
__device__ __forceinline__ uint32_t fj_group_direct(const IX &X, const FjGather<N32> &G, const uint4 (&q)[FJ_V], const bool (&okk)[FJ_V],
                                                    bool flip, uint32_t grp, uint32_t *gpre, uint64_t spec_base, uint4 *rec,
                                                    uint4 *out, uint64_t cap, bool &cannot)
{
    constexpr int V = FJ_V;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t c[V], first[V];
    const uint32_t nrec = fj_walk_group<N32>(X, G, q, okk, rec, c, first, cannot);
    uint32_t off[V], wrun = 0, cs = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        uint32_t tot;
        off[k] = wrun + wave_excl_scan_u32(c[k], &tot);
        wrun += tot;
        cs += c[k];
    }
#ifdef FJ_ABL_NOLB           // timing experiment only (wrong results): no chained scan, every group writes at its own 256 slots
    const uint64_t wbase = spec_base + (uint64_t)grp * 256u;
#else
    const uint64_t wbase = spec_base + fj_group_lookback(gpre, grp, wrun, lane);
#endif
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const uint64_t at = wbase + off[k];
        if (c[k] != 0 && at < cap) FJ_STORE_PAIR(&out[at], make_pair(flip, q[k].z, 0u, first[k], 0u));
    }
    fj_emit_records(flip, rec, nrec, off, wbase, out, cap, q);
    return cs;
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
// The whole body is a device function: k_join_fused runs it over units a host-launched plan wrote, k_small_join
// (rhj_small.hip.h) as the last phase of its single launch.
// SPEC (k_join_spec): the FOREIGN-KEY SPECULATION.  Hypothesis: every tuple of one relation (f.spec: 1 = S, 2 = R) has exactly
// one match.  Then the join's pairs of bucket b start at that relation's psum[b] and number its hist[b], whichever side
// probes: nothing is chained.  A unit whose probe side IS that relation writes its pairs straight from phase 1 — pair i of
// the unit at base + i, no stash, no emit pass — and checks that every tuple had exactly one match; a unit whose probe side
// is the other relation runs as usual, but emits at once at the predicted base and checks its total (such a bucket must be
// one unit).  Any check that fails raises ticket[4], the workgroups stop, and the ordinary kernel, always enqueued behind
// this one, does the join (it returns at once when the speculation held).  The last workgroup out also checks that the
// predicted totals add up to the relation's size (a bucket without partners has no unit to notice it).
template <bool MAYRES, bool N32, bool SPEC = false>
__device__ __forceinline__ void fj_body(const FusedArgs &f, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];
    __shared__ uint32_t sh_u;
    __shared__ uint32_t sh_ovf;
    __shared__ uint32_t sh_patch;
    __shared__ uint32_t sh_grab;
    __shared__ uint32_t sh_pick;
    __shared__ uint64_t sh_base;
    __shared__ uint32_t wsum[FJ_WAVES];
    __shared__ uint32_t gpre[SPEC ? FJ_GROUPS : 1];     // SPEC, the other relation probes: the groups' totals / prefixes (fj_group_lookback)
    const JoinArgs &a = f.j;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long *st = (unsigned long long *)f.status;
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const bool emitting = out != nullptr && FJ_ABLATE != 3;
    if ((a.summary->wide_row_ids == 0) != N32) return;                 // the other instantiation's launch does the join
    uint4 *stg = reinterpret_cast<uint4 *>(f.ovf + (size_t)gridDim.x * 2 * FJ_OVF_CAP) + ((size_t)blockIdx.x * FJ_WAVES + w) * FJ_REC_CAP;   // this wave's piece
    uint32_t pend = 0xffffffffu;                      // unit whose emit pass is deferred
    uint64_t pend_total = 0;
    bool pend_dup = false;
    const uint64_t *pend_ovf = nullptr;
    uint32_t *pend_table = nullptr;
    uint32_t pend_npatch = 0;

    for (uint32_t iter = 0;; ++iter) {
    __syncthreads();
    if (threadIdx.x == 0) {
        sh_u = atomicAdd(f.ticket, 1u); sh_ovf = 0; sh_grab = 0; sh_patch = 0;
        // (SPEC: a failed check anywhere ends the kernel — one thread reads the word, so the whole workgroup agrees)
        if (SPEC && __hip_atomic_load(f.ticket + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) sh_u = 0xffffffffu;
    }
    if (SPEC && threadIdx.x < FJ_GROUPS) gpre[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t u = sh_u;
    if (u >= a.summary->units || !a.summary->fused_ok) break;         // grid is an upper bound; tiled path takes over
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const uint64_t cR = a.histR[b], cS = a.histS[b];
    const bool flip = bucket_flip(a, b, cR, cS);                       // S is streamed (r_s == 1)
    const uint64_t ppos = (flip ? a.psumS[b] : a.psumR[b]) + un.off;   // position in the probe relation
    // SPEC: the probe side is the relation of the hypothesis (fkp); the unit's predicted first pair and pair count
#ifdef FJ_FLOOR              // timing experiment only: EVERY unit writes its pairs from the probe loop (no stash, no emit pass, no check)
    const bool fkp = SPEC;
#else
    const bool fkp = SPEC && (flip == (f.spec == 1u));
#endif
    const uint64_t spec_base = SPEC ? (f.spec == 1u ? a.psumS[b] : a.psumR[b]) + (fkp ? un.off : 0u) : 0u;
    const uint64_t spec_total = SPEC ? (fkp ? (uint64_t)un.count : (f.spec == 1u ? cS : cR)) : 0u;
    bool spec_bad = SPEC && !fkp && (un.off != 0 || un.count != (flip ? cS : cR));   // (a split bucket's later units cannot know their base)
    if (SPEC && threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(f.ticket + 6), (unsigned long long)spec_total);
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
    FjIndexT<MAYRES> X;
    X.ent = tbl + (RES ? 4u * bcp : 0u);
    X.hs = hs0;
    X.bits = f.radix_bits;
    X.dirw = X.ent + bcp + 8u;
    uint8_t *scnt = f.stash_cnt + (flip ? f.nR : 0) + ppos;
    uint2 *srow = reinterpret_cast<uint2 *>(f.stash_row + (flip ? f.nR : 0) + ppos);
    FjGather<N32> G;
    G.init(bdp, bpos, bc);
#ifdef FJ_PROBE_AUX
    FjGather<N32> PS;                                 // (A/B: the unit's probe tuples through a descriptor, policy FJ_PROBE_AUX)
    PS.init(prp, ppos, un.count);
#endif
    FjOvf O;                                          // double-buffered: the previous unit's emit may still be pending
    O.buf = f.ovf + ((size_t)blockIdx.x * 2 + (iter & 1u)) * FJ_OVF_CAP;
    O.table = f.ovf_base + ((size_t)blockIdx.x * 2 + (iter & 1u)) * (FJ_GROUPS * 16u);
    O.counter = &sh_ovf;
    O.gid = 0;

    if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 0] = __builtin_amdgcn_s_memrealtime();
    // ---- build
    if (RES) fj_build<true, N32>(X, bdp, bpos, bc, ltup, reinterpret_cast<uint32_t *>(O.buf), wsum, &sh_pick);
    else     fj_build<false, N32>(X, bdp, bpos, bc, ltup, reinterpret_cast<uint32_t *>(O.buf), wsum, &sh_pick, SPEC && !MAYRES && fkp);
    if (FJ_ABLATE == 1) continue;                      // timing experiment: build only
    if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 1] = __builtin_amdgcn_s_memrealtime();

    // ---- phase 1: count (+ stash of the first match when the build tuples are not resident)
    uint32_t mine = 0;
    bool needs_index = false;                         // the emit pass must walk the index again
    bool has_dup = false;                             // resident unit: some tuple has two or more matches (clean runs: fj_emit_res)
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
#ifdef FJ_PROBE_AUX
            q[k] = okk[k] ? PS.template stream<FJ_PROBE_AUX>(i) : make_uint4(0, 0, 0, 0);
#else
            q[k] = okk[k] ? (MAYRES ? pt_load<N32>(prp, ppos + i) : pt_load_nt<N32>(prp, ppos + i)) : make_uint4(0, 0, 0, 0);   // (nt: the gather kernels only)
#endif
        }
        O.gid = grp;
        uint32_t run[FJ_V], bm[FJ_V];
        if (SPEC && !MAYRES && !fkp) {                 // (MAYRES kernels — C4's — keep the stash for such units: this code's live values pushed their resident loops into scratch)
            // SPEC, the OTHER relation probes a gathered build side: the unit's first pair is known (the hypothesis' relation's
            // psum), so all a group needs is the match total of the groups in front of it — a chained scan among the waves of
            // this workgroup, in LDS — and its pairs go out from here (fj_group_direct): no stash, no emit pass.
            mine += fj_group_direct<N32>(X, G, q, okk, flip, grp, gpre, spec_base, stg, out, a.out_capacity, needs_index);
            continue;
        }
        if (RES) fj_count_res(X, ltup, q, okk, c, flo, fhi, fp, run);
        else     fj_count_batch<false, true, N32>(X, G, ltup, q, okk, c, flo, fhi, fp, bm, O);
        // (The failure word is looked at once a unit, not here: an agent-scope load a group made every wave wait for its
        // outstanding pair stores and gathers — the kernel was 18 % slower than without the speculation.  A failing
        // speculation costs 0.15 ms that way instead of 0.07, and after one that failed only every 16th join tries.)
        if (SPEC && fkp) {                             // one match each, or the speculation is off: the pairs go out right here
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = t0 + k * WAVE + lane;
                if (i < un.count) {
#if !defined(FJ_FLOOR) && !defined(FJ_ABL_G8)
                    spec_bad = spec_bad || c[k] != 1u;
#endif
                    const uint64_t at = spec_base + i;
                    if (at < a.out_capacity) FJ_STORE_PAIR(&out[at], make_pair(flip, q[k].z, N32 ? 0u : q[k].w, flo[k], fhi[k]));
                }
            }
            if (__ballot(spec_bad) != 0) {            // noticed at once, and by everybody
                if (lane == 0) __hip_atomic_store(f.ticket + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spec_bad = true;
                break;
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const uint32_t i = t0 + k * WAVE + lane;
            if (i < un.count) {
                // count byte: 0..126 exact, 127 = saturated (recounted in phase 2); bit 7 = some tag hit of
                // this tuple was a different key, so phase 2 must verify its candidates again
                scnt[i] = (uint8_t)(min(c[k], 127u) | (fp[k] ? 0x80u : 0u));
                // resident units: a tuple with two or more matches keeps its run (start | count << 16) instead of the first row id
                if (RES && c[k] >= 2u) fj_stash_put<N32>(srow, i, run[k], 0u, q[k].z);
                else                   fj_stash_put<N32>(srow, i, flo[k], fhi[k], q[k].z);
            }
            mine += c[k];
            if (RES) { needs_index = needs_index || (fp[k] && c[k] >= 2u); has_dup = has_dup || c[k] >= 2u; }
            else if (c[k] > FJ_OVF_J + 1u || (fp[k] && c[k] >= 2u && (MAYRES || bm[k] >= 65536u))) needs_index = true;   // matches beyond round 15: no
            else if (!MAYRES && fp[k] && c[k] >= 2u) {                                         // overflow slot (the patch list: gather kernels only)
                // a foreign key's tag hit between this tuple's matches: its overflow entries sit in the runs of the ROUNDS
                // they were found in, not of their ordinals — the emit pass needs the rounds (patch list in the buffer's tail)
                const uint32_t at = atomicAdd(&sh_patch, 1u);
                if (at < FJ_PATCH_CAP) reinterpret_cast<uint32_t *>(O.buf + FJ_OVF_ENT)[at] = (i << 16) | bm[k];
            }
        }
        {                                             // group total for the barrier-free emit pass
            uint32_t gt;
            wave_excl_scan_u32(c[0] + c[1] + c[2] + c[3], &gt);
            if (lane == 0) O.table[O.gid * 16u] = gt;
        }
    }

    if (SPEC) {                                       // a check failed (here or in another workgroup): stop
        if (__syncthreads_or(spec_bad)) {
            if (threadIdx.x == 0) __hip_atomic_store(f.ticket + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (FJ_DBG && threadIdx.x == 0) { FJ_DBG[(size_t)u * 8 + 3] = __builtin_amdgcn_s_memrealtime(); FJ_DBG[(size_t)u * 8 + 7] = fkp ? 1 : 2; }
        if (fkp) continue;                            // the pairs are out, one match each
    }
    // ---- unit total -> chained scan
    if (FJ_DBG && lane == 0) { if (w == 0) FJ_DBG[(size_t)u * 8 + 2] = __builtin_amdgcn_s_memrealtime(); }
    {
        uint32_t tot;
        wave_excl_scan_u32(mine, &tot);
        if (lane == 0) wsum[w] = tot;
    }
    bool unit_needs_index = __syncthreads_or(needs_index) != 0;   // also publishes wsum
    const uint32_t ovf_total = sh_ovf, npatch = sh_patch;
    unit_needs_index = unit_needs_index || ovf_total > FJ_OVF_ENT || npatch > FJ_PATCH_CAP;
    const bool unit_res_dup = RES && !unit_needs_index && __syncthreads_or(has_dup) != 0;
    uint64_t total = 0;
#pragma unroll
    for (int i = 0; i < FJ_WAVES; ++i) total += wsum[i];
    if (SPEC) {                                       // the other relation probes: the usual unit at the predicted base, emitted at once
#if defined(FJ_ABL_NOSER) || defined(FJ_ABL_G8)
        if (false) {
#else
        if (total != spec_total) {
#endif   // (workgroup-uniform)
            if (threadIdx.x == 0) __hip_atomic_store(f.ticket + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (threadIdx.x == 0) { a.unit_count[u] = total; sh_base = spec_base; }
        __syncthreads();
        if (!MAYRES) {
            // a gathered unit's pairs are out — unless a group had more second-and-later matches than its wave's piece holds or a
            // tuple more matches than an ordinal counts (hot keys among the build side's duplicates): the counts, and with them the
            // total just checked, are exact all the same, so the speculation stands and k_join_walk writes this unit's pairs again,
            // all of them, at the predicted base — from the index alone (flag 8: nothing was stashed)
            if (unit_needs_index && emitting && threadIdx.x == 0) {
                const uint32_t at = atomicAdd(f.ticket + 2, 1u);
                f.walk[at] = FjWalkItem{u, 8u, sh_base};
            }
            continue;
        }
        if (!emitting) continue;
        if (!unit_needs_index && !unit_res_dup) {
            if (ovf_total != 0) fj_emit_stream<true, N32>(f, u, sh_base, wsum, O.buf, O.table, &sh_grab, MAYRES ? 0u : npatch);
            else                fj_emit_stream<false, N32>(f, u, sh_base, wsum, O.buf, O.table, &sh_grab, 0u);
            continue;
        }
    } else if (threadIdx.x == 0) {
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
        if (pend_dup) fj_emit_stream<true, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab, MAYRES ? 0u : pend_npatch);
        else          fj_emit_stream<false, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab, 0u);
        if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)pend * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        pend = 0xffffffffu;
        __syncthreads();
        if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 7] = __builtin_amdgcn_s_memrealtime();
    }
    if (!SPEC && emitting && !unit_needs_index && !unit_res_dup) {   // this unit's emit pass needs no index: defer it
        pend = u;
        pend_total = total;
        pend_dup = ovf_total != 0;
        pend_ovf = O.buf;
        pend_table = O.table;
        pend_npatch = npatch;
        if (FJ_DBG && threadIdx.x == 0) { FJ_DBG[(size_t)u * 8 + 3] = FJ_DBG[(size_t)u * 8 + 4] = __builtin_amdgcn_s_memrealtime(); }
        continue;
    }

    if (!SPEC) {
        if (w == 0) {
            const uint64_t excl = u == 0 ? 0 : fj_lookback(st, u, lane);
            if (lane == 0) {
                if (u != 0) __hip_atomic_store(&st[u], (2ull << 62) | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh_base = excl;
            }
        }
        __syncthreads();
    }

    if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    if (!emitting) continue;
    if (unit_res_dup) {                               // resident build side, multi-match tuples in clean runs: output-centric emit
        fj_emit_res<N32>(f, X, ltup, u, sh_base, wsum, O.table, &sh_grab);
        if (FJ_DBG && threadIdx.x == 0) FJ_DBG[(size_t)u * 8 + 4] = __builtin_amdgcn_s_memrealtime();
        continue;
    }
    // ---- this unit's pairs need the index walked again: k_join_walk, enqueued behind this kernel, writes them
    if (threadIdx.x == 0) {
        const uint32_t at = atomicAdd(f.ticket + 2, 1u);
        f.walk[at] = FjWalkItem{u, (RES ? 1u : 0u) | (MAYRES ? 2u : 0u), sh_base};
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
        if (pend_dup) fj_emit_stream<true, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab, MAYRES ? 0u : pend_npatch);
        else          fj_emit_stream<false, N32>(f, pend, sh_base, wsum, pend_ovf, pend_table, &sh_grab, 0u);
    }
}

// The speculative kernel (fj_body<., ., SPEC>), 12-byte tuples only; always followed by k_join_fused with the same arguments.
template <bool MAYRES>
__global__ __launch_bounds__(FJ_BLOCK) void k_join_spec(FusedArgs f, uint32_t lds_bytes)
{
    fj_body<MAYRES, true, true>(f, lds_bytes);
    __syncthreads();
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(f.ticket + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {   // the last workgroup out
        __hip_atomic_store(f.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long predicted =
            __hip_atomic_load(reinterpret_cast<unsigned long long *>(f.ticket + 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the relation's tuples that take part (all of them, unless the join is one rank's share of a sharded join)
        const uint32_t last = (1u << f.radix_bits) - 1u;
        const uint64_t n_fk = f.spec == 1u ? f.j.psumS[last] + f.j.histS[last] : f.j.psumR[last] + f.j.histR[last];
        bool failed = __hip_atomic_load(f.ticket + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || predicted != n_fk;
#if defined(FJ_FLOOR) || defined(FJ_ABL_G8)
        failed = false;
#endif
        if (failed) {                                 // the ordinary kernel behind this one starts over
            __hip_atomic_store(f.ticket + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f.ticket + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
            const_cast<PlanSummary *>(f.j.summary)->matches = n_fk;
    }
}

template <bool MAYRES, bool N32>
__global__ __launch_bounds__(FJ_BLOCK) void k_join_fused(FusedArgs f, uint32_t lds_bytes)
{
    // behind k_join_spec: nothing to do when its speculation held (every workgroup reads the same word: all leave or none)
    if (f.spec && __hip_atomic_load(f.ticket + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    fj_body<MAYRES, N32>(f, lds_bytes);
    // The last workgroup out leaves the match total = inclusive prefix of the last unit (every unit publishes one
    // before it emits; nothing when the plan rejected the fused path: its unit list is the tiled one then).  Both
    // row-id instantiations of a join are launched and either may run first: the one that returns at once finds the
    // status words still clear or already complete, and the later launch's total is the one that stays.
    __syncthreads();
    if (threadIdx.x == 0) {
        // Release + acquire at agent scope on the counter: this workgroup's status words are out before it counts itself, and
        // the last one to arrive sees everybody's (the workgroups run on eight XCDs with an L2 each).
        if (__hip_atomic_fetch_add(f.ticket + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(f.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PlanSummary *sum = const_cast<PlanSummary *>(f.j.summary);
            const uint64_t n = sum->units;
            uint64_t total = 0;
            if (sum->fused_ok && n && n <= f.unit_bound) {
                const unsigned long long v = __hip_atomic_load((unsigned long long *)f.status + (n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // not an inclusive prefix: this launch did not run the units (the other row-id width's launch does and
                // leaves its total behind this one) — or the chain is broken, and then the host must not take 0 for an answer
                total = (v >> 62) == 2 ? (v & ((1ull << 62) - 1)) : FJ_NO_TOTAL;
            }
            sum->matches = total;
            if (f.host_summary) {
                const uint64_t *src = reinterpret_cast<const uint64_t *>(sum);
                static_assert(sizeof(PlanSummary) % 8 == 0, "summary words");
                for (uint32_t i = 0; i < sizeof(PlanSummary) / 8; ++i) {
                    const uint64_t v = i == offsetof(PlanSummary, matches) / 8 ? total : src[i];
                    __hip_atomic_store(f.host_summary + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                // behind the summary: the units left to k_join_walk (every workgroup is through: the list is complete) — the small
                // path's host launches that kernel only when there is something to walk
                const uint32_t nwalk = __hip_atomic_load(f.ticket + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(f.host_summary + sizeof(PlanSummary) / 8, (uint64_t)nwalk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---- k_join_walk: the pairs of the units k_join_fused left on the walk list.  ALWAYS enqueued behind k_join_fused and
// returns at once when the list is empty (no host round trip decides the launch).  With this code inside the fused kernel
// — one 1024-thread function at the 128-VGPR cap — the foreign-key loops paid for its live values: 20 spilled VGPRs and
// scratch on every launch (timing with the walk compiled out, profiles/README.md r02z: C3 -3.3 %, C2 -13 % of the kernel).
// A listed unit's index is built again (same geometry, same residency decision as in k_join_fused), then:
//   the first match comes from the stash (phase 1 kept its row id); further matches of a tuple are fetched in lockstep
//   rounds over its slot; when no tag hit of the tuple was a foreign key (stash bit 7 clear) its first candidate IS that
//   first match and is skipped unfetched.  Order: probe tuples in unit order, matches in slot order = descending build
//   position (rhjoin.c:141-250).
template <bool RES, bool N32, bool H32>
__device__ __forceinline__ void fj_walk_unit(const FusedArgs &f, uint32_t lds_bytes, uint32_t *tbl, uint32_t u, uint64_t base,
                                             uint32_t *wsum, uint32_t *sh_pick, bool nostash = false)
{
    const JoinArgs &a = f.j;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const uint64_t cap = a.out_capacity;
    const Unit un = a.units[u];
    const uint32_t b = un.bucket;
    const uint64_t cR = a.histR[b], cS = a.histS[b];
    const bool flip = bucket_flip(a, b, cR, cS);
    const uint64_t ppos = (flip ? a.psumS[b] : a.psumR[b]) + un.off;
    const rhj_tuple *prp = flip ? a.partS : a.partR;
    const rhj_tuple *bdp = flip ? a.partR : a.partS;
    const uint64_t bpos = flip ? a.psumR[b] : a.psumS[b];
    const uint32_t bc = (uint32_t)(flip ? cR : cS);
    const uint32_t bcp = (bc + 3u) & ~3u;
    uint32_t hs0 = bc < 64u ? 64u : bc;               // (the geometry of fj_body)
    {
        const uint32_t room = (lds_bytes - 64u - 4u * bcp) / 2u - 2u;
        if (hs0 > room) hs0 = room & ~1u;
    }
    uint4 *ltup = reinterpret_cast<uint4 *>(tbl);
    FjIndexT<H32> X;                                  // the hash of the kernel that wrote this unit's stash
    X.ent = tbl + (RES ? 4u * bcp : 0u);
    X.hs = hs0;
    X.bits = f.radix_bits;
    X.dirw = X.ent + bcp + 8u;
    const uint8_t *scnt = f.stash_cnt + (flip ? f.nR : 0) + ppos;
    const uint2 *srow = reinterpret_cast<const uint2 *>(f.stash_row + (flip ? f.nR : 0) + ppos);
    FjGather<N32> G;
    G.init(bdp, bpos, bc);
#ifdef FJ_PROBE_AUX
    FjGather<N32> PS;                                 // (A/B: the unit's probe tuples through a descriptor, policy FJ_PROBE_AUX)
    PS.init(prp, ppos, un.count);
#endif
    __syncthreads();                                  // the previous unit's index is no longer read
    fj_build<RES, N32>(X, bdp, bpos, bc, ltup, reinterpret_cast<uint32_t *>(f.ovf + (size_t)blockIdx.x * 2 * FJ_OVF_CAP), wsum, sh_pick);
    __syncthreads();
    uint64_t run = base;
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
                // nostash: nothing was parked for this unit — every tuple counts as "saturated, some tag hit foreign": recounted and walked in full
                const uint32_t sb = !okk[h][k] ? 0u : nostash ? 0xffu : scnt[i];
                c[h][k] = sb & 0x7fu;
                fpt[h][k] = RES || (sb & 0x80u) != 0;      // (a resident unit's stash keeps runs, not first matches, for multi-match tuples)
                const uint2 fr = (okk[h][k] && !nostash) ? srow[i] : make_uint2(0, 0);
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
                        const auto hh = X.hash(((uint64_t)q[h][k].y << 32) | q[h][k].x);
                        const uint32_t t = X.tag(hh), sl = X.slot(hh);
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
}

__global__ __launch_bounds__(FJ_BLOCK) void k_join_walk(FusedArgs f, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];
    __shared__ uint32_t sh_pick;
    __shared__ uint32_t wsum[FJ_WAVES];
    const uint32_t n = f.ticket[2];                   // written by k_join_fused, which is through
    if (n == 0 || !f.j.summary->fused_ok || f.j.out == nullptr) return;
    const bool narrow = f.j.summary->wide_row_ids == 0;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const FjWalkItem it = f.walk[i];
        const bool res = (it.flags & 1u) != 0;
        if (it.flags & 2u) {                          // (FjHashT<true> comes with the kernels that may keep build tuples resident)
            if (narrow) { if (res) fj_walk_unit<true, true, true>(f, lds_bytes, tbl, it.unit, it.base, wsum, &sh_pick);
                          else     fj_walk_unit<false, true, true>(f, lds_bytes, tbl, it.unit, it.base, wsum, &sh_pick); }
            else        { if (res) fj_walk_unit<true, false, true>(f, lds_bytes, tbl, it.unit, it.base, wsum, &sh_pick);
                          else     fj_walk_unit<false, false, true>(f, lds_bytes, tbl, it.unit, it.base, wsum, &sh_pick); }
        } else {
            if (narrow) fj_walk_unit<false, true, false>(f, lds_bytes, tbl, it.unit, it.base, wsum, &sh_pick, (it.flags & 8u) != 0);
            else        fj_walk_unit<false, false, false>(f, lds_bytes, tbl, it.unit, it.base, wsum, &sh_pick);
        }
    }
}

}  // namespace rhj
