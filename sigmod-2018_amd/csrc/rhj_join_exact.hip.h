// rhj_join_exact.hip.h — fused LDS join over an EXACT index: no verifying gather, no stash, no emit pass
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
#pragma once
#include "rhj_join_fused.hip.h"

namespace rhj {

// ------------------------------------------------------------------------------------------------ k_join_exact
//
// The fused kernel of rhj_join_fused.hip.h keeps (tag16, position) per build tuple in LDS and fetches the build tuple itself
// — 12 bytes out of a 293 KB bucket at 12 radix bits of a 100 M relation — to verify the key and to get the row id: one
// random line per probe tuple out of 9.4 MB of live buckets per XCD, which miss the 4 MB L2s.  That line rate (85 G/s for the
// whole chip, tools/gather_bench.py), not bytes, bounded the kernel (profiles/README.md r01g..r03zn).
//
// This kernel's LDS entry holds enough of the key to be EXACT, so the only thing left outside LDS is the build row id:
//   y        = ((key >> bits) * odd constant) mod 2^W, W = 64 - bits: a BIJECTION of the W key bits a bucket's keys differ in
//   slot     = floor(top 24 bits of y * hs / 2^24)
//   ent[e]   = low 32 bits of y, ext[e] = bits 32..39 of y            (5 bytes per build tuple)
// Two keys of one slot with equal 40 stored bits differ by at least 2^bits in their top 24 bits, a slot spans at most
// ceil(2^24 / hs) of those: with hs >= 2^(24 - bits) slots (4096 at 12 bits) equal (slot, ent, ext) means equal keys.
// The row ids go to a per-workgroup scratch array in entry order (4 bytes per build tuple, 98 KB for a C3 bucket, always the
// same addresses: 3.1 MB per XCD, which the L2 keeps), written by the build and read by ONE 4-byte gather per MATCH.
//
// Build: count per slot with the arrival number taken from the atomic, scan, then every tuple goes straight to
// start[slot] + arrival — no ordered insertion, no second atomic.  The entries of a slot are therefore in arrival order, and
// the matches of a probe tuple (rhjoin.c:141-217 hands them out in DESCENDING build position) are put in order when they are
// emitted: a stable partition keeps a bucket's tuples in input order, so where the build side's row ids increase along it —
// they do at every call site of the reference (row_id[i] = i, inter_res.c:202,225); the build checks it — descending
// position is descending row id, which the emit has in registers anyway.  A build side whose row ids do not increase, a
// build side beyond XJ_MAX_BUILD tuples or without room for the slots exactness needs, 16-byte tuples: the kernel raises the
// word the foreign-key speculation raises (ticket[4]) and the ordinary kernels enqueued behind it do the join.
//
// Units (k_join_spec's protocol, DESIGN.md 4.2: the foreign-key hypothesis gives every bucket's first pair and pair count):
//   the hypothesis' relation probes   one pass: lookup, one gather, pair i of the unit at base + i; every count must be 1
//   the other relation probes         pass A counts (LDS only, nothing stored), the group totals are scanned and checked against
//                                     the predicted total; pass B looks up again, gathers the row ids of up to four matches a
//                                     tuple at once, orders them in registers and writes the pairs.  No stash, no chain.
constexpr int      XJ_MAXB = 7;                                  // build batches whose arrival numbers stay in registers
constexpr uint32_t XJ_MAX_BUILD = XJ_MAXB * FJ_BATCH;            // 28 672 build tuples
constexpr uint32_t XJ_SCRATCH = 32768;                           // scratch row ids per workgroup
constexpr uint32_t XJ_HEAD = FJ_GROUPS * 4u;                     // bytes in front of the index: the unit's group totals
constexpr int      XJ_R = 4;                                     // matches of a tuple ordered in registers

struct XjIndex {
    uint32_t *ent;       // [bcp]
    uint8_t  *ext;       // [bcp]
    uint32_t *dirw;      // [(hs + 3) / 2] 16-bit slot starts, two a word
    uint32_t  hs;
    uint32_t  bits;
    __device__ __forceinline__ uint32_t H(uint32_t j) const { return reinterpret_cast<const uint16_t *>(dirw)[j]; }
    // the W-bit hash, left-aligned in 64 bits
    __device__ __forceinline__ uint64_t hash(uint64_t key) const { return ((key >> bits) * 0x9e3779b97f4a7c15ull) << bits; }
    __device__ __forceinline__ uint32_t slot(uint64_t y) const { return __umulhi((uint32_t)(y >> 32) & 0xffffff00u, hs); }
    __device__ __forceinline__ uint32_t lo32(uint64_t y) const { return (uint32_t)(y >> bits); }
    __device__ __forceinline__ uint32_t hi8(uint64_t y) const { return (uint32_t)(y >> (bits + 32u)) & 0xffu; }
};

// the scratch row ids of this workgroup, read through a buffer descriptor with sc1 (L1 bypass: the build's stores went to
// the L2, and the 32 KiB vector L1 may still hold the previous unit's lines of the same addresses)
struct XjRows {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ void init(const uint32_t *rows)
    {
        const uint64_t addr = (uint64_t)rows;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)(XJ_SCRATCH * 4u), 0x00020000);
    }
    __device__ __forceinline__ uint32_t load(uint32_t e) const
    {
        return __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(e * 4u), 0, 16 /* sc1 */);
    }
};

// Exact matches of `key` among the first XJ window of its slot: d0 = the slot's first entry, n its length, m = bit j set when
// entry d0 + j holds the key (j < min(n, FJ_WIN)).  Slots longer than the window: the caller goes on with xj_more.
__device__ __forceinline__ uint32_t xj_window(const XjIndex &X, uint32_t at, uint32_t n, uint32_t r32, uint32_t r8)
{
    uint32_t e[FJ_WIN];
#pragma unroll
    for (int j = 0; j < FJ_WIN; ++j) e[j] = X.ent[at + j];          // (the array is padded by a window)
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < FJ_WIN; ++j) m |= (e[j] == r32) ? (1u << j) : 0u;
    m &= (1u << min(n, (uint32_t)FJ_WIN)) - 1u;
    for (uint32_t mm = m; mm; mm &= mm - 1u) {                     // (one round nearly always: 32 equal bits are the key's)
        const uint32_t j = (uint32_t)__builtin_ctz(mm);
        if (X.ext[at + j] != r8) m &= ~(1u << j);
    }
    return m;
}

struct XjHit {               // one probe tuple's lookup
    uint32_t d0, n;          // its slot
    uint32_t r32, r8;        // what an entry must hold
    uint32_t m;              // exact hits among the slot's first window
    uint32_t c;              // exact hits in the whole slot
    uint32_t first;          // entry of the first exact hit (anything when c == 0)
};

__device__ __forceinline__ XjHit xj_lookup(const XjIndex &X, uint64_t key, bool ok)
{
    XjHit h;
    const uint64_t y = X.hash(key);
    const uint32_t s = X.slot(y);
    h.d0 = X.H(s + 1u);
    h.n = ok ? X.H(s + 2u) - h.d0 : 0u;
    h.r32 = X.lo32(y); h.r8 = X.hi8(y);
    h.m = xj_window(X, h.d0, h.n, h.r32, h.r8);
    h.c = (uint32_t)__popc(h.m);
    h.first = h.d0 + (h.m ? (uint32_t)__builtin_ctz(h.m) : 0u);
    for (uint32_t w0 = FJ_WIN; w0 < h.n; w0 += FJ_WIN) {           // rare: a slot beyond one window
        const uint32_t m2 = xj_window(X, h.d0 + w0, h.n - w0, h.r32, h.r8);
        if (h.c == 0 && m2) h.first = h.d0 + w0 + (uint32_t)__builtin_ctz(m2);
        h.c += (uint32_t)__popc(m2);
    }
    return h;
}

// Build the exact index of one bucket's build side (whole workgroup) and leave its row ids in `rows` in entry order.
// Returns (workgroup-uniform) whether the row ids increase along the build side.
__device__ __forceinline__ bool xj_build(const XjIndex &X, const Tuple12 *bd, uint32_t bc, uint32_t *rows, uint32_t *wsum, uint64_t *dbg)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t ndw = (X.hs + 3u) / 2u;
    for (uint32_t i = tid; i < ndw; i += FJ_BLOCK) X.dirw[i] = 0;
    if (tid < (uint32_t)FJ_WIN) X.ent[((bc + 3u) & ~3u) + tid] = 0;  // the pad a window may read (never counted: beyond n)
    __syncthreads();
    // ---- count: H[s + 1] += 1; the atomic's old value is this tuple's place inside its slot, parked in the still unused
    // entry array.  On the way: do the row ids increase with the position?
    bool mono = true;
    {
        Tuple12 t[FJ_V], tn[FJ_V];
        uint32_t nx[FJ_V], nxn[FJ_V];
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const uint32_t i = k * FJ_BLOCK + tid;
            t[k] = Tuple12{0, 0, 0}; nx[k] = 0xffffffffu;
            if (i < bc) t[k] = bd[i];
            if (i + 1u < bc) nx[k] = bd[i + 1u].rid;
        }
        for (uint32_t i0 = 0; i0 < bc; i0 += FJ_BATCH) {
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {               // next batch's loads fly while this batch is counted
                const uint32_t i = i0 + FJ_BATCH + k * FJ_BLOCK + tid;
                tn[k] = Tuple12{0, 0, 0}; nxn[k] = 0xffffffffu;
                if (i < bc) tn[k] = bd[i];
                if (i + 1u < bc) nxn[k] = bd[i + 1u].rid;
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = i0 + k * FJ_BLOCK + tid;
                if (i < bc) {
                    const uint64_t y = X.hash(((uint64_t)t[k].khi << 32) | t[k].klo);
                    const uint32_t j = X.slot(y) + 1u;
                    const uint32_t old = atomicAdd(&X.dirw[j >> 1], (j & 1u) ? 0x10000u : 1u);
                    X.ent[i] = (j & 1u) ? old >> 16 : old & 0xffffu;
                    mono = mono && (i + 1u >= bc || t[k].rid < nx[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) { t[k] = tn[k]; nx[k] = nxn[k]; }
        }
    }
    mono = __syncthreads_and(mono) != 0;
    if (dbg && tid == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
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
    // ---- fill: the parked arrival numbers pass through registers (the entry array is about to be overwritten in entry order)
    uint32_t ar[XJ_MAXB][FJ_V];
#pragma unroll
    for (int b = 0; b < XJ_MAXB; ++b)
#pragma unroll
        for (int k = 0; k < FJ_V; ++k) {
            const uint32_t i = (uint32_t)b * FJ_BATCH + k * FJ_BLOCK + tid;
            ar[b][k] = i < bc ? X.ent[i] : 0;
        }
    __syncthreads();                                   // also: the scanned slot starts are out
    if (dbg && tid == 0) dbg[2] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int b = 0; b < XJ_MAXB; ++b) {
        if ((uint32_t)b * FJ_BATCH < bc) {              // (workgroup-uniform)
            Tuple12 t[FJ_V];
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = (uint32_t)b * FJ_BATCH + k * FJ_BLOCK + tid;
                t[k] = Tuple12{0, 0, 0};
                if (i < bc) t[k] = bd[i];
            }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = (uint32_t)b * FJ_BATCH + k * FJ_BLOCK + tid;
                if (i < bc) {
                    const uint64_t y = X.hash(((uint64_t)t[k].khi << 32) | t[k].klo);
                    const uint32_t e = X.H(X.slot(y) + 1u) + ar[b][k];
                    X.ent[e] = X.lo32(y);
                    X.ext[e] = (uint8_t)X.hi8(y);
                    rows[e] = t[k].rid;
                }
            }
        }
    }
    __syncthreads();                                   // index and row ids are out (a barrier waits for the wave's stores)
    if (dbg && tid == 0) dbg[3] = __builtin_amdgcn_s_memrealtime();
    return mono;
}

// SPEC-only kernel body (see the header comment); `f.spec` names the hypothesis' relation as in k_join_spec.
__device__ __forceinline__ void xj_body(const FusedArgs &f, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];
    __shared__ uint32_t sh_u;
    __shared__ uint32_t sh_grab;
    __shared__ uint32_t wsum[FJ_WAVES];
    const JoinArgs &a = f.j;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const uint64_t cap = a.out_capacity;
    uint32_t *gtab = tbl;                                // [FJ_GROUPS] group totals, then exclusive starts
    uint32_t *rows = f.xrows + (size_t)blockIdx.x * XJ_SCRATCH;
    XjRows G;
    G.init(rows);
    // why: 1 = a check of the speculation failed, 2 = an input this kernel does not take (the host keeps the two apart)
    auto fail = [&](uint32_t why) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_max(f.ticket + 5, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f.ticket + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // what this kernel cannot do is known before the first unit: 16-byte tuples, a bucket beyond the exact index
    const uint32_t hs_min = f.radix_bits >= 24u ? 1u : 1u << (24u - f.radix_bits);
    {
        const uint64_t mb = a.summary->max_build;
        const uint64_t room = XJ_HEAD + 5u * (mb + 3u + FJ_WIN) + 64u + 2u * ((uint64_t)hs_min + 16u);
        if (a.summary->wide_row_ids != 0 || !a.summary->fused_ok || mb > XJ_MAX_BUILD || room > lds_bytes) { fail(2u); return; }
    }

    uint64_t *dbg_prev = nullptr;
    for (;;) {
        __syncthreads();
        if (dbg_prev && threadIdx.x == 0 && dbg_prev[6] == 0) dbg_prev[6] = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            sh_u = atomicAdd(f.ticket, 1u); sh_grab = 0;
            if (__hip_atomic_load(f.ticket + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) sh_u = 0xffffffffu;
        }
        __syncthreads();
        const uint32_t u = sh_u;
        if (u >= a.summary->units) break;
        const Unit un = a.units[u];
        const uint32_t b = un.bucket;
        const uint64_t cR = a.histR[b], cS = a.histS[b];
        const bool flip = bucket_flip(a, b, cR, cS);                       // S is streamed
        const uint64_t ppos = (flip ? a.psumS[b] : a.psumR[b]) + un.off;
        const bool fkp = flip == (f.spec == 1u);                           // the probe side is the hypothesis' relation
        const uint64_t spec_base = (f.spec == 1u ? a.psumS[b] : a.psumR[b]) + (fkp ? un.off : 0u);
        const uint64_t spec_total = fkp ? (uint64_t)un.count : (f.spec == 1u ? cS : cR);
        bool bad = !fkp && (un.off != 0 || un.count != (flip ? cS : cR));  // (a split bucket's later units cannot know their base)
        if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(f.ticket + 6), (unsigned long long)spec_total);
        const Tuple12 *pr = reinterpret_cast<const Tuple12 *>(flip ? a.partS : a.partR) + ppos;
        const Tuple12 *bd = reinterpret_cast<const Tuple12 *>(flip ? a.partR : a.partS) + (flip ? a.psumR[b] : a.psumS[b]);
        const uint32_t bc = (uint32_t)(flip ? cR : cS);
        const uint32_t bcp = (bc + 3u) & ~3u;
        // LDS: [group totals][entries 4 B x (bcp + 8)][ext 1 B x (bcp + 8)][slot starts 2 B x (hs + 3)]
        XjIndex X;
        X.ent = tbl + XJ_HEAD / 4u;
        X.ext = reinterpret_cast<uint8_t *>(X.ent + bcp + FJ_WIN);
        X.dirw = reinterpret_cast<uint32_t *>(X.ext) + (bcp + FJ_WIN) / 4u;
        X.bits = f.radix_bits;
        {
            const uint32_t used = XJ_HEAD + 5u * (bcp + FJ_WIN) + 64u;
            const uint32_t room = (lds_bytes - used) / 2u - 8u;            // 16-bit slot starts that still fit
            uint32_t hs0 = bc > hs_min ? bc : hs_min;
            if (hs0 < 64u) hs0 = 64u;
            if (hs0 > room) hs0 = room & ~1u;                              // (>= hs_min: checked against max_build above)
            if (hs0 > 65534u) hs0 = 65534u;
            X.hs = hs0;
        }
        uint64_t *dbg = FJ_DBG ? FJ_DBG + (size_t)u * 8 : nullptr;         // diagnostics build: phase stamps (tools/exp_xstamps.py)
        if (dbg && threadIdx.x == 0) { dbg[0] = __builtin_amdgcn_s_memrealtime(); dbg[6] = 0; dbg[7] = fkp; }
        dbg_prev = dbg;
        if (!xj_build(X, bd, bc, rows, wsum, dbg)) { fail(2u); break; }         // row ids that do not increase: not this kernel's (workgroup-uniform)

        const uint32_t ngroups = (un.count + 255u) >> 8;
        if (fkp) {
            // ---- the hypothesis' relation probes: one match each, pair i of the unit at base + i
            if (!bad) {
                for (;;) {
                    uint32_t grp = 0;
                    if (lane == 0) grp = atomicAdd(&sh_grab, 1u);
                    grp = __builtin_amdgcn_readfirstlane(grp);
                    if (grp >= ngroups) break;
                    const uint32_t t0 = grp << 8;
                    Tuple12 q[FJ_V];
                    XjHit h[FJ_V];
                    uint32_t r[FJ_V];
#pragma unroll
                    for (int k = 0; k < FJ_V; ++k) {
                        const uint32_t i = t0 + k * WAVE + lane;
                        q[k] = i < un.count ? pr[i] : Tuple12{0, 0, 0};
                    }
#pragma unroll
                    for (int k = 0; k < FJ_V; ++k) {
                        const uint32_t i = t0 + k * WAVE + lane;
                        h[k] = xj_lookup(X, ((uint64_t)q[k].khi << 32) | q[k].klo, i < un.count);
                        bad = bad || (i < un.count && h[k].c != 1u);
                    }
#pragma unroll
                    for (int k = 0; k < FJ_V; ++k) r[k] = h[k].c ? G.load(h[k].first) : 0u;
#pragma unroll
                    for (int k = 0; k < FJ_V; ++k) {
                        const uint32_t i = t0 + k * WAVE + lane;
                        const uint64_t at = spec_base + i;
                        if (i < un.count && at < cap) out[at] = make_pair(flip, q[k].rid, 0u, r[k], 0u);
                    }
                    if (__ballot(bad) != 0) { bad = true; break; }
                }
            }
            if (__syncthreads_or(bad)) { fail(1u); break; }
            if (dbg && threadIdx.x == 0) dbg[4] = dbg[5] = dbg[6] = __builtin_amdgcn_s_memrealtime();
            continue;
        }

        // ---- the other relation probes.  Pass A: counts, group totals
        uint32_t mine = 0;
        if (!bad) {
            for (;;) {
                uint32_t grp = 0;
                if (lane == 0) grp = atomicAdd(&sh_grab, 1u);
                grp = __builtin_amdgcn_readfirstlane(grp);
                if (grp >= ngroups) break;
                const uint32_t t0 = grp << 8;
                uint2 kq[FJ_V];
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    const uint32_t i = t0 + k * WAVE + lane;
                    kq[k] = make_uint2(0, 0);
                    if (i < un.count) kq[k] = make_uint2(pr[i].klo, pr[i].khi);
                }
                uint32_t cs = 0;
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    const uint32_t i = t0 + k * WAVE + lane;
                    cs += xj_lookup(X, ((uint64_t)kq[k].y << 32) | kq[k].x, i < un.count).c;
                }
                uint32_t gt;
                wave_excl_scan_u32(cs, &gt);
                if (lane == 0) gtab[grp] = gt;
                mine += cs;
            }
        }
        {
            uint32_t tot;
            wave_excl_scan_u32(mine, &tot);
            if (lane == 0) wsum[w] = tot;
        }
        if (__syncthreads_or(bad)) { fail(1u); break; }                     // also publishes wsum and gtab
        if (dbg && threadIdx.x == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
        uint64_t total = 0;
#pragma unroll
        for (int i = 0; i < FJ_WAVES; ++i) total += wsum[i];
        if (total != spec_total) { fail(1u); break; }                        // (workgroup-uniform)
        {                                                                  // group totals -> exclusive starts
            const uint32_t t = threadIdx.x;
            const uint32_t v = t < ngroups ? gtab[t] : 0u;
            uint32_t tot;
            uint32_t ex = wave_excl_scan_u32(v, &tot);
            __syncthreads();                                               // wsum reuse
            if (t == 0) { sh_grab = 0; a.unit_count[u] = total; }
            if (lane == 0) wsum[w] = tot;
            __syncthreads();
            for (uint32_t i = 0; i < w && i < FJ_GROUPS / WAVE; ++i) ex += wsum[i];
            if (t < ngroups) gtab[t] = ex;
            __syncthreads();
        }
        if (dbg && threadIdx.x == 0) dbg[5] = __builtin_amdgcn_s_memrealtime();
        // ---- pass B: look up again, order every tuple's matches by descending row id (= descending build position), write
        for (;;) {
            uint32_t grp = 0;
            if (lane == 0) grp = atomicAdd(&sh_grab, 1u);
            grp = __builtin_amdgcn_readfirstlane(grp);
            if (grp >= ngroups) break;
            const uint32_t t0 = grp << 8;
            Tuple12 q[FJ_V];
            XjHit h[FJ_V];
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = t0 + k * WAVE + lane;
                q[k] = i < un.count ? pr[i] : Tuple12{0, 0, 0};
            }
            uint32_t off[FJ_V], wrun = gtab[grp];
            bool fast[FJ_V], slow = false;                     // fast: all of the tuple's matches (at most XJ_R) sit in its slot's first window
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                const uint32_t i = t0 + k * WAVE + lane;
                h[k] = xj_lookup(X, ((uint64_t)q[k].khi << 32) | q[k].klo, i < un.count);
                uint32_t tot;
                off[k] = wrun + wave_excl_scan_u32(h[k].c, &tot);
                wrun += tot;
                fast[k] = h[k].c <= (uint32_t)XJ_R && h[k].c == (uint32_t)__popc(h[k].m);
                slow = slow || !fast[k];
            }
            uint32_t r[XJ_R][FJ_V];
#pragma unroll
            for (int j = 0; j < XJ_R; ++j)
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    r[j][k] = 0;
                    if (fast[k] && h[k].m) {
                        r[j][k] = G.load(h[k].d0 + (uint32_t)__builtin_ctz(h[k].m));
                        h[k].m &= h[k].m - 1u;
                    }
                }
#pragma unroll
            for (int k = 0; k < FJ_V; ++k) {
                // descending; the places beyond the count hold 0 and stay behind (a row id 0 among the matches is the smallest anyway)
#define XJ_CX(x, y) { const uint32_t hi_ = max(r[x][k], r[y][k]), lo_ = min(r[x][k], r[y][k]); r[x][k] = hi_; r[y][k] = lo_; }
                XJ_CX(0, 1) XJ_CX(2, 3) XJ_CX(0, 2) XJ_CX(1, 3) XJ_CX(1, 2)
#undef XJ_CX
                const uint64_t at = spec_base + off[k];
#pragma unroll
                for (int j = 0; j < XJ_R; ++j)
                    if (fast[k] && h[k].c > (uint32_t)j && at + j < cap) out[at + j] = make_pair(flip, q[k].rid, 0u, r[j][k], 0u);
            }
            if (__ballot(slow) != 0) {
                // rare: a tuple with more than four matches, or matches beyond its slot's first window — one pair a step, the
                // largest row id below the previous one (row ids of one bucket's build side are distinct: they increase)
#pragma unroll
                for (int k = 0; k < FJ_V; ++k) {
                    if (!fast[k]) {
                        uint64_t prev = 1ull << 32;
                        for (uint32_t j = 0; j < h[k].c; ++j) {
                            uint32_t best = 0;
                            for (uint32_t e = h[k].d0; e < h[k].d0 + h[k].n; ++e)
                                if (X.ent[e] == h[k].r32 && X.ext[e] == h[k].r8) {
                                    const uint32_t v = G.load(e);
                                    if ((uint64_t)v < prev && v >= best) best = v;
                                }
                            const uint64_t at = spec_base + off[k] + j;
                            if (at < cap) out[at] = make_pair(flip, q[k].rid, 0u, best, 0u);
                            prev = best;
                        }
                    }
                }
            }
        }
    }
}

// Always followed by the ordinary kernels with the same arguments (k_join_fused returns at once when ticket[4] stayed clear).
__global__ __launch_bounds__(FJ_BLOCK) void k_join_exact(FusedArgs f, uint32_t lds_bytes)
{
    xj_body(f, lds_bytes);
    __syncthreads();
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(f.ticket + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {   // the last workgroup out
        __hip_atomic_store(f.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long predicted =
            __hip_atomic_load(reinterpret_cast<unsigned long long *>(f.ticket + 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = (1u << f.radix_bits) - 1u;
        const uint64_t n_fk = f.spec == 1u ? f.j.psumS[last] + f.j.histS[last] : f.j.psumR[last] + f.j.histR[last];
        const bool failed = __hip_atomic_load(f.ticket + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || predicted != n_fk;
        if (failed) {                                 // the ordinary kernel behind this one starts over
            __hip_atomic_store(f.ticket + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f.ticket + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(f.ticket + 6), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else
            const_cast<PlanSummary *>(f.j.summary)->matches = n_fk;
    }
}

}  // namespace rhj
