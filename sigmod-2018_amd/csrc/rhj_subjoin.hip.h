// rhj_subjoin.hip.h — the sub-split join path (radix bits 9..15, row ids below 2^32).
//
// Why.  At 12 radix bits a 100 M build side gives 24.4 K-tuple buckets: 390 KB, more than a CU's LDS.  The fused
// kernel (rhj_kernels.hip.h) therefore keeps only a (tag, position) index in LDS and verifies every candidate with
// a random 16-byte gather from HBM/Infinity Cache (100 M gathers = 6.4 GB of line traffic for 1.6 GB of tuples), in
// one 1024-thread workgroup per CU.  Here the partition goes k bits FURTHER than the join's radix: bucket b of the
// join (rhjoin.c:42-57) is stored as 2^k sub-buckets (b, s), s = the next k key bits, each small enough that its
// build tuples live in LDS whole — no gather, small workgroups, several per CU:
//
//   pass 1   k_local_part (rhj_kernels.hip.h), unchanged: every 4096-tuple tile partitioned in place on the low `lo` bits
//   pass 2   k_hist_runs (unchanged, digit = the next hb + k bits), k_sub_colsum / k_sub_segscan / k_sub_apply (offsets in
//            the order (h, l, s, tile): bucket-major, sub-bucket inside the bucket), k_scatter_sub: 12-byte tuples to
//            their sub-bucket, and — because the canonical result order is the order of the probe tuples inside the
//            JOIN's bucket (stable partition on `bits` bits, preprocess.c:349-359), not inside the sub-bucket —
//            one byte per tuple at its CANONICAL position: the sub-bucket it went to (sseq)
//   K1       k_sub_join: one 512-thread workgroup per (sub-bucket, span of its probe side): CSR index + the build
//            tuples' upper key bits and row ids in LDS (14 B per build tuple, exact: the index tag is 19 literal key
//            bits, the rest of the key is compared in LDS), probe side streamed once; per probe tuple the match count
//            and {first match's build row id, probe row id} are stashed at the tuple's place in the sub-split array
//   scan     k_sub_bscan: matches per bucket -> first output position of every bucket
//   K2       k_sub_emit: one WAVE per bucket replays sseq: position P of the bucket came from sub-bucket s = sseq[P], and
//            is the (number of earlier positions with the same s)-th tuple of it — 2^k running counters, k ballots per
//            64 positions — so the stash is read back in canonical order (2^k coalesced streams) and the pairs are
//            written in the reference's order: bucket ascending, probe tuples in input order, build matches in
//            descending position (rhjoin.c:79, :141-250).
//
// What falls back to the fused / tiled paths (the host runs them after the plan said no): row ids of 2^32 or more,
// a sub-bucket whose build side exceeds the LDS cap (few distinct keys), radix bits below 9.
#pragma once
#include "rhj_kernels.hip.h"

namespace rhj {

// one relation through pass 2 of the sub-split partition
struct SubRel {
    RelArgs         r;          // in = pass-1 output (Tuple12), out = sub-split array (Tuple12), cnt = [tile2][2^hi] rows, runs ...
    uint8_t        *sseq;       // [n] sub-bucket of the tuple at each CANONICAL position of the relation's partition
    uint32_t       *segsum;     // [bins1][S][H] tuples of sub-bucket (h, l, s)
    uint32_t       *segbase;    // [bins1][S][H] first position of sub-bucket (h, l, s) in the sub-split array
};

struct SubGeom {                // lo + hb = the join's radix bits, kb = sub bits; pass-2 digit d = s << hb | h
    int lo, hb, kb;
};

__device__ __forceinline__ uint32_t sub_index(const SubGeom &g, uint32_t bucket, uint32_t s)
{
    const uint32_t l = bucket & ((1u << g.lo) - 1u), h = bucket >> g.lo;
    return (((l << g.kb) | s) << g.hb) | h;
}

// ---- offsets of pass 2: table A[l][j][d] (d = s << hb | h), wanted: exclusive prefix in the order (h, l, s, j) -------

// segsum[l][s][h] = sum over j of A[l][j][s][h]; one workgroup per (l, s)
__global__ __launch_bounds__(256) void k_sub_colsum(SubRel a0, SubRel a1, SubGeom g)
{
    __shared__ uint32_t part[256];
    const SubRel &a = blockIdx.y ? a1 : a0;
    const uint32_t H = 1u << g.hb, D = 1u << (g.hb + g.kb);
    const uint32_t l = blockIdx.x >> g.kb, s = blockIdx.x & ((1u << g.kb) - 1u);
    const uint32_t h = threadIdx.x & (H - 1u), jj = threadIdx.x >> g.hb, R = 256u >> g.hb;
    const uint32_t *row = a.r.cnt + (size_t)l * a.r.groups * D + s * H + h;
    uint32_t acc = 0;
#pragma unroll 4
    for (uint32_t j = jj; j < a.r.groups; j += R) acc += row[(size_t)j * D];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < H) {
        uint32_t t = 0;
        for (uint32_t q = 0; q < R; ++q) t += part[q * H + threadIdx.x];
        a.segsum[(size_t)blockIdx.x * H + threadIdx.x] = t;
    }
}

// segbase[l][s][h] = tuples in front of sub-bucket (h, l, s) in the order (h, l, s).  One workgroup per relation;
// wave w takes the columns h = w, w + 16, ...: scan over the segments (l, s) 64 at a time, then the columns' bases.
__global__ __launch_bounds__(1024) void k_sub_segscan(SubRel a0, SubRel a1, SubGeom g)
{
    __shared__ uint64_t coltot[256];
    __shared__ uint64_t sm[1024 / 64 + 1];
    const SubRel &a = blockIdx.x ? a1 : a0;
    const uint32_t H = 1u << g.hb, segs = 1u << (g.lo + g.kb);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t h = w; h < H; h += 16) {
        uint32_t carry = 0;
        for (uint32_t s0 = 0; s0 < segs; s0 += 64) {
            const uint32_t sg = s0 + lane;
            const uint32_t v = sg < segs ? a.segsum[(size_t)sg * H + h] : 0;
            uint32_t tot;
            const uint32_t e = wave_excl_scan_u32(v, &tot);
            if (sg < segs) a.segbase[(size_t)sg * H + h] = carry + e;
            carry += tot;
        }
        if (lane == 0) coltot[h] = carry;
    }
    __syncthreads();
    const uint64_t mine = threadIdx.x < H ? coltot[threadIdx.x] : 0;
    const uint64_t base = block_excl_scan<1024>(mine, nullptr, sm);
    if (threadIdx.x < H) coltot[threadIdx.x] = base;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < segs * H; i += 1024) a.segbase[i] += (uint32_t)coltot[i & (H - 1u)];
}

// A[l][j][s][h] (counts) -> first output position of tile (l, j)'s tuples of digit (s, h); one workgroup per (l, s):
// thread (jj, h) owns a contiguous range of j
__global__ __launch_bounds__(256) void k_sub_apply(SubRel a0, SubRel a1, SubGeom g)
{
    __shared__ uint32_t part[256];
    const SubRel &a = blockIdx.y ? a1 : a0;
    const uint32_t H = 1u << g.hb, D = 1u << (g.hb + g.kb);
    const uint32_t l = blockIdx.x >> g.kb, s = blockIdx.x & ((1u << g.kb) - 1u);
    const uint32_t h = threadIdx.x & (H - 1u), jj = threadIdx.x >> g.hb, R = 256u >> g.hb;
    const uint32_t per = (a.r.groups + R - 1u) / R;
    const uint32_t j0 = min(jj * per, a.r.groups), j1 = min(j0 + per, a.r.groups);
    uint32_t *row = a.r.cnt + (size_t)l * a.r.groups * D + s * H + h;
    uint32_t acc = 0;
#pragma unroll 4
    for (uint32_t j = j0; j < j1; ++j) acc += row[(size_t)j * D];
    part[threadIdx.x] = acc;
    __syncthreads();
    uint32_t run = a.segbase[(size_t)blockIdx.x * H + h];
    for (uint32_t q = 0; q < jj; ++q) run += part[q * H + h];
#pragma unroll 4
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t c = row[(size_t)j * D];
        row[(size_t)j * D] = run;
        run += c;
    }
}

// ---- pass-2 scatter into sub-buckets ------------------------------------------------------------------------------
// k_scatter_runs (rhj_kernels.hip.h) with two ranks per tuple from the same ballots: its rank among the tile's tuples
// of the same (h, s) — where it goes — and among those of the same h — its canonical position, where its sseq byte
// goes.  The tile is staged in LDS in (h, s) order as three word arrays (12-byte tuples), the sseq bytes in h order.
#ifndef SS_ABL
#define SS_ABL 0        // timing experiments only: 1 no sseq store, 2 no h-rank / sseq staging at all, 3 also no tuple store
#endif
#ifndef SS_BLOCK
#define SS_BLOCK 512    // threads per workgroup of k_scatter_sub
#endif
#ifndef SS_V
#define SS_V 4          // tuples per thread and batch (8: 2.45 ms on C3, 4: 2.04 ms — six waves per SIMD instead of four)
#endif
#ifndef SS_MINW
#define SS_MINW 6       // waves per SIMD the register allocation has to allow
#endif
constexpr int SS_TILE = SS_BLOCK * SS_V;
constexpr int SS_WAVES = SS_BLOCK / WAVE;
constexpr size_t SS_LDS_BYTES = (size_t)SS_TILE * 12 + SS_TILE + (size_t)2 * SS_WAVES * 256 * 4 + (size_t)6 * 256 * 4 +
                                (SS_BLOCK / 64 + 2) * 8 + (2 * PT_MAX_GROUP + 1) * 4 + 64;

__global__ __launch_bounds__(SS_BLOCK, SS_MINW) void k_scatter_sub(SubRel a0, SubRel a1, SubGeom g, uint32_t search0, const PlanSummary *summary)
{
    if (summary->wide_row_ids != 0) return;               // 16-byte intermediates: this path does not run
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_klo = reinterpret_cast<uint32_t *>(smem);                      // [SS_TILE] x 3
    uint32_t *s_khi = s_klo + SS_TILE, *s_rid = s_khi + SS_TILE;
    uint8_t  *s_sq = reinterpret_cast<uint8_t *>(s_rid + SS_TILE);            // [SS_TILE] digit byte in canonical order
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(s_sq + SS_TILE);            // [SS_WAVES][256] per digit d = s << hb | h
    uint32_t *wcnth = wcnt + SS_WAVES * 256;                                  // [SS_WAVES][256] per h
    uint32_t *dstart = wcnth + SS_WAVES * 256;                                // [256] LDS start of digit d's run (runs in (h, s) order)
    uint32_t *delta = dstart + 256;                                           // [256] output position - LDS position, digit d
    uint32_t *gbase = delta + 256;                                            // [256] next output position of digit d
    uint32_t *pdelta = gbase + 256;                                           // [256] canonical position - LDS position, h
    uint32_t *pbase = pdelta + 256;                                           // [256] next canonical position of h
    uint32_t *hstart = pbase + 256;                                           // [256] LDS start of h's run; (tile prologue: scratch)
    uint64_t *sm = reinterpret_cast<uint64_t *>(hstart + 256);                // scan scratch [SS_BLOCK / 64 + 1]
    uint32_t *runoff = reinterpret_cast<uint32_t *>(sm + SS_BLOCK / 64 + 2);  // [PT_MAX_GROUP + 1]
    uint32_t *rbase = runoff + PT_MAX_GROUP + 1;                              // [PT_MAX_GROUP]

    const SubRel &a = blockIdx.y ? a1 : a0;
    const RelArgs &r = a.r;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    const int hb = g.hb, kb = g.kb, lo = g.lo, db = g.hb + g.kb;
    const uint32_t H = 1u << hb, S = 1u << kb, D = 1u << db;
    const Tuple12 *in = reinterpret_cast<const Tuple12 *>(r.in);
    Tuple12 *out = reinterpret_cast<Tuple12 *>(r.out);

    // tile order over the XCDs as in k_scatter_runs (blocks of consecutive tiles per XCD, round-robin)
    const uint32_t xcd = blockIdx.x & 7u, per_xcd = gridDim.x >> 3;
    const uint32_t slot = blockIdx.x >> 3;
    const uint32_t tstep = 8u * per_xcd;
    const uint32_t t_end = r.tiles;
    const uint32_t t_first = xcd * per_xcd + slot;
    // thread x < D: the digit it describes in the scans is the x-th in (h, s) order, d = (x & (S - 1)) << hb | x >> kb
    const uint32_t my_d = ((threadIdx.x & (S - 1u)) << hb) | (threadIdx.x >> kb);
    uint32_t nphys = 0, nlen = 0, ngb = 0, nsb = 0;
    if (t_first < t_end) {
        pt_run_of(r, t_first, threadIdx.x, nphys, nlen);
        if (threadIdx.x < D) {
            ngb = r.cnt[(size_t)t_first * D + threadIdx.x];
            nsb = a.segbase[(size_t)(t_first / r.groups) * D + threadIdx.x];
        }
    }
    for (uint32_t tile2 = t_first; tile2 < t_end; tile2 += tstep) {
    uint32_t total;
    {
        const uint32_t phys = nphys, len = nlen;
        if (threadIdx.x < D) { gbase[threadIdx.x] = ngb; hstart[threadIdx.x] = ngb - nsb; }     // by raw digit d = threadIdx.x
        uint64_t tot64;
        const uint32_t off = (uint32_t)block_excl_scan<SS_BLOCK>(len, &tot64, sm);      // (its barriers publish hstart)
        total = (uint32_t)tot64;
        if (threadIdx.x < PT_MAX_GROUP) { runoff[threadIdx.x] = threadIdx.x < r.group ? off : total; rbase[threadIdx.x] = phys - off; }
        if (threadIdx.x == 0) runoff[PT_MAX_GROUP] = total;
        if (threadIdx.x < H) {                        // canonical start of the tile's h-run: bucket start + what the
            uint32_t p = nsb;                         // tile's predecessors put into each sub-bucket (thread x < H: s = 0)
            for (uint32_t s = 0; s < S; ++s) p += hstart[s * H + threadIdx.x];
            pbase[threadIdx.x] = p;
        }
        const uint32_t nt = tile2 + tstep;
        nphys = 0; nlen = 0;
        if (nt < t_end) {
            pt_run_of(r, nt, threadIdx.x, nphys, nlen);
            if (threadIdx.x < D) {
                ngb = r.cnt[(size_t)nt * D + threadIdx.x];
                nsb = a.segbase[(size_t)(nt / r.groups) * D + threadIdx.x];
            }
        }
    }
    __syncthreads();

    for (uint32_t sb = 0; sb < total; sb += SS_TILE) {
        const uint32_t count = min((uint32_t)SS_TILE, total - sb);
        for (uint32_t i = threadIdx.x; i < 2 * SS_WAVES * 256; i += SS_BLOCK) wcnt[i] = 0;      // wcnt and wcnth

        uint32_t tk[SS_V], th[SS_V], tr[SS_V];
        bool ok[SS_V];
        uint32_t pos = 0;
#pragma unroll
        for (int k = 0; k < SS_V; ++k) {
            const uint32_t i = w * (WAVE * SS_V) + k * WAVE + lane;
            ok[k] = i < count;
            const uint32_t e = sb + i;
            if (k == 0) {
                for (uint32_t s2 = search0; s2 >= 1; s2 >>= 1)
                    if (runoff[pos + s2] <= e) pos += s2;
            } else {
                if (runoff[pos + 1] <= e) ++pos;
                if (runoff[pos + 1] <= e) ++pos;
                if (runoff[pos + 1] <= e) {
                    pos = 0;
                    for (uint32_t s2 = search0; s2 >= 1; s2 >>= 1)
                        if (runoff[pos + s2] <= e) pos += s2;
                }
            }
            tk[k] = th[k] = tr[k] = 0;
            if (ok[k]) { const Tuple12 x = in[rbase[pos] + e]; tk[k] = x.klo; th[k] = x.khi; tr[k] = x.rid; }
        }
        __syncthreads();

        uint32_t rk[SS_V];                            // rank among digit d | rank among h << 12 | d << 24 (ranks < 4096)
        uint32_t *mycnt = wcnt + w * 256, *mycnth = wcnth + w * 256;
#pragma unroll
        for (int k = 0; k < SS_V; ++k) {
            const uint64_t key = ((uint64_t)th[k] << 32) | tk[k];
            const uint32_t d = (uint32_t)(key >> lo) & (D - 1u);
            // match-any over the digit's bits, h bits first: the lanes with the same h after hb ballots, the same d after all
            const uint64_t valid = __ballot(ok[k]);
            uint32_t plo = (uint32_t)valid, phi = (uint32_t)(valid >> 32), hlo = plo, hhi = phi;
#pragma unroll
            for (int b = 0; b < PT_MAX_BITS; ++b) {
                if (b < db) {                                          // wave-uniform
                    const uint32_t pb = ok[k] ? 0u - ((d >> b) & 1u) : 0u;
                    const uint64_t m = __ballot(pb != 0);
                    plo &= ~((uint32_t)m ^ pb);
                    phi &= ~((uint32_t)(m >> 32) ^ pb);
                    if (b == hb - 1) { hlo = plo; hhi = phi; }
                }
            }
            const uint64_t pp = ((uint64_t)phi << 32) | plo, ph = ((uint64_t)hhi << 32) | hlo;
            const uint32_t h = d & (H - 1u);
            // the whole group reads its counter, then its lowest lane adds the group (a wave's LDS operations are in order)
            const uint32_t old = mycnt[d];
            const uint32_t rank = (uint32_t)__popcll(pp & lt);
            if (ok[k] && rank == 0) mycnt[d] = old + (uint32_t)__popcll(pp);
            uint32_t oldh = 0, rankh = 0;
            if (SS_ABL < 2) {
                oldh = mycnth[h];
                rankh = (uint32_t)__popcll(ph & lt);
                if (ok[k] && rankh == 0) mycnth[h] = oldh + (uint32_t)__popcll(ph);
            }
            rk[k] = (old + rank) | ((oldh + rankh) << 12) | (d << 24);
        }
        __syncthreads();

        uint64_t mytotal = 0;
        uint32_t mytotalh = 0;
        if (threadIdx.x < D) {                        // digit my_d: x-th in (h, s) order
            uint32_t run = 0;
            for (int ww = 0; ww < SS_WAVES; ++ww) {
                const uint32_t c = wcnt[ww * 256 + my_d];
                wcnt[ww * 256 + my_d] = run;
                run += c;
            }
            mytotal = run;
        }
        if (threadIdx.x < H) {
            uint32_t run = 0;
            for (int ww = 0; ww < SS_WAVES; ++ww) {
                const uint32_t c = wcnth[ww * 256 + threadIdx.x];
                wcnth[ww * 256 + threadIdx.x] = run;
                run += c;
            }
            mytotalh = run;
        }
        const uint64_t ds = block_excl_scan<SS_BLOCK>(mytotal, nullptr, sm);
        if (threadIdx.x < D) {
            dstart[my_d] = (uint32_t)ds;
            const uint32_t gb = gbase[my_d];
            delta[my_d] = gb - (uint32_t)ds;                        // mod 2^32
            gbase[my_d] = gb + (uint32_t)mytotal;
        }
        __syncthreads();
        if (threadIdx.x < H) {
            const uint32_t hs = dstart[threadIdx.x];                // the h-run starts where its s = 0 run starts (d = h)
            const uint32_t pb = pbase[threadIdx.x];
            hstart[threadIdx.x] = hs;
            pdelta[threadIdx.x] = pb - hs;
            pbase[threadIdx.x] = pb + mytotalh;
        }
        __syncthreads();

#pragma unroll
        for (int k = 0; k < SS_V; ++k) {
            if (ok[k]) {
                const uint32_t d = rk[k] >> 24;
                const uint32_t p = dstart[d] + mycnt[d] + (rk[k] & 0xfffu);
                s_klo[p] = tk[k]; s_khi[p] = th[k]; s_rid[p] = tr[k];
                const uint32_t h = d & (H - 1u);
                if (SS_ABL < 2) s_sq[hstart[h] + mycnth[h] + ((rk[k] >> 12) & 0xfffu)] = (uint8_t)d;
            }
        }
        __syncthreads();

#pragma unroll 2
        for (int k = 0; k < SS_V; ++k) {
            const uint32_t p = k * SS_BLOCK + threadIdx.x;
            if (p < count) {
                const uint32_t klo = s_klo[p], khi = s_khi[p];
                const uint32_t d = (uint32_t)((((uint64_t)khi << 32) | klo) >> lo) & (D - 1u);
                if (SS_ABL < 3) out[delta[d] + p] = Tuple12{klo, khi, s_rid[p]};
                if (SS_ABL < 1) {
                    const uint32_t q = s_sq[p];
                    a.sseq[pdelta[q & (H - 1u)] + p] = (uint8_t)(q >> hb);
                }
            }
        }
        __syncthreads();
    }
    }   // grid-stride loop
}

// ---- plan ---------------------------------------------------------------------------------------------------------

constexpr int SJ_BLOCK = 512;
constexpr int SJ_NB = 7;                             // build tuples per thread at most
constexpr uint32_t SJ_CAP = SJ_BLOCK * SJ_NB;        // 3584 build tuples per sub-bucket, 13 B each in LDS: three workgroups per CU
#ifndef SJ_PVN
#define SJ_PVN 4
#endif
constexpr int SJ_PV = SJ_PVN;                        // probe tuples per thread and batch
constexpr uint32_t SJ_SPAN = 8192;                   // probe tuples per unit
#ifndef SJ_ABL
#define SJ_ABL 0        // timing experiments only: 1 no arena work (tuples with several matches keep the first one only)
#endif
constexpr uint32_t SJ_LONG = 16;                     // slots above this are filled in arrival order and ranked afterwards
constexpr uint32_t SJ_MAX_SLOTS = SJ_CAP / 2 + 64;   // one slot per two build tuples
constexpr size_t SJ_LDS_BYTES = (size_t)(SJ_CAP + 8) * 4 + (size_t)SJ_CAP * 8 + ((SJ_MAX_SLOTS + 3) / 2 + 2) * 4;

struct SjDesc {                  // one sub-bucket (b, s)
    uint32_t build_off, bc;      // its build side in the build relation's sub-split array
    uint32_t probe_off, pc;      // its probe side; pc = 0: nothing to do (bucket inactive or no probe tuples)
    uint32_t bucket, flip;       // flip: S is the probe side (rhjoin.c:86: R probes when histR >= histS)
    uint32_t pad[2];
};
struct SjBucket {                // one bucket of the join's radix
    uint32_t p0, np;             // its probe side in canonical positions: first position, count (0: inactive)
    uint32_t flip, pad;
};
struct SjExtra { uint32_t bs, off; };                // a further span of a long probe side

struct SjSummary {               // lives behind the PlanSummary in the same device block; zeroed before the plan
    uint32_t bad;                // a build side above SJ_CAP, a bucket too long for one K2 wave, arena overflow: the caller falls back
    uint32_t extra;              // entries of the extra-unit list
    uint32_t max_build;
    uint32_t pad;
    unsigned long long arena_used;
    uint64_t matches;
};

struct SjArgs {
    const Tuple12 *partR, *partS;
    const uint32_t *segsumR, *segsumS, *segbaseR, *segbaseS;
    const uint8_t *sseqR, *sseqS;
    SjDesc        *desc;         // [2^t]
    SjBucket      *bdesc;        // [2^bits]
    SjExtra       *extra;
    const PlanSummary *summary;
    SjSummary     *sj;
    uint8_t       *stash_cnt;    // [nR + nS] matches of the probe tuple (255: the count heads its arena run)
    uint2         *stash_row;    // [nR + nS] {build row id of the only match | arena offset of the matches, probe row id}
    uint32_t      *arena;        // build row ids of the tuples with two or more matches, descending build position per tuple
    uint64_t       arena_cap;
    unsigned long long *btotal;  // [2^bits] matches per bucket
    uint64_t      *obase;        // [2^bits] first output position of the bucket
    rhj_result_tuple *out;
    uint64_t       out_capacity;
    uint64_t       nR;
    SubGeom        g;
    uint32_t       max_bucket;   // K2: probe positions one wave may walk
    uint32_t       extra_cap;
};

// one thread per sub-bucket: descriptors of the sub-bucket and (s = 0) of its bucket, further spans of long probe sides
__global__ __launch_bounds__(256) void k_sub_plan(SjArgs a)
{
    const uint32_t S = 1u << a.g.kb;
    const uint32_t bs = blockIdx.x * 256 + threadIdx.x;
    if (bs >= (1u << (a.g.lo + a.g.hb + a.g.kb))) return;
    const uint32_t b = bs >> a.g.kb, s = bs & (S - 1u);
    uint64_t cR = 0, cS = 0;
    for (uint32_t ss = 0; ss < S; ++ss) {
        const uint32_t i = sub_index(a.g, b, ss);
        cR += a.segsumR[i]; cS += a.segsumS[i];
    }
    const bool active = cR != 0 && cS != 0, flip = cR < cS;                  // rhjoin.c:82, :86
    const uint32_t i = sub_index(a.g, b, s);
    SjDesc d;
    d.build_off = flip ? a.segbaseR[i] : a.segbaseS[i];
    d.bc = flip ? a.segsumR[i] : a.segsumS[i];
    d.probe_off = flip ? a.segbaseS[i] : a.segbaseR[i];
    d.pc = active ? (flip ? a.segsumS[i] : a.segsumR[i]) : 0u;
    d.bucket = b; d.flip = flip ? 1u : 0u; d.pad[0] = d.pad[1] = 0;
    a.desc[bs] = d;
    if (d.pc != 0) {
        atomicMax(&a.sj->max_build, d.bc);
        if (d.bc > SJ_CAP) atomicOr(&a.sj->bad, 1u);
        const uint32_t nx = (d.pc - 1u) / SJ_SPAN;
        if (nx) {
            const uint32_t at = atomicAdd(&a.sj->extra, nx);
            for (uint32_t c = 0; c < nx; ++c)
                if (at + c < a.extra_cap) a.extra[at + c] = SjExtra{bs, (c + 1u) * SJ_SPAN};
            if (at + nx > a.extra_cap) atomicOr(&a.sj->bad, 2u);
        }
    }
    if (s == 0) {
        SjBucket q;
        q.p0 = d.probe_off; q.np = active ? (uint32_t)(flip ? cS : cR) : 0u; q.flip = d.flip; q.pad = 0;
        a.bdesc[b] = q;
        if (q.np > a.max_bucket) atomicOr(&a.sj->bad, 4u);
    }
}

__device__ __forceinline__ bool sj_usable(const SjArgs &a)
{
    return a.sj->bad == 0 && a.summary->wide_row_ids == 0 && a.summary->row_id_overflow == 0;
}

// ---- K1: join of one sub-bucket ---------------------------------------------------------------------------------------
// LDS index of the sub-bucket's build side (CSR by hash slot, as FjIndex):
//   ent[p]   ((tag19 << 13 | position) + 1), the entries of one slot contiguous and DESCENDING: equal keys have equal
//            tags, so the positions of one key come out descending — the order in which the reference's chain hands
//            out the matches of a key (CreateIndex walks last->first and appends at the tail, rhjoin.c:219-250).
//            tag19 = key bits [t, t + 19), t = lo + hb + kb >= 13 the bits all tuples of the sub-bucket share
//   khi[i]   key bits [32, 64) of build tuple i;  tag and khi together are every key bit the sub-bucket leaves open:
//            a candidate is verified in LDS, exactly, without touching the build tuple in memory
//   rid[i]   its row id (below 2^32 on this path)
//   H[s + 1] 16-bit start of slot s in ent[]; one slot per two build tuples
struct SjIndex {
    uint32_t *ent, *khi, *rid, *dirw;
    uint32_t  hs;
    __device__ __forceinline__ uint32_t H(uint32_t j) const { return reinterpret_cast<const uint16_t *>(dirw)[j]; }
};

__device__ __forceinline__ uint32_t sj_hash(uint64_t x)      // x = the key without its t shared bits
{
    uint32_t f = (uint32_t)x ^ ((uint32_t)(x >> 32) * 0x9e3779b1u);
    f ^= f >> 16; f *= 0x85ebca6bu; f ^= f >> 13; f *= 0xc2b2ae35u; f ^= f >> 16;
    return f;
}

// matches of one probe key in slot order = descending build position per key.  WRITE: their row ids go to dst[0..),
// otherwise the first one is returned in `first`.
template <bool WRITE>
__device__ __forceinline__ uint32_t sj_walk(const SjIndex &X, uint32_t klo, uint32_t khi, int t, uint32_t &first, uint32_t *dst)
{
    const uint64_t x = (((uint64_t)khi << 32) | klo) >> t;
    const uint32_t sl = __umulhi(sj_hash(x), X.hs);
    const uint32_t tag = (uint32_t)x & 0x7ffffu;
    uint32_t st = X.H(sl + 1u);
    const uint32_t en = X.H(sl + 2u);
    uint32_t c = 0;
    for (; st < en; st += 4) {
        uint32_t e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = X.ent[st + j] - 1u;          // pad cells behind the array: 0 - 1 never matches
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (st + j < en && (e[j] >> 13) == tag) {
                const uint32_t p = e[j] & 0x1fffu;
                if (X.khi[p] == khi) {
                    if (WRITE) dst[c] = X.rid[p];
                    else if (c == 0) first = X.rid[p];
                    ++c;
                }
            }
        }
    }
    return c;
}

__global__ __launch_bounds__(SJ_BLOCK) void k_sub_join(SjArgs a, uint32_t nsub)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sj_lds[];
    __shared__ uint32_t wsum[SJ_BLOCK / 64];
    __shared__ uint32_t sh_pick;
    __shared__ unsigned long long sh_chunk;            // first arena entry of the batch's run
    if (!sj_usable(a)) return;
    uint32_t bs = blockIdx.x, off = 0;
    if (bs >= nsub) {
        const uint32_t x = bs - nsub;
        if (x >= a.sj->extra) return;
        const SjExtra e = a.extra[x];
        bs = e.bs; off = e.off;
    }
    const SjDesc d = a.desc[bs];
    if (off >= d.pc) return;
    const uint32_t cnt = min(SJ_SPAN, d.pc - off);
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t = a.g.lo + a.g.hb + a.g.kb;
    const bool flip = d.flip != 0;
    const Tuple12 *bd = (flip ? a.partR : a.partS) + d.build_off;
    const uint32_t bc = d.bc;
    const uint32_t q0 = d.probe_off + off;
    const Tuple12 *pr = (flip ? a.partS : a.partR) + q0;
    uint8_t *scnt = a.stash_cnt + (flip ? a.nR : 0) + q0;
    uint2 *srow = a.stash_row + (flip ? a.nR : 0) + q0;

    SjIndex X;
    X.ent = sj_lds;
    X.khi = X.ent + SJ_CAP + 8;
    X.rid = X.khi + SJ_CAP;
    X.dirw = X.rid + SJ_CAP;
    X.hs = bc < 128u ? 64u : bc / 2u;
    const uint32_t ndw = (X.hs + 3u) / 2u;

    // both sides' first loads go out before anything else: the build tuples, and the first batch of probe tuples,
    // whose latency hides behind the build
    Tuple12 bt[SJ_NB];
#pragma unroll
    for (int j = 0; j < SJ_NB; ++j) {
        const uint32_t i = j * SJ_BLOCK + tid;
        bt[j] = Tuple12{0, 0, 0};
        if (i < bc) bt[j] = bd[i];
    }
    Tuple12 q[SJ_PV];
#pragma unroll
    for (int k = 0; k < SJ_PV; ++k) {
        const uint32_t i = k * SJ_BLOCK + tid;
        q[k] = Tuple12{0, 0, 0};
        if (i < cnt) q[k] = pr[i];
    }
    // ---- build
    for (uint32_t i = tid; i < ndw; i += SJ_BLOCK) X.dirw[i] = 0;
    for (uint32_t i = tid; i < bc + 8u; i += SJ_BLOCK) X.ent[i] = 0;
    __syncthreads();
    uint32_t sw[SJ_NB];                               // slot << 19 | tag of this thread's build tuples
#pragma unroll
    for (int j = 0; j < SJ_NB; ++j) {
        const uint32_t i = j * SJ_BLOCK + tid;
        sw[j] = 0;
        if (i < bc) {
            const uint64_t x = (((uint64_t)bt[j].khi << 32) | bt[j].klo) >> t;
            const uint32_t sl = __umulhi(sj_hash(x), X.hs);
            sw[j] = (sl << 19) | ((uint32_t)x & 0x7ffffu);
            X.khi[i] = bt[j].khi;
            X.rid[i] = bt[j].rid;
            const uint32_t jj = sl + 1u;
            atomicAdd(&X.dirw[jj >> 1], (jj & 1u) ? 0x10000u : 1u);
        }
    }
    __syncthreads();
    {   // exclusive scan over the halfwords: H[s + 1] = start of slot s, H[hs + 1] = bc
        const uint32_t chunk = (ndw + SJ_BLOCK - 1u) / SJ_BLOCK;
        const uint32_t lo_ = min(tid * chunk, ndw), hi_ = min(lo_ + chunk, ndw);
        uint32_t sum = 0;
        for (uint32_t i = lo_; i < hi_; ++i) { const uint32_t v = X.dirw[i]; sum += (v & 0xffffu) + (v >> 16); }
        uint32_t tot;
        uint32_t run = wave_excl_scan_u32(sum, &tot);
        if (lane == 0) wsum[w] = tot;
        __syncthreads();
        for (uint32_t i = 0; i < w; ++i) run += wsum[i];
        for (uint32_t i = lo_; i < hi_; ++i) {
            const uint32_t v = X.dirw[i];
            const uint32_t a0 = run; run += v & 0xffffu;
            const uint32_t a1 = run; run += v >> 16;
            X.dirw[i] = a0 | (a1 << 16);
        }
    }
    __syncthreads();
    // fill: ordered insertion into the slot's range (atomicMax on the cell, go on with the smaller value; exactly n
    // values enter n cells).  Long slots (many duplicates of one key) take places in arrival order — the range's last
    // cell counts the arrivals until the last arrival overwrites it — and are ranked afterwards.
    bool has_long = false;
#pragma unroll
    for (int j = 0; j < SJ_NB; ++j) {
        const uint32_t i = j * SJ_BLOCK + tid;
        if (i < bc) {
            const uint32_t sl = sw[j] >> 19;
            uint32_t v = (((sw[j] & 0x7ffffu) << 13) | i) + 1u;
            const uint32_t st = X.H(sl + 1u), n = X.H(sl + 2u) - st;
            if (n <= SJ_LONG) {
                for (uint32_t p = st;; ++p) {
                    const uint32_t old = atomicMax(&X.ent[p], v);
                    if (old == 0) break;
                    v = min(old, v);
                }
            } else {
                has_long = true;
                const uint32_t arrival = atomicAdd(&X.ent[st + n - 1u], 1u);
                X.ent[st + arrival] = v;              // arrival n - 1: everybody has counted, the counter cell is free
            }
        }
    }
    if (__syncthreads_or(has_long)) {
        for (uint32_t next = 0;;) {                   // long slots one at a time, ranked by the whole workgroup, in place
            if (tid == 0) sh_pick = 0xffffffffu;
            __syncthreads();
            for (uint32_t sl = tid; sl < X.hs; sl += SJ_BLOCK)
                if (sl >= next && X.H(sl + 2u) - X.H(sl + 1u) > SJ_LONG) { atomicMin(&sh_pick, sl); break; }
            __syncthreads();
            const uint32_t pick = sh_pick;
            if (pick == 0xffffffffu) break;
            const uint32_t st = X.H(pick + 1u), n = X.H(pick + 2u) - st;
            uint32_t v[SJ_NB], rk[SJ_NB];
#pragma unroll
            for (int j = 0; j < SJ_NB; ++j) {
                const uint32_t i = j * SJ_BLOCK + tid;
                v[j] = 0; rk[j] = 0;
                if (i < n) {
                    v[j] = X.ent[st + i];
                    for (uint32_t jj = 0; jj < n; ++jj) rk[j] += X.ent[st + jj] > v[j];
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SJ_NB; ++j)
                if ((uint32_t)j * SJ_BLOCK + tid < n) X.ent[st + rk[j]] = v[j];
            next = pick + 1u;
            __syncthreads();
        }
    }

    // ---- probe: the unit's probe tuples, SJ_PV per thread and batch.  The stash is written in the order of the
    // sub-split array (coalesced); k_sub_emit reads it back in canonical order.  A tuple with two or more matches
    // leaves all of them in the arena, and walks its slot a second time to write the row ids.
    uint32_t mine = 0;
    for (uint32_t t0 = 0; t0 < cnt; t0 += SJ_BLOCK * SJ_PV) {
        Tuple12 nq[SJ_PV];
#pragma unroll
        for (int k = 0; k < SJ_PV; ++k) {             // next batch in flight while this one is probed
            const uint32_t i = t0 + SJ_BLOCK * SJ_PV + k * SJ_BLOCK + tid;
            nq[k] = Tuple12{0, 0, 0};
            if (i < cnt) nq[k] = pr[i];
        }
        uint32_t c[SJ_PV], first[SJ_PV], need = 0;
#pragma unroll
        for (int k = 0; k < SJ_PV; ++k) {
            const uint32_t i = t0 + k * SJ_BLOCK + tid;
            c[k] = 0; first[k] = 0;
            if (i < cnt) c[k] = sj_walk<false>(X, q[k].klo, q[k].khi, t, first[k], nullptr);
            mine += c[k];
            need += c[k] >= 2u ? c[k] + (c[k] >= 255u ? 1u : 0u) : 0u;
        }
#if SJ_ABL != 1
        // a batch with such tuples takes one run of the arena: block scan of the needs, one device atomic
        if (__syncthreads_or(need != 0)) {
            uint32_t wtot;
            uint32_t ex = wave_excl_scan_u32(need, &wtot);
            if (lane == 0) wsum[w] = wtot;
            __syncthreads();
            uint32_t btot = 0;
#pragma unroll
            for (int i = 0; i < SJ_BLOCK / 64; ++i) { if ((uint32_t)i < w) ex += wsum[i]; btot += wsum[i]; }
            if (tid == 0) sh_chunk = atomicAdd(&a.sj->arena_used, (unsigned long long)btot);
            __syncthreads();
            uint64_t at = sh_chunk + ex;
#pragma unroll
            for (int k = 0; k < SJ_PV; ++k) {
                if (c[k] >= 2u) {
                    const uint32_t n = c[k] + (c[k] >= 255u ? 1u : 0u);
                    if (at + n <= a.arena_cap) {
                        uint32_t *dst = a.arena + at;
                        if (c[k] >= 255u) *dst++ = c[k];
                        uint32_t dummy;
                        sj_walk<true>(X, q[k].klo, q[k].khi, t, dummy, dst);
                    }
                    first[k] = (uint32_t)at;          // arena_cap <= 2^32 entries; an overflow makes the caller fall back
                    at += n;
                }
            }
        }
#endif
#pragma unroll
        for (int k = 0; k < SJ_PV; ++k) {
            const uint32_t i = t0 + k * SJ_BLOCK + tid;
            if (i < cnt) {
                scnt[i] = (uint8_t)min(c[k], 255u);
                srow[i] = make_uint2(first[k], q[k].rid);
            }
        }
#pragma unroll
        for (int k = 0; k < SJ_PV; ++k) q[k] = nq[k];
    }
    {
        uint32_t tot;
        wave_excl_scan_u32(mine, &tot);
        if (lane == 0 && tot) atomicAdd(&a.btotal[d.bucket], (unsigned long long)tot);
    }
}

// matches per bucket -> first output position of each bucket, total
__global__ __launch_bounds__(1024) void k_sub_bscan(SjArgs a)
{
    __shared__ uint64_t sm[1024 / 64 + 1];
    const uint32_t bins = 1u << (a.g.lo + a.g.hb);
    const uint32_t per = (bins + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, bins);
    uint64_t mine = 0;
    for (uint32_t b = b0; b < b1; ++b) mine += a.btotal[b];
    uint64_t tot;
    uint64_t base = block_excl_scan<1024>(mine, &tot, sm);
    for (uint32_t b = b0; b < b1; ++b) { a.obase[b] = base; base += a.btotal[b]; }
    if (threadIdx.x == 0) {
        if (a.sj->arena_used > a.arena_cap) atomicOr(&a.sj->bad, 8u);
        a.sj->matches = sj_usable(a) ? tot : 0;
    }
}

// ---- K2: pairs in canonical order, one workgroup per bucket ----------------------------------------------------------
// The bucket's canonical positions are walked 2048 at a time (4 waves x 8 rounds of 64).  Position p came from
// sub-bucket s = sseq[p] and is that sub-bucket's next tuple: k ballots per round give its rank among the round's
// positions of the same s, one register per wave (lane s) counts the rounds before, a 4 x 2^k table in LDS the waves
// before, and a running counter (lane s again) everything before this step.  The stash is then read at those places —
// 8 gathers per lane in flight — and the pairs leave in canonical order behind the bucket's first output position.
constexpr int SE_BLOCK = 256;
constexpr int SE_WAVES = SE_BLOCK / WAVE;
#ifndef SE_ROUNDS
#define SE_ROUNDS 4
#endif
#ifndef SE_PREFETCH
#define SE_PREFETCH 1
#endif
constexpr int SE_R = SE_ROUNDS;                       // rounds per wave and step
constexpr uint32_t SE_STEP = SE_WAVES * SE_R * WAVE;  // 2048 positions

__device__ __forceinline__ uint4 sj_pair(bool flip, uint32_t probe_rid, uint32_t build_rid)
{
    return flip ? make_uint4(build_rid, 0u, probe_rid, 0u) : make_uint4(probe_rid, 0u, build_rid, 0u);   // (row_idR, row_idS), rhjoin.c:169-178
}

__global__ __launch_bounds__(SE_BLOCK) void k_sub_emit(SjArgs a)
{
    __shared__ uint32_t s_cnt[2][SE_WAVES][32];
    __shared__ uint32_t s_tot[SE_WAVES];
    if (!sj_usable(a)) return;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t S = 1u << a.g.kb;
    const uint32_t b = blockIdx.x;
    const SjBucket bd = a.bdesc[b];
    const uint32_t np = bd.np;
    if (np == 0) return;
    const bool flip = bd.flip != 0;
    const uint8_t *sseq = (flip ? a.sseqS : a.sseqR) + bd.p0;
    const uint8_t *scnt = a.stash_cnt + (flip ? a.nR : 0);
    const uint2 *srow = a.stash_row + (flip ? a.nR : 0);
    uint32_t carry = lane < S ? a.desc[(b << a.g.kb) | lane].probe_off : 0;   // lane s: next tuple of sub-bucket s
    uint4 *out = reinterpret_cast<uint4 *>(a.out);
    const uint64_t cap = a.out_capacity;
    const uint64_t lt = lanemask_lt();
    uint64_t at = a.obase[b];

    uint32_t sq[SE_R];
#pragma unroll
    for (int r = 0; r < SE_R; ++r) {
        const uint32_t p = w * (SE_R * WAVE) + lane + r * WAVE;
        sq[r] = p < np ? sseq[p] : 0xffu;
    }
    for (uint32_t base = 0, it = 0; base < np; base += SE_STEP, ++it) {
        uint32_t nsq[SE_R];
#pragma unroll
        for (int r = 0; r < SE_R; ++r) {              // the next step's bytes are in flight during this one
            const uint32_t p = base + SE_STEP + w * (SE_R * WAVE) + lane + r * WAVE;
            nsq[r] = 0xffu;
            if (SE_PREFETCH) nsq[r] = p < np ? sseq[p] : 0xffu;
        }
        uint32_t q[SE_R], wcnt = 0;                   // lane s: positions of sub-bucket s in this wave's earlier rounds
#pragma unroll
        for (int r = 0; r < SE_R; ++r) {
            const bool ok = sq[r] != 0xffu;
            uint64_t peers = __ballot(ok), mineS = peers;         // mineS: the positions whose sub-bucket is this LANE's number
            for (int bit = 0; bit < a.g.kb; ++bit) {
                const uint64_t m = __ballot(ok && ((sq[r] >> bit) & 1u));
                peers &= ((sq[r] >> bit) & 1u) ? m : ~m;
                mineS &= ((lane >> bit) & 1u) ? m : ~m;
            }
            q[r] = __shfl(wcnt, ok ? (int)sq[r] : 0, 64) + (uint32_t)__popcll(peers & lt);
            wcnt += (uint32_t)__popcll(mineS);
        }
        if (lane < S) s_cnt[it & 1][w][lane] = wcnt;
        __syncthreads();
        uint32_t mybase = carry;
        if (lane < S) {
#pragma unroll
            for (int ww = 0; ww < SE_WAVES; ++ww) {
                const uint32_t v = s_cnt[it & 1][ww][lane];
                if ((uint32_t)ww < w) mybase += v;
                carry += v;
            }
        }
        uint32_t c[SE_R], rx[SE_R], ry[SE_R];         // matches, {build row id | arena offset}, probe row id
#pragma unroll
        for (int r = 0; r < SE_R; ++r) {
            const bool ok = sq[r] != 0xffu;
            q[r] += __shfl(mybase, ok ? (int)sq[r] : 0, 64);
            c[r] = ok ? scnt[q[r]] : 0;
            const uint2 row = ok ? srow[q[r]] : make_uint2(0, 0);
            rx[r] = row.x; ry[r] = row.y;
        }
        uint32_t tot[SE_R], wtot = 0;                 // matches of each round of this wave
#pragma unroll
        for (int r = 0; r < SE_R; ++r) {
            if (__ballot(c[r] == 255u) != 0) {          // the count heads the arena run
                const bool big = c[r] == 255u;
                const uint32_t real = big ? a.arena[rx[r]] : c[r];
                rx[r] += big ? 1u : 0u;
                c[r] = real;
            }
            if (__ballot(c[r] > 1u) == 0) tot[r] = (uint32_t)__popcll(__ballot(c[r] != 0));
            else { uint32_t t; wave_excl_scan_u32(c[r], &t); tot[r] = t; }
            wtot += tot[r];
        }
        if (lane == 0) s_tot[w] = wtot;
        __syncthreads();
        uint64_t wat = at;
#pragma unroll
        for (int ww = 0; ww < SE_WAVES; ++ww) {
            const uint32_t v = s_tot[ww];
            if ((uint32_t)ww < w) wat += v;
            at += v;
        }
#pragma unroll
        for (int r = 0; r < SE_R; ++r) {
            if (tot[r] != 0 && __ballot(c[r] > 1u) == 0) {           // zero or one match per tuple: offsets from one ballot
                const uint64_t mm = __ballot(c[r] != 0);
                const uint64_t dst = wat + (uint32_t)__popcll(mm & lt);
                if (c[r] != 0 && dst < cap) out[dst] = sj_pair(flip, ry[r], rx[r]);
            } else if (tot[r] != 0) {
                // some tuple has several matches: the lanes turn to the OUTPUT positions of this round, 64 at a time;
                // position o belongs to the tuple whose inclusive prefix is the first above o (binary search over the
                // lanes) and is match o - (its exclusive prefix) of that tuple's arena run.  (Fetching the first four
                // matches of every multi-match tuple of all rounds up front, so that this loop waits for memory once,
                // measured 1.13 ms against 1.01 ms on C3: 17 more VGPRs, nothing gained.)
                uint32_t t;
                const uint32_t excl = wave_excl_scan_u32(c[r], &t);
                const uint32_t incl = excl + c[r];
                for (uint32_t o0 = 0; o0 < t; o0 += WAVE) {
                    const uint32_t o = o0 + lane;
                    uint32_t lo_ = 0;
#pragma unroll
                    for (int stp = 32; stp >= 1; stp >>= 1) {
                        const uint32_t v = __shfl(incl, (int)(lo_ + stp - 1u), 64);
                        if (v <= o) lo_ += stp;
                    }
                    const int src = (int)min(lo_, 63u);
                    const uint32_t ci = __shfl(c[r], src, 64), xi = __shfl(rx[r], src, 64), pi = __shfl(ry[r], src, 64), ei = __shfl(excl, src, 64);
                    if (o < t) {
                        const uint32_t brid = ci == 1u ? xi : a.arena[xi + (o - ei)];
                        const uint64_t dst = wat + o;
                        if (dst < cap) out[dst] = sj_pair(flip, pi, brid);
                    }
                }
            }
            wat += tot[r];
        }
#pragma unroll
        for (int r = 0; r < SE_R; ++r) {
            if (SE_PREFETCH) sq[r] = nsq[r];
            else { const uint32_t p = base + SE_STEP + w * (SE_R * WAVE) + lane + r * WAVE; sq[r] = p < np ? sseq[p] : 0xffu; }
        }
    }
}

}  // namespace rhj
