// rhj_lowradix.hip.h — a join on FEW radix bits over inputs too big for them: canonical order kept, work done on finer buckets
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
//
// The reference ships N_LSB 4 (structs.h:11): 16 buckets.  The result order is a function of that width — bucket ascending,
// inside a bucket the probe side (R when histR >= histS, rhjoin.c:86) in input order, per probe tuple the matches in
// descending build position — so the library cannot silently join on 12 bits instead.  But a 100 M x 100 M
// join on 4 bits has buckets of 6 M tuples, far beyond the LDS index: the tiled path with hash tables in HBM took 21 ms.
//
// Here the join RUNS on r + k bits and EMITS in the order of r bits:
//   partition   two passes in run form with pass 1 on exactly the caller's r bits and pass 2 on the next k bits (k <= 8):
//               bucket (s, b) of the r + k-bit partition is sub-bucket s of the caller's bucket b, and — because pass 1 is
//               tile-local and stable — the sequence in which pass 2 READS the tuples of bucket b (tile by tile, each
//               tile's run of digit b) is exactly the canonical order of bucket b's tuples;
//   join        the fused kernel on the r + k-bit buckets (k_join_fused<false, true>), with the probe side of every sub-bucket
//               chosen by ITS CALLER'S bucket (JoinArgs::parent_flip) — equal keys share a sub-bucket, so the matches of a probe
//               tuple and their order are those of the r-bit join; its pairs go to a scratch list in (s, b) order, and a
//               tuple with several matches leaves the place of its pairs there in its stash row;
//   emit        k_lr_emit REPLAYS pass 2 over the probe side of each bucket b, a pass-2 tile ("chunk") per workgroup: the same run
//               tables, digit bytes and stable ranks say where every tuple of the canonical sequence went (sub-bucket s,
//               position q), its stash entry there says what it matched, and the pairs are written bucket by bucket in
//               canonical order — every chunk at the exclusive prefix of the chunks' match totals (k_lr_totals, k_offsets_*).
// What is refused (the host then takes the tiled path): row ids of 2^32 or more, a unit the fused kernel hands to k_join_walk
// (a probe tuple with more than 16 matches), 2^32 pairs.
#pragma once
#include "rhj_common.hip.h"
#include "rhj_partition.hip.h"
#include "rhj_join_tiled.hip.h"
#include "rhj_join_fused.hip.h"

namespace rhj {

struct LrArgs {
    JoinArgs        j;               // hist / psum of the r + k-bit partition, parent_flip / parent_mask, out, out_capacity
    RelArgs         p2R, p2S;        // both relations as pass 2 saw them: runs, digit bytes, scanned tile counts
    const uint8_t  *stash_cnt;       // [nR + nS] match count per probe tuple of the internal join (0 where no build side exists)
    const uint2    *stash_row;       // [nR + nS] {build row id of the only match | place of the tuple's pairs in tmp, probe row id}
    const uint4    *tmp;             // the internal join's pairs, in (s, b) order
    uint64_t       *ctotal;          // [slots] pairs of every chunk (k_lr_totals), then their exclusive prefix (k_offsets_*)
    uint32_t       *bad;             // (spare: nothing raises it)
    uint64_t        nR;
    uint32_t        r_bits, k_bits;
    uint32_t        slots_per_bucket;   // max(groups of R, groups of S): chunk slot = bucket * slots_per_bucket + chunk
    uint32_t        search0;         // first step of the run search (largest power of two <= group)
};

// parent_flip[b] = 1 when S probes in the caller's bucket b (histR < histS over all its sub-buckets, rhjoin.c:86);
// parent_flip[2^r + b] = 1 when the bucket takes part at all (rhjoin.c:82).  One workgroup per bucket.
__global__ __launch_bounds__(256) void k_lr_parent(const uint64_t *histR, const uint64_t *histS, int r_bits, int k_bits, uint8_t *parent_flip)
{
    __shared__ uint64_t sm[256 / 64 + 1];
    const uint32_t b = blockIdx.x;
    uint64_t cR = 0, cS = 0;
    for (uint32_t s = threadIdx.x; s < (1u << k_bits); s += 256) { cR += histR[(s << r_bits) | b]; cS += histS[(s << r_bits) | b]; }
    uint64_t tR, tS;
    block_excl_scan<256>(cR, &tR, sm);
    block_excl_scan<256>(cS, &tS, sm);
    if (threadIdx.x == 0) {
        parent_flip[b] = tR < tS ? 1 : 0;
        parent_flip[(1u << r_bits) + b] = (tR != 0 && tS != 0) ? 1 : 0;
    }
}

// A chunk = one pass-2 tile (pass-1 digit b, tile group j) of the probe relation of the caller's bucket b: the tuples pass 2
// read there are a stretch of bucket b's canonical sequence.  Pass 2 sent those of sub-bucket s to positions
// [first(s), first(s) + n(s)) of the partitioned relation — the tile's scanned counts say where.
struct LrChunk {
    const RelArgs *r;
    uint32_t tile2;
    bool     flip, active;
};
__device__ __forceinline__ LrChunk lr_chunk(const LrArgs &a, uint32_t slot)
{
    LrChunk c;
    const uint32_t b = slot / a.slots_per_bucket, chunk = slot % a.slots_per_bucket;
    c.flip = a.j.parent_flip[b] != 0;
    c.r = c.flip ? &a.p2S : &a.p2R;
    c.active = a.j.parent_flip[(1u << a.r_bits) + b] != 0 && chunk < c.r->groups;
    c.tile2 = b * c.r->groups + chunk;
    return c;
}
// sub-bucket s of the chunk: first position and number of its tuples in the partitioned probe relation
__device__ __forceinline__ void lr_stream(const LrArgs &a, const LrChunk &c, uint32_t slot, uint32_t s, uint32_t &first, uint32_t &n)
{
    const uint32_t bins = 1u << a.k_bits;
    const uint32_t b = slot / a.slots_per_bucket, chunk = slot % a.slots_per_bucket;
    first = pt_start(*c.r, c.tile2, s, bins);
    uint32_t end;
    if (chunk + 1u < c.r->groups) end = pt_start(*c.r, c.tile2 + 1u, s, bins);
    else {                                           // the bucket's last chunk: up to the end of sub-bucket (s, b)
        const uint32_t sub = (s << a.r_bits) | b;
        end = (uint32_t)((c.flip ? a.j.psumS[sub] : a.j.psumR[sub]) + (c.flip ? a.j.histS[sub] : a.j.histR[sub]));
    }
    n = end - first;
}

// ctotal[slot] = pairs the chunk's tuples produced in the internal join (sum of their count bytes).  Sixteen lanes a
// sub-bucket: its bytes are consecutive, so a wave load touches four lines (a lane per sub-bucket touched sixty-four and
// the kernel ran at the L2's request rate: 0.45 ms for 100 M tuples).
__global__ __launch_bounds__(256) void k_lr_totals(LrArgs a)
{
    __shared__ uint64_t sm[256 / 64 + 1];
    __shared__ uint32_t s_first[1u << PT_MAX_BITS], s_n[1u << PT_MAX_BITS];
    const uint32_t slot = blockIdx.x;
    const LrChunk c = lr_chunk(a, slot);
    if (!c.active) { if (threadIdx.x == 0) a.ctotal[slot] = 0; return; }
    const uint32_t bins = 1u << a.k_bits;
    for (uint32_t s = threadIdx.x; s < bins; s += 256) lr_stream(a, c, slot, s, s_first[s], s_n[s]);
    __syncthreads();
    const uint8_t *cnt = a.stash_cnt + (c.flip ? a.nR : 0);
    uint64_t sum = 0;
    for (uint32_t s0 = 0; s0 < bins; s0 += 64) {      // four passes of 16 sub-buckets each in flight per thread
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t s = s0 + q * 16 + (threadIdx.x >> 4);
            v[q] = 0;
            if (s < bins) {
                const uint32_t f = s_first[s], n = s_n[s];
                for (uint32_t i = threadIdx.x & 15u; i < n; i += 16) v[q] += cnt[f + i] & 0x7fu;
            }
        }
        sum += (uint64_t)v[0] + v[1] + v[2] + v[3];
    }
    uint64_t tot;
    block_excl_scan<256>(sum, &tot, sm);
    if (threadIdx.x == 0) a.ctotal[slot] = tot;
}

constexpr int LR_BLOCK = PT_BLOCK;                // the geometry of pass 2: 512 threads x 8 elements = one batch
constexpr int LR_V = SR_V;
__host__ __device__ constexpr size_t lr_lds_bytes(int k_bits)
{
    return (size_t)SR_TILE * 9 + ((size_t)PT_WAVES + 3) * ((size_t)1 << k_bits) * 4 + (LR_BLOCK / 64 + 2) * 8 + (SR_RUNOFF + PT_MAX_GROUP) * 4 + 64;
}

// The pairs of one chunk per workgroup, at ctotal[slot] (by now the exclusive prefix over the chunks in canonical order).
// A chunk is walked in batches of 4096 elements like a tile of pass 2 (on uniform keys: one batch; the tiles of a hot key's
// bucket are longer), the positions of the batch's first tuple of every sub-bucket carried from batch to batch.
__global__ __launch_bounds__(LR_BLOCK, 4) void k_lr_emit(LrArgs a, uint32_t nslots)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t bins = 1u << a.k_bits;
    uint2    *l_row = reinterpret_cast<uint2 *>(smem);                         // [SR_TILE] stash rows of the batch's tuples, sub-bucket by sub-bucket
    uint8_t  *l_cnt = reinterpret_cast<uint8_t *>(l_row + SR_TILE);            // [SR_TILE] their match counts
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(l_cnt + SR_TILE);            // [PT_WAVES][bins] per-wave digit counts -> prefixes over the waves
    uint32_t *lstart = wcnt + PT_WAVES * bins;                                 // [bins] first LDS slot of sub-bucket s in this batch
    uint32_t *first = lstart + bins;                                           // [bins] position of the batch's first tuple of s in the partitioned relation
    uint32_t *nb = first + bins;                                               // [bins] tuples of s in this batch
    uint64_t *sm = reinterpret_cast<uint64_t *>(nb + bins);                    // scan scratch
    uint32_t *runoff = reinterpret_cast<uint32_t *>(sm + LR_BLOCK / 64 + 2);   // [SR_RUNOFF]
    uint32_t *rbase = runoff + SR_RUNOFF;                                      // [PT_MAX_GROUP]
    __shared__ uint32_t wsum[PT_WAVES];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    uint4 *out = reinterpret_cast<uint4 *>(a.j.out);
    const uint64_t cap = a.j.out_capacity;

    for (uint32_t slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
        const LrChunk ch = lr_chunk(a, slot);
        if (!ch.active) continue;                     // (workgroup-uniform)
        const RelArgs &r = *ch.r;
        const bool flip = ch.flip;
        const uint64_t sbase = flip ? a.nR : 0;
        __syncthreads();                              // the previous chunk's LDS is no longer read
        // ---- the chunk's run table; where pass 2 sent its tuples, sub-bucket by sub-bucket
        uint32_t total;
        {
            uint32_t phys, len;
            pt_run_of(r, ch.tile2, threadIdx.x, phys, len);
            if (threadIdx.x < bins) { uint32_t f, n; lr_stream(a, ch, slot, threadIdx.x, f, n); first[threadIdx.x] = f; }
            uint64_t tot64;
            const uint32_t off = (uint32_t)block_excl_scan<LR_BLOCK>(len, &tot64, sm);
            total = (uint32_t)tot64;
            if (threadIdx.x < SR_RUNOFF) runoff[threadIdx.x] = threadIdx.x < r.group ? off : total;
            if (threadIdx.x < PT_MAX_GROUP) rbase[threadIdx.x] = phys - off;
        }
        uint64_t at_batch = a.ctotal[slot];           // first output position of the batch
        for (uint32_t sb = 0; sb < total; sb += SR_TILE) {
            const uint32_t count = min((uint32_t)SR_TILE, total - sb);
            for (uint32_t i = threadIdx.x; i < PT_WAVES * bins; i += LR_BLOCK) wcnt[i] = 0;
            __syncthreads();                          // run table / counters / the previous batch's LDS
            // ---- every element of the batch, in canonical order (wave, round, lane): the sub-bucket it went to
            uint32_t sq[LR_V];
            {
                uint32_t pos = 0;
                {
                    const uint32_t e0 = sb + w * (WAVE * LR_V) + (lane & 7u) * WAVE;
                    for (uint32_t s2 = a.search0; s2 >= 1; s2 >>= 1)
                        if (runoff[pos + s2] <= e0) pos += s2;
                }
#pragma unroll
                for (int k = 0; k < LR_V; ++k) {
                    const uint32_t i = w * (WAVE * LR_V) + k * WAVE + lane;
                    const uint32_t e = sb + i;
                    const uint32_t j0 = (uint32_t)__builtin_amdgcn_readlane((int)pos, k);
                    const uint32_t a1 = runoff[j0 + 1], a2 = runoff[j0 + 2], a3 = runoff[j0 + 3], a4 = runoff[j0 + 4];
                    uint32_t j = j0 + (a1 <= e ? 1u : 0u) + (a2 <= e ? 1u : 0u) + (a3 <= e ? 1u : 0u);
                    if (a4 <= sb + w * (WAVE * LR_V) + (uint32_t)k * WAVE + (WAVE - 1)) {
                        j = j0;
                        while (runoff[j + 1] <= e && j + 1 < r.group) ++j;
                    }
                    sq[k] = 0;
                    if (i < count) sq[k] = r.dig_in[rbase[j] + e];
                }
            }
            // ---- its stable rank among the batch's elements of the same sub-bucket: what pass 2 computed (k_scatter_runs)
            uint32_t rk[LR_V];
            uint32_t *mycnt = wcnt + w * bins;
#pragma unroll
            for (int k = 0; k < LR_V; ++k) {
                const bool ok = w * (WAVE * LR_V) + k * WAVE + lane < count;
                const uint64_t peers = digit_peers(sq[k], ok, (int)a.k_bits);
                const uint32_t rank = (uint32_t)__popcll(peers & lt);
                const uint32_t old = mycnt[sq[k]];        // the whole group reads its counter, its lowest lane adds the group
                if (ok && rank == 0) mycnt[sq[k]] = old + (uint32_t)__popcll(peers);
                rk[k] = old + rank;
            }
            __syncthreads();
            uint32_t mine = 0;
            if (threadIdx.x < bins) {                 // per sub-bucket: exclusive prefix over the waves, tuples in this batch
                uint32_t run = 0;
                for (int ww = 0; ww < PT_WAVES; ++ww) {
                    const uint32_t c = wcnt[ww * bins + threadIdx.x];
                    wcnt[ww * bins + threadIdx.x] = run;
                    run += c;
                }
                nb[threadIdx.x] = run;
                mine = run;
            }
            const uint32_t ls = (uint32_t)block_excl_scan<LR_BLOCK>((uint64_t)mine, nullptr, sm);
            if (threadIdx.x < bins) lstart[threadIdx.x] = ls;
            __syncthreads();
            // ---- what the internal join left for these tuples, read stream by stream (sixteen lanes a sub-bucket: its entries
            // are consecutive) into LDS; read tuple by tuple in canonical order they are 64 different lines per wave load
            for (uint32_t s0 = 0; s0 < bins; s0 += LR_BLOCK / 16) {
                const uint32_t s = s0 + (threadIdx.x >> 4);
                if (s < bins) {
                    const uint32_t f = first[s], lsx = lstart[s], n = nb[s];
                    for (uint32_t i = threadIdx.x & 15u; i < n; i += 16) {
                        l_row[lsx + i] = a.stash_row[sbase + f + i];
                        l_cnt[lsx + i] = a.stash_cnt[sbase + f + i];
                    }
                }
            }
            __syncthreads();
            // ---- matches of every element; offsets of the pairs in canonical order
            uint32_t c[LR_V], off[LR_V];
            uint2 row[LR_V];
            uint32_t wrun = 0;
#pragma unroll
            for (int k = 0; k < LR_V; ++k) {
                const bool ok = w * (WAVE * LR_V) + k * WAVE + lane < count;
                c[k] = 0; row[k] = make_uint2(0, 0);
                if (ok) {
                    const uint32_t at = lstart[sq[k]] + mycnt[sq[k]] + rk[k];
                    c[k] = l_cnt[at] & 0x7fu;
                    row[k] = l_row[at];
                }
            }
#pragma unroll
            for (int k = 0; k < LR_V; ++k) {
                uint32_t tot;
                off[k] = wrun + wave_excl_scan_u32(c[k], &tot);
                wrun += tot;
            }
            if (lane == 0) wsum[w] = wrun;
            __syncthreads();
            uint32_t wbase = 0, btotal = 0;
#pragma unroll
            for (int i = 0; i < PT_WAVES; ++i) {
                const uint32_t v = wsum[i];
                if ((uint32_t)i < w) wbase += v;
                btotal += v;
            }
            const uint64_t at0 = at_batch + wbase;
            // ---- the pairs: (row_idR, row_idS), rhjoin.c:169-178; several matches of a tuple are copied from the internal join's list
#pragma unroll
            for (int k = 0; k < LR_V; ++k) {
                const uint64_t at = at0 + off[k];
                if (c[k] == 1u) {
                    if (at < cap) out[at] = make_pair(flip, row[k].y, 0u, row[k].x, 0u);
                } else if (c[k] >= 2u) {
                    for (uint32_t i = 0; i < c[k]; ++i)
                        if (at + i < cap) out[at + i] = a.tmp[(uint64_t)row[k].x + i];
                }
            }
            at_batch += btotal;
            if (threadIdx.x < bins) first[threadIdx.x] += nb[threadIdx.x];     // the next batch's tuples of s follow (read again behind the barrier at the loop's top)
        }
    }
}

}  // namespace rhj
