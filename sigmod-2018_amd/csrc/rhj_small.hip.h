// rhj_small.hip.h — the partition of a small join in two launches, or in one
// (part of the device code of librhj.so; rhj_kernels.hip.h includes all of it)
//
// 1M x 1M at 8 radix bits moves 48 MB: microseconds at HBM speed.  Such a join is bound by the number of dependent
// launches and by what each of them waits for, not by bytes, so the one-pass partition (preprocess.c:302-362) is cut
// to two launches and the plan rides along:
//   k_small_hist     per-tile digit histograms, tiles of 8192 tuples        -> cnt[tile][digit]
//                    (and the words the join kernel expects zeroed: ticket, unit status — no host memset)
//   k_small_scatter  every workgroup sums the digit columns of cnt for itself (a few hundred rows from L2: the tile
//                    starts and the bucket starts in one go — no scan kernel), then the stable LDS-staged scatter of
//                    its tile; one extra workgroup leaves hist / psum and runs the plan meanwhile
// and k_join_fused follows as the third launch; its last workgroup out leaves the match total and the plan summary
// in pinned host memory, so the host only waits for the stream.  Relations of one or two tiles (16 K tuples: most
// joins of the contest's workload) skip k_small_hist: the scatter workgroups count the digits of all the relation's
// tiles themselves (small_selfhist) and clear the join kernel's words.
//
// (A single persistent kernel with grid-wide barriers between these phases was built first and measured slower,
// 0.30 ms against 0.19 ms: on MI355X a grid barrier costs 6 us bare and 20-50 us with the agent-scope release /
// acquire fences that make one phase's stores visible to the other XCDs — tools/micro/gridbar.hip, profiles/README.md.)
#pragma once
#include "rhj_common.hip.h"
#include "rhj_partition.hip.h"
#include "rhj_join_tiled.hip.h"

namespace rhj {

constexpr int SM_BLOCK = 1024;
constexpr int SM_V = 8;
constexpr int SM_TILE = SM_BLOCK * SM_V;          // 8192 tuples = 128 KiB staged in LDS
constexpr int SM_WAVES = SM_BLOCK / WAVE;
constexpr uint32_t SM_MAX_TILES = 512;            // per relation: 4 M tuples
constexpr uint32_t SM_PARTS = 16;                 // column sums: slices of the tile range summed in parallel

#ifdef RHJ_INSTRUMENT
__device__ uint64_t g_sm_dbg[4 * 2048];           // diagnostics build: [kernel][workgroup] start / end stamps (100 MHz)
#define SM_STAMP(kern, slot) do { if (threadIdx.x == 0) g_sm_dbg[((kern) * 1024 + blockIdx.y * gridDim.x + blockIdx.x) * 2 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SM_FINE(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 128) g_sm_dbg[2 * 2048 + blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SM_STAMP(kern, slot) do { } while (0)
#define SM_FINE(slot) do { } while (0)
#endif

__host__ __device__ constexpr size_t small_lds_bytes(int bits)
{
    // [stage SM_TILE x 16 B (column-sum partials before that)][wcnt SM_WAVES x bins][dstart][delta][gstart]
    return (size_t)SM_TILE * 16 + ((size_t)SM_WAVES + 3) * ((size_t)1 << bits) * 4;
}

__global__ __launch_bounds__(SM_BLOCK) void k_small_hist(RelArgs r0, RelArgs r1, int bits, uint64_t *zero_words, uint64_t n_zero)
{
    __shared__ uint32_t tile_h[1u << PT_MAX_BITS];
    const uint32_t tid = threadIdx.x;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    SM_STAMP(0, 0);
    {
        const uint64_t wgs = (uint64_t)gridDim.x * gridDim.y, me = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
        for (uint64_t i = me * SM_BLOCK + tid; i < n_zero; i += wgs * SM_BLOCK) zero_words[i] = 0;
    }
    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t t = blockIdx.x;
    if (t >= r.tiles) return;
    if (tid < bins) tile_h[tid] = 0;
    __syncthreads();
    const uint64_t beg = (uint64_t)t * SM_TILE;
    const uint64_t end = min(beg + (uint64_t)SM_TILE, r.n);
#pragma unroll
    for (int k = 0; k < SM_V; ++k) {
        const uint64_t i = beg + (uint32_t)k * SM_BLOCK + tid;
        if (i < end) atomicAdd(&tile_h[(uint32_t)r.in[i].value & mask], 1u);
    }
    __syncthreads();
    if (tid < bins) r.cnt[(size_t)t * bins + tid] = tile_h[tid];
    SM_STAMP(0, 1);
}

// Sums of a relation's digit columns: for digit d (threads 0..bins-1 get the result) the tuples of digit d in the
// tiles before `t` and in all tiles.  A thread takes four digits of up to SM_ROWS rows of its slice of the tiles with
// 16-byte loads that are all in flight together (the rows were written by the previous kernel on other XCDs: each
// dependent round of loads costs ~2 us, and four rounds of scalar loads were half of this kernel's time).
// part[] = 2 x SM_PARTS x bins words of LDS.
constexpr uint32_t SM_ROWS = 8;
__device__ __forceinline__ void small_colsum(const RelArgs &r, uint32_t t, int bits, uint32_t *part, uint32_t &before, uint32_t &all)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t Q;
    if (bits >= 2) {
        const uint32_t lanes = bins >> 2;                                 // threads per slice, four digits each
        Q = min((uint32_t)SM_BLOCK / lanes, SM_PARTS);
        const uint32_t q = tid / lanes, d4 = (tid % lanes) * 4u;
        if (q < Q) {
            const uint32_t per = (r.tiles + Q - 1u) / Q;
            const uint32_t row0 = min(q * per, r.tiles), row1 = min(row0 + per, r.tiles);
            const uint4 *col = reinterpret_cast<const uint4 *>(r.cnt + d4);
            uint4 bf = make_uint4(0, 0, 0, 0), al = make_uint4(0, 0, 0, 0);
            for (uint32_t rb = row0; rb < row1; rb += SM_ROWS) {
                uint4 v[SM_ROWS];
#pragma unroll
                for (uint32_t j = 0; j < SM_ROWS; ++j) {
                    v[j] = make_uint4(0, 0, 0, 0);
                    if (rb + j < row1) v[j] = col[(size_t)(rb + j) * (bins >> 2)];
                }
#pragma unroll
                for (uint32_t j = 0; j < SM_ROWS; ++j) {
                    al.x += v[j].x; al.y += v[j].y; al.z += v[j].z; al.w += v[j].w;
                    if (rb + j < t) { bf.x += v[j].x; bf.y += v[j].y; bf.z += v[j].z; bf.w += v[j].w; }
                }
            }
            *reinterpret_cast<uint4 *>(part + q * bins + d4) = bf;
            *reinterpret_cast<uint4 *>(part + (Q + q) * bins + d4) = al;
        }
    } else {
        Q = min((uint32_t)SM_BLOCK >> bits, SM_PARTS);
        const uint32_t q = tid >> bits, d = tid & mask;
        if (q < Q) {
            const uint32_t per = (r.tiles + Q - 1u) / Q;
            const uint32_t row0 = min(q * per, r.tiles), row1 = min(row0 + per, r.tiles);
            const uint32_t *col = r.cnt + d;
            uint32_t bf = 0, al = 0;
            for (uint32_t row = row0; row < row1; ++row) {
                const uint32_t v = col[(size_t)row * bins];
                al += v;
                bf += row < t ? v : 0u;
            }
            part[q * bins + d] = bf;
            part[(Q + q) * bins + d] = al;
        }
    }
    __syncthreads();
    before = 0; all = 0;
    if (tid < bins)
        for (uint32_t qq = 0; qq < Q; ++qq) { before += part[qq * bins + tid]; all += part[(Q + qq) * bins + tid]; }
    __syncthreads();
}

// Relations of at most SM_SELF_TILES tiles need no histogram launch: a workgroup counts the digits of ALL the relation's
// tiles itself (16 K tuples, L2-resident after the first workgroup) — `all` — and of the tiles before `t` — `before`.
// h[] = 2 x bins words of LDS.  (The 88 joins of the contest's `small` workload are mostly this size: 57 us -> 48 us each.)
constexpr uint32_t SM_SELF_TILES = 2;
__device__ __forceinline__ void small_selfhist(const RelArgs &r, uint32_t t, int bits, uint32_t *h, uint32_t &before, uint32_t &all)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    if (tid < 2u * bins) h[tid] = 0;
    __syncthreads();
    for (uint32_t tt = 0; tt < r.tiles; ++tt) {
        const uint64_t beg = (uint64_t)tt * SM_TILE;
        const uint64_t end = min(beg + (uint64_t)SM_TILE, r.n);
#pragma unroll
        for (int k = 0; k < SM_V; ++k) {
            const uint64_t i = beg + (uint32_t)k * SM_BLOCK + tid;
            if (i < end) {
                const uint32_t d = (uint32_t)r.in[i].value & mask;
                atomicAdd(&h[d], 1u);
                if (tt < t) atomicAdd(&h[bins + d], 1u);
            }
        }
    }
    __syncthreads();
    before = 0; all = 0;
    if (tid < bins) { all = h[tid]; before = h[bins + tid]; }
    __syncthreads();
}

// grid (max tiles + 1, 2): workgroup (x, rel) scatters tile x of relation rel; workgroup (max tiles, 0) is the plan's.
// self_hist: no k_small_hist launch went before (both relations have at most SM_SELF_TILES tiles): the digit counts are
// taken from the tuples, and this launch clears the join kernel's words.
__global__ __launch_bounds__(SM_BLOCK) void k_small_scatter(RelArgs r0, RelArgs r1, int bits, uint64_t *hist, uint64_t *psum, PlanArgs plan,
                                                            int self_hist, uint64_t *zero_words, uint64_t n_zero)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint64_t sm[SM_BLOCK / 64 + 1];
    __shared__ unsigned long long red[2];
    uint4    *stage = reinterpret_cast<uint4 *>(smem);                                // [SM_TILE]
    uint32_t *part = reinterpret_cast<uint32_t *>(smem);                              // column-sum partials, before the stage is used
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + (size_t)SM_TILE * 16);       // [SM_WAVES][bins]
    uint32_t *dstart = wcnt + (size_t)SM_WAVES * bins;                                // [bins]
    uint32_t *delta = dstart + bins;                                                  // [bins]
    uint32_t *gstart = delta + bins;                                                  // [bins]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    SM_STAMP(1, 0);
    if (self_hist) {
        const uint64_t wgs = (uint64_t)gridDim.x * gridDim.y, me = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
        for (uint64_t i = me * SM_BLOCK + tid; i < n_zero; i += wgs * SM_BLOCK) zero_words[i] = 0;
    }

    if (blockIdx.x == gridDim.x - 1u) {
        // ---- the plan's workgroup: bucket sizes and starts of both relations, then the fused path's plan over them:
        // one unit per span_lds probe tuples of every bucket both sides are present in, probe side = R when
        // cR >= cS (rhjoin.c:86); fused_ok only if every such bucket's build side fits the LDS index (otherwise the
        // host plans again for the tiled path, k_plan)
        if (blockIdx.y != 0) return;
        uint64_t c[2] = {0, 0};
        if (self_hist) {
            for (int rel = 0; rel < 2; ++rel) {
                const RelArgs &r = rel ? r1 : r0;
                uint32_t before, all;
                small_selfhist(r, 0u, bits, part, before, all);
                const uint64_t ex = block_excl_scan<SM_BLOCK>(tid < bins ? (uint64_t)all : 0ull, nullptr, sm);
                if (tid < bins) {
                    hist[(size_t)rel * bins + tid] = all;
                    psum[(size_t)rel * bins + tid] = ex;
                }
                c[rel] = all;
            }
        } else {
            // both relations at once: threads 0..511 sum R's digit columns, 512..1023 S's (loads as in small_colsum)
            const uint32_t rel = tid >> 9, tt = tid & 511u;
            const RelArgs &r = rel ? r1 : r0;
            uint32_t Q;
            if (bits >= 2) {
                const uint32_t lanes = bins >> 2;
                Q = min(512u / lanes, SM_PARTS);
                const uint32_t q = tt / lanes, d4 = (tt % lanes) * 4u;
                if (q < Q) {
                    const uint32_t per = (r.tiles + Q - 1u) / Q;
                    const uint32_t row0 = min(q * per, r.tiles), row1 = min(row0 + per, r.tiles);
                    const uint4 *col = reinterpret_cast<const uint4 *>(r.cnt + d4);
                    uint4 al = make_uint4(0, 0, 0, 0);
                    for (uint32_t rb = row0; rb < row1; rb += 2 * SM_ROWS) {
                        uint4 v[2 * SM_ROWS];
#pragma unroll
                        for (uint32_t j = 0; j < 2 * SM_ROWS; ++j) {
                            v[j] = make_uint4(0, 0, 0, 0);
                            if (rb + j < row1) v[j] = col[(size_t)(rb + j) * (bins >> 2)];
                        }
#pragma unroll
                        for (uint32_t j = 0; j < 2 * SM_ROWS; ++j) { al.x += v[j].x; al.y += v[j].y; al.z += v[j].z; al.w += v[j].w; }
                    }
                    *reinterpret_cast<uint4 *>(part + (rel * Q + q) * bins + d4) = al;
                }
            } else {
                Q = min(512u >> bits, SM_PARTS);
                const uint32_t q = tt >> bits, d = tt & mask;
                if (q < Q) {
                    const uint32_t per = (r.tiles + Q - 1u) / Q;
                    const uint32_t row0 = min(q * per, r.tiles), row1 = min(row0 + per, r.tiles);
                    const uint32_t *col = r.cnt + d;
                    uint32_t al = 0;
                    for (uint32_t row = row0; row < row1; ++row) al += col[(size_t)row * bins];
                    part[(rel * Q + q) * bins + d] = al;
                }
            }
            __syncthreads();
            uint64_t tot = 0;
            if (tt < bins)
                for (uint32_t qq = 0; qq < Q; ++qq) tot += part[(rel * Q + qq) * bins + tt];
            // one scan over R's digits followed by S's: S's own prefix = that - all of R
            const uint64_t ex = block_excl_scan<SM_BLOCK>(tt < bins ? tot : 0ull, nullptr, sm);
            uint32_t *both = part + 2u * SM_PARTS * bins;                 // [2][bins] behind the partials
            if (tt < bins) {
                hist[(size_t)rel * bins + tt] = tot;
                psum[(size_t)rel * bins + tt] = rel ? ex - r0.n : ex;
                both[rel * bins + tt] = (uint32_t)tot;
            }
            __syncthreads();
            if (tid < bins) { c[0] = both[tid]; c[1] = both[bins + tid]; }
        }
        const bool active = tid < bins && c[0] != 0 && c[1] != 0;
        const uint64_t pc = c[0] >= c[1] ? c[0] : c[1], bc = c[0] >= c[1] ? c[1] : c[0];
        const uint64_t span = plan.span_lds;
        const uint64_t nu = active ? (pc + span - 1) / span : 0;
        if (tid < 2) red[tid] = 0;
        uint64_t tot_u;
        uint64_t ubase = block_excl_scan<SM_BLOCK>(nu, &tot_u, sm);       // (its barriers publish red's zeroes)
        if (active) atomicMax(&red[0], (unsigned long long)bc);
        const int too_big = __syncthreads_or(active && bc > plan.lds_cap);
        if (active)
            for (uint64_t o = 0; o < pc; o += span) {
                Unit u; u.off = o; u.bucket = tid; u.count = (uint32_t)min(span, pc - o);
                plan.units[ubase++] = u;
            }
        if (tid == 0) {
            PlanSummary s;
            s.units = tot_u; s.build_units = too_big ? 1 : 0; s.hbm_slots = 0; s.lds_buckets = 0;
            s.tab32_slots = 0; s.max_lds_slots = 0; s.max_build = red[0]; s.matches = 0;
            s.fused_ok = too_big ? 0 : 1;
            s.wide_row_ids = 1; s.row_id_overflow = 0;                    // one pass: 16-byte tuples throughout
            *plan.summary = s;
        }
        SM_STAMP(1, 1);
        return;
    }

    const RelArgs &r = blockIdx.y ? r1 : r0;
    const uint32_t t = blockIdx.x;
    if (t >= r.tiles) return;
    const uint64_t lt = lanemask_lt();
    const uint64_t beg = (uint64_t)t * SM_TILE;
    const uint32_t count = (uint32_t)min((uint64_t)SM_TILE, r.n - beg);
    const uint4 *in = reinterpret_cast<const uint4 *>(r.in) + beg;
    SM_FINE(0);
    uint4 tp[SM_V];                                     // the tile's loads fly while the columns are summed
    bool ok[SM_V];
#pragma unroll
    for (int k = 0; k < SM_V; ++k) {
        const uint32_t i = w * (WAVE * SM_V) + (uint32_t)k * WAVE + lane;             // tile order (wave, round, lane)
        ok[k] = i < count;
        tp[k] = make_uint4(0, 0, 0, 0);
        if (ok[k]) tp[k] = in[i];
    }
    {
        // where this tile's tuples of digit d go: bucket start + the digit's tuples in earlier tiles
        uint32_t before, all;
        if (self_hist) small_selfhist(r, t, bits, part, before, all);
        else           small_colsum(r, t, bits, part, before, all);
        const uint64_t ex = block_excl_scan<SM_BLOCK>(tid < bins ? (uint64_t)all : 0ull, nullptr, sm);
        if (tid < bins) gstart[tid] = (uint32_t)ex + before;
    }
    SM_FINE(1);
    for (uint32_t i = tid; i < (uint32_t)SM_WAVES * bins; i += SM_BLOCK) wcnt[i] = 0;
    __syncthreads();
    SM_FINE(2);
    // stable rank of a tuple inside its digit: earlier waves + earlier rounds of this wave + lower lanes of this
    // round, from one match-any per round and per-wave LDS counters (no atomics: k_scatter_lds, rhj_partition.hip.h)
    uint32_t lrank[SM_V], dig[SM_V];
    uint32_t *mycnt = wcnt + w * bins;
#pragma unroll
    for (int k = 0; k < SM_V; ++k) {
        const uint32_t d = tp[k].x & mask;
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
    SM_FINE(3);
    uint64_t mytotal = 0;                               // per digit: exclusive prefix over waves, digit totals
    if (tid < bins) {
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < SM_WAVES; ++ww) {
            const uint32_t c = wcnt[ww * bins + tid];
            wcnt[ww * bins + tid] = run;
            run += c;
        }
        mytotal = run;
    }
    const uint64_t ds = block_excl_scan<SM_BLOCK>(mytotal, nullptr, sm);
    if (tid < bins) {
        dstart[tid] = (uint32_t)ds;
        delta[tid] = gstart[tid] - (uint32_t)ds;        // mod 2^32
    }
    __syncthreads();
    SM_FINE(4);
#pragma unroll
    for (int k = 0; k < SM_V; ++k)
        if (ok[k]) stage[dstart[dig[k]] + mycnt[dig[k]] + lrank[k]] = tp[k];
    __syncthreads();
    SM_FINE(5);
    uint4 *out = reinterpret_cast<uint4 *>(r.out);
#pragma unroll
    for (int k = 0; k < SM_V; ++k) {
        const uint32_t p = (uint32_t)k * SM_BLOCK + tid;
        if (p < count) {
            const uint4 v = stage[p];
            out[delta[v.x & mask] + p] = v;
        }
    }
    SM_FINE(6);
    SM_STAMP(1, 1);
}

}  // namespace rhj
