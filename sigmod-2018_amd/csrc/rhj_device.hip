// rhj_device.hip — launch orchestration, device workspace and staging behind the
// C-ABI of include/rhj.h.  The host-facing reference signatures (RadixHashJoin,
// Filter, result lists) live in rhj_abi.c and call the rhj_host_* helpers below.
//
// There is no CPU fallback: any HIP failure is reported on stderr and returned as
// an error; a missing GPU makes every entry point fail.
#include "rhj_kernels.hip.h"
#include "rhj_shard_kernels.hip.h"
#include "rhj_internal.h"
#include <mutex>
#include <thread>
#include <atomic>
#include <vector>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <utility>

using namespace rhj;

#define HIP_TRY(x)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "rhj: %s -> %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                              \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

namespace {

// RHJ_DEBUG_SYNC=1: synchronise after every launch and name it on stderr (the last name printed before
// a "Memory access fault" is the kernel that faulted).  Diagnostic runs only.
static int g_debug_sync = -1;
static void debug_after_launch(const char *what, hipStream_t s)
{
    if (g_debug_sync < 0) g_debug_sync = getenv("RHJ_DEBUG_SYNC") ? 1 : 0;
    if (!g_debug_sync) return;
    fprintf(stderr, "rhj: launched %s\n", what);
    fflush(stderr);
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) fprintf(stderr, "rhj: %s -> %s\n", what, hipGetErrorString(e));
}
#define RHJ_LAUNCH(kernel, grid, block, lds, stream, ...)                       \
    do {                                                                         \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);       \
        debug_after_launch(#kernel, stream);                                     \
    } while (0)

// kernels compiled apart for sharded joins (template <bool RANGED>): the ordinary form when the range is the whole radix
#define RHJ_LAUNCH_RANGED(kernel, ranged, grid, block, lds, stream, ...)        \
    do {                                                                         \
        if (ranged) RHJ_LAUNCH((kernel<true>), grid, block, lds, stream, __VA_ARGS__);   \
        else        RHJ_LAUNCH((kernel<false>), grid, block, lds, stream, __VA_ARGS__);  \
    } while (0)

struct Buf {
    void  *p = nullptr;
    size_t cap = 0;
};

constexpr int MAX_BITS = 15;
constexpr uint32_t LDS_BUDGET = 160 * 1024;       // bytes per workgroup on gfx950

enum Stage { ST_HIST, ST_SCAN, ST_SCATTER, ST_PLAN, ST_BUILD, ST_COUNT, ST_OFFSETS, ST_PROBE, ST_END, ST_N };

struct Ctx {
    bool        ready = false;
    int         device = 0;
    hipStream_t stream = nullptr;
    bool        own_stream = false;
    bool        stream_set = false;  // rhj_set_stream() was called (possibly with the null stream)
    int         bits = 4;
    int         null_on_empty = 0;
    int         force_hbm = 0;
    int         ablate = 0;
    int         order_any = 0;       // 1: pair order not needed, the library picks the radix (env RHJ_ORDER=any, rhj_set_order(1))
    int         no_fused = 0;
    int         force_fused = 0;     // rhj_set_fused(2): the fused path even where the tiled one is expected to be faster (tiny buckets)
    int         no_resident = 0;
    int         wide_row_ids = 0;    // 1: never use 12-byte intermediates (env RHJ_WIDE_ROW_IDS)
    int         timing = 2;          // 0: no events, rhj_get_stats() times are zero; 1: whole join only; 2: per stage (env RHJ_TIMING, rhj_set_timing)
    bool        stamps = false;      // env RHJ_STAMPS (diagnostics build): in-kernel phase stamps of the fused kernel, read once at load time
    int         no_count_in_pass1 = 0;   // 1: pass 2's counts from the digit bytes (k_hist_runs) at every radix width (env RHJ_NO_COUNT_IN_PASS1; A/B)
    int         no_spec = 0;         // 1: never try the foreign-key speculation (env RHJ_NO_SPEC; rhj_set_spec(0))
    int         spec_score = 2;      // > 0: try it (a speculation that holds adds 1, up to 4; one that fails takes 2 off)
    int         spec_skipped = 0;    // joins not speculated on since the score went to zero: every 16th tries again
    int         last_spec = 0;       // the last join: 0 not tried, 1 held, 2 failed (rhj_last_spec)
    int         no_exact = 1;        // 1: never launch k_join_exact (the default: it measured slower than the gather kernels, profiles/README.md r04a; env RHJ_EXACT=1 / rhj_set_exact(1) turn it on)
    int         exact_score = 2;     // > 0: launch it where it applies (a join it did adds 1, up to 4; one it handed back for its input takes 2 off)
    int         exact_skipped = 0;   // eligible joins not given to it since the score went to zero: every 16th tries again
    int         cols_input = 0;      // this join's inputs are key columns (rhj_join_keys_device): pass 1 of the two-pass partition reads 8 bytes a tuple
    int         last_exact = 0;      // the last join: 0 not launched, 1 k_join_exact did the join, 2 it handed over (rhj_last_exact)
    int         msd = 0;             // RHJ_MSD=1: pass 1 of the two-pass partition takes the HIGH bits of the radix, pass 2 the low ones (A/B)
    int         lo_override = 0;     // RHJ_LO_BITS: pass-1 digit bits of the two-pass partition (experiments; default bits / 2)
    int         seen_wide = 0;       // a join of this process needed 16-byte intermediates: launch those kernels from now on
    int         no_lowradix = 0;     // 1: never take the low-radix path (env RHJ_NO_LOWRADIX; rhj_set_lowradix(0)): big joins on few bits go tiled
    int         no_small = 0;        // 1: never take the three-launch path for small joins (env RHJ_NO_SMALL, rhj_set_small(0))
    uint32_t    small_tiles = 512;   // largest relation, in 8192-tuple tiles, the small path takes (env RHJ_SMALL_TILES; at most SM_MAX_TILES)
    uint32_t    range_lo = 0, range_span = 0;   // rhj_join_device_range: the buckets this call joins (span 0: all of them)
    uint64_t    slice_skip = 0, slice_end = 0;  // rhj_join_device_slice: the first bucket's probe tuples from slice_skip on, the last bucket's before slice_end (0: all)
    int         cus = 256;           // compute units of the device (one fused workgroup each)
    uint64_t    node_pairs = 65535;
    hipEvent_t  ev[ST_N + 1] = {};
    hipEvent_t  ev_x[4] = {};
    Buf partR, partS, tmpR, tmpS, cntR, cntS, chunk, histpsum, passhp, units, bunits, ldsb, meta, summary,
        ucount, ubase, uflag, tab32, tab64, stash_cnt, stash_row, status, dbg, bsum, digR, digS, ovf, ovf_base, runR, runS, walk, xrows, lr_tmp, lr_words, lr_status, stripR, stripS, slice_tot, sbase;
    Buf inR, inS, out, fcol_sel, fmask, ftile, fbase, fout;
    void *pin = nullptr;            // small pinned block for read-backs
    void *pin_ring[4] = {nullptr, nullptr, nullptr, nullptr};   // D2H staging of result pairs (16 MiB each)
    hipEvent_t ev_ring[4] = {};
    // REGISTERED host columns (rhj_register_relation_map / the resident InitRelationMap) -> device copy.
    // Nothing else is cached: an unregistered column is uploaded on every call that names it, so a caller
    // that frees a column and gets the same address back never meets the old contents.
    struct Column { void *dev; size_t rows; };
    std::map<const void *, Column> columns;
    std::map<const void *, size_t> pinned;                       // hipHostRegister'ed host ranges (base -> bytes)
    int pin_refusals = 0;                                        // ranges of 64 KiB or more the host refused to pin
    Buf fcol;                                                    // staging of an unregistered column (host Filter())
    std::multimap<size_t, void *> free_blocks;                  // rhj_dev_alloc: cached blocks by size
    std::map<void *, size_t> live_blocks;                       // rhj_dev_alloc: blocks handed out
    rhj_stats stats = {};
};

// One context per device.  g_all[0] is the library's context (every entry point of round 1..3 works on it); g_all[1..] belong to
// the further devices of rhj_set_devices(n) and are only ever touched by the worker threads of a multi-device join, each of
// which makes its device's context the current one of ITS thread.  `g` stays the name of "the context this thread works on".
constexpr int MAX_DEVICES = 8;
Ctx g_all[MAX_DEVICES];
thread_local Ctx *g_cur = &g_all[0];
#define g (*g_cur)
int g_ndev = 1;                      // devices a join is sharded over (rhj_set_devices, env RHJ_DEVICES)
int g_ndev_env = 0;                  // RHJ_DEVICES as read at load time (applied by the first call that can shard)
int g_balance = 0;                   // rhj_join_devices: 1 cuts the bucket ranges by histR + histS instead of equal widths, 2 also cuts INSIDE hot buckets (rhj_set_devices_balance, env RHJ_DEVICES_BALANCE=hist / slice)
int g_same_device = 0;               // RHJ_DEVICES_SAME=1 (tests): every context on the library's own device — n streams and workspaces on one GPU

// environment defaults are read once at load time; the rhj_set_* calls override them
struct EnvDefaults {
    EnvDefaults()
    {
        const char *e;
        if ((e = getenv("RHJ_DEVICE"))) g.device = atoi(e);
        if ((e = getenv("RHJ_DEVICES"))) g_ndev_env = atoi(e);
        if ((e = getenv("RHJ_DEVICES_SAME"))) g_same_device = atoi(e);
        if ((e = getenv("RHJ_DEVICES_BALANCE"))) g_balance = strcmp(e, "slice") == 0 ? 2 : strcmp(e, "hist") == 0 ? 1 : 0;
        if ((e = getenv("RHJ_RADIX_BITS"))) { int b = atoi(e); if (b >= 1 && b <= 15) g.bits = b; }
        if ((e = getenv("RHJ_EMPTY"))) g.null_on_empty = (strcmp(e, "null") == 0);
        if ((e = getenv("RHJ_FORCE_HBM_TABLE"))) g.force_hbm = atoi(e);
        if ((e = getenv("RHJ_ABLATE"))) g.ablate = atoi(e);
        if ((e = getenv("RHJ_ORDER"))) g.order_any = (strcmp(e, "any") == 0);
        if ((e = getenv("RHJ_NO_FUSED"))) g.no_fused = atoi(e);
        if ((e = getenv("RHJ_FORCE_FUSED"))) g.force_fused = atoi(e);
        if ((e = getenv("RHJ_NO_RESIDENT"))) g.no_resident = atoi(e);
        if ((e = getenv("RHJ_WIDE_ROW_IDS"))) g.wide_row_ids = atoi(e);
        if ((e = getenv("RHJ_NODE_PAIRS"))) g.node_pairs = strtoull(e, nullptr, 10);
        if ((e = getenv("RHJ_NO_SMALL"))) g.no_small = atoi(e);
        if ((e = getenv("RHJ_NO_LOWRADIX"))) g.no_lowradix = atoi(e);
        if ((e = getenv("RHJ_NO_SPEC"))) g.no_spec = atoi(e);
        if ((e = getenv("RHJ_EXACT"))) g.no_exact = !atoi(e);
        if ((e = getenv("RHJ_LO_BITS"))) g.lo_override = atoi(e);
        if ((e = getenv("RHJ_MSD"))) g.msd = atoi(e);
        if ((e = getenv("RHJ_NO_COUNT_IN_PASS1"))) g.no_count_in_pass1 = atoi(e);
        g.stamps = getenv("RHJ_STAMPS") != nullptr;
        if ((e = getenv("RHJ_TIMING"))) g.timing = atoi(e);
        if ((e = getenv("RHJ_SMALL_TILES"))) { g.small_tiles = (uint32_t)atoi(e); if (g.small_tiles > SM_MAX_TILES) g.small_tiles = SM_MAX_TILES; }
    }
} env_defaults;

// Stage events by timing level (rhj_set_timing): 2 = all of them, 1 = first and last of a join only, 0 = none.  A record
// between two kernels keeps the second from being fed while the first drains (~6 us, tools/timeline.sh).
static inline bool stage_on(int st) { return g.timing >= 2 || (g.timing >= 1 && (st == ST_HIST || st == ST_END)); }
#define RHJ_STAGE(st) do { if (stage_on(st)) HIP_TRY(hipEventRecord(g.ev[st], g.stream)); } while (0)

int ensure(Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    static const bool trace = getenv("RHJ_TRACE") != nullptr;
    if (trace) fprintf(stderr, "rhj-trace:   workspace buffer grows %zu -> %zu bytes (hipFree + hipMalloc)\n", b.cap, bytes);
    const size_t was = b.cap;
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    // first allocation: what is asked for and an eighth; after that at least double — a query plan's joins come in every
    // size and each regrowth is a hipFree + hipMalloc (two device-wide synchronisations): 41 of them in the `small` run
    size_t want = bytes + bytes / 8 + 4096;
    if (was && was < ((size_t)1 << 30) && want < 2 * was) want = 2 * was;     // (above 1 GiB a regrowth is not a per-query event: no doubling)
    if (hipMalloc(&b.p, want) != hipSuccess) {
        // the generous size does not fit: what was asked for, exactly (a C4-scale buffer of 16 GB must not fail for its slack)
        (void)hipGetLastError();
        b.p = nullptr;
        want = bytes;
        HIP_TRY(hipMalloc(&b.p, want));
    }
    b.cap = want;
    return 0;
}

int ctx_init()
{
    // HIP's current device is per thread: every entry point (all of them come through here under the API
    // lock) selects the library's device on the calling thread before it allocates, records or launches
    if (g.ready) { HIP_TRY(hipSetDevice(g.device)); return 0; }
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) { fprintf(stderr, "rhj: no HIP device visible; this library has no CPU path\n"); return -1; }
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipDeviceGetAttribute(&g.cus, hipDeviceAttributeMultiprocessorCount, g.device));
    if (g.cus <= 0) g.cus = 256;
    if (!g.stream_set) { HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking)); g.own_stream = true; }
    for (auto &ev : g.ev) HIP_TRY(hipEventCreate(&ev));
    for (auto &ev : g.ev_x) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipHostMalloc(&g.pin, 4096, hipHostMallocDefault));
    // dynamic LDS above 64 KiB has to be requested per kernel
    HIP_TRY(hipFuncSetAttribute((const void *)k_build_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
    {
        const void *fused[6] = {(const void *)k_join_fused<false, false>, (const void *)k_join_fused<false, true>,
                                (const void *)k_join_fused<true, false>, (const void *)k_join_fused<true, true>,
                                (const void *)k_join_spec<false>, (const void *)k_join_spec<true>};
        for (const void *k : fused)
            HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_BUDGET - FJ_LDS_EXTRA)));
        HIP_TRY(hipFuncSetAttribute((const void *)k_join_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_BUDGET - FJ_LDS_EXTRA)));
        HIP_TRY(hipFuncSetAttribute((const void *)k_join_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_BUDGET - FJ_LDS_EXTRA)));
    }
    HIP_TRY(hipFuncSetAttribute((const void *)k_small_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes(PT_MAX_BITS)));
    HIP_TRY(hipFuncSetAttribute((const void *)k_bucket_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4u << MAX_BITS)));
    HIP_TRY(hipFuncSetAttribute((const void *)k_bucket_psum<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8u << 14)));
    HIP_TRY(hipFuncSetAttribute((const void *)k_scatter_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
    HIP_TRY(hipFuncSetAttribute((const void *)k_scatter_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
    {
        const void *lp[] = {(const void *)k_local_part<false, true, false>, (const void *)k_local_part<false, true, true>,
                            (const void *)k_local_part<false, false, true>, (const void *)k_local_part<true, true, false>,
                            (const void *)k_local_part<true, true, true>,   (const void *)k_local_part<true, false, true>,
                            (const void *)k_local_part<false, true, false, true>, (const void *)k_local_part<false, true, true, true>,
                            (const void *)k_local_part<false, false, true, true>};
        for (const void *k : lp) HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
    }
    HIP_TRY(hipFuncSetAttribute((const void *)k_scatter_runs<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
    HIP_TRY(hipFuncSetAttribute((const void *)k_scatter_runs<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
    HIP_TRY(hipFuncSetAttribute((const void *)k_scatter_runs<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
    HIP_TRY(hipFuncSetAttribute((const void *)k_lr_emit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr_lds_bytes(PT_MAX_BITS)));
    g.ready = true;
    return 0;
}

struct PartState {
    bool     launch_wide = false;    // the 16-byte kernels of the two-pass partition were launched (else only the 12-byte ones)
    RelArgs  p2[2];          // two-pass partition: the relations as pass 2 saw them (runs, digit bytes, scanned tile counts) —
                             // the low-radix path replays pass 2's order when it emits (rhj_lowradix.hip.h)
    int      lo_bits = 0;    // two-pass partition: digit bits of pass 1 (0: half of the radix); the low-radix path passes the caller's radix
    RelArgs  r[2];           // in = caller's input, out = final partitioned array
    rhj_tuple *tmp[2];       // intermediate of the two-pass path
    uint64_t *hist, *psum;   // [2][bins] of the join's radix (filled by run_partition)
    const PlanArgs *plan = nullptr;   // the join's plan arguments: a small one-pass partition runs the plan in its scan launch
    bool plan_done = false;
};

uint32_t tiles_for(uint64_t n)
{
    const uint64_t t = (n + PT_TILE - 1) / PT_TILE;
    return (uint32_t)(t ? t : 1);
}

size_t scatter_lds_bytes(int bits)
{
    const size_t bins = (size_t)1 << bits;
    return (size_t)PT_TILE * 16 + (PT_WAVES + 2) * bins * 4 + (PT_BLOCK / 64 + 1) * 8 + 16;
}

size_t scatter_runs_lds_bytes(int bits)
{
    const size_t bins = (size_t)1 << bits;
    return (size_t)SR_TILE * 16 + (PT_WAVES + 3) * bins * 4 + (PT_BLOCK / 64 + 2) * 8 + 2 * (SR_RUNOFF + PT_MAX_GROUP) * 4 + 16;   // (run tables double-buffered)
}

// one stable pass over both relations (radix bits <= 8): per-tile histogram, scan, LDS-staged scatter
int partition_pass(RelArgs r0, RelArgs r1, int nrel, int bits, uint64_t *hist, uint64_t *psum, const PlanArgs *plan = nullptr,
                   bool *plan_done = nullptr)
{
    const uint32_t bins = 1u << bits;
    uint32_t max_tiles = r0.tiles;
    if (nrel > 1 && r1.tiles > max_tiles) max_tiles = r1.tiles;
    if (plan && nrel == 2 && max_tiles <= SMALL_TILES) {
        // small join: histogram, {scans + plan} in one single-workgroup launch, scatter — three launches instead of seven
        RHJ_STAGE(ST_HIST);
        RHJ_LAUNCH_RANGED(k_hist_tiles, r0.range_span, dim3(max_tiles, nrel), dim3(256), (size_t)bins * 4, g.stream, r0, r1, 0, bits);
        RHJ_STAGE(ST_SCAN);
        RHJ_LAUNCH(k_small_scan_plan, dim3(1), dim3(1024), 0, g.stream, r0, r1, bits, hist, psum, *plan);
        RHJ_STAGE(ST_SCATTER);
        RHJ_LAUNCH_RANGED(k_scatter_lds, r0.range_span, dim3(max_tiles, nrel), dim3(PT_BLOCK), scatter_lds_bytes(bits), g.stream, r0, r1, 0, bits);
        HIP_TRY(hipGetLastError());
        *plan_done = true;
        return 0;
    }
    uint32_t chunks = (max_tiles + 15) / 16;                  // >= 16 tiles per chunk, at most 512 chunks
    if (chunks > 512) chunks = 512;
    if (chunks < 1) chunks = 1;
    if (ensure(g.chunk, (size_t)2 * chunks * bins * 8)) return -1;
    const uint32_t hist_grid = max_tiles < 2048 ? max_tiles : 2048;
    RHJ_STAGE(ST_HIST);
    RHJ_LAUNCH_RANGED(k_hist_tiles, r0.range_span, dim3(hist_grid, nrel), dim3(256), (size_t)bins * 4, g.stream, r0, r1, 0, bits);
    RHJ_STAGE(ST_SCAN);
    RHJ_LAUNCH(k_scan_chunks, dim3((bins + 255) / 256, chunks, nrel), dim3(256), 0, g.stream, r0, r1, bits,
                       chunks, (uint64_t *)g.chunk.p);
    RHJ_LAUNCH(k_scan_bins, dim3(bins, nrel), dim3(WAVE), 0, g.stream, bits, chunks, (uint64_t *)g.chunk.p, hist);
    RHJ_LAUNCH(k_scan_psum, dim3(nrel), dim3(1024), 0, g.stream, bits, (const uint64_t *)hist, psum);
    RHJ_LAUNCH(k_scan_apply, dim3((bins + 255) / 256, chunks, nrel), dim3(256), 0, g.stream, r0, r1, bits,
                       chunks, (const uint64_t *)g.chunk.p, (const uint64_t *)psum);
    RHJ_STAGE(ST_SCATTER);
    RHJ_LAUNCH_RANGED(k_scatter_lds, r0.range_span, dim3(max_tiles, nrel), dim3(PT_BLOCK), scatter_lds_bytes(bits), g.stream, r0, r1, 0, bits);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Stable radix partition of one or two relations on the low `bits` bits.  bits <= 8: one pass
// (per-tile histogram, scan, LDS-staged scatter).  bits 9..15: two LSD passes in run form — pass 1
// partitions every tile in place on the low half of the bits (no histogram, no offsets), pass 2
// moves the runs to their final places on the high half; the bucket histogram of the full radix is
// the column sum of pass 2's per-tile counts.
// Stage events: one pass  ST_HIST..ST_SCAN histogram, ST_SCAN..ST_SCATTER scan, ST_SCATTER..ST_PLAN scatter;
//               two passes ST_HIST..ST_SCAN pass 1, ST_SCAN..ST_SCATTER pass-2 histogram + scan,
//                          ST_SCATTER..ST_PLAN pass-2 scatter.
// final12: with 12-byte intermediates (row ids below 2^32) the FINAL arrays hold Tuple12 as well — the join's own
// partition on its fused path; rhj_partition_device() hands out rhj_tuple and passes false.
int run_partition(PartState &ps, int bits, int nrel, bool force_wide, bool final12)
{
    const uint32_t bins = 1u << bits;
    if (ensure(g.histpsum, (size_t)4 * bins * 8) || ensure(g.passhp, (size_t)4 * 256 * 8)) return -1;
    ps.hist = (uint64_t *)g.histpsum.p;
    ps.psum = ps.hist + 2 * bins;
    RelArgs none = RelArgs{};
    for (int i = 0; i < nrel; ++i) ps.r[i].tiles = tiles_for(ps.r[i].n);
    if (ensure(g.summary, sizeof(PlanSummary))) return -1;
    if (ensure(g.cntR, (size_t)ps.r[0].tiles * 256 * 4)) return -1;
    ps.r[0].cnt = (uint32_t *)g.cntR.p;
    if (nrel > 1) {
        if (ensure(g.cntS, (size_t)ps.r[1].tiles * 256 * 4)) return -1;
        ps.r[1].cnt = (uint32_t *)g.cntS.p;
    }
    if (bits <= PT_MAX_BITS && !ps.lo_bits) {
        // one pass: no 12-byte intermediates and nothing that checks the row ids, so everything downstream stays wide
        HIP_TRY(hipMemsetAsync(&((PlanSummary *)g.summary.p)->wide_row_ids, 0, 8, g.stream));    // wide_row_ids, row_id_overflow
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)&((PlanSummary *)g.summary.p)->wide_row_ids, 1, 1, g.stream));
        ps.launch_wide = true;
        return partition_pass(ps.r[0], nrel > 1 ? ps.r[1] : none, nrel, bits, ps.hist, ps.psum, ps.plan, &ps.plan_done);
    }

    // ---- two passes in run form (k_local_part .. k_scatter_runs in rhj_kernels.hip.h)
    int lo = ps.lo_bits ? ps.lo_bits : bits / 2;
    if (!ps.lo_bits && g.lo_override > 0 && g.lo_override < bits && bits - g.lo_override <= PT_MAX_BITS && g.lo_override <= PT_MAX_BITS) lo = g.lo_override;   // (RHJ_LO_BITS: experiments)
    const int hi = bits - lo;
    // which end of the radix pass 1 takes: the low `lo` bits (pass 2 then gathers by them and scatters by the high `hi`), or —
    // msd — the high `lo` bits, pass 2 the low `hi` (same buckets, same order inside them: either pass is stable and a pass-2
    // tile reads its runs in tile order).  The low-radix path counts on pass 1 taking the caller's own low bits.
    const bool msd = g.msd && !ps.lo_bits;
    const int sh1 = msd ? hi : 0, sh2 = msd ? 0 : lo;
    const uint32_t bins1 = 1u << lo, bins2 = 1u << hi;
    if (ensure(g.slice_tot, (size_t)2 * bins * FH_SLICES * 4) || ensure(g.sbase, (size_t)2 * bins * FH_SLICES * 4)) return -1;
    RelArgs a0 = ps.r[0], a1 = nrel > 1 ? ps.r[1] : none;
    Buf *digb[2] = {&g.digR, &g.digS}, *runb[2] = {&g.runR, &g.runS}, *cntb[2] = {&g.cntR, &g.cntS}, *partb[2] = {&g.stripR, &g.stripS};
    // pass 1 counts pass 2's digits itself while the (digit, digit) cells fit beside its staging tile (k_local_part); the
    // digit bytes then serve the low-radix emit only
    const bool count_in_pass1 = bits <= 12 && !g.no_count_in_pass1;
    const bool want_dig = !count_in_pass1 || ps.lo_bits != 0;
    RelArgs *ar[2] = {&a0, &a1};
    uint32_t group = 15u * bins1 / 16u;               // a pass-2 tile averages 15/16 of 4096 tuples on uniform keys
    // (the low-radix path replays pass 2 one batch per tile and gives up on a tile beyond 4096 tuples: 7/8 of a batch on
    // average puts the limit 8 standard deviations away on uniform keys — at 15/16 it was 4, and 100 M tuples have 49 K tiles)
    if (ps.lo_bits) group = 7u * bins1 / 8u;
    if (group < 1) group = 1;
    if (group > PT_MAX_GROUP - 1) group = PT_MAX_GROUP - 1;
    uint32_t most_tiles = 0;
    for (int i = 0; i < nrel; ++i) most_tiles = ps.r[i].tiles > most_tiles ? ps.r[i].tiles : most_tiles;
    const uint32_t strip_tiles = most_tiles >= 2048 ? PT_STRIP : most_tiles >= 1024 ? (PT_STRIP < 2 ? PT_STRIP : 2u) : 1u;
    for (int i = 0; i < nrel; ++i) {
        RelArgs &a = *ar[i];
        a.tiles1 = a.tiles;
        a.group = group;
        a.groups = (a.tiles1 + group - 1) / group;
        // strips of 4 tiles when there are enough of them to fill the chip twice over; small relations keep one tile a workgroup
        // (200 K .. 1 M tuples lost 3-6 % of the join to strips of 4: 75 workgroups of four tiles each instead of 245 of one)
        a.strip = strip_tiles;
        a.parts = (group + a.strip - 1) / a.strip;
        a.per = (a.groups + FH_SLICES - 1) / FH_SLICES;
        a.sbase = (uint32_t *)g.sbase.p + (size_t)i * bins * FH_SLICES;
        if ((want_dig && ensure(*digb[i], a.n + 64)) || ensure(*runb[i], (size_t)a.tiles1 * (bins1 + 1) * 2 + 64) ||
            ensure(*cntb[i], (size_t)bins1 * a.groups * bins2 * 4) ||
            (count_in_pass1 && ensure(*partb[i], (size_t)bins * a.groups * a.parts * 2)))
            return -1;
        a.out = ps.tmp[i];
        a.part = count_in_pass1 ? (uint16_t *)partb[i]->p : nullptr;
        a.dig_out = want_dig ? (uint8_t *)digb[i]->p : nullptr;
        a.runs = (uint16_t *)runb[i]->p;
        a.cnt = (uint32_t *)cntb[i]->p;
    }
    RelArgs b0 = a0, b1 = a1;
    RelArgs *br[2] = {&b0, &b1};
    uint32_t max1 = 0, max2 = 0;
    for (int i = 0; i < nrel; ++i) {
        RelArgs &b = *br[i];
        b.in = ps.tmp[i];
        b.out = ps.r[i].out;
        b.dig_in = want_dig ? (const uint8_t *)digb[i]->p : nullptr;
        b.dig_out = nullptr;
        b.tiles = bins1 * b.groups;                   // pass-2 tiles
        if (ar[i]->tiles > max1) max1 = ar[i]->tiles;
        if (b.tiles > max2) max2 = b.tiles;
    }
    // 12-byte intermediates when the row ids fit 32 bits: a sample decides on the device and the pass
    // kernels read the decision there (no host round trip; pass 2 is compiled once per format).  The host
    // launches the 12-byte kernels alone until a join of this process turned out to need the 16-byte ones
    // (g.seen_wide): the sample then reports a wide input as an overflow and the caller runs again wide.
    const bool launch_narrow = !force_wide;
    ps.launch_wide = force_wide || g.seen_wide;
    if (ensure(g.summary, sizeof(PlanSummary))) return -1;
    PlanSummary *dsum = (PlanSummary *)g.summary.p;
    RHJ_STAGE(ST_HIST);
    RHJ_LAUNCH(k_rowid_sample, dim3(1), dim3(1024), 0, g.stream, a0, a1, nrel, force_wide ? 1 : 0, ps.launch_wide ? 0 : 1, dsum, g.cols_input);
    {
        const uint32_t h2_off = (uint32_t)((scatter_lds_bytes(lo) + 15) & ~(size_t)15);
        const size_t lds1 = count_in_pass1 ? h2_off + ((size_t)2 << bits) : scatter_lds_bytes(lo);
        uint32_t strips = 0;
        for (int i = 0; i < nrel; ++i) strips = ar[i]->groups * ar[i]->parts > strips ? ar[i]->groups * ar[i]->parts : strips;
        const dim3 grid1(count_in_pass1 ? strips : max1, nrel);
        const bool ranged = a0.range_span != 0;
#define RHJ_LP(R, H, D) RHJ_LAUNCH((k_local_part<R, H, D>), grid1, dim3(PT_BLOCK), lds1, g.stream, a0, a1, sh1, lo, sh2, hi, dsum, h2_off)
#define RHJ_LPC(H, D) RHJ_LAUNCH((k_local_part<false, H, D, true>), grid1, dim3(PT_BLOCK), lds1, g.stream, a0, a1, sh1, lo, sh2, hi, dsum, h2_off)
        if (g.cols_input) { if (!count_in_pass1) RHJ_LPC(false, true); else if (want_dig) RHJ_LPC(true, true); else RHJ_LPC(true, false); }   // (never ranged: join_keys)
        else if (ranged)  { if (!count_in_pass1) RHJ_LP(true, false, true); else if (want_dig) RHJ_LP(true, true, true); else RHJ_LP(true, true, false); }
        else              { if (!count_in_pass1) RHJ_LP(false, false, true); else if (want_dig) RHJ_LP(false, true, true); else RHJ_LP(false, true, false); }
#undef RHJ_LPC
#undef RHJ_LP
    }
    RHJ_STAGE(ST_SCAN);
    if (count_in_pass1)
        RHJ_LAUNCH((k_group_scan<true>), dim3(bins1, nrel, FH_SLICES), dim3(1024), 0, g.stream, b0, b1, hi, (uint32_t *)g.slice_tot.p, msd ? 1 : 0);
    else {
        const uint32_t hw = (max2 + HR_BLOCK / WAVE - 1) / (HR_BLOCK / WAVE);    // one wave per pass-2 tile
        RHJ_LAUNCH(k_hist_runs, dim3(hw < 4096 ? hw : 4096, nrel), dim3(HR_BLOCK), (size_t)bins2 * 4 * (HR_BLOCK / WAVE), g.stream,
                   b0, b1, hi);
        RHJ_LAUNCH((k_group_scan<false>), dim3(bins1, nrel, FH_SLICES), dim3(1024), 0, g.stream, b0, b1, hi, (uint32_t *)g.slice_tot.p, msd ? 1 : 0);
    }
    const int staged = bits >= 13;
#define RHJ_BP(P) RHJ_LAUNCH((k_bucket_psum<P>), dim3(nrel, FH_SLICES), dim3(1024), staged ? (size_t)(bins < 16384u ? bins : 16384u) * 8 : 0, g.stream, lo, hi, \
                            (const uint32_t *)g.slice_tot.p, (uint32_t *)g.sbase.p, ps.hist, ps.psum, staged, msd ? 1 : 0)
    if (bits <= 10) RHJ_BP(1); else if (bits == 11) RHJ_BP(2); else if (bits == 12) RHJ_BP(4); else RHJ_BP(0);
#undef RHJ_BP
    RHJ_STAGE(ST_SCATTER);
    uint32_t search0 = 1;                             // largest power of two <= group: first step of the run search
    while (search0 * 2 <= group) search0 *= 2;
    {
        uint32_t per_cu = (uint32_t)(LDS_BUDGET / scatter_runs_lds_bytes(hi));   // the workgroups that are resident together
        if (per_cu > (uint32_t)SR_MINW * 256u / PT_BLOCK) per_cu = (uint32_t)SR_MINW * 256u / PT_BLOCK;
        if (per_cu < 1) per_cu = 1;
        const uint32_t sgrid = (uint32_t)g.cus * per_cu;
        const uint32_t want = ((max2 < sgrid ? max2 : sgrid) + 7u) & ~7u;     // a multiple of the 8 XCDs
        if (launch_narrow && final12)
            RHJ_LAUNCH((k_scatter_runs<true, true>), dim3(want, nrel), dim3(PT_BLOCK), scatter_runs_lds_bytes(hi), g.stream, b0, b1,
                       sh2, hi, search0, (const PlanSummary *)dsum);
        else if (launch_narrow)
            RHJ_LAUNCH((k_scatter_runs<true, false>), dim3(want, nrel), dim3(PT_BLOCK), scatter_runs_lds_bytes(hi), g.stream, b0, b1,
                       sh2, hi, search0, (const PlanSummary *)dsum);
        if (ps.launch_wide)
            RHJ_LAUNCH((k_scatter_runs<false, false>), dim3(want, nrel), dim3(PT_BLOCK), scatter_runs_lds_bytes(hi), g.stream, b0, b1,
                       sh2, hi, search0, (const PlanSummary *)dsum);
    }
    ps.p2[0] = b0; ps.p2[1] = b1;
    HIP_TRY(hipGetLastError());
    return 0;
}

// exclusive scan of up to max_n u64 counts (device-side count in *n_ptr when given)
int launch_offsets(const uint64_t *cnt, uint64_t *base, const uint64_t *n_ptr, uint64_t n_fixed, uint64_t max_n,
                   uint64_t *total_out)
{
    const uint32_t nblocks = (uint32_t)((max_n + 1023) / 1024 ? (max_n + 1023) / 1024 : 1);
    if (ensure(g.bsum, (size_t)nblocks * 8)) return -1;
    RHJ_LAUNCH(k_offsets_local, dim3(nblocks), dim3(1024), 0, g.stream, cnt, base, n_ptr, n_fixed, (uint64_t *)g.bsum.p);
    RHJ_LAUNCH(k_offsets_blocks, dim3(1), dim3(1024), 0, g.stream, (uint64_t *)g.bsum.p, nblocks, total_out);
    RHJ_LAUNCH(k_offsets_add, dim3(nblocks), dim3(1024), 0, g.stream, base, n_ptr, n_fixed, (const uint64_t *)g.bsum.p);
    return 0;
}

float ev_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}
float stage_ms(int a, int b) { return stage_on(a) && stage_on(b) ? ev_ms(g.ev[a], g.ev[b]) : 0.f; }

// The whole device-side join.  out == nullptr && use_ctx_out: the pairs land in the
// context's own buffer (grown after the count pass), returned through *ctx_out.
int join_device_once(const rhj_tuple *dR, uint64_t nR, const rhj_tuple *dS, uint64_t nS, rhj_result_tuple *out,
                     uint64_t out_capacity, bool use_ctx_out, rhj_result_tuple **ctx_out, uint64_t *matches,
                     bool force_wide, bool *overflow)
{
    *overflow = false;
    if (ctx_init()) return -1;
    rhj_stats &st = g.stats;
    const float keep_h2d = st.ms_h2d;
    memset(&st, 0, sizeof(st));
    st.ms_h2d = keep_h2d;
    st.n_r = nR; st.n_s = nS; st.radix_bits = g.bits;
    *matches = 0;
    if (ctx_out) *ctx_out = nullptr;
    if (nR == 0 || nS == 0) return 0;                         // rhjoin.c:15-16
    if (nR >= (1ull << 32) || nS >= (1ull << 32)) {
        fprintf(stderr, "rhj: relations of 2^32 tuples or more are not supported (offsets are 32-bit; the reference "
                        "itself is limited to 2^31-1, SURVEY.md finding 9)\n");
        return -2;
    }
    const int bits = g.bits;
    const uint32_t bins = 1u << bits;

    PartState ps;
    if (ensure(g.partR, nR * sizeof(rhj_tuple)) || ensure(g.partS, nS * sizeof(rhj_tuple))) return -1;
    ps.r[0] = RelArgs{dR, (rhj_tuple *)g.partR.p, nullptr, nR, 0, 0, nullptr, nullptr};
    ps.r[1] = RelArgs{dS, (rhj_tuple *)g.partS.p, nullptr, nS, 0, 0, nullptr, nullptr};
    const bool ranged = g.range_span != 0;                // a rank's share of a sharded join: the partition drops the other buckets
    for (int i = 0; i < 2; ++i) { ps.r[i].range_lo = g.range_lo; ps.r[i].range_span = g.range_span; }
    ps.tmp[0] = ps.tmp[1] = nullptr;
    if (bits > PT_MAX_BITS) {
        if (ensure(g.tmpR, nR * sizeof(rhj_tuple)) || ensure(g.tmpS, nS * sizeof(rhj_tuple))) return -1;
        ps.tmp[0] = (rhj_tuple *)g.tmpR.p; ps.tmp[1] = (rhj_tuple *)g.tmpS.p;
    }
    // ---- plan arguments (the plan runs behind the partition; a small one-pass partition runs it in its scan launch)
    const uint32_t build_chunk = 4096;
    // Thousands of buckets of a few hundred tuples: a fused unit costs ~15 us whatever its size (a dozen barriers and
    // dependent round trips, one unit per CU at a time), the tiled path's 256-thread probe units run eight to a CU —
    // 1M x 1M at 15 bits 2.2 ms fused, 0.88 ms tiled; from ~512 tuples per bucket on the fused path is ahead again
    // (tools/exp_twopass_sizes.py).  rhj_set_fused(2) keeps the fused path regardless.
    const bool tiny_buckets = bins >= 4096 && (nR > nS ? nR : nS) < (uint64_t)512 * bins;
    const bool want_fused = !g.no_fused && !g.force_hbm && (g.force_fused || !tiny_buckets);
    const uint32_t lds_max_slots = LDS_BUDGET / 4 / 4 * 4;                 // tiled path: k_build_lds owns the whole LDS
    uint32_t lds_cap = (uint32_t)((uint64_t)lds_max_slots * 4 / 5);        // load factor <= 0.8
    if (want_fused) lds_cap = (LDS_BUDGET - FJ_LDS_EXTRA - 128) * 2 / 9;   // fused: 4 B entry + >= 0.5 B of slot starts per build tuple
    if (lds_cap > 65534) lds_cap = 65534;                                  // 16-bit position + 1
    if (g.force_hbm) lds_cap = 0;
    const uint64_t nmin = nR < nS ? nR : nS;
    const uint64_t max_units = (uint64_t)bins + (nR + nS) / PR_UNIT + 2;
    const uint64_t max_bunits = (uint64_t)bins + nmin / build_chunk + 2;
    const uint64_t max_tab32 = nmin + nmin / 2 + (uint64_t)80 * bins + 64;
    if (ensure(g.units, max_units * sizeof(Unit)) || ensure(g.bunits, max_bunits * sizeof(Unit)) ||
        ensure(g.ldsb, (size_t)bins * 4) || ensure(g.meta, (size_t)bins * sizeof(BucketMeta)) ||
        ensure(g.summary, sizeof(PlanSummary) + 64) || ensure(g.ucount, max_units * 8) || ensure(g.ubase, max_units * 8) ||
        ensure(g.uflag, max_units * 4) || ensure(g.histpsum, (size_t)4 * bins * 8))
        return -1;
    PlanArgs pa;
    pa.histR = (uint64_t *)g.histpsum.p; pa.histS = pa.histR + bins;
    pa.units = (Unit *)g.units.p; pa.build_units = (Unit *)g.bunits.p; pa.lds_buckets = (uint32_t *)g.ldsb.p;
    pa.meta = (BucketMeta *)g.meta.p; pa.summary = (PlanSummary *)g.summary.p;
    pa.lds_cap = lds_cap; pa.lds_max_slots = lds_max_slots; pa.build_chunk = build_chunk;
    pa.parent_mask = 0; pa.parent_flip = nullptr; pa.zero = nullptr; pa.zero_words = 0;
    const bool sliced = ranged && (g.slice_skip != 0 || g.slice_end != 0);
    if (sliced) {
        if (g.slice_skip) { pa.slice_b0 = g.range_lo; pa.slice_o0 = g.slice_skip; }
        if (g.slice_end) { pa.slice_b1 = g.range_lo + g.range_span - 1u; pa.slice_o1 = g.slice_end; }
    }
    // probe tuples per fused unit: whole buckets when there are plenty of them, smaller spans (each unit
    // rebuilds its bucket's index) when a low radix would otherwise leave most CUs idle
    uint32_t fused_span = FJ_SPAN;
    if (bins < 256) {                                         // fewer buckets than CUs
        uint64_t want = ((nR > nS ? nR : nS) / 512 + FJ_BATCH - 1) / FJ_BATCH * FJ_BATCH;
        if (want < FJ_BATCH) want = FJ_BATCH;
        if (want < fused_span) fused_span = (uint32_t)want;
    }
    pa.span_lds = want_fused ? fused_span : PR_UNIT;

    // the fused path reads 12-byte partitioned tuples when the row ids fit 32 bits; the tiled path reads rhj_tuple
    if (!want_fused) force_wide = true;
    ps.plan = &pa;
    ps.hist = (uint64_t *)g.histpsum.p;
    ps.psum = ps.hist + 2 * bins;
    PlanSummary *hs = (PlanSummary *)g.pin;
    PlanSummary plan;

    JoinArgs ja;
    ja.partR = (const rhj_tuple *)g.partR.p; ja.partS = (const rhj_tuple *)g.partS.p;
    ja.histR = ps.hist; ja.histS = ps.hist + bins; ja.psumR = ps.psum; ja.psumS = ps.psum + bins;
    ja.units = (const Unit *)g.units.p; ja.meta = (const BucketMeta *)g.meta.p;
    ja.summary = (const PlanSummary *)g.summary.p;
    ja.tab32 = nullptr; ja.tab64 = nullptr;
    ja.unit_count = (uint64_t *)g.ucount.p; ja.unit_base = (const uint64_t *)g.ubase.p;
    ja.unit_flag = (uint32_t *)g.uflag.p;
    ja.out = nullptr; ja.out_capacity = 0;
    ja.ablate = (uint32_t)g.ablate; ja.parent_mask = 0; ja.parent_flip = nullptr;
    ja.stash_cnt = nullptr; ja.stash_row = nullptr; ja.stash_nR = nR;

    // ---- small joins: two launches for the partition (the plan rides in the second), the fused join third, and
    // no host memset, no total kernel, no read-back copy (rhj_small.hip.h)
    bool partitioned = false;      // the small path partitioned and planned, but some bucket needs the tiled path
    const uint32_t stilesR = (uint32_t)((nR + SM_TILE - 1) / SM_TILE), stilesS = (uint32_t)((nS + SM_TILE - 1) / SM_TILE);
    if (want_fused && bits <= PT_MAX_BITS && !g.no_small && !ranged && stilesR <= g.small_tiles && stilesS <= g.small_tiles &&
        !g.stamps) {
        const uint64_t unit_bound = (uint64_t)bins + (nR + nS) / fused_span + 2;
        const uint64_t status_words = unit_bound + 1 + 8;                 // 8 ticket words in front
        RelArgs a0 = ps.r[0], a1 = ps.r[1];
        a0.tiles = stilesR; a1.tiles = stilesS;
        if (ensure(g.cntR, (size_t)a0.tiles * 256 * 4) || ensure(g.cntS, (size_t)a1.tiles * 256 * 4) ||
            ensure(g.stash_cnt, nR + nS + 64) || ensure(g.stash_row, (nR + nS + 8) * 8) ||
            ensure(g.status, status_words * 8 + 64) ||
            ensure(g.ovf, fj_ovf_bytes((size_t)g.cus)) || ensure(g.ovf_base, (size_t)g.cus * 2 * FJ_GROUPS * 16 * 4) ||
            ensure(g.walk, (unit_bound + 1) * sizeof(FjWalkItem)))
            return -1;
        a0.cnt = (uint32_t *)g.cntR.p; a1.cnt = (uint32_t *)g.cntS.p;
        FusedArgs fa;
        fa.stash_cnt = (uint8_t *)g.stash_cnt.p; fa.stash_row = (uint64_t *)g.stash_row.p;
        fa.status = (uint64_t *)g.status.p + 8;
        fa.ticket = (uint32_t *)g.status.p;
        fa.nR = nR;
        fa.allow_resident = !g.no_resident; fa.radix_bits = (uint32_t)bits; fa.lr_mode = 0; fa.spec = 0; fa.xrows = nullptr;
        fa.unit_bound = unit_bound;
        fa.host_summary = (uint64_t *)g.pin;
        fa.dbg = nullptr;
        fa.ovf = (uint64_t *)g.ovf.p; fa.ovf_base = (uint32_t *)g.ovf_base.p; fa.walk = (FjWalkItem *)g.walk.p;
        const uint32_t fused_lds = LDS_BUDGET - FJ_LDS_EXTRA;
        if (use_ctx_out) {
            const uint64_t guess = (nR > nS ? nR : nS) + 1024;
            if (g.out.cap < guess * sizeof(rhj_result_tuple) && ensure(g.out, guess * sizeof(rhj_result_tuple))) return -1;
            out = (rhj_result_tuple *)g.out.p;
            out_capacity = g.out.cap / sizeof(rhj_result_tuple);
        }
        const uint32_t max_tiles = a0.tiles > a1.tiles ? a0.tiles : a1.tiles;
        const unsigned fgrid = (unsigned)(unit_bound < (uint64_t)g.cus ? unit_bound : (uint64_t)g.cus);
        bool small_done = false;
        uint64_t M = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            ja.out = out; ja.out_capacity = out ? out_capacity : 0;
            fa.j = ja;
            RHJ_STAGE(ST_HIST);
            // relations of one or two tiles: the scatter workgroups count the digits themselves, two launches in all
            const int self_hist = max_tiles <= SM_SELF_TILES;
            if (!self_hist)
                RHJ_LAUNCH(k_small_hist, dim3(max_tiles, 2), dim3(SM_BLOCK), 0, g.stream, a0, a1, bits, (uint64_t *)g.status.p, status_words);
            RHJ_STAGE(ST_SCATTER);
            RHJ_LAUNCH(k_small_scatter, dim3(max_tiles + 1, 2), dim3(SM_BLOCK), small_lds_bytes(bits), g.stream, a0, a1, bits, ps.hist,
                       ps.psum, pa, self_hist, (uint64_t *)g.status.p, status_words);
            RHJ_STAGE(ST_PROBE);
            if (nmin / bins <= 7000 && !g.no_resident)
                RHJ_LAUNCH((k_join_fused<true, false>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
            else
                RHJ_LAUNCH((k_join_fused<false, false>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
            RHJ_STAGE(ST_END);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(g.stream));
            plan = *hs;                                // written by the join kernel's last workgroup (system-scope stores)
            if (plan.fused_ok && out && ((const uint64_t *)g.pin)[sizeof(PlanSummary) / 8] != 0) {
                // some unit's pairs need the index walked again (a probe tuple with more than 16 matches, ...): the host is
                // waiting on this stream anyway, so the second kernel is launched only now — a launch that returns at once
                // costs a 0.08 ms join 5 % (the two-pass path, whose joins are milliseconds, always enqueues it)
                RHJ_LAUNCH(k_join_walk, dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
                RHJ_STAGE(ST_END);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(g.stream));
            }
            if (plan.fused_ok && plan.matches == FJ_NO_TOTAL) { fprintf(stderr, "rhj: fused join left no match total (chained scan incomplete)\n"); return -1; }
            if (!plan.fused_ok) { partitioned = true; break; }
            small_done = true;
            M = plan.matches;
            if (!use_ctx_out || M <= out_capacity) break;
            if (ensure(g.out, M * sizeof(rhj_result_tuple))) return -1;    // rare: fan-out above the guess
            out = (rhj_result_tuple *)g.out.p;
            out_capacity = M;
        }
        if (small_done) {
            *overflow = false;
            st.units = plan.units; st.hbm_units = 0; st.max_build = plan.max_build;
            st.reserved = 3;
            *matches = M;
            st.matches = M;
            if (ctx_out) *ctx_out = out;
            st.ms_hist = stage_ms(ST_HIST, ST_SCATTER);
            st.ms_scan = 0.f;                          // no scan launch: every scatter workgroup sums the columns it needs
            st.ms_scatter = stage_ms(ST_SCATTER, ST_PROBE);
            st.ms_probe = stage_ms(ST_PROBE, ST_END);
            st.ms_total = stage_ms(ST_HIST, ST_END);
            return (!use_ctx_out && out && M > out_capacity) ? 1 : 0;
        }
        if (use_ctx_out) { out = nullptr; out_capacity = 0; }
    }

    const uint64_t unit_bound = (uint64_t)bins + (nR + nS) / fused_span + 2;      // fused path: the host-side bound on the unit count
    if (!partitioned) {
        if (run_partition(ps, bits, 2, force_wide, want_fused && !force_wide)) return -1;
        RHJ_STAGE(ST_PLAN);
        if (want_fused) {                             // the plan clears the fused kernel's ticket and status words
            if (ensure(g.status, (unit_bound + 1) * 8 + 64)) return -1;
            pa.zero = (uint32_t *)g.status.p; pa.zero_words = (uint32_t)(((unit_bound + 1) * 8 + 64) / 4);
        }
        if (!ps.plan_done) RHJ_LAUNCH(k_plan, dim3(bits >= 11 ? 8 : 1), dim3(1024), 0, g.stream, pa, bits);
        else if (want_fused) HIP_TRY(hipMemsetAsync(g.status.p, 0, (unit_bound + 1) * 8 + 64, g.stream));
        pa.zero = nullptr; pa.zero_words = 0;
    } else {
        RHJ_STAGE(ST_HIST);
        RHJ_STAGE(ST_SCAN);
        RHJ_STAGE(ST_SCATTER);
        RHJ_STAGE(ST_PLAN);
        pa.span_lds = PR_UNIT;                         // plan again with tile-granular units
        RHJ_LAUNCH(k_plan, dim3(1), dim3(1024), 0, g.stream, pa, bits);
    }

    if (want_fused && !partitioned) {
        // ---- fused LDS path: build + probe + emit in one kernel, chained output offsets.  Launched
        // without waiting for the plan: the grid is the host-side upper bound on the unit count, the
        // LDS allocation the maximum, and the kernel itself returns when the plan found a bucket that
        // does not fit LDS (then the tiled path below takes over).  One host sync per join.
        if (ensure(g.stash_cnt, nR + nS + 64) || ensure(g.stash_row, (nR + nS + 8) * 8) ||
            ensure(g.status, (unit_bound + 1) * 8 + 64) ||
            ensure(g.ovf, fj_ovf_bytes((size_t)g.cus)) || ensure(g.ovf_base, (size_t)g.cus * 2 * FJ_GROUPS * 16 * 4) ||
            ensure(g.walk, (unit_bound + 1) * sizeof(FjWalkItem)))
            return -1;
        FusedArgs fa;
        fa.stash_cnt = (uint8_t *)g.stash_cnt.p; fa.stash_row = (uint64_t *)g.stash_row.p;
        fa.status = (uint64_t *)g.status.p + 8;               // words 0..7 hold the ticket
        fa.ticket = (uint32_t *)g.status.p;
        fa.nR = nR;
        fa.allow_resident = !g.no_resident; fa.radix_bits = (uint32_t)bits; fa.lr_mode = 0; fa.spec = 0; fa.xrows = nullptr;
        fa.unit_bound = unit_bound;
        fa.host_summary = nullptr;
        fa.dbg = nullptr;
        fa.ovf = (uint64_t *)g.ovf.p; fa.ovf_base = (uint32_t *)g.ovf_base.p; fa.walk = (FjWalkItem *)g.walk.p;
        if (g.stamps) {                                       // diagnostic runs only
            if (ensure(g.dbg, (unit_bound + 1) * 64)) return -1;
            fa.dbg = (uint64_t *)g.dbg.p;
        }
        uint64_t M = 0;
        const uint32_t fused_lds = LDS_BUDGET - FJ_LDS_EXTRA;
        if (use_ctx_out) {
            const uint64_t guess = (nR > nS ? nR : nS) + 1024;
            if (g.out.cap < guess * sizeof(rhj_result_tuple) && ensure(g.out, guess * sizeof(rhj_result_tuple))) return -1;
            out = (rhj_result_tuple *)g.out.p;
            out_capacity = g.out.cap / sizeof(rhj_result_tuple);
        }
        RHJ_STAGE(ST_PROBE);                          // (no separate build / count / offsets stages on this path)
        bool fused_done = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            ja.out = out; ja.out_capacity = out ? out_capacity : 0;
            fa.j = ja;
            if (attempt) HIP_TRY(hipMemsetAsync(g.status.p, 0, (unit_bound + 1) * 8 + 64, g.stream));   // (first: cleared by the plan)
            // workgroups are persistent (ticket loop) and the LDS request leaves room for one per CU
            const unsigned fgrid = (unsigned)(unit_bound < (uint64_t)g.cus ? unit_bound : (uint64_t)g.cus);
            // the resident variant only when an average bucket could fit beside the index (~7.4 K tuples)
            // The stash width follows the partition's row-id decision (summary->wide_row_ids, known only on the device):
            // the instantiation that does not match returns at once, and the 16-byte one is launched only when the
            // partition's 16-byte kernels were (one-pass partitions, forced-wide runs, a process that has seen wide row ids).
            const bool maybe_narrow = bits > PT_MAX_BITS && !force_wide;
            const bool maybe_wide = !maybe_narrow || ps.launch_wide;
            // The foreign-key speculation (k_join_spec): the bigger relation's tuples have one match each?  Then nothing is
            // stashed for the units that relation probes and nothing is chained.  Tried while it keeps holding; a failed try
            // costs the time to the first unit that notices (tens of microseconds), so after failures only every 16th join tries.
            // (A buffer below the relation's size is no obstacle: pairs beyond it are dropped as on every path and the caller hears
            // the count.  A rank's share of a sharded join — rhj_join_device_range — has a buffer for its share and buckets of the
            // whole join's size: the per-bucket rule is the same.)
            // (Not on a slice: a bucket cut between two devices has units that start inside it, and the last workgroup's check
            // adds up to the whole relation.)
            bool try_spec = attempt == 0 && maybe_narrow && out != nullptr && g.no_spec <= 0 && !g.ablate && !sliced &&
                            (nS >= nR ? nS : nR) / bins >= 4096;     // (units of 2.4 K tuples: 10M x 10M at 12 bits lost 7 % to its per-unit extras)
            if (try_spec && g.spec_score <= 0 && g.no_spec >= 0 && ++g.spec_skipped < 16) try_spec = false;   // (RHJ_NO_SPEC=-1: always try — to time a failing one)
            fa.spec = try_spec ? (nS >= nR ? 1u : 2u) : 0u;
            // k_join_exact (rhj_join_exact.hip.h) runs the speculation over an index that needs no verifying gather: build sides
            // the gather kernels would take (beyond the LDS-resident ones), enough radix bits for its slots to make the 40 stored
            // hash bits exact.  It hands over (ticket[4], reason in ticket[5]) what it does not take; a join it handed back for
            // its INPUT (row ids that do not increase, a bucket beyond its index) makes the next eligible joins skip it.
            bool try_exact = try_spec && g.no_exact <= 0 && !(nmin / bins <= 7000 && !g.no_resident) && bits >= 10 &&
                             nmin / bins <= XJ_MAX_BUILD;
            if (try_exact && g.exact_score <= 0 && g.no_exact >= 0 && ++g.exact_skipped < 16) try_exact = false;
            if (try_exact) {
                if (ensure(g.xrows, (size_t)g.cus * XJ_SCRATCH * 4)) return -1;
                fa.xrows = (uint32_t *)g.xrows.p;
                g.exact_skipped = 0;
            }
            if (try_spec) {
                g.spec_skipped = 0;
                if (try_exact)
                    RHJ_LAUNCH(k_join_exact, dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
                else if (nmin / bins <= 7000 && !g.no_resident)
                    RHJ_LAUNCH((k_join_spec<true>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
                else
                    RHJ_LAUNCH((k_join_spec<false>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
            }
            if (nmin / bins <= 7000 && !g.no_resident) {
                if (maybe_narrow)
                    RHJ_LAUNCH((k_join_fused<true, true>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
                if (maybe_wide)
                    RHJ_LAUNCH((k_join_fused<true, false>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
            } else {
                if (maybe_narrow)
                    RHJ_LAUNCH((k_join_fused<false, true>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
                if (maybe_wide)
                    RHJ_LAUNCH((k_join_fused<false, false>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
            }
            RHJ_LAUNCH(k_join_walk, dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);   // returns at once when no unit needs it
            RHJ_STAGE(ST_END);
            HIP_TRY(hipMemcpyAsync(hs, g.summary.p, sizeof(PlanSummary), hipMemcpyDeviceToHost, g.stream));
            uint32_t *spec_words = (uint32_t *)((char *)g.pin + 512);       // the ticket words: word 4 = the speculation failed
            if (try_spec) HIP_TRY(hipMemcpyAsync(spec_words, g.status.p, 32, hipMemcpyDeviceToHost, g.stream));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(g.stream));
            plan = *hs;
            g.last_spec = 0;
            g.last_exact = 0;
            if (try_spec) {
                const bool held = spec_words[4] == 0;
                const bool not_taken = try_exact && !held && spec_words[5] == 2;     // k_join_exact's input, not the hypothesis
                g.last_spec = held ? 1 : 2;
                if (try_exact) {
                    g.last_exact = held ? 1 : 2;
                    g.exact_score = not_taken ? g.exact_score - 2 : (g.exact_score < 4 ? g.exact_score + 1 : 4);
                    if (g.exact_score < -2) g.exact_score = -2;
                }
                if (!not_taken) {
                    g.spec_score = held ? (g.spec_score < 4 ? g.spec_score + 1 : 4) : g.spec_score - 2;
                    if (g.spec_score < -2) g.spec_score = -2;
                }
            }
            if (plan.row_id_overflow) { *overflow = true; return 0; }     // (also: wide row ids met the 12-byte kernels alone)
            if (plan.fused_ok && plan.matches == FJ_NO_TOTAL && !g.ablate) { fprintf(stderr, "rhj: fused join left no match total (chained scan incomplete)\n"); return -1; }
            if (!plan.fused_ok) {
                // a bucket needs the tiled path, which reads 16-byte tuples: partition again wide if this one was narrow
                if (bits > PT_MAX_BITS && !force_wide && !plan.wide_row_ids) { *overflow = true; return 0; }
                break;
            }
            fused_done = true;
            M = plan.matches;
            if (!use_ctx_out || M <= out_capacity) break;
            if (ensure(g.out, M * sizeof(rhj_result_tuple))) return -1;    // rare: fan-out above the guess
            out = (rhj_result_tuple *)g.out.p;
            out_capacity = M;
        }
        if (fused_done) {
            *overflow = plan.row_id_overflow != 0;
            st.units = plan.units; st.hbm_units = 0; st.max_build = plan.max_build;
            st.reserved = 1;
            *matches = M;
            st.matches = M;
            if (ctx_out) *ctx_out = out;
            st.ms_hist = stage_ms(ST_HIST, ST_SCAN);
            st.ms_scan = stage_ms(ST_SCAN, ST_SCATTER);
            st.ms_scatter = stage_ms(ST_SCATTER, ST_PLAN);
            st.ms_plan = stage_ms(ST_PLAN, ST_PROBE);
            st.ms_probe = stage_ms(ST_PROBE, ST_END);
            st.ms_total = stage_ms(ST_HIST, ST_END);
            return (!use_ctx_out && out && M > out_capacity) ? 1 : 0;
        }
        // some bucket needs an HBM table: plan again with tile-granular units
        pa.span_lds = PR_UNIT;
        RHJ_LAUNCH(k_plan, dim3(1), dim3(1024), 0, g.stream, pa, bits);
        if (use_ctx_out) { out = nullptr; out_capacity = 0; }
    }
    HIP_TRY(hipMemcpyAsync(hs, g.summary.p, sizeof(PlanSummary), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));                  // tiled path: launch geometry
    plan = *hs;
    st.units = plan.units; st.hbm_units = plan.build_units; st.max_build = plan.max_build;
    st.table_slots = plan.hbm_slots + plan.tab32_slots;

    if (ensure(g.tab32, max_tab32 * 4) || ensure(g.stash_cnt, nR + nS + 64) || ensure(g.stash_row, (nR + nS + 8) * 8)) return -1;
    ja.tab32 = (uint32_t *)g.tab32.p;
    ja.stash_cnt = (uint8_t *)g.stash_cnt.p; ja.stash_row = (uint64_t *)g.stash_row.p;
    RHJ_STAGE(ST_BUILD);
    if (plan.hbm_slots) {
        if (ensure(g.tab64, plan.hbm_slots * 8)) return -1;
        ja.tab64 = (uint64_t *)g.tab64.p;
        HIP_TRY(hipMemsetAsync(g.tab64.p, 0, plan.hbm_slots * 8, g.stream));
        RHJ_LAUNCH(k_build_hbm, dim3((unsigned)plan.build_units), dim3(256), 0, g.stream, ja,
                           (const Unit *)g.bunits.p);
    }
    if (plan.lds_buckets)
        RHJ_LAUNCH(k_build_lds, dim3((unsigned)plan.lds_buckets), dim3(BL_BLOCK), (size_t)plan.max_lds_slots * 4,
                           g.stream, ja, (const uint32_t *)g.ldsb.p);

    const unsigned probe_grid = (unsigned)((plan.units + 7) / 8 * 8);
    RHJ_STAGE(ST_COUNT);
    if (plan.units)
        RHJ_LAUNCH((k_probe<false>), dim3(probe_grid), dim3(PR_BLOCK), 0, g.stream, ja);
    RHJ_STAGE(ST_OFFSETS);
    if (launch_offsets((const uint64_t *)g.ucount.p, (uint64_t *)g.ubase.p,
                       (const uint64_t *)&((PlanSummary *)g.summary.p)->units, 0, plan.units,
                       &((PlanSummary *)g.summary.p)->matches))
        return -1;
    HIP_TRY(hipMemcpyAsync(hs, g.summary.p, sizeof(PlanSummary), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));                  // sync #2: match count -> output size
    *overflow = hs->row_id_overflow != 0;
    if (*overflow) return 0;                                  // the caller runs the join again with wide intermediates
    const uint64_t M = hs->matches;
    *matches = M;
    st.matches = M;

    int rc = 0;
    if (use_ctx_out) {
        if (ensure(g.out, (M ? M : 1) * sizeof(rhj_result_tuple))) return -1;
        out = (rhj_result_tuple *)g.out.p;
        out_capacity = M;
        if (ctx_out) *ctx_out = out;
    } else if (M > out_capacity) {
        rc = 1;
    }
    RHJ_STAGE(ST_PROBE);
    if (plan.units && M && out && out_capacity) {
        ja.out = out; ja.out_capacity = out_capacity;
        RHJ_LAUNCH((k_probe<true>), dim3(probe_grid), dim3(PR_BLOCK), 0, g.stream, ja);
    }
    RHJ_STAGE(ST_END);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g.stream));
    st.ms_hist = stage_ms(ST_HIST, ST_SCAN);
    st.ms_scan = stage_ms(ST_SCAN, ST_SCATTER);
    st.ms_scatter = stage_ms(ST_SCATTER, ST_PLAN);
    st.ms_plan = stage_ms(ST_PLAN, ST_BUILD);
    st.ms_build = stage_ms(ST_BUILD, ST_COUNT);
    st.ms_count = stage_ms(ST_COUNT, ST_OFFSETS);
    st.ms_offsets = stage_ms(ST_OFFSETS, ST_PROBE);
    st.ms_probe = stage_ms(ST_PROBE, ST_END);
    st.ms_total = stage_ms(ST_HIST, ST_END);
    return rc;
}

// ---- the low-radix path (rhj_lowradix.hip.h) ------------------------------------------------------------------------
// A canonical join on r <= 8 radix bits whose buckets' build sides are beyond the LDS index runs on r + k bits and is
// emitted in the order of r bits.  Returns 2 when the path does not apply or gave up (the caller takes the tiled path).
static int lowradix_sub_bits(int r, uint64_t nR, uint64_t nS)
{
    const uint64_t nmin = nR < nS ? nR : nS;
    if (r > PT_MAX_BITS || (nmin >> r) <= 33000) return 0;                // the fused path takes such buckets as they are (its LDS index holds 36 K build tuples)
    int k = 1;
    while (k < PT_MAX_BITS && r + k < MAX_BITS && (nmin >> (r + k)) > 20000) ++k;
    if ((nmin >> (r + k)) > 30000) return 0;                              // even 8 more bits leave the build sides too big
    return k;
}

int join_device_lr(const rhj_tuple *dR, uint64_t nR, const rhj_tuple *dS, uint64_t nS, rhj_result_tuple *out,
                   uint64_t out_capacity, bool use_ctx_out, rhj_result_tuple **ctx_out, uint64_t *matches, int kb)
{
    rhj_stats &st = g.stats;
    const int r = g.bits, T = r + kb;
    const uint32_t bins = 1u << T;
    PartState ps;
    if (ensure(g.partR, nR * sizeof(rhj_tuple)) || ensure(g.partS, nS * sizeof(rhj_tuple)) ||
        ensure(g.tmpR, nR * sizeof(rhj_tuple)) || ensure(g.tmpS, nS * sizeof(rhj_tuple)) || ensure(g.lr_words, 4096))
        return -1;
    ps.r[0] = RelArgs{dR, (rhj_tuple *)g.partR.p, nullptr, nR, 0, 0, nullptr, nullptr};
    ps.r[1] = RelArgs{dS, (rhj_tuple *)g.partS.p, nullptr, nS, 0, 0, nullptr, nullptr};
    ps.tmp[0] = (rhj_tuple *)g.tmpR.p; ps.tmp[1] = (rhj_tuple *)g.tmpS.p;
    for (int i = 0; i < 2; ++i) { ps.r[i].range_lo = g.range_lo; ps.r[i].range_span = g.range_span; ps.r[i].range_bits = (uint32_t)r; }   // (a share: the CALLER's buckets)
    ps.lo_bits = r;                                            // pass 1 on exactly the caller's bits: pass 2 reads in canonical order
    uint32_t *words = (uint32_t *)g.lr_words.p;                // word 0: a pass-2 tile / chunk beyond one batch; word 1: k_lr_emit's ticket
    uint8_t *parent_flip = (uint8_t *)(words + 16);            // [2 << r]
    HIP_TRY(hipMemsetAsync(words, 0, 64, g.stream));
    if (run_partition(ps, T, 2, false, true)) return -1;
    RHJ_STAGE(ST_PLAN);
    RHJ_LAUNCH(k_lr_parent, dim3(1u << r), dim3(256), 0, g.stream, (const uint64_t *)ps.hist, (const uint64_t *)(ps.hist + bins), r, kb, parent_flip);

    const uint32_t build_chunk = 4096;
    uint32_t lds_cap = (LDS_BUDGET - FJ_LDS_EXTRA - 128) * 2 / 9;
    if (lds_cap > 65534) lds_cap = 65534;
    const uint64_t nmin = nR < nS ? nR : nS;
    const uint64_t max_units = (uint64_t)bins + (nR + nS) / PR_UNIT + 2;
    const uint64_t max_bunits = (uint64_t)bins + nmin / build_chunk + 2;
    const uint64_t unit_bound = (uint64_t)bins + (nR + nS) / FJ_SPAN + 2;
    if (ensure(g.units, max_units * sizeof(Unit)) || ensure(g.bunits, max_bunits * sizeof(Unit)) || ensure(g.ldsb, (size_t)bins * 4) ||
        ensure(g.meta, (size_t)bins * sizeof(BucketMeta)) || ensure(g.summary, sizeof(PlanSummary) + 64) ||
        ensure(g.ucount, max_units * 8) || ensure(g.ubase, max_units * 8) || ensure(g.uflag, max_units * 4) ||
        ensure(g.stash_cnt, nR + nS + 64) || ensure(g.stash_row, (nR + nS + 8) * 8) || ensure(g.status, (unit_bound + 1) * 8 + 64) ||
        ensure(g.ovf, fj_ovf_bytes((size_t)g.cus)) || ensure(g.ovf_base, (size_t)g.cus * 2 * FJ_GROUPS * 16 * 4) ||
        ensure(g.walk, (unit_bound + 1) * sizeof(FjWalkItem)))
        return -1;
    PlanArgs pa;
    pa.histR = ps.hist; pa.histS = ps.hist + bins;
    pa.units = (Unit *)g.units.p; pa.build_units = (Unit *)g.bunits.p; pa.lds_buckets = (uint32_t *)g.ldsb.p;
    pa.meta = (BucketMeta *)g.meta.p; pa.summary = (PlanSummary *)g.summary.p;
    pa.lds_cap = lds_cap; pa.lds_max_slots = LDS_BUDGET / 4 / 4 * 4; pa.build_chunk = build_chunk; pa.span_lds = FJ_SPAN;
    pa.parent_mask = (1u << r) - 1u; pa.parent_flip = parent_flip; pa.zero = nullptr; pa.zero_words = 0;
    RHJ_LAUNCH(k_plan, dim3(T >= 11 ? 8 : 1), dim3(1024), 0, g.stream, pa, T);

    JoinArgs ja;
    ja.partR = (const rhj_tuple *)g.partR.p; ja.partS = (const rhj_tuple *)g.partS.p;
    ja.histR = ps.hist; ja.histS = ps.hist + bins; ja.psumR = ps.psum; ja.psumS = ps.psum + bins;
    ja.units = (const Unit *)g.units.p; ja.meta = (const BucketMeta *)g.meta.p; ja.summary = (const PlanSummary *)g.summary.p;
    ja.tab32 = nullptr; ja.tab64 = nullptr;
    ja.unit_count = (uint64_t *)g.ucount.p; ja.unit_base = (const uint64_t *)g.ubase.p; ja.unit_flag = (uint32_t *)g.uflag.p;
    ja.ablate = 0; ja.parent_mask = (1u << r) - 1u; ja.parent_flip = parent_flip;
    ja.stash_cnt = nullptr; ja.stash_row = nullptr; ja.stash_nR = nR;
    FusedArgs fa;
    fa.stash_cnt = (uint8_t *)g.stash_cnt.p; fa.stash_row = (uint64_t *)g.stash_row.p;
    fa.status = (uint64_t *)g.status.p + 8; fa.ticket = (uint32_t *)g.status.p;
    fa.nR = nR; fa.allow_resident = 0; fa.radix_bits = (uint32_t)T; fa.lr_mode = 1; fa.spec = 0; fa.xrows = nullptr;
    fa.unit_bound = unit_bound; fa.host_summary = nullptr; fa.dbg = nullptr;
    fa.ovf = (uint64_t *)g.ovf.p; fa.ovf_base = (uint32_t *)g.ovf_base.p; fa.walk = (FjWalkItem *)g.walk.p;
    const uint32_t fused_lds = LDS_BUDGET - FJ_LDS_EXTRA;
    const unsigned fgrid = (unsigned)(unit_bound < (uint64_t)g.cus ? unit_bound : (uint64_t)g.cus);
    const bool count_only = !use_ctx_out && out == nullptr;
    struct Back { PlanSummary p; uint32_t ticket[4]; uint32_t words[4]; } *hb = (Back *)g.pin;
    uint64_t M = 0;
    RHJ_STAGE(ST_PROBE);
    for (int attempt = 0; attempt < 2; ++attempt) {
        // the internal join's pairs: a scratch list of their own (the emit pass reads it while it writes the caller's)
        if (!count_only) {
            const uint64_t guess = attempt ? M : (nR > nS ? nR : nS) + 1024;
            if (ensure(g.lr_tmp, guess * sizeof(rhj_result_tuple))) return -1;
        }
        ja.out = count_only ? nullptr : (rhj_result_tuple *)g.lr_tmp.p;
        ja.out_capacity = count_only ? 0 : g.lr_tmp.cap / sizeof(rhj_result_tuple);
        fa.j = ja;
        HIP_TRY(hipMemsetAsync(g.status.p, 0, (unit_bound + 1) * 8 + 64, g.stream));
        HIP_TRY(hipMemsetAsync(g.stash_cnt.p, 0, nR + nS, g.stream));      // probe tuples of sub-buckets without a build side match nothing
        RHJ_LAUNCH((k_join_fused<false, true>), dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
        RHJ_LAUNCH(k_join_walk, dim3(fgrid), dim3(FJ_BLOCK), fused_lds, g.stream, fa, fused_lds);
        HIP_TRY(hipMemcpyAsync(&hb->p, g.summary.p, sizeof(PlanSummary), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(hb->ticket, g.status.p, 16, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(hb->words, words, 16, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g.stream));
        // what this path refuses: wide row ids, a build side beyond the LDS index, a unit that needed the index walk (its
        // tuples' pairs are not where their stash rows say), a pass-2 tile of several batches, 2^32 pairs or more
        if (hb->p.wide_row_ids || hb->p.row_id_overflow || !hb->p.fused_ok || hb->ticket[2] != 0 || hb->words[0] != 0 ||
            hb->p.matches == FJ_NO_TOTAL || hb->p.matches >= (1ull << 32)) {
            static const bool trace = getenv("RHJ_TRACE") != nullptr;
            if (trace) fprintf(stderr, "rhj-trace:   low-radix path gives up: wide ids %u/%u, fused_ok %llu, walk units %u, big tile %u, matches %llu\n",
                               hb->p.wide_row_ids, hb->p.row_id_overflow, (unsigned long long)hb->p.fused_ok, hb->ticket[2], hb->words[0],
                               (unsigned long long)hb->p.matches);
            return 2;
        }
        M = hb->p.matches;
        if (count_only || M * sizeof(rhj_result_tuple) <= g.lr_tmp.cap) break;
    }
    st.units = hb->p.units; st.hbm_units = 0; st.max_build = hb->p.max_build;
    st.reserved = 4;                                           // path id: low-radix
    *matches = M;
    st.matches = M;
    int rc = 0;
    if (!count_only && M) {
        if (use_ctx_out) {
            if (ensure(g.out, M * sizeof(rhj_result_tuple))) return -1;
            out = (rhj_result_tuple *)g.out.p;
            out_capacity = M;
            if (ctx_out) *ctx_out = out;
        } else if (M > out_capacity) {
            rc = 1;
        }
        LrArgs la;
        la.j = ja; la.j.out = out; la.j.out_capacity = out_capacity;
        la.p2R = ps.p2[0]; la.p2S = ps.p2[1];
        la.stash_cnt = (const uint8_t *)g.stash_cnt.p; la.stash_row = (const uint2 *)g.stash_row.p;
        la.tmp = (const uint4 *)g.lr_tmp.p;
        la.nR = nR; la.r_bits = (uint32_t)r; la.k_bits = (uint32_t)kb;
        la.slots_per_bucket = ps.p2[0].groups > ps.p2[1].groups ? ps.p2[0].groups : ps.p2[1].groups;
        uint32_t search0 = 1;
        while (search0 * 2 <= ps.p2[0].group) search0 *= 2;
        la.search0 = search0;
        const uint32_t nslots = la.slots_per_bucket << r;
        if (ensure(g.lr_status, (size_t)nslots * 8 + 64)) return -1;
        la.ctotal = (uint64_t *)g.lr_status.p;
        la.bad = words;
        RHJ_LAUNCH(k_lr_totals, dim3(nslots), dim3(256), 0, g.stream, la);
        if (launch_offsets((const uint64_t *)g.lr_status.p, (uint64_t *)g.lr_status.p, nullptr, nslots, nslots,
                           (uint64_t *)(words + 4)))
            return -1;
        const unsigned egrid = (unsigned)(nslots < (uint32_t)g.cus * 8u ? nslots : (uint32_t)g.cus * 8u);
        RHJ_LAUNCH(k_lr_emit, dim3(egrid), dim3(LR_BLOCK), lr_lds_bytes(kb), g.stream, la, nslots);
        RHJ_STAGE(ST_END);
        HIP_TRY(hipMemcpyAsync(hb->words, words, 16, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g.stream));
        if (hb->words[0] != 0) return 2;
    } else {
        RHJ_STAGE(ST_END);
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    st.radix_bits = r;
    st.ms_hist = stage_ms(ST_HIST, ST_SCAN);
    st.ms_scan = stage_ms(ST_SCAN, ST_SCATTER);
    st.ms_scatter = stage_ms(ST_SCATTER, ST_PLAN);
    st.ms_plan = stage_ms(ST_PLAN, ST_PROBE);
    st.ms_probe = stage_ms(ST_PROBE, ST_END);
    st.ms_total = stage_ms(ST_HIST, ST_END);
    return rc;
}

static int auto_radix_bits(uint64_t nR, uint64_t nS)
{
    const uint64_t nmin = nR < nS ? nR : nS, nmax = nR < nS ? nS : nR;
    const uint64_t target = nmax >= 4 * nmin ? 6500 : 16000;
    int b = 0;
    while (b < MAX_BITS && (nmin >> b) > target) ++b;
    while (b < PT_MAX_BITS && (nmin >> (b + 1)) >= 512) ++b;
    // 13 bits only where 12 would leave buckets beyond the gather kernels' LDS index: up to 12 bits pass 1 counts pass 2's digits
    // itself, and since the speculative kernel writes every unit's pairs from phase 1 (round 4) 100M x 100M is faster on 12 bits
    // (24.4 K a bucket: 4.13 ms) than on 13 (4.27) or 14 (4.54); tools/exp_auto_bits.py: 10M / 30M / 50M best on 10 / 11 / 12
    if (b == 13 && nmax < 4 * nmin && (nmin >> 12) <= 28000) b = 12;
    return b < 1 ? 1 : b;
}

static int join_device_radix(const rhj_tuple *dR, uint64_t nR, const rhj_tuple *dS, uint64_t nS, rhj_result_tuple *out,
                             uint64_t out_capacity, bool use_ctx_out, rhj_result_tuple **ctx_out, uint64_t *matches);

int join_device(const rhj_tuple *dR, uint64_t nR, const rhj_tuple *dS, uint64_t nS, rhj_result_tuple *out,
                uint64_t out_capacity, bool use_ctx_out, rhj_result_tuple **ctx_out, uint64_t *matches)
{
    if (!g.order_any || nR == 0 || nS == 0) return join_device_radix(dR, nR, dS, nS, out, out_capacity, use_ctx_out, ctx_out, matches);
    const int callers = g.bits;
    g.bits = auto_radix_bits(nR, nS);
    const int rc = join_device_radix(dR, nR, dS, nS, out, out_capacity, use_ctx_out, ctx_out, matches);
    g.bits = callers;
    return rc;
}

static int join_device_radix(const rhj_tuple *dR, uint64_t nR, const rhj_tuple *dS, uint64_t nS, rhj_result_tuple *out,
                             uint64_t out_capacity, bool use_ctx_out, rhj_result_tuple **ctx_out, uint64_t *matches)
{
    bool overflow = false;
    {
        // few radix bits over big inputs, canonical order wanted: run on finer buckets, emit in the caller's order
        // (also for a rank's share of a sharded join — the partition's first pass, on the caller's bits, drops the other buckets —
        // but not for a share cut inside a bucket: that is a matter of the plan's units, and these are sub-buckets)
        const int kb = (!g.no_lowradix && !g.no_fused && !g.force_hbm && !g.wide_row_ids && !g.slice_skip && !g.slice_end && nR < (1ull << 32) && nS < (1ull << 32))
                           ? lowradix_sub_bits(g.bits, nR, nS) : 0;
        if (kb) {
            if (ctx_init()) return -1;
            const float keep_h2d = g.stats.ms_h2d;
            memset(&g.stats, 0, sizeof(g.stats));
            g.stats.ms_h2d = keep_h2d;
            g.stats.n_r = nR; g.stats.n_s = nS; g.stats.radix_bits = g.bits;
            *matches = 0;
            if (ctx_out) *ctx_out = nullptr;
            const int rc3 = join_device_lr(dR, nR, dS, nS, out, out_capacity, use_ctx_out, ctx_out, matches, kb);
            if (rc3 != 2) return rc3;
        }
    }
    int rc = join_device_once(dR, nR, dS, nS, out, out_capacity, use_ctx_out, ctx_out, matches, g.wide_row_ids != 0, &overflow);
    if (rc >= 0 && overflow) g.seen_wide = 1;
    if (rc >= 0 && overflow)
        rc = join_device_once(dR, nR, dS, nS, out, out_capacity, use_ctx_out, ctx_out, matches, true, &overflow);
    return rc;
}

// k_filter_write is grid-stride, one wave per two tiles: enough workgroups to fill the chip a few times over
unsigned filter_write_grid(uint64_t tiles)
{
    const uint64_t want = (tiles + 7) / 8, cap = (uint64_t)g.cus * 32;
    return (unsigned)(want < cap ? want : cap);
}

int op_code(char op)
{
    return op == '<' ? 0 : op == '>' ? 1 : op == '=' ? 2 : -1;
}

// masks + tile counts -> the ascending index list and the hit total (both filters end here)
int filter_write_out(uint64_t n, uint64_t tiles, uint64_t *total, uint64_t *d_out, uint64_t *hits)
{
    if (tiles <= FILTER_SELF_TILES) {                  // the write waves sum the tile counts themselves, the total lands in pinned memory:
        RHJ_LAUNCH((k_filter_write<true>), dim3(filter_write_grid(tiles)), dim3(256), 0, g.stream, n, (const uint64_t *)g.fmask.p,   // no scan
                   (const uint64_t *)g.ftile.p, d_out, (unsigned long long *)g.pin);                                                  // launches, no copy
        RHJ_STAGE(ST_END);
    } else {
        if (launch_offsets((const uint64_t *)g.ftile.p, (uint64_t *)g.fbase.p, nullptr, tiles, tiles, total)) return -1;
        RHJ_LAUNCH((k_filter_write<false>), dim3(filter_write_grid(tiles)), dim3(256), 0, g.stream, n, (const uint64_t *)g.fmask.p,
                   (const uint64_t *)g.fbase.p, d_out, (unsigned long long *)nullptr);
        RHJ_STAGE(ST_END);
        HIP_TRY(hipMemcpyAsync(g.pin, total, 8, hipMemcpyDeviceToHost, g.stream));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g.stream));
    *hits = *(volatile uint64_t *)g.pin;
    return 0;
}

int filter_device(const uint64_t *d_col, const uint64_t *d_sel, uint64_t n, char op, uint64_t value, uint64_t *d_out,
                  bool use_ctx_out, uint64_t **ctx_out, uint64_t *hits)
{
    if (ctx_init()) return -1;
    *hits = 0;
    if (ctx_out) *ctx_out = nullptr;
    const int oc = op_code(op);
    if (oc < 0) return -3;
    if (n == 0) return 0;
    const uint64_t tiles = (n + FILTER_TILE - 1) / FILTER_TILE;
    if (ensure(g.fmask, ((n + 63) / 64 + FILTER_ROUNDS * 8 + 8) * 8) || ensure(g.ftile, tiles * 8) ||
        ensure(g.fbase, tiles * 8) || ensure(g.summary, sizeof(PlanSummary)))
        return -1;
    if (use_ctx_out) {
        if (ensure(g.fout, n * 8)) return -1;
        d_out = (uint64_t *)g.fout.p;
        if (ctx_out) *ctx_out = d_out;
    }
    uint64_t *total = &((PlanSummary *)g.summary.p)->matches;
    RHJ_STAGE(ST_HIST);
    RHJ_LAUNCH(k_filter_mask, dim3((unsigned)tiles), dim3(256), 0, g.stream, d_col, d_sel, n, oc, value,
                       (uint64_t *)g.fmask.p, (uint64_t *)g.ftile.p);
    if (filter_write_out(n, tiles, total, d_out, hits)) return -1;
    memset(&g.stats, 0, sizeof(g.stats));
    g.stats.n_r = n; g.stats.matches = *hits;
    g.stats.ms_total = g.stats.ms_probe = stage_ms(ST_HIST, ST_END);
    return 0;
}

int filter_eq2_device(const uint64_t *colA, const uint64_t *selA, const uint64_t *colB, const uint64_t *selB, uint64_t n,
                      uint64_t *d_out, uint64_t *hits)
{
    if (ctx_init()) return -1;
    *hits = 0;
    if (n == 0) return 0;
    const uint64_t tiles = (n + FILTER_TILE - 1) / FILTER_TILE;
    if (ensure(g.fmask, ((n + 63) / 64 + FILTER_ROUNDS * 8 + 8) * 8) || ensure(g.ftile, tiles * 8) ||
        ensure(g.fbase, tiles * 8) || ensure(g.summary, sizeof(PlanSummary)))
        return -1;
    uint64_t *total = &((PlanSummary *)g.summary.p)->matches;
    RHJ_STAGE(ST_HIST);
    RHJ_LAUNCH(k_filter_mask_eq2, dim3((unsigned)tiles), dim3(256), 0, g.stream, colA, selA, colB, selB, n,
               (uint64_t *)g.fmask.p, (uint64_t *)g.ftile.p);
    if (filter_write_out(n, tiles, total, d_out, hits)) return -1;
    memset(&g.stats, 0, sizeof(g.stats));
    g.stats.n_r = n; g.stats.matches = *hits;
    g.stats.ms_total = g.stats.ms_probe = stage_ms(ST_HIST, ST_END);
    return 0;
}

// stable selection of the tuples whose bucket lies in [bucket_lo, bucket_hi) (on the calling thread's context, no API lock)
static int select_range(const rhj_tuple *d_in, uint64_t n, uint32_t bucket_lo, uint32_t bucket_hi, rhj_tuple *d_out, uint64_t capacity,
                        uint64_t *count)
{
    if (ctx_init()) return -1;
    *count = 0;
    if (n == 0 || bucket_hi <= bucket_lo) return 0;
    const uint32_t mask = (1u << g.bits) - 1u;
    const uint64_t tiles = (n + SH_TILE - 1) / SH_TILE;
    if (ensure(g.ftile, tiles * 8) || ensure(g.fbase, tiles * 8) || ensure(g.summary, sizeof(PlanSummary))) return -1;
    uint64_t *total = &((PlanSummary *)g.summary.p)->matches;
    RHJ_LAUNCH(k_select_count, dim3((unsigned)tiles), dim3(SH_BLOCK), 0, g.stream, d_in, n, mask, bucket_lo, bucket_hi,
               (uint64_t *)g.ftile.p);
    if (launch_offsets((const uint64_t *)g.ftile.p, (uint64_t *)g.fbase.p, nullptr, tiles, tiles, total)) return -1;
    RHJ_LAUNCH(k_select_write, dim3((unsigned)tiles), dim3(SH_BLOCK), 0, g.stream, d_in, n, mask, bucket_lo, bucket_hi,
               (const uint64_t *)g.fbase.p, d_out, capacity);
    HIP_TRY(hipMemcpyAsync(g.pin, total, 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g.stream));
    *count = *(uint64_t *)g.pin;
    return *count > capacity ? 1 : 0;
}

// (on the calling thread's context, without the API lock: the public entry below, and the per-device workers of a multi-device join)
static int join_range(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS, uint32_t bucket_lo, uint32_t bucket_hi,
                      rhj_result_tuple *d_out, uint64_t out_capacity, bool use_ctx_out, rhj_result_tuple **ctx_out, uint64_t *matches,
                      uint64_t first_skip = 0, uint64_t last_end = 0)
{
    if (matches) *matches = 0;
    if (ctx_out) *ctx_out = nullptr;
    if (g.order_any) { fprintf(stderr, "rhj_join_device_range: bucket numbers belong to the caller's radix; not with RHJ_ORDER=any\n"); return -3; }
    const uint32_t bins = 1u << g.bits;
    if (bucket_hi > bins) bucket_hi = bins;
    if (bucket_lo >= bucket_hi) return 0;                     // an empty range joins nothing
    uint64_t m = 0;
    const bool whole = bucket_lo == 0 && bucket_hi == bins && !first_skip && !last_end;
    g.range_lo = bucket_lo;
    g.range_span = whole ? 0u : bucket_hi - bucket_lo;        // the whole radix is the ordinary join (small path and all)
    g.slice_skip = first_skip; g.slice_end = last_end;
    const int rc = join_device(d_R, nR, d_S, nS, d_out, out_capacity, use_ctx_out, ctx_out, &m);
    g.range_lo = 0; g.range_span = 0; g.slice_skip = 0; g.slice_end = 0;
    if (matches) *matches = m;
    return rc;
}

// ---- several devices behind the C interface (RHJ_DEVICES; 8e: bucket b of R only meets bucket b of S) ------------
// The knobs live in the library's own context; the others take them over at the start of every multi-device call.
static void adopt_knobs(Ctx &d, const Ctx &s)
{
    d.bits = s.bits; d.null_on_empty = s.null_on_empty; d.force_hbm = s.force_hbm; d.ablate = s.ablate; d.order_any = s.order_any;
    d.no_fused = s.no_fused; d.force_fused = s.force_fused; d.no_resident = s.no_resident; d.wide_row_ids = s.wide_row_ids;
    d.timing = s.timing; d.no_count_in_pass1 = s.no_count_in_pass1; d.no_spec = s.no_spec; d.no_exact = s.no_exact;
    d.lo_override = s.lo_override; d.msd = s.msd; d.no_lowradix = s.no_lowradix; d.no_small = s.no_small; d.small_tiles = s.small_tiles;
    d.node_pairs = s.node_pairs;
}

static int set_devices(int n)
{
    if (n < 1 || n > MAX_DEVICES) return -1;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return -1;
    const int base = g_all[0].device;
    if (!g_same_device && base + n > count) {
        fprintf(stderr, "rhj_set_devices(%d): devices %d..%d asked for, %d visible\n", n, base, base + n - 1, count);
        return -1;
    }
    for (int d = 1; d < n; ++d) {
        const int ordinal = g_same_device ? base : base + d;
        if (g_all[d].ready && g_all[d].device != ordinal) return -1;          // (a context does not move to another device)
        g_all[d].device = ordinal;
    }
    if (!g_same_device)
        for (int a = 0; a < n; ++a)                                           // best effort: peer copies of pair lists go direct over xGMI
            for (int b = 0; b < n; ++b)
                if (a != b && hipSetDevice(base + a) == hipSuccess && hipDeviceEnablePeerAccess(base + b, 0) != hipSuccess) (void)hipGetLastError();
    (void)hipSetDevice(base);
    g_ndev = n;
    return 0;
}

static int devices_ready()
{
    if (g_ndev_env > 1 && g_ndev == 1) { const int want = g_ndev_env; g_ndev_env = 0; return set_devices(want); }
    return 0;
}

// the d-th of n contiguous bucket ranges of equal width (shard.equal_ranges: no histogram, no read-back)
static inline uint32_t range_cut(uint32_t bins, int n, int d) { return (uint32_t)((uint64_t)bins * (uint64_t)d / (uint64_t)n); }

// Runs fn(d) for every device d of the set on a thread of its own whose current context is device d's (d = 0 included: the
// calling thread holds the API lock and only waits).  One device: on the calling thread.
template <class F> static void on_devices(int n, F fn)
{
    if (n == 1) { fn(0); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)n);
    for (int d = 0; d < n; ++d)
        pool.emplace_back([d, &fn]() {
            g_cur = &g_all[d];
            if (d) adopt_knobs(g_all[d], g_all[0]);
            try { fn(d); } catch (...) { }             // (nothing may leave a thread; fn records its own result, preset to failure)
        });
    for (auto &t : pool) t.join();
}

}  // namespace

// ------------------------------------------------------------------ C-ABI

extern "C" {

// (rhj_api_lock / rhj_api_unlock: rhj_host.cpp)

int rhj_set_radix_bits(int bits)
{
    RhjApiLock api_lock;
    if (bits < 1 || bits > MAX_BITS) return -1;
    g.bits = bits;
    return 0;
}
int rhj_get_radix_bits(void) { return g.bits; }
void rhj_set_empty_mode(int null_on_empty) { g.null_on_empty = null_on_empty; }
void rhj_set_node_pairs(uint64_t pairs) { g.node_pairs = pairs; }
void rhj_set_force_hbm_table(int on) { g.force_hbm = on; }
void rhj_set_fused(int on) { g.no_fused = !on; g.force_fused = on >= 2; }
void rhj_set_resident(int on) { g.no_resident = !on; }
void rhj_set_small(int on) { g.no_small = !on; }
void rhj_set_lowradix(int on) { g.no_lowradix = !on; }
void rhj_set_spec(int on) { g.no_spec = !on; g.spec_score = 2; g.spec_skipped = 0; }
int rhj_last_spec(void) { return g.last_spec; }
void rhj_set_exact(int on) { g.no_exact = !on; g.exact_score = 2; g.exact_skipped = 0; }
int rhj_last_exact(void) { return g.last_exact; }
void rhj_set_count_in_pass1(int on) { g.no_count_in_pass1 = !on; }
void rhj_set_order(int any) { g.order_any = any != 0; }
int rhj_auto_radix_bits(uint64_t nR, uint64_t nS) { return auto_radix_bits(nR, nS); }
int rhj_get_order(void) { return g.order_any; }
void rhj_set_timing(int level) { g.timing = level < 0 ? 0 : level > 2 ? 2 : level; }
/* diagnostic: copy the per-unit phase stamps of the last fused run (RHJ_STAMPS=1) */
int rhj_debug_stamps(uint64_t *host, uint64_t units)
{
    RhjApiLock api_lock;
    if (!g.dbg.p) return -1;
    return hipMemcpy(host, g.dbg.p, units * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#ifdef RHJ_INSTRUMENT
/* diagnostics build only: per-workgroup start / end stamps of the small path's two partition kernels */
int rhj_debug_small_stamps(uint64_t *host)
{
    RhjApiLock api_lock;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sm_dbg), sizeof(uint64_t) * 4 * 2048) == hipSuccess ? 0 : -1;
}
/* diagnostics build only: phase stamps of pass 2's workgroups (tools/exp_sr_stamps.py) */
int rhj_debug_sr_stamps(uint64_t *host)
{
    RhjApiLock api_lock;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sr_dbg), sizeof(uint64_t) * 256 * 16) == hipSuccess ? 0 : -1;
}
/* diagnostics build only: rate of random 16-byte gathers from per-workgroup regions (tools/gather_bench.py) */
int rhj_debug_gather_bench(uint32_t region_elems, uint32_t rounds, uint32_t stream_per_round, uint32_t wgs, float *ms)
{
    RhjApiLock api_lock;
    if (ctx_init()) return -1;
    const size_t reg_bytes = (size_t)wgs * region_elems * 16, str_bytes = (size_t)wgs * rounds * (stream_per_round ? stream_per_round : 1) * FJ_BLOCK * 16;
    void *reg = nullptr, *str = nullptr, *sink = nullptr;
    if (hipMalloc(&reg, reg_bytes) != hipSuccess || hipMalloc(&str, str_bytes) != hipSuccess || hipMalloc(&sink, FJ_BLOCK * 16) != hipSuccess) return -1;
    HIP_TRY(hipMemsetAsync(reg, 1, reg_bytes, g.stream));
    HIP_TRY(hipMemsetAsync(str, 2, str_bytes, g.stream));
    for (int rep = 0; rep < 3; ++rep) {
        RHJ_STAGE(ST_HIST);
        RHJ_LAUNCH(k_gather_bench, dim3(wgs), dim3(FJ_BLOCK), 0, g.stream, (const rhj_tuple *)reg, region_elems, rounds,
                   (const uint4 *)str, stream_per_round, (uint4 *)sink);
        RHJ_STAGE(ST_END);
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    *ms = stage_ms(ST_HIST, ST_END);
    (void)hipFree(reg); (void)hipFree(str); (void)hipFree(sink);
    return 0;
}
#endif
int rhj_set_device(int ordinal)
{
    RhjApiLock api_lock;
    if (g.ready) return -1;
    g.device = ordinal;
    return 0;
}
int rhj_get_device(void) { return g.device; }
void rhj_set_stream(void *s)
{
    RhjApiLock api_lock;
    // s is a hipStream_t; NULL is HIP's default (null) stream, which is what PyTorch's default
    // stream is: work the caller queued there is ordered before ours.  Until this is called the
    // library launches on a non-blocking stream of its own.
    if (g.own_stream && g.stream) { (void)hipStreamDestroy(g.stream); g.own_stream = false; }
    g.stream = (hipStream_t)s;
    g.stream_set = true;
}
const rhj_stats *rhj_last_stats(void) { return &g.stats; }
const char *rhj_version(void) { return "rhj-mi355x 0.1 (gfx950)"; }

int rhj_join_device(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS, rhj_result_tuple *d_out,
                    uint64_t out_capacity, uint64_t *matches)
{
    RhjApiLock api_lock;
    uint64_t m = 0;
    const int rc = join_device(d_R, nR, d_S, nS, d_out, out_capacity, false, nullptr, &m);
    if (matches) *matches = m;
    return rc;
}

/* The join of two relations given as KEY COLUMNS: tuple i of a relation is {keys[i], i} — what GetRelation makes of a base
 * relation (inter_res.c:199-204, :223-227: row_id = i) — without materialising the 16-byte tuples: on the two-pass partition
 * (9..15 radix bits) pass 1 reads the columns themselves, 8 bytes a tuple instead of 16 (the north_star's "coalesced uint64
 * column loads"); on every other path the tuples are built first (one streaming kernel) and the ordinary join runs.  Same
 * pairs, same order as rhj_join_device on the materialised relations. */
int rhj_join_keys_device(const uint64_t *d_keysR, uint64_t nR, const uint64_t *d_keysS, uint64_t nS, rhj_result_tuple *d_out,
                         uint64_t out_capacity, uint64_t *matches)
{
    RhjApiLock api_lock;
    uint64_t m = 0;
    if (matches) *matches = 0;
    if (nR >= (1ull << 32) || nS >= (1ull << 32)) return -2;
    int rc;
    const bool lowradix = !g.no_lowradix && lowradix_sub_bits(g.bits, nR, nS) != 0;
    if (!g.order_any && g.bits > PT_MAX_BITS && !lowradix && !g.wide_row_ids && !g.no_fused && !g.force_hbm && nR && nS) {
        g.cols_input = 1;
        rc = join_device((const rhj_tuple *)d_keysR, nR, (const rhj_tuple *)d_keysS, nS, d_out, out_capacity, false, nullptr, &m);
        g.cols_input = 0;
    } else {
        if (ctx_init() || ensure(g.inR, (nR ? nR : 1) * sizeof(rhj_tuple)) || ensure(g.inS, (nS ? nS : 1) * sizeof(rhj_tuple))) return -1;
        if (rhj_build_relation_device(d_keysR, nullptr, nR, (rhj_tuple *)g.inR.p) || rhj_build_relation_device(d_keysS, nullptr, nS, (rhj_tuple *)g.inS.p)) return -1;
        rc = join_device((const rhj_tuple *)g.inR.p, nR, (const rhj_tuple *)g.inS.p, nS, d_out, out_capacity, false, nullptr, &m);
    }
    if (matches) *matches = m;
    return rc;
}

/* One rank's share of a join sharded by bucket range (bucket b of R only meets bucket b of S, rhjoin.c:42-57):
 * the canonical result restricted to the buckets [bucket_lo, bucket_hi) of the current radix.  The first partition pass
 * drops the other buckets' tuples while it reads the relations — one read of each relation, no selection pass and no
 * host round trip in front of the join; the concatenation of the ranks' results in range order is the canonical result. */
int rhj_join_device_range(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS, uint32_t bucket_lo, uint32_t bucket_hi,
                          rhj_result_tuple *d_out, uint64_t out_capacity, uint64_t *matches)
{
    RhjApiLock api_lock;
    return join_range(d_R, nR, d_S, nS, bucket_lo, bucket_hi, d_out, out_capacity, false, nullptr, matches);
}

/* A share cut INSIDE buckets (a hot bucket joined by several devices): the buckets [bucket_lo, bucket_hi) as above, but of the
 * first one only the probe tuples from first_skip on take part and of the last one (bucket_hi - 1) only those before last_end
 * (0: all of them) — positions among the tuples of the bucket's probe side (R when |R_b| >= |S_b|, rhjoin.c:86) in partition
 * order; the build side of a cut bucket is whole on every device that has a piece of it.  The result is the canonical list's
 * pairs of exactly those probe tuples (rhjoin.c:141-217 walks a bucket's probe tuples in order), so shares that tile the
 * (bucket, probe position) space concatenate to the canonical result. */
int rhj_join_device_slice(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS, uint32_t bucket_lo, uint32_t bucket_hi,
                          uint64_t first_skip, uint64_t last_end, rhj_result_tuple *d_out, uint64_t out_capacity, uint64_t *matches)
{
    RhjApiLock api_lock;
    return join_range(d_R, nR, d_S, nS, bucket_lo, bucket_hi, d_out, out_capacity, false, nullptr, matches, first_skip, last_end);
}

int rhj_set_devices(int n) { RhjApiLock api_lock; g_ndev_env = 0; return set_devices(n); }
/* Contiguous bucket ranges for n devices balanced by histR + histS (skewed keys): cuts[d] .. cuts[d + 1] is device d's range.
 * shard.bucket_ranges in C: the first bucket at which the running total reaches d / n of all tuples; no device needed. */
int rhj_plan_device_ranges(const uint64_t *histR, const uint64_t *histS, int bits, int n, uint32_t *cuts)
{
    if (bits < 1 || bits > MAX_BITS || n < 1 || n > MAX_DEVICES) return -1;
    const uint32_t bins = 1u << bits;
    double total = 0.0;
    for (uint32_t b = 0; b < bins; ++b) total += (double)histR[b] + (double)histS[b];
    cuts[0] = 0;
    uint32_t at = 0;
    double cum = 0.0;                                  // tuples in front of bucket `at`
    for (int d = 1; d < n; ++d) {
        const double target = total * (double)d / (double)n;
        while (at < bins && cum < target) { cum += (double)histR[at] + (double)histS[at]; ++at; }
        cuts[d] = at;
    }
    cuts[n] = bins;
    return 0;
}
/* The same with cuts INSIDE hot buckets: device d joins from (cut_bucket[d], cut_off[d]) up to (cut_bucket[d + 1], cut_off[d + 1])
 * in (bucket, probe position) order — rhj_cut_to_slice turns two neighbouring cuts into rhj_join_device_slice's arguments.  A cut
 * falls inside a bucket only when that bucket holds at least 1 / (2 n) of all tuples and both relations have tuples in it: the
 * device in front takes the bucket's build side and the probe tuples up to the cut (a multiple of 256; not a sliver: a cut that would
 * leave less than an eighth of the probe side on one side goes to that boundary), so that its tuples reach d / n of the total; every
 * other cut is the range planner's bucket boundary.  shard.bucket_slices in C (integer arithmetic). */
int rhj_plan_device_slices(const uint64_t *histR, const uint64_t *histS, int bits, int n, uint32_t *cut_bucket, uint64_t *cut_off)
{
    if (bits < 1 || bits > MAX_BITS || n < 1 || n > MAX_DEVICES) return -1;
    const uint32_t bins = 1u << bits;
    unsigned __int128 total = 0;
    for (uint32_t b = 0; b < bins; ++b) total += (unsigned __int128)histR[b] + histS[b];
    cut_bucket[0] = 0; cut_off[0] = 0;
    uint32_t at = 0;
    unsigned __int128 cum = 0;                         // tuples in front of bucket `at`
    for (int d = 1; d < n; ++d) {
        const unsigned __int128 target = total * (unsigned)d / (unsigned)n;
        while (at < bins && cum + histR[at] + histS[at] <= target) { cum += (unsigned __int128)histR[at] + histS[at]; ++at; }
        cut_bucket[d] = at; cut_off[d] = 0;            // (all buckets in front of `at` end at or before the target)
        if (at == bins) continue;
        const uint64_t hR = histR[at], hS = histS[at];
        const unsigned __int128 w = (unsigned __int128)hR + hS;
        const bool hot = hR != 0 && hS != 0 && w * 2u * (unsigned)n >= total;
        if (!hot) {
            if (cum < target) { cum += w; ++at; cut_bucket[d] = at; }      // the first boundary at or behind the target
            continue;
        }
        const uint64_t pc = hR >= hS ? hR : hS, bc = hR >= hS ? hS : hR;
        uint64_t off = target > cum + bc ? (uint64_t)(target - cum - bc) : 0u;
        off &= ~(uint64_t)255;
        if (off * 8u < pc) off = 0;                    // (a sliver of a bucket is not worth a second partition of its build side:
        else if ((pc - off) * 8u < pc) off = pc;       //  less than an eighth of the probe side on either side goes to the boundary)
        if (off >= pc) { cum += w; ++at; cut_bucket[d] = at; }
        else cut_off[d] = off;
    }
    cut_bucket[n] = bins; cut_off[n] = 0;
    return 0;
}
/* cuts d and d + 1 of rhj_plan_device_slices -> the arguments of rhj_join_device_slice */
void rhj_cut_to_slice(uint32_t b0, uint64_t o0, uint32_t b1, uint64_t o1, uint32_t *bucket_lo, uint32_t *bucket_hi, uint64_t *first_skip,
                      uint64_t *last_end)
{
    *bucket_lo = b0; *first_skip = o0;
    if (o1) { *bucket_hi = b1 + 1u; *last_end = o1; } else { *bucket_hi = b1; *last_end = 0; }
}
void rhj_set_devices_balance(int mode) { RhjApiLock api_lock; g_balance = mode < 0 ? 0 : mode > 2 ? 2 : mode; }

/* the bucket range device d of n joins at `bits` radix bits (no device needed: the planning half of rhj_join_devices) */
int rhj_device_range(int bits, int n, int d, uint32_t *lo, uint32_t *hi)
{
    if (bits < 1 || bits > MAX_BITS || n < 1 || n > MAX_DEVICES || d < 0 || d >= n) return -1;
    *lo = range_cut(1u << bits, n, d);
    *hi = range_cut(1u << bits, n, d + 1);
    return 0;
}
int rhj_get_devices(void) { RhjApiLock api_lock; (void)devices_ready(); return g_ndev; }

/* One join over the devices of rhj_set_devices(n).  d_R[d] / d_S[d]: the relations ON device d (replicated: a device-resident
 * column store per GPU).  Device d joins the d-th of n equal-width bucket ranges with its own context, stream and workspace,
 * driven by a host thread of its own (rhj_join_device_range's one call per rank, inside one process), and leaves its pairs in
 * out[d] (capacity[d] of them; matches[d] = their number, the call returns 1 when some list did not fit).  The concatenation
 * of the lists in device order is the canonical result. */
int rhj_join_devices(const rhj_tuple *const *d_R, uint64_t nR, const rhj_tuple *const *d_S, uint64_t nS,
                     rhj_result_tuple *const *out, const uint64_t *capacity, uint64_t *matches)
{
    RhjApiLock api_lock;
    if (devices_ready()) return -1;
    const int n = g_ndev;
    const int bits = g_all[0].bits;
    const uint32_t bins = 1u << bits;
    uint32_t cuts[MAX_DEVICES + 1];
    uint64_t offs[MAX_DEVICES + 1] = {0};
    for (int d = 0; d <= n; ++d) cuts[d] = range_cut(bins, n, d);
    if (g_balance && n > 1 && nR && nS) {
        // skewed keys: the library's own device counts both relations' buckets (two launches, one read-back of 2^bits words each)
        if (ctx_init() || ensure(g.passhp, (size_t)2 * bins * 8)) return -1;
        uint64_t *d_h = (uint64_t *)g.passhp.p;
        std::vector<uint64_t> h((size_t)2 * bins);
        if (rhj_bucket_histogram_device(d_R[0], nR, d_h) || rhj_bucket_histogram_device(d_S[0], nS, d_h + bins)) return -1;
        HIP_TRY(hipMemcpyAsync(h.data(), d_h, (size_t)2 * bins * 8, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        if (g_balance == 2 ? rhj_plan_device_slices(h.data(), h.data() + bins, bits, n, cuts, offs)
                           : rhj_plan_device_ranges(h.data(), h.data() + bins, bits, n, cuts)) return -1;
    }
    int rcs[MAX_DEVICES];
    for (int d = 0; d < MAX_DEVICES; ++d) rcs[d] = -1;
    try {
        on_devices(n, [&](int d) {
            uint32_t lo, hi;
            uint64_t skip, end;
            rhj_cut_to_slice(cuts[d], offs[d], cuts[d + 1], offs[d + 1], &lo, &hi, &skip, &end);
            rcs[d] = join_range(d_R[d], nR, d_S[d], nS, lo, hi, out[d], capacity[d], false, nullptr, &matches[d], skip, end);
        });
    } catch (...) { return -1; }
    int rc = 0;
    for (int d = 0; d < n; ++d) { if (rcs[d] < 0) return rcs[d]; if (rcs[d] > rc) rc = rcs[d]; }
    return rc;
}

/* The whole pair list on one device of the set: device d's list (lists[d], matches[d] pairs, as rhj_join_devices left them) is
 * copied to dst + (the pairs of the devices in front of d) — exact sizes, peer to peer (xGMI), on dst_device's stream.  Only a
 * caller whose next operator lives on ONE device needs it (north_star: a match list crosses only when its consumer is elsewhere). */
int rhj_gather_pairs_devices(const rhj_result_tuple *const *lists, const uint64_t *matches, int dst_device, rhj_result_tuple *dst,
                             uint64_t capacity, uint64_t *total)
{
    RhjApiLock api_lock;
    if (devices_ready() || dst_device < 0 || dst_device >= g_ndev) return -1;
    uint64_t sum = 0;
    for (int d = 0; d < g_ndev; ++d) sum += matches[d];
    if (total) *total = sum;
    if (sum > capacity) return 1;
    Ctx *keep = g_cur;
    g_cur = &g_all[dst_device];
    int rc = ctx_init();
    uint64_t at = 0;
    for (int d = 0; d < g_ndev && !rc; ++d) {
        if (matches[d] && (lists[d] != dst + at || d != dst_device))
            if (hipMemcpyPeerAsync(dst + at, g_all[dst_device].device, lists[d], g_all[d].device, matches[d] * sizeof(rhj_result_tuple), g.stream) != hipSuccess) rc = -1;
        at += matches[d];
    }
    if (!rc && hipStreamSynchronize(g.stream) != hipSuccess) rc = -1;
    g_cur = keep;
    (void)hipSetDevice(g.device);
    return rc;
}

int rhj_partition_device(const rhj_tuple *d_in, uint64_t n, rhj_tuple *d_out, uint64_t *h_hist, int64_t *h_psum)
{
    RhjApiLock api_lock;
    if (ctx_init()) return -1;
    const int bits = g.bits;
    const uint32_t bins = 1u << bits;
    if (n >= (1ull << 32)) return -2;
    PartState ps;
    ps.r[0] = RelArgs{d_in, d_out, nullptr, n, 0, 0, nullptr, nullptr};
    ps.tmp[0] = ps.tmp[1] = nullptr;
    if (bits > PT_MAX_BITS) {
        if (ensure(g.tmpR, (n ? n : 1) * sizeof(rhj_tuple))) return -1;
        ps.tmp[0] = (rhj_tuple *)g.tmpR.p;
    }
    uint64_t *hh = (uint64_t *)malloc((size_t)2 * bins * 8);
    if (!hh) return -1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (run_partition(ps, bits, 1, attempt == 1 || g.wide_row_ids != 0, false)) { free(hh); return -1; }
        RHJ_STAGE(ST_PLAN);
        PlanSummary *hs = (PlanSummary *)g.pin;
        HIP_TRY(hipMemcpyAsync(hh, ps.hist, (size_t)bins * 8, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(hh + bins, ps.psum, (size_t)bins * 8, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(hs, g.summary.p, sizeof(PlanSummary), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        if (!hs->row_id_overflow) break;                      // else: a row id above 2^32 - 1 met a 12-byte intermediate
        g.seen_wide = 1;
    }
    for (uint32_t b = 0; b < bins; ++b) {
        if (h_hist) h_hist[b] = hh[b];
        if (h_psum) h_psum[b] = hh[b] ? (int64_t)hh[bins + b] : -1;      // preprocess.c:336-347
    }
    free(hh);
    memset(&g.stats, 0, sizeof(g.stats));
    g.stats.n_r = n; g.stats.radix_bits = bits;
    g.stats.ms_hist = stage_ms(ST_HIST, ST_SCAN);
    g.stats.ms_scan = stage_ms(ST_SCAN, ST_SCATTER);
    g.stats.ms_scatter = stage_ms(ST_SCATTER, ST_PLAN);
    g.stats.ms_total = stage_ms(ST_HIST, ST_PLAN);
    return 0;
}

int rhj_filter_device(const uint64_t *d_col, const uint64_t *d_sel, uint64_t n, char op, uint64_t value,
                      uint64_t *d_out, uint64_t *hits)
{
    RhjApiLock api_lock;
    uint64_t h = 0;
    const int rc = filter_device(d_col, d_sel, n, op, value, d_out, false, nullptr, &h);
    if (hits) *hits = h;
    return rc;
}

/* ---- bucket-range sharding of one join across GPUs (host side: sigmod-2018_amd/shard.py) ---- */

int rhj_bucket_histogram_device(const rhj_tuple *d_in, uint64_t n, uint64_t *d_hist)
{
    RhjApiLock api_lock;
    if (ctx_init()) return -1;
    const int bits = g.bits;
    HIP_TRY(hipMemsetAsync(d_hist, 0, ((size_t)8) << bits, g.stream));
    if (n) {
        uint64_t blocks = (n + SH_BLOCK * 16 - 1) / (SH_BLOCK * 16);
        if (blocks > (uint64_t)g.cus * 4) blocks = (uint64_t)g.cus * 4;
        RHJ_LAUNCH(k_bucket_hist, dim3((unsigned)blocks), dim3(SH_BLOCK), ((size_t)4) << bits, g.stream, d_in, n, bits,
                   (unsigned long long *)d_hist);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int rhj_select_bucket_range_device(const rhj_tuple *d_in, uint64_t n, uint32_t bucket_lo, uint32_t bucket_hi,
                                   rhj_tuple *d_out, uint64_t capacity, uint64_t *count)
{
    RhjApiLock api_lock;
    return select_range(d_in, n, bucket_lo, bucket_hi, d_out, capacity, count);
}

static void release_current()
{
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    Buf *all[] = {&g.partR, &g.partS, &g.tmpR, &g.tmpS, &g.cntR, &g.cntS, &g.chunk, &g.histpsum, &g.passhp,
                  &g.units, &g.bunits, &g.ldsb, &g.meta, &g.summary, &g.ucount, &g.ubase, &g.uflag, &g.bsum, &g.digR, &g.digS, &g.ovf, &g.ovf_base, &g.walk, &g.xrows, &g.lr_tmp, &g.lr_words, &g.lr_status, &g.runR, &g.runS, &g.stripR, &g.stripS, &g.slice_tot, &g.sbase, &g.stash_cnt, &g.stash_row, &g.status, &g.tab32,
                  &g.tab64, &g.inR, &g.inS, &g.out, &g.fcol, &g.fcol_sel, &g.fmask, &g.ftile, &g.fbase, &g.fout};
    for (Buf *b : all) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    for (auto &kv : g.columns) (void)hipFree(kv.second.dev);
    g.columns.clear();
    for (auto &kv : g.pinned) (void)hipHostUnregister((void *)kv.first);
    g.pinned.clear();
    for (auto &kv : g.free_blocks) (void)hipFree(kv.second);
    g.free_blocks.clear();
    for (auto &kv : g.live_blocks) (void)hipFree(kv.first);
    g.live_blocks.clear();
}

void rhj_release(void)
{
    RhjApiLock api_lock;
    rhj_host_pool_release();
    Ctx *keep = g_cur;
    for (int d = MAX_DEVICES - 1; d >= 0; --d) { g_cur = &g_all[d]; release_current(); }   // (the library's own device last: it stays current)
    g_cur = keep;
}

// ---- host-side staging used by rhj_abi.c (not part of the public header) ----

// The pairs of the calling thread's context, [d_out, d_out + M), into the caller-visible nodes as elements [base, base + M) of the
// list.  The nodes are plain malloc memory (FreeResult = free(buff); free(node), results.c:144-153): freshly mapped pages, so
// whoever writes them first pays the page faults — a single thread filling them measured 5.8 GB/s (44 ms for 256 MB).  All
// nodes are allocated up front (untouched), the pairs come through a ring of pinned staging blocks (the copy of block i+1..
// runs while block i is moved), and every block is moved into the nodes by several host threads, each faulting in its own pages.
static int pairs_to_nodes(const rhj_result_tuple *d_out, uint64_t base, uint64_t M, uint64_t node_pairs, char *const *nodes, unsigned nthreads)
{
    if (M == 0) return 0;
    constexpr int RING = 4;
    const uint64_t blk = (uint64_t)1 << 20;                               // pairs per staging block (16 MiB)
    if (!g.pin_ring[0]) {
        for (int i = 0; i < RING; ++i) {
            HIP_TRY(hipHostMalloc(&g.pin_ring[i], blk * sizeof(rhj_result_tuple), hipHostMallocDefault));
            HIP_TRY(hipEventCreate(&g.ev_ring[i]));
        }
    }
    HIP_TRY(hipEventRecord(g.ev_x[2], g.stream));
    if (nthreads < 1 || M * sizeof(rhj_result_tuple) < ((size_t)8 << 20)) nthreads = 1;
    // the ring protocol and the mover threads are host-only code (rhj_host.cpp: rhj_move_blocks_at, built and run under the
    // sanitizers on the CPU); this side only starts the copy of a block and waits for it
    struct Copy { const rhj_result_tuple *d_out; uint64_t M, blk; Ctx *c; } cp = {d_out, M, blk, g_cur};
    auto issue = [](void *c, uint64_t b) -> int {
        const Copy *k = (const Copy *)c;
        const uint64_t cnt = k->M - b * k->blk < k->blk ? k->M - b * k->blk : k->blk;
        if (hipMemcpyAsync(k->c->pin_ring[b % RING], k->d_out + b * k->blk, cnt * sizeof(rhj_result_tuple), hipMemcpyDeviceToHost, k->c->stream) != hipSuccess) return -1;
        return hipEventRecord(k->c->ev_ring[b % RING], k->c->stream) == hipSuccess ? 0 : -1;
    };
    auto landed = [](void *c, uint64_t b) -> int { return hipEventSynchronize(((const Copy *)c)->c->ev_ring[b % RING]) == hipSuccess ? 0 : -1; };
    char *staging[RING];
    for (int i = 0; i < RING; ++i) staging[i] = (char *)g.pin_ring[i];
    if (rhj_move_blocks_at(base, M, sizeof(rhj_result_tuple), node_pairs, nodes, blk, RING, staging, nthreads, issue, landed, &cp)) {
        (void)hipGetLastError();
        return -1;
    }
    HIP_TRY(hipEventRecord(g.ev_x[3], g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    g.stats.ms_d2h = ev_ms(g.ev_x[2], g.ev_x[3]);
    return 0;
}

// Upload both relations, join, and copy the pairs back into `node_pairs`-sized
// chunks handed to `sink(ctx, chunk_index, ptr_to_fill, pairs)`-allocated memory.
// With rhj_set_devices(n > 1) / RHJ_DEVICES: every device uploads both relations over its own link, joins the d-th of n
// equal-width bucket ranges (the first partition pass drops the others while it reads), and its pairs go to their place in the
// list — behind the pairs of the devices in front of it — over its own link again: n uploads side by side, 1/n of the kernel
// work and of the read-back each.
int rhj_host_join(const rhj_tuple *R, uint64_t nR, const rhj_tuple *S, uint64_t nS, uint64_t *matches,
                  void *(*alloc_chunk)(void *ctx, uint64_t pairs), void *ctx, uint64_t node_pairs)
{
    RhjApiLock api_lock;
    *matches = 0;
    if (devices_ready()) return -1;
    if (ctx_init()) return -1;
    if (nR == 0 || nS == 0) return 0;
    const int n = g_all[0].order_any ? 1 : g_ndev;             // (bucket ranges belong to the caller's radix: not with RHJ_ORDER=any)
    const uint32_t bins = 1u << g_all[0].bits;
    int rcs[MAX_DEVICES] = {0};
    uint64_t Ms[MAX_DEVICES] = {0};
    rhj_result_tuple *d_outs[MAX_DEVICES] = {nullptr};
    auto upload_and_join = [&](int d) {
        rcs[d] = -1;
        if (ctx_init()) return;
        if (ensure(g.inR, nR * sizeof(rhj_tuple)) || ensure(g.inS, nS * sizeof(rhj_tuple))) return;
        if (hipEventRecord(g.ev_x[0], g.stream) != hipSuccess ||
            hipMemcpyAsync(g.inR.p, R, nR * sizeof(rhj_tuple), hipMemcpyHostToDevice, g.stream) != hipSuccess ||
            hipMemcpyAsync(g.inS.p, S, nS * sizeof(rhj_tuple), hipMemcpyHostToDevice, g.stream) != hipSuccess ||
            hipEventRecord(g.ev_x[1], g.stream) != hipSuccess) return;
        if (n == 1) rcs[d] = join_device((const rhj_tuple *)g.inR.p, nR, (const rhj_tuple *)g.inS.p, nS, nullptr, 0, true, &d_outs[d], &Ms[d]);
        else        rcs[d] = join_range((const rhj_tuple *)g.inR.p, nR, (const rhj_tuple *)g.inS.p, nS, range_cut(bins, n, d), range_cut(bins, n, d + 1),
                                        nullptr, 0, true, &d_outs[d], &Ms[d]);
        if (rcs[d] >= 0) g.stats.ms_h2d = ev_ms(g.ev_x[0], g.ev_x[1]);
    };
    try { on_devices(n, upload_and_join); } catch (...) { return -1; }
    uint64_t M = 0, base[MAX_DEVICES] = {0};
    for (int d = 0; d < n; ++d) { if (rcs[d] < 0) return rcs[d]; base[d] = M; M += Ms[d]; }
    *matches = M;
    if (M == 0) return 0;
    if (node_pairs == 0) node_pairs = M;
    const uint64_t nnodes = (M + node_pairs - 1) / node_pairs;
    std::vector<char *> nodes((size_t)nnodes);
    for (uint64_t i = 0; i < nnodes; ++i) {
        const uint64_t cnt = M - i * node_pairs < node_pairs ? M - i * node_pairs : node_pairs;
        nodes[(size_t)i] = (char *)alloc_chunk(ctx, cnt);
        if (!nodes[(size_t)i]) { fprintf(stderr, "rhj: out of host memory for %llu result pairs\n", (unsigned long long)cnt); return -1; }
    }
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads > 8) nthreads = 8;
    nthreads = nthreads / (unsigned)n ? nthreads / (unsigned)n : 1u;      // the devices' movers share the host's cores
    try {
        for (int d = 0; d < n; ++d) rcs[d] = -1;
        on_devices(n, [&](int d) { rcs[d] = pairs_to_nodes(d_outs[d], base[d], Ms[d], node_pairs, nodes.data(), nthreads); });
    } catch (...) { return -1; }
    for (int d = 0; d < n; ++d) if (rcs[d]) return -1;
    return 0;
}

// The device copy of a registered column, or nullptr.
static const uint64_t *registered_column(const uint64_t *host_col, uint64_t rows)
{
    auto it = g.columns.find((const void *)host_col);
    if (it == g.columns.end() || it->second.rows != rows) return nullptr;
    return (const uint64_t *)it->second.dev;
}

// Pin [base, base + bytes) for the H2D copy (relation_map.c:28-50: the columns of a relation are one
// contiguous block of the PROT_READ | MAP_PRIVATE file mapping, hence the read-only flag first).
// Returns whether the range is pinned now; a range that cannot be pinned is copied pageable.
// hipHostRegister of a host range the library copies columns from.  A read-only file mapping (InitRelationMap's
// PROT_READ | MAP_PRIVATE block) takes the read-only flag; memory the caller owns takes either.  Which flag the host
// accepted is traced (RHJ_TRACE=1) and counted: rhj_pinned_ranges(), rhj_pin_refusals().
static bool pin_range(const void *base, size_t bytes)
{
    static const bool trace = getenv("RHJ_TRACE") != nullptr;
    if (bytes < (64u << 10)) return false;                              // registration costs more than it saves
    if (g.pinned.count(base)) return true;
    const char *how = "read-only";
    hipError_t e = hipHostRegister((void *)base, bytes, hipHostRegisterReadOnly);
    if (e != hipSuccess) { (void)hipGetLastError(); how = "default"; e = hipHostRegister((void *)base, bytes, hipHostRegisterDefault); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ++g.pin_refusals;
        if (trace) fprintf(stderr, "rhj-trace:   hipHostRegister refused %zu bytes at %p (%s): pageable copy\n", bytes, base, hipGetErrorString(e));
        return false;
    }
    if (trace) fprintf(stderr, "rhj-trace:   hipHostRegister(%s) pinned %zu bytes at %p\n", how, bytes, base);
    g.pinned[base] = bytes;
    return true;
}

static void unpin_range(const void *base)
{
    auto it = g.pinned.find(base);
    if (it == g.pinned.end()) return;
    (void)hipHostUnregister((void *)base);
    g.pinned.erase(it);
}

static int register_column(const uint64_t *host_col, uint64_t rows)
{
    auto it = g.columns.find((const void *)host_col);
    if (it != g.columns.end()) {
        if (it->second.rows == rows) return 0;
        (void)hipStreamSynchronize(g.stream);                           // registered again with another length: replace
        (void)hipFree(it->second.dev);
        g.columns.erase(it);
    }
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, (rows ? rows : 1) * 8));
    if (hipMemcpyAsync(d, host_col, rows * 8, hipMemcpyHostToDevice, g.stream) != hipSuccess) { (void)hipFree(d); return -1; }
    g.columns[(const void *)host_col] = Ctx::Column{d, (size_t)rows};
    return 0;
}

static void unregister_column(const uint64_t *host_col)
{
    auto it = g.columns.find((const void *)host_col);
    if (it == g.columns.end()) return;
    (void)hipFree(it->second.dev);
    g.columns.erase(it);
}

// are the columns of this relation one contiguous block (the layout of relation_map.c:39-50)?
static bool contiguous_columns(const rhj_relation_map *rm)
{
    for (uint64_t c = 1; c < rm->num_columns; ++c)
        if (rm->columns[c] != rm->columns[c - 1] + rm->num_tuples) return false;
    return rm->num_columns > 0;
}

int rhj_register_relation_map(const rhj_relation_map *map, int num_relations)
{
    RhjApiLock api_lock;
    if (ctx_init()) return -1;
    for (int r = 0; r < num_relations; ++r) {
        const rhj_relation_map *rm = &map[r];
        if (contiguous_columns(rm)) pin_range(rm->columns[0], (size_t)rm->num_columns * rm->num_tuples * 8);
        else for (uint64_t c = 0; c < rm->num_columns; ++c) pin_range(rm->columns[c], (size_t)rm->num_tuples * 8);
        for (uint64_t c = 0; c < rm->num_columns; ++c)
            if (register_column(rm->columns[c], rm->num_tuples)) return -1;
    }
    HIP_TRY(hipStreamSynchronize(g.stream));
    return 0;
}

int rhj_unregister_relation_map(const rhj_relation_map *map, int num_relations)
{
    RhjApiLock api_lock;
    if (!g.ready) return 0;
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipStreamSynchronize(g.stream));                             // queued work may still read the device copies
    for (int r = 0; r < num_relations; ++r)
        for (uint64_t c = 0; c < map[r].num_columns; ++c) {
            unregister_column(map[r].columns[c]);
            unpin_range(map[r].columns[c]);
        }
    return 0;
}

int rhj_registered_columns(void) { return (int)g.columns.size(); }
int rhj_pinned_ranges(void) { return (int)g.pinned.size(); }
int rhj_pin_refusals(void) { return g.pin_refusals; }

// Filter on a host column (its registered device copy, or uploaded for this call) with an optional
// host row-id indirection vector; ids come back in `node_ids`-sized chunks.
int rhj_host_filter(const uint64_t *col, uint64_t col_rows, const uint64_t *sel, uint64_t n, char op, uint64_t value,
                    uint64_t *hits, void *(*alloc_chunk)(void *ctx, uint64_t ids), void *ctx, uint64_t node_ids)
{
    RhjApiLock api_lock;
    *hits = 0;
    if (ctx_init()) return -1;
    if (op_code(op) < 0) return -3;
    if (n == 0) return 0;
    const uint64_t *d_col = registered_column(col, col_rows);
    if (!d_col) {                                      // not registered: upload it for this call only
        if (ensure(g.fcol, (col_rows ? col_rows : 1) * 8)) return -1;
        HIP_TRY(hipMemcpyAsync(g.fcol.p, col, col_rows * 8, hipMemcpyHostToDevice, g.stream));
        d_col = (const uint64_t *)g.fcol.p;
    }
    const uint64_t *d_sel = nullptr;
    if (sel) {
        if (ensure(g.fcol_sel, n * 8)) return -1;
        HIP_TRY(hipMemcpyAsync(g.fcol_sel.p, sel, n * 8, hipMemcpyHostToDevice, g.stream));
        d_sel = (const uint64_t *)g.fcol_sel.p;
    }
    uint64_t *d_out = nullptr, h = 0;
    const int rc = filter_device(d_col, d_sel, n, op, value, nullptr, true, &d_out, &h);
    if (rc) return rc;
    *hits = h;
    for (uint64_t at = 0; at < h; at += node_ids) {
        const uint64_t cnt = h - at < node_ids ? h - at : node_ids;
        void *dst = alloc_chunk(ctx, cnt);
        if (!dst) return -1;
        HIP_TRY(hipMemcpyAsync(dst, d_out + at, cnt * 8, hipMemcpyDeviceToHost, g.stream));
    }
    HIP_TRY(hipStreamSynchronize(g.stream));
    return 0;
}

int rhj_host_null_on_empty(void) { return g.null_on_empty; }
uint64_t rhj_host_node_pairs(void) { return g.node_pairs; }

// ---- device-side services for rhj_inter.hip (device-resident intermediate results) ----

// Device blocks of the intermediate results.  Everything that touches them is queued on the one
// library stream, so a freed block can be handed out again at once: the new user's work is ordered
// behind the old user's by the stream itself.  Blocks are cached by size (never returned to the
// driver before rhj_release()), which keeps hipMalloc/hipFree — both device-wide synchronisations —
// out of the per-operator path.  (hipMallocAsync/hipFreeAsync were tried first: on this stack a
// block freed with work still queued came back with stale contents visible to later kernels.)
void *rhj_dev_alloc(size_t bytes)
{
    RhjApiLock api_lock;
    if (ctx_init()) return nullptr;
    // size classes 2^k and 1.5 * 2^k (a query plan asks for a few dozen different sizes; with 1 MiB granules every one of them
    // was a hipMalloc — a device-wide synchronisation of 50-150 us — the first time: 79 during the `small` run), and a cached
    // block of up to twice the class is taken before the driver is asked
    size_t want = 256;
    const size_t need = bytes < 256 ? 256 : bytes;
    while (want < need) { const size_t mid = want + want / 2; if (want >= ((size_t)1 << 20) && mid >= need) { want = mid; break; } want <<= 1; }
    auto it = g.free_blocks.lower_bound(want);
    if (it != g.free_blocks.end() && it->first <= 2 * want) {
        void *p = it->second;
        g.live_blocks[p] = it->first;
        g.free_blocks.erase(it);
        return p;
    }
    void *p = nullptr;
    static const bool trace = getenv("RHJ_TRACE") != nullptr;
    if (trace) fprintf(stderr, "rhj-trace:   hipMalloc %zu bytes (no cached block of that size)\n", want);
    if (hipMalloc(&p, want) != hipSuccess) {
        // out of device memory: give the cached blocks back and try once more
        (void)hipStreamSynchronize(g.stream);
        for (auto &kv : g.free_blocks) (void)hipFree(kv.second);
        g.free_blocks.clear();
        if (hipMalloc(&p, want) != hipSuccess) {
            fprintf(stderr, "rhj: device allocation of %zu bytes failed\n", bytes);
            return nullptr;
        }
    }
    g.live_blocks[p] = want;
    return p;
}
void rhj_dev_free(void *p)
{
    RhjApiLock api_lock;
    if (!p) return;
    auto it = g.live_blocks.find(p);
    if (it == g.live_blocks.end()) return;
    g.free_blocks.emplace(it->second, p);
    g.live_blocks.erase(it);
}
void *rhj_dev_stream(void) { return ctx_init() ? nullptr : (void *)g.stream; }
// Workspace of joins over relations of up to `rows` tuples a side, allocated now (InitRelationMap knows the base relations'
// sizes: the first joins of a plan then find their buffers in place instead of growing them one by one).
int rhj_dev_reserve(uint64_t rows)
{
    RhjApiLock api_lock;
    if (ctx_init() || rows == 0 || rows >= (1ull << 32)) return -1;
    const uint64_t tiles = (rows + SM_TILE - 1) / SM_TILE + 1, units = 256 + 2 * rows / FJ_BATCH + 16;
    if (ensure(g.partR, rows * sizeof(rhj_tuple)) || ensure(g.partS, rows * sizeof(rhj_tuple)) ||
        ensure(g.cntR, (size_t)tiles * 256 * 4) || ensure(g.cntS, (size_t)tiles * 256 * 4) ||
        ensure(g.stash_cnt, 2 * rows + 64) || ensure(g.stash_row, (2 * rows + 8) * 8) ||
        ensure(g.status, (units + 9) * 8 + 64) || ensure(g.units, units * sizeof(Unit)) ||
        ensure(g.ovf, fj_ovf_bytes((size_t)g.cus)) || ensure(g.ovf_base, (size_t)g.cus * 2 * FJ_GROUPS * 16 * 4))
        return -1;
    return 0;
}
// Device copy of a host column for the resident operators: the registered copy, or — for a column nobody
// registered — a fresh block uploaded now, returned in *temp for the caller to rhj_dev_free() once the
// kernels that read it are queued (blocks are reused in stream order).
const uint64_t *rhj_dev_column(const uint64_t *host_col, uint64_t rows, void **temp)
{
    RhjApiLock api_lock;
    *temp = nullptr;
    if (ctx_init()) return nullptr;
    if (const uint64_t *d = registered_column(host_col, rows)) return d;
    void *blk = rhj_dev_alloc((rows ? rows : 1) * 8);
    if (!blk) return nullptr;
    if (hipMemcpyAsync(blk, host_col, rows * 8, hipMemcpyHostToDevice, g.stream) != hipSuccess) { rhj_dev_free(blk); return nullptr; }
    *temp = blk;
    return (const uint64_t *)blk;
}
int rhj_dev_register_column(const uint64_t *host_col, uint64_t rows, const void *pin_base, uint64_t pin_bytes)
{
    RhjApiLock api_lock;
    if (ctx_init()) return -1;
    if (pin_base) pin_range(pin_base, (size_t)pin_bytes);
    return register_column(host_col, rows);
}
void rhj_dev_unregister_column(const uint64_t *host_col)
{
    RhjApiLock api_lock;
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    unregister_column(host_col);
    unpin_range(host_col);
}
int rhj_dev_join(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS, rhj_result_tuple **out, uint64_t *matches)
{
    RhjApiLock api_lock;
    *out = nullptr; *matches = 0;
    return join_device(d_R, nR, d_S, nS, nullptr, 0, true, out, matches);
}
int rhj_filter_eq2_device(const uint64_t *d_colA, const uint64_t *d_selA, const uint64_t *d_colB, const uint64_t *d_selB,
                          uint64_t n, uint64_t *d_out, uint64_t *hits)
{
    RhjApiLock api_lock;
    uint64_t h = 0;
    const int rc = filter_eq2_device(d_colA, d_selA, d_colB, d_selB, n, d_out, &h);
    if (hits) *hits = h;
    return rc;
}

}  // extern "C"
