/*
 * rhj.h — C-ABI of the MI355X-native radix hash join / filter scan.
 *
 * This is the drop-in boundary for ONE hot path of VagelisN/Sigmod-2018:
 * RadixHashJoin() (rhjoin.c + preprocess.c) and Filter() (filter.c), with the
 * result-list layout of results.c.  Everything here is plain C: pointers and
 * sizes, no torch / HIP types in any signature.
 *
 * Layout compatibility (all sizes checked by static asserts in rhj_abi.c):
 *   rhj_tuple        == reference `tuple`          structs.h:15-19   16 B
 *   rhj_relation     == reference `relation`       structs.h:25-29   16 B
 *   rhj_result       == reference `result`         structs.h:37-43   24 B
 *   rhj_result_tuple == reference `result_tuple`   structs.h:46-50   16 B
 *   rhj_inter_data / rhj_inter_res == `inter_data` / `inter_res`   structs.h:97-111
 *   rhj_column_stats / rhj_relation_map == `column_stats` / `relation_map` structs.h:120-138
 *   rhj_filter_pred  == reference `filter_pred`    structs.h:141-147 16 B
 *
 * When RHJ_REFERENCE_NAMES is defined before including this header the
 * reference's own type names are typedef'd onto these, so a translation unit
 * of the reference's caller side (query.c, inter_res.c) can be compiled
 * against this header instead of structs.h for the types on this path.
 */
#ifndef RHJ_H
#define RHJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference-compatible data layout ---------------------------------- */

typedef struct rhj_tuple {          /* structs.h:15-19 */
    uint64_t value;
    uint64_t row_id;
} rhj_tuple;

typedef struct rhj_relation {       /* structs.h:25-29 */
    rhj_tuple *tuples;
    uint64_t   num_tuples;
} rhj_relation;

typedef struct rhj_result {         /* structs.h:37-43 */
    char              *buff;         /* current_load elements, malloc'd   */
    struct rhj_result *next;
    uint64_t           current_load; /* elements in buff                  */
} rhj_result;

typedef struct rhj_result_tuple {   /* structs.h:46-50 */
    uint64_t row_idR;
    uint64_t row_idS;
} rhj_result_tuple;

typedef struct rhj_inter_data {     /* structs.h:97-101 */
    uint64_t   num_tuples;
    uint64_t **table;                /* table[rel] == NULL: rel inactive  */
} rhj_inter_data;

typedef struct rhj_inter_res {      /* structs.h:106-111 */
    rhj_inter_data       *data;
    int                   num_of_relations;
    struct rhj_inter_res *next;
} rhj_inter_res;

typedef struct rhj_column_stats {   /* structs.h:120-126 */
    uint64_t l, u;
    double   f, d;
} rhj_column_stats;

typedef struct rhj_relation_map {   /* structs.h:132-138 */
    uint64_t          num_tuples;
    uint64_t          num_columns;
    uint64_t        **columns;       /* column-major u64, relation_map.c:39-50 */
    rhj_column_stats *col_stats;
} rhj_relation_map;

typedef struct rhj_filter_pred {    /* structs.h:141-147 */
    int  relation;
    int  column;
    int  value;                      /* atoi() of the constant, query.c:239,247 */
    char comperator;                 /* '<' '>' '='  (spelling is the reference's) */
} rhj_filter_pred;

/* The reference's scheduler (structs.h:209-222) is opaque here: the GPU path
 * owns its own parallelism and tolerates sched == NULL (handler.c:60-63 passes
 * NULL when built with THREADS == 1). */
struct scheduler;

/* ---- the two boundary functions (names and signatures are the reference's) */

/* rhjoin.h:9, defined rhjoin.c:13-111.  NULL when either input is empty
 * (rhjoin.c:15-16).  Result order is the canonical order of SURVEY.md A.1:
 * bucket ascending, probe-side tuple in input order, build-side matches in
 * descending input position.  Zero matches: see rhj_set_empty_mode(). */
rhj_result *RadixHashJoin(rhj_relation *relR, rhj_relation *relS,
                          struct scheduler *sched);

/* filter.h:9, defined filter.c:92-190.  Ascending list of u64 indices i with
 * col[i] OP (uint64_t)(int)value; through head->data->table[rel][i] when the
 * relation is already active in the intermediate result.  NULL on zero hits.
 * An unknown comparator prints the reference's message and exit(2)s
 * (filter.c:184-186). */
rhj_result *Filter(rhj_inter_res *head, rhj_filter_pred *filter_p,
                   rhj_relation_map *map, int *query_relations);

/* ---- result-list API (results.h:7-25), same names, same behaviour -------- */
rhj_result       *InsertResult(rhj_result **head, rhj_result_tuple *res_tuple);  /* results.c:8-46    */
rhj_result       *InsertRowIdResult(rhj_result **head, uint64_t *row_id);        /* results.c:155-192 */
int               GetResultNum(rhj_result *res);                                 /* results.c:65-76   */
uint64_t          FindResultRowId(rhj_result *res, int num);                     /* results.c:48-64   */
rhj_result_tuple *FindResultTuples(rhj_result *head, int num);                   /* results.c:126-142 */
void              FreeResult(rhj_result *head);                                  /* results.c:144-153 */
void              PrintResult(rhj_result *head);                                 /* results.c:78-100  */
void              FreeRelation(rhj_relation *rel);                               /* preprocess.c:213-218 */
/* handler.c:62,102 call these; the GPU path needs no worker threads, so they
 * allocate/free an inert token (scheduler.c:9-25, :88-107). */
int               SchedulerInit(struct scheduler **sched, int num_of_threads);
int               SchedulerDestroy(struct scheduler *sched);

/* ---- knobs the reference hard-codes in structs.h:8-12 -------------------- */

/* N_LSB (structs.h:11), 1..15 here; default 4 or env RHJ_RADIX_BITS.  Returns 0
 * or -1 on a value outside the supported range. */
int  rhj_set_radix_bits(int bits);
int  rhj_get_radix_bits(void);
/* Zero-match convention (SURVEY.md finding 5): 0 = non-NULL empty head (as
 * shipped, THREADS 4: rhjoin.c:356-359), 1 = NULL (THREADS 1).  Env RHJ_EMPTY. */
void rhj_set_empty_mode(int null_on_empty);
/* Pairs per host result node; default 65535 (MergeResults' strict '<' at
 * rhjoin.c:371 caps a 1 MiB node at 65535 pairs).  0 = one node for all. */
void rhj_set_node_pairs(uint64_t pairs_per_node);
/* Device ordinal (default 0 or env RHJ_DEVICE); must precede the first call. */
int  rhj_set_device(int ordinal);       /* -1 once the context exists (one device per process) */
int  rhj_get_device(void);
/* Launch all work on this hipStream_t (passed as void*).  NULL is HIP's default stream (work the
 * caller queued there, e.g. through PyTorch's default stream, is then ordered before the join);
 * without this call the library uses a non-blocking stream of its own and the caller must have
 * synchronised its producers. */
void rhj_set_stream(void *hip_stream);
/* Path selection (results are identical on every path; tests run all three):
 *   fused (default)  one workgroup per bucket keeps its tag table in LDS, builds, probes and
 *                    emits in one kernel; chosen when every bucket's build side fits LDS;
 *   tiled            tag tables in HBM, tile-granular probe units, count + emit kernels;
 *                    taken automatically when some bucket is too large for LDS, when the buckets are tiny
 *                    (4096 or more of them with fewer than 512 tuples each on average: a fused unit has
 *                    a fixed cost) unless rhj_set_fused(2) / env RHJ_FORCE_FUSED=1 insists, or with
 *                    rhj_set_fused(0) / env RHJ_NO_FUSED=1;
 *   rhj_set_force_hbm_table(1) additionally builds every table with global atomics
 *                    (64-bit entries), the path of buckets beyond 65534 build tuples. */
void rhj_set_fused(int on);
void rhj_set_force_hbm_table(int on);
/* Fused path only: 1 (default) copies a bucket's build tuples into LDS when they fit beside
 * the index (<= ~7 K tuples: no global gather at all), 0 always gathers them from HBM. */
void rhj_set_resident(int on);
/* 1 (default): a join on at most 8 radix bits whose relations hold at most ~8 M tuples together runs histogram,
 * scan, plan, scatter and the fused join as the phases of ONE kernel launch (csrc/rhj_small.hip.h); 0: the same
 * steps as separate launches.  Results are identical either way (env RHJ_NO_SMALL=1). */
void rhj_set_small(int on);
/* 1 (default): a join on at most 8 radix bits whose buckets' build sides are beyond the LDS index (the reference's 4 bits
 * from ~0.5 M tuples per relation on) runs on r + k bits internally and is emitted in the canonical order of the r bits in
 * force (csrc/rhj_lowradix.hip.h); 0: such joins take the tiled path with hash tables in HBM.  Results are identical
 * either way (env RHJ_NO_LOWRADIX=1). */
void rhj_set_lowradix(int on);
/* 1 (default): the two-pass partition's first pass counts the second pass' digits itself at radix widths up to 12 bits
 * (csrc/rhj_partition.hip.h, k_local_part); 0: a kernel of its own counts them from one byte per tuple at every width, as
 * it does at 13..15 bits.  Results are identical either way (env RHJ_NO_COUNT_IN_PASS1=1; for A/B and tests). */
void rhj_set_count_in_pass1(int on);
/* 1 (default): big joins first try the foreign-key speculation — every tuple of the bigger relation has exactly one match
 * (csrc/rhj_join_fused.hip.h, k_join_spec): pairs of the units that relation probes are written without stash or chained
 * offsets; checked on the device, and the ordinary kernel takes over in the same call when it does not hold.  Results are
 * identical either way (env RHJ_NO_SPEC=1).  rhj_last_spec(): the last join — 0 not tried, 1 held, 2 failed. */
void rhj_set_spec(int on);
int  rhj_last_spec(void);
/* 0 (default) / 1 (env RHJ_EXACT=1): where the speculation is tried on build sides that are not LDS-resident (radix widths of
 * 10 bits and more, 12-byte tuples), k_join_exact (csrc/rhj_join_exact.hip.h) runs it instead of k_join_spec: its LDS index holds
 * 40 bits of a bijective hash of the key bits a bucket's keys differ in and is exact by itself — no gather of the build tuple,
 * only of its row id, from an L2-resident scratch array.  It takes build sides whose row ids increase with the input position
 * (the reference's always do: inter_res.c:202,225) and hands everything else to the ordinary kernels in the same call.
 * Results are identical either way.  Off by default because it measured slower than the gather kernel (DESIGN.md 4.2,
 * profiles/README.md r04a); kept as the measured alternative, with its tests.
 * rhj_last_exact(): the last join — 0 not launched, 1 it did the join, 2 it handed over. */
void rhj_set_exact(int on);
int  rhj_last_exact(void);

/* ---- several GPUs of one node behind this interface (SURVEY.md 8b "RHJ_DEVICES", 8e; rhjoin.c:42-57: bucket b of R only meets
 * bucket b of S, so contiguous bucket ranges are independent joins and their pair lists, concatenated in range order, ARE the
 * canonical result).  rhj_set_devices(n) / env RHJ_DEVICES=n: the library's device (rhj_set_device) and the n - 1 ordinals
 * behind it, each with a context, stream and workspace of its own inside this one process; -1 when they are not there.
 * From then on
 *   RadixHashJoin() with host relations   every device uploads both relations over its own link, joins the d-th of n
 *                                         equal-width bucket ranges (its first partition pass drops the other buckets while
 *                                         it reads) and moves its pairs to their place in the one result list; same list,
 *                                         bit for bit, as on one device;
 *   rhj_join_devices()                    the same for relations that are already on the devices (one replica each: a
 *                                         device-resident column store per GPU): device d's pairs stay in out[d];
 *   rhj_gather_pairs_devices()            the whole list on one of the devices, exact-size peer copies over xGMI — only for a
 *                                         caller whose next operator lives on that device.
 * Every device is driven by a host thread of its own; the caller's thread holds the library's lock and waits.  Everything
 * else (rhj_join_device, Filter, the device-resident intermediate results) stays on the library's own device.
 * rhj_device_range(): the bucket range device d of n takes at `bits` radix bits (needs no GPU).
 * Env RHJ_DEVICES_SAME=1 (tests): all n contexts on the library's device — the sharded path at n > 1 on a one-GPU box. */
int  rhj_set_devices(int n);
int  rhj_get_devices(void);
int  rhj_device_range(int bits, int n, int d, uint32_t *lo, uint32_t *hi);
/* rhj_set_devices_balance(1) / env RHJ_DEVICES_BALANCE=hist: rhj_join_devices cuts the ranges by histR + histS (two histogram
 * launches on the library's device and one read-back) instead of equal widths — for skewed keys; rhj_plan_device_ranges() is
 * that plan (cuts[d] .. cuts[d + 1] = device d's buckets; needs no GPU).  These ranges are cut BETWEEN buckets. */
void rhj_set_devices_balance(int mode);
int  rhj_plan_device_ranges(const uint64_t *histR, const uint64_t *histS, int bits, int n, uint32_t *cuts);
/* rhj_set_devices_balance(2) / RHJ_DEVICES_BALANCE=slice: the cuts may also fall INSIDE a hot bucket (one that holds at least
 * 1 / (2 n) of all tuples): the devices on either side of such a cut both partition the bucket's build side and share its probe
 * tuples ("a hot bucket's probe side across GPUs with the build side replicated") — a join on few radix bits, or
 * one with a key that dominates, still spreads over all devices.  rhj_plan_device_slices() is that plan: device d joins from
 * (cut_bucket[d], cut_off[d]) to (cut_bucket[d + 1], cut_off[d + 1]) in (bucket, position among the bucket's probe tuples)
 * order, n + 1 cuts; rhj_cut_to_slice() turns two neighbouring cuts into the arguments of rhj_join_device_slice().  No GPU needed. */
int  rhj_plan_device_slices(const uint64_t *histR, const uint64_t *histS, int bits, int n, uint32_t *cut_bucket, uint64_t *cut_off);
void rhj_cut_to_slice(uint32_t b0, uint64_t o0, uint32_t b1, uint64_t o1, uint32_t *bucket_lo, uint32_t *bucket_hi,
                      uint64_t *first_skip, uint64_t *last_end);
int  rhj_join_devices(const rhj_tuple *const *d_R, uint64_t nR, const rhj_tuple *const *d_S, uint64_t nS,
                      rhj_result_tuple *const *out, const uint64_t *capacity, uint64_t *matches);
int  rhj_gather_pairs_devices(const rhj_result_tuple *const *lists, const uint64_t *matches, int dst_device,
                              rhj_result_tuple *dst, uint64_t capacity, uint64_t *total);
/* Pair order (env RHJ_ORDER=canonical|any).  0 = canonical (default): the reference's order for the
 * radix width in force — bucket ascending, probe side = R iff cR >= cS, probe tuples in input order, build matches
 * in descending position (rhjoin.c:42-57,86,141-250).  1 = any: the same pairs in the canonical order of a radix
 * width the library picks from the relation sizes (rhj_get_stats().radix_bits says which): for callers whose
 * answers do not depend on the order — the reference's query executor is one: its view sums are order-free
 * (inter_res.c:320-339), tests/test_gpu_dropin.py runs it this way — and who therefore need not run big joins on
 * buckets sized for a CPU cache (100M x 100M on the reference's 4 bits: 7.1 ms in its canonical order, 4.1 ms on the 12 bits the library picks).
 * rhj_partition_device() always uses the radix width in force. */
void rhj_set_order(int any);
int  rhj_get_order(void);
/* the radix width order mode "any" would use for relations of these sizes (pure function: needs no device) */
int  rhj_auto_radix_bits(uint64_t nR, uint64_t nS);
/* How much of a join rhj_get_stats() times with HIP events: 2 (default) every stage, 1 the whole join only, 0 nothing
 * (all ms_* zero).  An event between two launches keeps the second kernel from being fed while the first drains, ~6 us
 * each: 10 % of a 1M x 1M join, 1 % of 100M x 100M (env RHJ_TIMING).  A caller that never reads the stage times — the
 * reference's engine — loses nothing with 0. */
void rhj_set_timing(int level);

/* ---- device-resident entry points (what RadixHashJoin()/Filter() call
 *      after staging; bench.py and the parity tests call them directly) ----- */

typedef struct rhj_stats {
    /* per-stage GPU time of the LAST call in milliseconds (hipEvent pairs on
     * the launch stream) and the launch counts behind them */
    float ms_hist, ms_scan, ms_scatter, ms_plan, ms_build, ms_count, ms_offsets,
          ms_probe, ms_total;
    float ms_h2d, ms_d2h;
    uint64_t n_r, n_s, matches;
    uint64_t units, hbm_units, max_build, table_slots;
    int radix_bits;
    int reserved;      /* path of the last join: 0 tiled, 1 fused, 3 small (fused join behind the two- or three-launch partition of csrc/rhj_small.hip.h), 4 low-radix (csrc/rhj_lowradix.hip.h) */
} rhj_stats;

/* Join two device-resident AoS relations (rhj_tuple[nR], rhj_tuple[nS]).
 * d_out receives up to out_capacity rhj_result_tuple in canonical order.
 * *matches gets the exact match count even when it exceeds out_capacity (then
 * nothing beyond capacity is written and the return value is 1).  Returns 0 on
 * success, <0 on a HIP error (message on stderr). d_out may be NULL with
 * capacity 0 to count only. */
int rhj_join_device(const rhj_tuple *d_R, uint64_t nR,
                    const rhj_tuple *d_S, uint64_t nS,
                    rhj_result_tuple *d_out, uint64_t out_capacity,
                    uint64_t *matches);

/* The join of two relations given as KEY COLUMNS: tuple i of a relation is {keys[i], i} — what GetRelation makes of a base
 * relation (inter_res.c:199-204, :223-227: row_id = i).  On the two-pass partition (9..15 radix bits) its first pass reads the
 * columns themselves, 8 bytes a tuple instead of the 16 of an rhj_tuple; elsewhere the tuples are built first and the ordinary
 * join runs.  Same pairs, same order as rhj_join_device() on the materialised relations. */
int  rhj_join_keys_device(const uint64_t *d_keysR, uint64_t nR, const uint64_t *d_keysS, uint64_t nS, rhj_result_tuple *d_out,
                          uint64_t out_capacity, uint64_t *matches);

/* The stable radix partition alone (SerialReorderArray, preprocess.c:302-362):
 * d_out[n] partitioned tuples, h_hist[2^bits] counts, h_psum[2^bits] starts
 * (-1 for an empty bucket as preprocess.c:336-347 leaves it).  Host arrays may
 * be NULL. */
int rhj_partition_device(const rhj_tuple *d_in, uint64_t n, rhj_tuple *d_out,
                         uint64_t *h_hist, int64_t *h_psum);

/* Filter scan on a device-resident column.  d_sel == NULL: scan col[0..n);
 * else scan col[d_sel[i]] for i in [0,n).  d_out gets the ascending indices i
 * (capacity n).  op is '<', '>' or '='; value is already converted as the
 * reference does ((uint64_t)(int)value, filter.c:116).  */
int rhj_filter_device(const uint64_t *d_col, const uint64_t *d_sel, uint64_t n,
                      char op, uint64_t value, uint64_t *d_out, uint64_t *hits);

/* Bucket-range sharding of ONE join over the GPUs of a node (bucket b of R only meets
 * bucket b of S, rhjoin.c:42-57): the bucket histogram of a device-resident relation on the current
 * radix bits (d_hist: 2^bits u64 on the device; what HistJob counts, preprocess.c:181-195), and the
 * stable selection of the tuples whose bucket lies in [bucket_lo, bucket_hi) — input order and row ids
 * kept, so rhj_join_device() of the selections is the canonical result restricted to that range and
 * the concatenation over the ranks' ranges is the canonical result.  Returns 1 (and the count needed)
 * when capacity was too small. */
int rhj_bucket_histogram_device(const rhj_tuple *d_in, uint64_t n, uint64_t *d_hist);
/* One rank's share in ONE call: the canonical result restricted to the buckets [bucket_lo, bucket_hi) of the current radix,
 * with the relations handed over whole — the join's first partition pass drops the other ranks' buckets while it reads
 * them (one read of each relation per rank; no selection pass, no host round trip in front of the join).  Same return
 * values as rhj_join_device; -3 when the pair order was left to the library (RHJ_ORDER=any: bucket numbers are the
 * caller's radix). */
int rhj_join_device_range(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS,
                          uint32_t bucket_lo, uint32_t bucket_hi,
                          rhj_result_tuple *d_out, uint64_t out_capacity, uint64_t *matches);
/* A share cut inside buckets: the buckets [bucket_lo, bucket_hi), of the first only the probe tuples from first_skip on, of the
 * last (bucket_hi - 1) only those before last_end (0: all) — positions among the tuples of the bucket's probe side (R when
 * |R_b| >= |S_b|, rhjoin.c:86) in partition order.  The result: the canonical list's pairs of exactly those probe tuples
 * (rhjoin.c:141-217 walks a bucket's probe tuples in order), so shares that tile (bucket, position) concatenate to the
 * canonical result.  The foreign-key speculation is not tried on a share that cuts a bucket. */
int rhj_join_device_slice(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS,
                          uint32_t bucket_lo, uint32_t bucket_hi, uint64_t first_skip, uint64_t last_end,
                          rhj_result_tuple *d_out, uint64_t out_capacity, uint64_t *matches);
int rhj_select_bucket_range_device(const rhj_tuple *d_in, uint64_t n, uint32_t bucket_lo, uint32_t bucket_hi,
                                   rhj_tuple *d_out, uint64_t capacity, uint64_t *count);

/* Register a host column store (relation_map.c:28-50 layout: the columns of one relation are one
 * contiguous column-major block of the read-only file mapping): the block is pinned with
 * hipHostRegister (read-only flag first; a range that cannot be pinned is copied pageable), every
 * column is copied to the device once with hipMemcpyAsync, and Filter() and the resident operators
 * read that copy until rhj_unregister_relation_map() (or rhj_release()) drops it.  The registry is
 * the ONLY cache of host columns: a column that was not registered is uploaded on every call that
 * names it, so reusing a host address for other data is always safe.  A caller must unregister a
 * map before it frees or rewrites its columns. */
int  rhj_register_relation_map(const rhj_relation_map *map, int num_relations);
int  rhj_unregister_relation_map(const rhj_relation_map *map, int num_relations);
int  rhj_registered_columns(void);     /* registered columns / pinned host ranges right now */
int  rhj_pinned_ranges(void);
int  rhj_pin_refusals(void);      /* ranges of 64 KiB or more the host refused to pin (copied pageable instead) */
void rhj_release(void);                /* drops the registry, every device buffer and the workspace */

const rhj_stats *rhj_last_stats(void);
const char      *rhj_version(void);

#ifdef RHJ_REFERENCE_NAMES
typedef rhj_tuple        tuple;
typedef rhj_relation     relation;
typedef rhj_result       result;
typedef rhj_result_tuple result_tuple;
typedef rhj_inter_data   inter_data;
typedef rhj_inter_res    inter_res;
typedef rhj_column_stats column_stats;
typedef rhj_relation_map relation_map;
typedef rhj_filter_pred  filter_pred;
typedef struct scheduler scheduler;
#endif

#ifdef __cplusplus
}
#endif
#endif /* RHJ_H */
