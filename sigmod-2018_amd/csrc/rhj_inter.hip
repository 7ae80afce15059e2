// rhj_inter.hip — device-resident intermediate results (include/rhj_inter.h): the reference's
// inter_res.c and the helper half of filter.c restated over row-id tables that live on the GPU.
// Control flow (which node, which relation is active, what becomes NULL) follows the reference
// function by function; every loop over tuples is one gather / scan / reduction kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "rhj.h"
#include "rhj_inter.h"
#include "rhj_internal.h"

namespace {

constexpr int MAX_TABLES = 16;                      // relations per query the gather kernel takes in one launch

struct GatherArgs {
    uint64_t       *dst[MAX_TABLES];
    const uint64_t *src[MAX_TABLES];                 // nullptr: write the index itself
    int             ntab;
};

// dst[t][i] = src[t][idx[i * stride]]: the index is read once per row, every table is a coalesced
// write and a random 8-byte read (inter_res.c:95-101,131-137,304-313; filter.c:73-77)
__global__ __launch_bounds__(256) void k_gather_tables(GatherArgs a, const uint64_t *idx, int stride, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = idx[i * (uint64_t)stride];
#pragma unroll 4
    for (int t = 0; t < a.ntab; ++t) a.dst[t][i] = a.src[t] ? a.src[t][p] : p;
}

// tuples[i] = {col[sel ? sel[i] : i], i}   (inter_res.c:199-204, :223-227)
__global__ __launch_bounds__(256) void k_build_relation(const uint64_t *col, const uint64_t *sel, uint64_t n, rhj_tuple *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = col[sel ? sel[i] : i];
    reinterpret_cast<ulonglong2 *>(out)[i] = make_ulonglong2(v, i);
}

// sum += col[sel[i]] with wrap-around (inter_res.c:329-333)
__global__ __launch_bounds__(256) void k_sum_gather(const uint64_t *col, const uint64_t *sel, uint64_t n, unsigned long long *sum)
{
    __shared__ unsigned long long part[4];
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        s += col[sel ? sel[i] : i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}

// All view sums of a query in ONE launch (inter_res.c:320-339 sums them one view after the other; a launch, a memset, a pageable
// 8-byte copy and a stream wait per view were 78 us a query of `small`).  Grid: blocks x views.  Every block adds its part to the
// view's device word; the last block out (a ticket) hands the sums to pinned host memory and leaves the words and the ticket
// zero for the next query — no memset, no copy: the host waits for the stream and reads.
constexpr int SUM_VIEWS = 8;
struct SumViews {
    const uint64_t *col[SUM_VIEWS];
    const uint64_t *sel[SUM_VIEWS];
    uint64_t        n[SUM_VIEWS];
};
__global__ __launch_bounds__(256) void k_sum_views(SumViews v, unsigned long long *d_sums, uint32_t *d_ticket, unsigned long long *h_sums)
{
    __shared__ unsigned long long part[4];
    const int view = blockIdx.y;
    const uint64_t *col = v.col[view], *sel = v.sel[view];
    const uint64_t n = v.n[view];
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        s += col[sel ? sel[i] : i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long mine = part[0] + part[1] + part[2] + part[3];
        if (mine) __hip_atomic_fetch_add(&d_sums[view], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // release + acquire on the ticket: this block's part is out before it counts itself, the last one sees everybody's
        if (__hip_atomic_fetch_add(d_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1u) {
            for (unsigned i = 0; i < gridDim.y; ++i) {
                const unsigned long long t = __hip_atomic_exchange(&d_sums[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&h_sums[i], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(d_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// new[z][i * n2 + j] = a[z][i] if a[z] else b[z][j]   (inter_res.c:409-421)
struct CartArgs {
    uint64_t       *dst[MAX_TABLES];
    const uint64_t *a[MAX_TABLES];
    const uint64_t *b[MAX_TABLES];
    int             ntab;
};
__global__ __launch_bounds__(256) void k_cartesian(CartArgs c, uint64_t n1, uint64_t n2)
{
    const uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= n1 * n2) return;
    const uint64_t i = x / n2, j = x % n2;
    for (int t = 0; t < c.ntab; ++t) c.dst[t][x] = c.a[t] ? c.a[t][i] : c.b[t][j];
}

// column statistics (relation_map.c:53-84): min / max, then one flag per value of the range
__global__ __launch_bounds__(256) void k_col_minmax(const uint64_t *col, uint64_t n, unsigned long long *mm /*[min, max]*/)
{
    __shared__ unsigned long long lo4[4], hi4[4];
    unsigned long long lo = ~0ull, hi = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const unsigned long long v = col[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long a = __shfl_xor(lo, d, 64), b = __shfl_xor(hi, d, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) { lo4[threadIdx.x >> 6] = lo; hi4[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = lo4[w] < lo ? lo4[w] : lo; hi = hi4[w] > hi ? hi4[w] : hi; }
        lo = lo4[0] < lo ? lo4[0] : lo; hi = hi4[0] > hi ? hi4[0] : hi;
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}
__global__ __launch_bounds__(256) void k_col_flags(const uint64_t *col, uint64_t n, uint64_t lo, uint64_t fold, uint8_t *flags)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = col[i] - lo;
    flags[fold ? x % fold : x] = 1;                  // everybody stores the same byte
}
__global__ __launch_bounds__(256) void k_count_flags(const uint8_t *flags, uint64_t n, unsigned long long *count)
{
    __shared__ unsigned long long part[4];
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) s += flags[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, part[0] + part[1] + part[2] + part[3]);
}

std::unordered_set<const void *> g_rel, g_inter;
// resident result nodes, each with the host slot FindResultTuples hands out for it (a pointer into the list is what results.c
// returns; the list is on the device here, so the caller gets a copy that lives as long as the list — one per list, so that
// elements of two live results do not alias)
std::unordered_map<const void *, rhj_result_tuple> g_res;

// RHJ_TRACE=1: one line per call on stderr (which operator, which relations, how many rows)
bool tracing() { static const bool on = getenv("RHJ_TRACE") != nullptr; return on; }
double trace_ms()
{
    static struct timespec t0 = {0, 0};
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    if (t0.tv_sec == 0 && t0.tv_nsec == 0) t0 = t;
    return (t.tv_sec - t0.tv_sec) * 1e3 + (t.tv_nsec - t0.tv_nsec) * 1e-6;
}
#define TRACE(...) do { if (tracing()) { fprintf(stderr, "rhj-trace %9.2f ms: ", trace_ms()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
void trace_nodes(const rhj_inter_res *h)
{
    if (!tracing()) return;
    for (int k = 0; h; h = h->next, ++k) {
        fprintf(stderr, "rhj-trace:   node %d rows %lu active", k, (unsigned long)h->data->num_tuples);
        for (int j = 0; j < h->num_of_relations; ++j) if (h->data->table[j]) fprintf(stderr, " %d", j);
        fputc('\n', stderr);
    }
}

hipStream_t stream() { return (hipStream_t)rhj_dev_stream(); }

[[noreturn]] void die(const char *what)
{
    // same convention as the rest of the boundary: no error channel, print and exit(2)
    fprintf(stderr, "rhj: %s failed on the device path; there is no CPU fallback\n", what);
    exit(2);
}

void check(hipError_t e, const char *what)
{
    if (e != hipSuccess) { fprintf(stderr, "rhj: %s: %s\n", what, hipGetErrorString(e)); die(what); }
}

uint64_t *alloc_ids(uint64_t n)
{
    uint64_t *p = (uint64_t *)rhj_dev_alloc(n * 8);
    if (!p) die("device allocation");
    return p;
}

rhj_result *make_result(void *dbuff, uint64_t count)
{
    rhj_result *r = (rhj_result *)malloc(sizeof(rhj_result));
    r->buff = (char *)dbuff;
    r->next = nullptr;
    r->current_load = count;
    g_res.emplace(r, rhj_result_tuple{0, 0});
    return r;
}

uint64_t result_count(const rhj_result *res)
{
    uint64_t n = 0;
    for (; res; res = res->next) n += res->current_load;   // results.c:65-76 (there as int)
    return n;
}

const uint64_t *result_ids(const rhj_result *res)
{
    if (res == nullptr) return nullptr;
    if (!rhj_resident_result(res)) die("a host result list handed to the device-resident intermediate results");
    return (const uint64_t *)res->buff;
}

void gather(uint64_t *const *dst, const uint64_t *const *src, int ntab, const uint64_t *idx, int stride, uint64_t n)
{
    if (rhj_gather_tables_device(dst, src, ntab, idx, stride, n)) die("gather");
}

// A column on the device for the duration of one operator: the registered copy (InitRelationMap,
// rhj_register_relation_map) or, for a column nobody registered, an upload that is released when the
// operator has queued its kernels.  Nothing is cached by host address.
struct DevColumn {
    const uint64_t *p;
    void *temp;
    DevColumn(const rhj_relation_map *rm, int column)
    {
        p = rhj_dev_column(rm->columns[column], rm->num_tuples, &temp);
        if (!p) die("staging a column on the device");
    }
    ~DevColumn() { if (temp) rhj_dev_free(temp); }
    DevColumn(const DevColumn &) = delete;
    DevColumn &operator=(const DevColumn &) = delete;
    operator const uint64_t *() const { return p; }
};

// the first node in which `rel` is active (inter_res.c:184-190, :243-249; filter.c:98-104)
rhj_inter_res *node_of(rhj_inter_res *inter, int rel)
{
    while (inter && inter->data->table[rel] == nullptr) inter = inter->next;
    return inter;
}

// Rebuild `node` with `count` rows: every active relation j gets new[j][i] = old[j][idx[i * stride]];
// `fresh_rel` (or -1) becomes active with new[fresh_rel][i] = fresh_idx[i * stride].
// inter_res.c:64-104 / :106-140 / filter.c:45-81.
void rebuild_node(rhj_inter_res *node, uint64_t count, const uint64_t *idx, int stride, int fresh_rel, const uint64_t *fresh_idx)
{
    const int nrel = node->num_of_relations;
    rhj_inter_data *nd = nullptr;
    InitInterData(&nd, nrel, (int)count);
    nd->num_tuples = count;
    std::vector<uint64_t *> dst;
    std::vector<const uint64_t *> src;
    for (int j = 0; j < nrel; ++j) {
        if (node->data->table[j] != nullptr && j != fresh_rel) {
            nd->table[j] = alloc_ids(count);
            dst.push_back(nd->table[j]);
            src.push_back(node->data->table[j]);
        }
    }
    for (size_t at = 0; at < dst.size(); at += MAX_TABLES)
        gather(dst.data() + at, src.data() + at, (int)std::min<size_t>(MAX_TABLES, dst.size() - at), idx, stride, count);
    if (fresh_rel >= 0) {
        nd->table[fresh_rel] = alloc_ids(count);
        uint64_t *d = nd->table[fresh_rel];
        const uint64_t *s = nullptr;
        gather(&d, &s, 1, fresh_idx, stride, count);
    }
    FreeInterData(node->data, nrel);
    node->data = nd;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ identity of resident objects

int rhj_resident_relation(const rhj_relation *rel)
{
    RhjApiLock api_lock;
    return rel && g_rel.count(rel);
}
int rhj_resident_result(const rhj_result *res)
{
    RhjApiLock api_lock;
    return res && g_res.count(res);
}
int rhj_resident_inter(const rhj_inter_res *head)
{
    RhjApiLock api_lock;
    return head && g_inter.count(head);
}

// ------------------------------------------------------------------ device entry points

int rhj_gather_tables_device(uint64_t *const *dst, const uint64_t *const *src, int ntab, const uint64_t *idx,
                             int idx_stride, uint64_t n)
{
    RhjApiLock api_lock;
    if (ntab < 0 || ntab > MAX_TABLES) return -2;
    if (n == 0 || ntab == 0) return 0;
    GatherArgs a;
    a.ntab = ntab;
    for (int t = 0; t < ntab; ++t) { a.dst[t] = dst[t]; a.src[t] = src[t]; }
    hipLaunchKernelGGL(k_gather_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(), a, idx, idx_stride, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rhj_build_relation_device(const uint64_t *d_col, const uint64_t *d_sel, uint64_t n, rhj_tuple *d_tuples)
{
    RhjApiLock api_lock;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_build_relation, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(), d_col, d_sel, n, d_tuples);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rhj_sum_gather_device(const uint64_t *d_col, const uint64_t *d_sel, uint64_t n, uint64_t *sum)
{
    RhjApiLock api_lock;
    *sum = 0;
    if (n == 0) return 0;
    unsigned long long *d_sum = (unsigned long long *)rhj_dev_alloc(8);
    if (!d_sum) return -1;
    hipStream_t s = stream();
    if (hipMemsetAsync(d_sum, 0, 8, s) != hipSuccess) return -1;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_sum_gather, dim3((unsigned)blocks), dim3(256), 0, s, d_col, d_sel, n, d_sum);
    unsigned long long h = 0;
    if (hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    rhj_dev_free(d_sum);
    *sum = h;
    return 0;
}

// up to SUM_VIEWS (column, row-id list or null, rows) sums in one launch; sums[i] receives view i's
int rhj_sum_views_device(int views, const uint64_t *const *d_cols, const uint64_t *const *d_sels, const uint64_t *ns, uint64_t *sums)
{
    RhjApiLock api_lock;
    static unsigned long long *d_words = nullptr;     // [SUM_VIEWS] sums + the ticket behind them, zero between calls
    static unsigned long long *h_words = nullptr;     // pinned
    if (views <= 0) return 0;
    if (views > SUM_VIEWS) return -1;
    hipStream_t s = stream();
    if (!d_words) {                                   // both blocks and the cleared words, or nothing: published only when all three calls went through
        unsigned long long *d = nullptr, *h = nullptr;
        if (hipMalloc((void **)&d, (SUM_VIEWS + 1) * 8) != hipSuccess) return -1;
        if (hipHostMalloc((void **)&h, SUM_VIEWS * 8, hipHostMallocDefault) != hipSuccess) { (void)hipFree(d); return -1; }
        if (hipMemsetAsync(d, 0, (SUM_VIEWS + 1) * 8, s) != hipSuccess) { (void)hipFree(d); (void)hipHostFree(h); return -1; }
        d_words = d; h_words = h;
    }
    SumViews v;
    uint64_t most = 0;
    for (int i = 0; i < SUM_VIEWS; ++i) {
        v.col[i] = i < views ? d_cols[i] : nullptr; v.sel[i] = i < views ? d_sels[i] : nullptr; v.n[i] = i < views ? ns[i] : 0;
        if (v.n[i] > most) most = v.n[i];
    }
    uint64_t blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sum_views, dim3((unsigned)blocks, (unsigned)views), dim3(256), 0, s, v, d_words, (uint32_t *)(d_words + SUM_VIEWS), h_words);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        // the kernel's last block out leaves the words zero for the next call; after a failed launch or wait nobody did
        if (hipMemsetAsync(d_words, 0, (SUM_VIEWS + 1) * 8, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            (void)hipFree(d_words); (void)hipHostFree(h_words);
            d_words = nullptr; h_words = nullptr;             // (start over on the next call)
        }
        return -1;
    }
    for (int i = 0; i < views; ++i) sums[i] = ((volatile unsigned long long *)h_words)[i];
    return 0;
}

// ------------------------------------------------------------------ the reference's boundary, resident side

rhj_result *rhj_resident_join(rhj_relation *relR, rhj_relation *relS)
{
    RhjApiLock api_lock;
    uint64_t m = 0;
    TRACE("RadixHashJoin %lu x %lu", (unsigned long)relR->num_tuples, (unsigned long)relS->num_tuples);
    // The pairs go straight into the block the result list will own (no copy out of the join's workspace): sized for one
    // pair per tuple of the larger side; a join with more matches than that says so (rc 1, exact count) and runs once
    // more into a block of the right size — 4 of the 88 joins of `small`.
    uint64_t cap = (relR->num_tuples > relS->num_tuples ? relR->num_tuples : relS->num_tuples) + 1024;
    void *d = rhj_dev_alloc(cap * sizeof(rhj_result_tuple));
    if (!d) die("RadixHashJoin");
    int rc = rhj_join_device(relR->tuples, relR->num_tuples, relS->tuples, relS->num_tuples, (rhj_result_tuple *)d, cap, &m);
    if (rc == 1) {
        rhj_dev_free(d);
        cap = m;
        d = rhj_dev_alloc(cap * sizeof(rhj_result_tuple));
        if (!d) die("RadixHashJoin");
        rc = rhj_join_device(relR->tuples, relR->num_tuples, relS->tuples, relS->num_tuples, (rhj_result_tuple *)d, cap, &m);
    }
    if (rc != 0) die("RadixHashJoin");
    if (m == 0) {
        rhj_dev_free(d);
        if (rhj_host_null_on_empty()) return nullptr;          // THREADS 1 behaviour
        return make_result(rhj_dev_alloc(8), 0);               // as shipped: an empty head (rhjoin.c:356-359)
    }
    // A selective join over big inputs would keep a block sized for its INPUT alive for as long as the intermediate result
    // lives (1.6 GB for a handful of pairs of 100 M-row relations, and every live intermediate of a plan the same): above a
    // few megabytes, a list that fills less than half of its block moves to a block of its own size (one device copy).
    if (cap * sizeof(rhj_result_tuple) > ((size_t)8 << 20) && m < cap / 2) {
        void *exact = rhj_dev_alloc(m * sizeof(rhj_result_tuple));
        if (exact) {
            if (hipMemcpyAsync(exact, d, m * sizeof(rhj_result_tuple), hipMemcpyDeviceToDevice, stream()) != hipSuccess) die("RadixHashJoin");
            rhj_dev_free(d);                                   // (stream-ordered: the copy is queued in front of any reuse)
            d = exact;
        }
    }
    return make_result(d, m);
}

rhj_result *rhj_resident_filter(rhj_inter_res *head, rhj_filter_pred *filter_p, rhj_relation_map *map, int *query_relations)
{
    RhjApiLock api_lock;
    const int relation = filter_p->relation;
    const rhj_relation_map *rm = &map[query_relations[relation]];
    const char op = filter_p->comperator;
    if (op != '<' && op != '>' && op != '=') {                  // filter.c:184-186
        printf("Wrong comperator in filter function\n");
        exit(2);
    }
    const DevColumn col(rm, filter_p->column);                   // filter.c:96
    rhj_inter_res *node = node_of(head, relation);               // filter.c:98-104
    const uint64_t *sel = node ? node->data->table[relation] : nullptr;
    const uint64_t n = node ? node->data->num_tuples : rm->num_tuples;
    const uint64_t value = (uint64_t)(int64_t)filter_p->value;   // int -> u64, filter.c:116
    if (n == 0) return nullptr;
    uint64_t *ids = alloc_ids(n), hits = 0;
    TRACE("Filter over %lu rows", (unsigned long)n);
    if (rhj_filter_device(col, sel, n, op, value, ids, &hits)) die("Filter");
    if (hits == 0) { rhj_dev_free(ids); return nullptr; }        // filter.c:94,189
    return make_result(ids, hits);
}

void rhj_resident_free_result(rhj_result *res)
{
    RhjApiLock api_lock;
    while (res) {
        rhj_result *t = res;
        res = res->next;
        g_res.erase(t);
        rhj_dev_free(t->buff);
        free(t);
    }
}

void rhj_resident_free_relation(rhj_relation *rel)
{
    RhjApiLock api_lock;
    g_rel.erase(rel);
    rhj_dev_free(rel->tuples);
    free(rel);
}

// the host copy of the element FindResultTuples last fetched from this resident list (unordered_map nodes do not move)
rhj_result_tuple *rhj_resident_slot(const rhj_result *res)
{
    RhjApiLock api_lock;
    auto it = g_res.find(res);
    return it == g_res.end() ? nullptr : &it->second;
}

// element `index` of a resident result, for the results.c accessors (results.c:48-64, :126-142)
int rhj_resident_fetch(const rhj_result *res, uint64_t elem_bytes, uint64_t index, void *dst)
{
    RhjApiLock api_lock;
    if (index >= res->current_load) return -1;
    hipStream_t s = stream();
    if (hipMemcpyAsync(dst, res->buff + index * elem_bytes, elem_bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------ inter_res.c

int InitInterData(rhj_inter_data **head, int num_of_relations, int num_tuples)      // inter_res.c:10-15
{
    RhjApiLock api_lock;
    (*head) = (rhj_inter_data *)malloc(sizeof(rhj_inter_data));
    (*head)->num_tuples = (uint64_t)num_tuples;
    (*head)->table = (uint64_t **)calloc((size_t)num_of_relations, sizeof(uint64_t *));
    return 0;
}

void FreeInterData(rhj_inter_data *head, int num_of_relations)                       // inter_res.c:17-24
{
    RhjApiLock api_lock;
    for (int i = 0; i < num_of_relations; ++i)
        if (head->table[i] != nullptr) rhj_dev_free(head->table[i]);
    free(head->table);
    free(head);
}

int InitInterResults(rhj_inter_res **head, int num_of_rel)                           // inter_res.c:26-32
{
    RhjApiLock api_lock;
    (*head) = (rhj_inter_res *)malloc(sizeof(rhj_inter_res));
    (*head)->next = nullptr;
    (*head)->num_of_relations = num_of_rel;
    InitInterData(&(*head)->data, num_of_rel, 0);
    g_inter.insert(*head);
    return 0;
}

void FreeInterResults(rhj_inter_res *var)                                            // inter_res.c:175-180
{
    RhjApiLock api_lock;
    if (var->next != nullptr) FreeInterResults(var->next);
    FreeInterData(var->data, var->num_of_relations);
    g_inter.erase(var);
    free(var);
}

void PrintInterResults(rhj_inter_res *head)                                          // inter_res.c:154-173
{
    RhjApiLock api_lock;
    int index = 0;
    hipStream_t s = stream();
    while (head != nullptr) {
        const uint64_t n = head->data->num_tuples;
        std::vector<std::vector<uint64_t>> host((size_t)head->num_of_relations);
        for (int j = 0; j < head->num_of_relations; ++j)
            if (head->data->table[j] != nullptr) {
                host[j].resize(n);
                check(hipMemcpyAsync(host[j].data(), head->data->table[j], n * 8, hipMemcpyDeviceToHost, s), "PrintInterResults");
            }
        check(hipStreamSynchronize(s), "PrintInterResults");
        printf("Intermediate results node[%d]: \n", index);
        for (uint64_t i = 0; i < n; i++) {
            printf("Tuple: %5lu|||||", (unsigned long)i);
            for (int j = 0; j < head->num_of_relations; j++) {
                if (head->data->table[j] != nullptr) printf(" %5lu |", (unsigned long)host[j][i]);
                else printf(" NULL |");
            }
            printf("\n");
        }
        printf("--------------------------------------------------\n");
        head = head->next;
        index++;
    }
}

int InsertJoinToInterResults(rhj_inter_res *head, int rel1, int rel2, rhj_result *res)   // inter_res.c:34-152
{
    RhjApiLock api_lock;
    const uint64_t count = result_count(res);
    const uint64_t *pairs = result_ids(res);                    // {row_idR, row_idS} per element
    TRACE("InsertJoinToInterResults rel %d %d pairs %lu", rel1, rel2, (unsigned long)count);
    trace_nodes(head);
    do {
        if (head->data->num_tuples == 0) {
            // first instance of this node: both sides of the pairs become its tables (:39-62)
            head->data->num_tuples = count;
            head->data->table[rel1] = alloc_ids(count);
            head->data->table[rel2] = alloc_ids(count);
            const uint64_t *none = nullptr;
            gather(&head->data->table[rel1], &none, 1, pairs, 2, count);
            gather(&head->data->table[rel2], &none, 1, pairs + 1, 2, count);
            return 1;
        }
        if (head->data->table[rel1] != nullptr && head->data->table[rel2] == nullptr) {
            // rel1 active: row_idR indexes the node's rows, row_idS is rel2's new column (:64-104)
            rebuild_node(head, count, pairs, 2, rel2, pairs + 1);
            return 1;
        }
        if (head->data->table[rel2] != nullptr && head->data->table[rel1] == nullptr) {
            // rel2 active: the mirror case (:106-140)
            rebuild_node(head, count, pairs + 1, 2, rel1, pairs);
            return 1;
        }
        if (head->next == nullptr) break;
        head = head->next;
    } while (1);
    InitInterResults(&(head->next), head->num_of_relations);   // :146-150
    InsertJoinToInterResults(head->next, rel1, rel2, res);
    return 0;
}

rhj_relation *ScanInterResults(int given_rel, int column, rhj_inter_res *inter, rhj_relation_map *map, int *query_relations)
{                                                               // inter_res.c:182-206
    RhjApiLock api_lock;
    rhj_inter_res *node = node_of(inter, given_rel);
    if (node == nullptr) return nullptr;
    const rhj_relation_map *rm = &map[query_relations[given_rel]];
    rhj_relation *rel = (rhj_relation *)malloc(sizeof(rhj_relation));
    rel->num_tuples = node->data->num_tuples;
    rel->tuples = (rhj_tuple *)rhj_dev_alloc(rel->num_tuples * sizeof(rhj_tuple));
    if (!rel->tuples) die("GetRelation");
    if (rhj_build_relation_device(DevColumn(rm, column), node->data->table[given_rel], rel->num_tuples, rel->tuples)) die("GetRelation");
    g_rel.insert(rel);
    return rel;
}

rhj_relation *GetRelation(int given_rel, int column, rhj_inter_res *inter, rhj_relation_map *map, int *query_relations)
{                                                               // inter_res.c:208-231
    RhjApiLock api_lock;
    rhj_relation *rel = nullptr;
    TRACE("GetRelation rel %d col %d", given_rel, column);
    if (inter != nullptr && (rel = ScanInterResults(given_rel, column, inter, map, query_relations)) != nullptr) return rel;
    const rhj_relation_map *rm = &map[query_relations[given_rel]];
    rel = (rhj_relation *)malloc(sizeof(rhj_relation));
    rel->num_tuples = rm->num_tuples;
    rel->tuples = (rhj_tuple *)rhj_dev_alloc(rel->num_tuples * sizeof(rhj_tuple));
    if (!rel->tuples) die("GetRelation");
    if (rhj_build_relation_device(DevColumn(rm, column), nullptr, rel->num_tuples, rel->tuples)) die("GetRelation");
    g_rel.insert(rel);
    return rel;
}

rhj_result *SelfJoin(int given_rel, int column1, int column2, rhj_inter_res **inter, rhj_relation_map *map, int *query_relations)
{                                                               // inter_res.c:234-263 (intended semantics, see rhj_inter.h)
    RhjApiLock api_lock;
    const rhj_relation_map *rm = &map[query_relations[given_rel]];
    const DevColumn c1(rm, column1), c2(rm, column2);
    rhj_inter_res *node = node_of(*inter, given_rel);
    const uint64_t *sel = node ? node->data->table[given_rel] : nullptr;
    const uint64_t n = node ? node->data->num_tuples : rm->num_tuples;
    if (n == 0) return nullptr;
    uint64_t *ids = alloc_ids(n), hits = 0;
    if (rhj_filter_eq2_device(c1, sel, c2, sel, n, ids, &hits)) die("SelfJoin");
    if (hits == 0) { rhj_dev_free(ids); return nullptr; }
    return make_result(ids, hits);
}

void Merge(rhj_inter_res **head, rhj_inter_res **node, int rel_num)                // inter_res.c:287-318
{
    RhjApiLock api_lock;
    rhj_inter_res *h = *head, *victim = (*node)->next;
    const int nrel = h->num_of_relations;
    TRACE("Merge on rel %d: head rows %lu, victim rows %lu", rel_num, (unsigned long)h->data->num_tuples, (unsigned long)victim->data->num_tuples);
    const uint64_t n = h->data->num_tuples;
    // head's rel_num column holds row positions of `victim`; pull every relation active there
    std::vector<uint64_t *> dst;
    std::vector<const uint64_t *> src;
    for (int j = 0; j < nrel; ++j) {
        if (victim->data->table[j] == nullptr) continue;
        if (h->data->table[j] == nullptr) h->data->table[j] = alloc_ids(n);
        if (j == rel_num) continue;                             // the index column itself goes last
        dst.push_back(h->data->table[j]);
        src.push_back(victim->data->table[j]);
    }
    const uint64_t *idx = h->data->table[rel_num];
    for (size_t at = 0; at < dst.size(); at += MAX_TABLES)
        gather(dst.data() + at, src.data() + at, (int)std::min<size_t>(MAX_TABLES, dst.size() - at), idx, 1, n);
    {   // table[rel_num][i] = victim[rel_num][table[rel_num][i]]: element-wise in place, like the reference's loop
        uint64_t *d = h->data->table[rel_num];
        const uint64_t *s = victim->data->table[rel_num];
        gather(&d, &s, 1, idx, 1, n);
    }
    (*node)->next = victim->next;
    FreeInterData(victim->data, nrel);
    g_inter.erase(victim);
    free(victim);
}

static bool inter_active(const rhj_inter_res *node, int rel) { return node->data->table[rel] != nullptr; }

// inter_res.c:265-284.  A relation that is active in a node AND in a later node ties the two together: the later node is
// merged into the earlier one through that relation (Merge unlinks it), for every node of the list in turn.  The walk over
// the later nodes steps past the node that follows a merged one without testing it in the same sweep — as the reference's
// does; which node meets which first decides the row order of the merged tables, so the sweep order is part of the result.
void MergeInterNodes(rhj_inter_res **inter)
{
    RhjApiLock api_lock;
    for (rhj_inter_res **owner = inter; *owner != nullptr && (*owner)->next != nullptr; owner = &(*owner)->next) {
        for (int rel = 0; rel < (*owner)->num_of_relations; ++rel) {
            if (!inter_active(*owner, rel)) continue;
            for (rhj_inter_res *before = *owner; before != nullptr && before->next != nullptr; before = before->next)
                if (inter_active(before->next, rel)) Merge(owner, &before, rel);    // before->next is gone afterwards
        }
        if ((*owner)->next == nullptr) break;
    }
}

void CalculateQueryResults(rhj_inter_res *inter, rhj_relation_map *map, rhj_batch_listnode *query)   // inter_res.c:320-339
{
    RhjApiLock api_lock;
    TRACE("CalculateQueryResults");
    trace_nodes(inter);
    const int nv = query->views->num_of_elements;
    for (int v0 = 0; v0 < nv; v0 += SUM_VIEWS) {      // (all views of the query in one launch)
        const int m = nv - v0 < SUM_VIEWS ? nv - v0 : SUM_VIEWS;
        const uint64_t *cols[SUM_VIEWS], *sels[SUM_VIEWS];
        uint64_t ns[SUM_VIEWS], sums[SUM_VIEWS];
        for (int i = 0; i < m; i++) {
            const int index = query->views->data[v0 + i][0] - '0';
            const int relation = query->relations[index];
            const int column = query->views->data[v0 + i][2] - '0';
            cols[i] = DevColumn(&map[relation], column);
            sels[i] = inter->data->table[index];
            ns[i] = inter->data->num_tuples;
        }
        if (rhj_sum_views_device(m, cols, sels, ns, sums)) die("CalculateQueryResults");
        for (int i = 0; i < m; i++) {
            printf("%lu", (unsigned long)sums[i]);
            if (v0 + i != nv - 1) printf(" ");
        }
    }
    printf("\n");
}

void PrintNullResults(rhj_batch_listnode *query)                                    // inter_res.c:341-350
{
    RhjApiLock api_lock;
    TRACE("PrintNullResults");
    for (int i = 0; i < query->views->num_of_elements; i++) {
        printf("NULL");
        if (i != query->views->num_of_elements - 1) printf(" ");
    }
    printf("\n");
}

int AreActiveInInter(rhj_inter_res *inter, int rel1, int rel2)                      // inter_res.c:352-361
{
    RhjApiLock api_lock;
    while (inter != nullptr) {
        if (inter->data->table[rel1] != nullptr && inter->data->table[rel2] != nullptr) return 1;
        inter = inter->next;
    }
    return 0;
}

int JoinInterNode(rhj_inter_res **inter, rhj_relation_map *rel_map, int rel1, int col1, int rel2, int col2, int *relations)
{                                                               // inter_res.c:363-389
    RhjApiLock api_lock;
    rhj_inter_res *node = (*inter);
    while (node != nullptr) {
        if (node->data->table[rel1] != nullptr && node->data->table[rel2] != nullptr) break;
        node = node->next;
    }
    if (node == nullptr) return 0;
    const uint64_t n = node->data->num_tuples;
    rhj_result *res = nullptr;
    TRACE("JoinInterNode %d.%d = %d.%d over %lu rows", rel1, col1, rel2, col2, (unsigned long)n);
    trace_nodes(*inter);
    if (n != 0) {
        uint64_t *ids = alloc_ids(n), hits = 0;
        const DevColumn cA(&rel_map[relations[rel1]], col1), cB(&rel_map[relations[rel2]], col2);
        if (rhj_filter_eq2_device(cA, node->data->table[rel1], cB, node->data->table[rel2], n, ids, &hits))
            die("JoinInterNode");
        TRACE("JoinInterNode hits %lu", (unsigned long)hits);
        if (hits == 0) rhj_dev_free(ids);
        else res = make_result(ids, hits);                      // positions inside the node (:381)
    }
    InsertSingleRowIdsToInterResult(inter, rel1, res);
    if (res) rhj_resident_free_result(res);
    return 1;
}

void CartesianInterResults(rhj_inter_res **inter)                                   // inter_res.c:391-428
{
    RhjApiLock api_lock;
    rhj_inter_res *temp = (*inter);
    if (temp->next == nullptr) return;
    TRACE("CartesianInterResults");
    CartesianInterResults(&temp->next);
    rhj_inter_res *next_node = temp->next;
    const uint64_t n1 = temp->data->num_tuples, n2 = next_node->data->num_tuples;
    if (n1 * n2 == 0) return;
    const int nrel = temp->num_of_relations;
    rhj_inter_data *nd = nullptr;
    InitInterData(&nd, nrel, 0);
    nd->num_tuples = n1 * n2;
    CartArgs c;
    c.ntab = 0;
    for (int z = 0; z < nrel; ++z) {
        if (temp->data->table[z] == nullptr && next_node->data->table[z] == nullptr) continue;
        nd->table[z] = alloc_ids(n1 * n2);
        c.dst[c.ntab] = nd->table[z];
        c.a[c.ntab] = temp->data->table[z];
        c.b[c.ntab] = next_node->data->table[z];
        if (++c.ntab == MAX_TABLES) {
            hipLaunchKernelGGL(k_cartesian, dim3((unsigned)((n1 * n2 + 255) / 256)), dim3(256), 0, stream(), c, n1, n2);
            c.ntab = 0;
        }
    }
    if (c.ntab) hipLaunchKernelGGL(k_cartesian, dim3((unsigned)((n1 * n2 + 255) / 256)), dim3(256), 0, stream(), c, n1, n2);
    check(hipGetLastError(), "CartesianInterResults");
    FreeInterData(temp->data, nrel);
    temp->data = nd;
    FreeInterData(next_node->data, nrel);
    g_inter.erase(next_node);
    free(next_node);
    temp->next = nullptr;
}

// ------------------------------------------------------------------ filter.c:11-89

int InsertSingleRowIdsToInterResult(rhj_inter_res **head, int relation_num, rhj_result *res)
{
    RhjApiLock api_lock;
    const uint64_t count = result_count(res);
    const uint64_t *ids = result_ids(res);
    rhj_inter_res *node = *head, *last = nullptr;
    TRACE("InsertSingleRowIdsToInterResult rel %d ids %lu", relation_num, (unsigned long)count);
    for (; node != nullptr; last = node, node = node->next) {
        if (node->data->num_tuples == 0) {
            // first instance of the node: the ids are the relation's row ids (filter.c:19-40)
            node->data->num_tuples = count;
            node->data->table[relation_num] = alloc_ids(count);
            const uint64_t *none = nullptr;
            gather(&node->data->table[relation_num], &none, 1, ids, 1, count);
            return 1;
        }
        if (node->data->table[relation_num] != nullptr) {
            // already active: keep the rows the ids name, for every active relation (filter.c:42-82)
            rebuild_node(node, count, ids, 1, -1, nullptr);
            return 1;
        }
    }
    // not active anywhere and the first node is taken: a new node at the end of the list
    // (filter.c:83-88 dereferences the NULL it walked to; this is what it means to do)
    InitInterResults(&last->next, last->num_of_relations);
    return InsertSingleRowIdsToInterResult(&last->next, relation_num, res);
}


// ------------------------------------------------------------------ relation_map.c

int rhj_column_stats_device(const uint64_t *d_col, uint64_t n, uint64_t *l, uint64_t *u, double *d)
{
    RhjApiLock api_lock;
    if (n == 0) { *l = *u = 0; *d = 0; return 0; }
    hipStream_t s = stream();
    unsigned long long *acc = (unsigned long long *)rhj_dev_alloc(32);      // min, max, count
    if (!acc) return -1;
    const unsigned long long init[3] = {~0ull, 0ull, 0ull};
    unsigned long long h[3];
    if (hipMemcpyAsync(acc, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_col_minmax, dim3((unsigned)blocks), dim3(256), 0, s, d_col, n, acc);
    if (hipMemcpyAsync(h, acc, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    *l = h[0]; *u = h[1];
    // relation_map.c:66-84: one flag per value of [l, u], at most 50 000 000 flags; a larger (or exactly that
    // large) range is folded modulo 5 000 000 into the same 50 000 000-entry array before counting
    uint64_t size = h[1] - h[0] + 1;
    if (size > 50000000ull || size == 0) size = 50000000ull;
    const uint64_t fold = size < 50000000ull ? 0 : 5000000ull;
    uint8_t *flags = (uint8_t *)rhj_dev_alloc(size);
    if (!flags) return -1;
    if (hipMemsetAsync(flags, 0, size, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_col_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_col, n, h[0], fold, flags);
    uint64_t cb = (size + 255) / 256;
    if (cb > 2048) cb = 2048;
    hipLaunchKernelGGL(k_count_flags, dim3((unsigned)cb), dim3(256), 0, s, flags, size, acc + 2);
    if (hipMemcpyAsync(h + 2, acc + 2, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    *d = (double)h[2];
    rhj_dev_free(flags);
    rhj_dev_free(acc);
    return 0;
}

// A relation file (relation_map.c:21-50): two 64-bit words — rows, columns — then the columns one after the other.
struct RelationFileHeader { uint64_t rows, columns; };

// maps a relation file read-only; the descriptor stays open in the list node as in the reference (FreeRelationList closes it)
static const RelationFileHeader *map_relation_file(rhj_relation_listnode *node, size_t *bytes)
{
    node->fd = open(node->filename, O_RDONLY);
    if (node->fd < 0) return nullptr;
    struct stat info;
    void *base = MAP_FAILED;
    const RelationFileHeader *h = nullptr;
    if (fstat(node->fd, &info) == 0 && (size_t)info.st_size >= sizeof(RelationFileHeader))
        base = mmap(nullptr, (size_t)info.st_size, PROT_READ, MAP_PRIVATE, node->fd, 0);
    if (base == MAP_FAILED) fprintf(stderr, "rhj: cannot map relation file %s\n", node->filename);
    else {
        h = static_cast<const RelationFileHeader *>(base);
        // rows x columns x 8 bytes behind the header, without trusting the product of two 64-bit words of the file not to wrap
        const uint64_t words = ((uint64_t)info.st_size - sizeof(RelationFileHeader)) / sizeof(uint64_t);
        if (h->columns != 0 && h->rows > words / h->columns) {
            fprintf(stderr, "rhj: relation file %s is shorter than its header says\n", node->filename);
            munmap(base, (size_t)info.st_size);
            h = nullptr;
        }
    }
    if (h == nullptr) { close(node->fd); node->fd = -1; return nullptr; }   // nothing of a refused file stays open or mapped
    *bytes = (size_t)info.st_size;
    return h;
}

int InitRelationMap(rhj_relation_listnode *head, rhj_relation_map *rel_map)           // relation_map.c:13-88
{
    RhjApiLock api_lock;
    TRACE("InitRelationMap begins");
    int loaded = 0;
    uint64_t most_rows = 0;
    for (rhj_relation_listnode *node = head; node != nullptr; node = node->next, ++loaded) {
        size_t bytes = 0;
        const RelationFileHeader *file = map_relation_file(node, &bytes);
        if (file == nullptr) return 1;
        rhj_relation_map &rm = rel_map[loaded];
        rm.num_tuples = file->rows;
        rm.num_columns = file->columns;
        rm.col_stats = (rhj_column_stats *)malloc(rm.num_columns * sizeof(rhj_column_stats));
        rm.columns = (uint64_t **)malloc(rm.num_columns * sizeof(uint64_t *));
        uint64_t *first = const_cast<uint64_t *>(reinterpret_cast<const uint64_t *>(file + 1));
        for (uint64_t c = 0; c < rm.num_columns; ++c) {
            rm.columns[c] = first + c * rm.num_tuples;
            // the device copy every later operator reads (registered until FreeRelationMap; the relation's columns are one
            // block of the file mapping, pinned for the copy), and the optimiser's statistics computed on it
            if (rhj_dev_register_column(rm.columns[c], rm.num_tuples, c == 0 ? (const void *)first : nullptr, rm.num_columns * rm.num_tuples * 8))
                die("InitRelationMap");
            rhj_column_stats *st = &rm.col_stats[c];
            st->f = (double)rm.num_tuples;
            if (rhj_column_stats_device(DevColumn(&rm, (int)c), rm.num_tuples, &st->l, &st->u, &st->d)) die("InitRelationMap");
        }
        if (rm.num_tuples > most_rows) most_rows = rm.num_tuples;
        TRACE("InitRelationMap: relation %d loaded", loaded);
    }
    if (most_rows) (void)rhj_dev_reserve(most_rows);    // the joins' workspace for inputs of the base relations' size
    return 0;
}

void FreeRelationMap(rhj_relation_map *rel_map, int map_size)                        // relation_map.c:90-98
{
    RhjApiLock api_lock;
    for (int i = 0; i < map_size; ++i) {
        for (uint64_t j = 0; j < rel_map[i].num_columns; ++j) rhj_dev_unregister_column(rel_map[i].columns[j]);
        free(rel_map[i].columns);
        free(rel_map[i].col_stats);
    }
    free(rel_map);
}

void PrintRelationMap(rhj_relation_map *rel_map, int map_size)                       // relation_map.c:100-115
{
    RhjApiLock api_lock;
    for (int i = 0; i < map_size; ++i) {
        printf("Printing Relation: %d\n", i);
        printf("%lu %lu\n", (unsigned long)rel_map[i].num_tuples, (unsigned long)rel_map[i].num_columns);
        for (uint64_t j = 0; j < rel_map[i].num_columns; ++j) {
            printf("Printing Column: %d\n", (int)j);
            for (uint64_t k = 0; k < rel_map[i].num_tuples; ++k) printf(" %lu\n", (unsigned long)rel_map[i].columns[j][k]);
        }
    }
}

}  // extern "C"
