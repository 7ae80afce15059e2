// host_san_driver.cpp — the host-only side of librhj.so (csrc/rhj_abi.c: result lists, node pool, reference entry points;
// csrc/rhj_host.cpp: API lock, ring-of-staging-blocks mover threads) under -fsanitize=address,undefined / thread, on a
// machine without a GPU: `make -C sigmod-2018_amd asan tsan` builds and runs this.  The device side is replaced by the
// stand-ins below: rhj_host_join joins on the host (nested loops over a hash of R) and hands the pairs to the REAL
// rhj_move_blocks through a ring of staging buffers, the way rhj_device.hip hands it the D2H copies.
#include "rhj.h"
#include "rhj_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <unordered_map>
#include <vector>

// ---- stand-ins for the device side (rhj_device.hip / rhj_inter.hip) ------------------------------------------------------
static uint64_t g_node_pairs = 65535;
extern "C" {
int rhj_host_null_on_empty(void) { return 0; }
uint64_t rhj_host_node_pairs(void) { return g_node_pairs; }
int rhj_resident_relation(const rhj_relation *) { return 0; }
int rhj_resident_result(const rhj_result *) { return 0; }
int rhj_resident_inter(const rhj_inter_res *) { return 0; }
rhj_result *rhj_resident_join(rhj_relation *, rhj_relation *) { return nullptr; }
rhj_result *rhj_resident_filter(rhj_inter_res *, rhj_filter_pred *, rhj_relation_map *, int *) { return nullptr; }
void rhj_resident_free_result(rhj_result *) {}
void rhj_resident_free_relation(rhj_relation *) {}
int rhj_resident_fetch(const rhj_result *, uint64_t, uint64_t, void *) { return -1; }
rhj_result_tuple *rhj_resident_slot(const rhj_result *) { return nullptr; }

struct FakeCopy { const rhj_result_tuple *src; uint64_t total, blk; char *const *staging; int ring; };
static int fake_issue(void *c, uint64_t b)
{
    const FakeCopy *k = (const FakeCopy *)c;
    const uint64_t cnt = k->total - b * k->blk < k->blk ? k->total - b * k->blk : k->blk;
    memcpy(k->staging[b % (uint64_t)k->ring], k->src + b * k->blk, cnt * sizeof(rhj_result_tuple));   // the "D2H copy"
    return 0;
}
static int fake_wait(void *, uint64_t) { return 0; }

int rhj_host_join(const rhj_tuple *R, uint64_t nR, const rhj_tuple *S, uint64_t nS, uint64_t *matches,
                  void *(*alloc_chunk)(void *ctx, uint64_t pairs), void *ctx, uint64_t node_pairs)
{
    RhjApiLock api_lock;
    std::unordered_multimap<uint64_t, uint64_t> idx;
    for (uint64_t i = 0; i < nR; ++i) idx.emplace(R[i].value, R[i].row_id);
    std::vector<rhj_result_tuple> out;
    for (uint64_t j = 0; j < nS; ++j) {
        auto rng = idx.equal_range(S[j].value);
        for (auto it = rng.first; it != rng.second; ++it) out.push_back(rhj_result_tuple{it->second, S[j].row_id});
    }
    const uint64_t M = out.size();
    *matches = M;
    if (M == 0) return 0;
    if (node_pairs == 0) node_pairs = M;
    const uint64_t nnodes = (M + node_pairs - 1) / node_pairs;
    std::vector<char *> nodes((size_t)nnodes);
    for (uint64_t i = 0; i < nnodes; ++i) {
        nodes[(size_t)i] = (char *)alloc_chunk(ctx, M - i * node_pairs < node_pairs ? M - i * node_pairs : node_pairs);
        if (!nodes[(size_t)i]) return -1;
    }
    const int ring = 4;
    const uint64_t blk = 3001;                         // small odd blocks: many ring wrap-arounds, slices that straddle nodes
    std::vector<std::vector<char>> bufs(ring, std::vector<char>(blk * sizeof(rhj_result_tuple)));
    char *staging[ring];
    for (int i = 0; i < ring; ++i) staging[i] = bufs[(size_t)i].data();
    FakeCopy cp = {out.data(), M, blk, staging, ring};
    return rhj_move_blocks(M, sizeof(rhj_result_tuple), node_pairs, nodes.data(), blk, ring, staging, 4, fake_issue, fake_wait, &cp);
}

int rhj_host_filter(const uint64_t *col, uint64_t, const uint64_t *sel, uint64_t n, char op, uint64_t value, uint64_t *hits,
                    void *(*alloc_chunk)(void *ctx, uint64_t ids), void *ctx, uint64_t node_ids)
{
    RhjApiLock api_lock;
    std::vector<uint64_t> ids;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t v = col[sel ? sel[i] : i];
        if (op == '<' ? v < value : op == '>' ? v > value : v == value) ids.push_back(i);
    }
    *hits = ids.size();
    for (uint64_t at = 0; at < ids.size(); at += node_ids) {
        const uint64_t cnt = ids.size() - at < node_ids ? ids.size() - at : node_ids;
        char *dst = (char *)alloc_chunk(ctx, cnt);
        if (!dst) return -1;
        memcpy(dst, ids.data() + at, cnt * 8);
    }
    return 0;
}
}   // extern "C"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_san_driver: %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static rhj_relation *make_relation(uint64_t n, uint64_t domain, uint64_t seed)
{
    rhj_relation *r = (rhj_relation *)malloc(sizeof(rhj_relation));
    r->tuples = (rhj_tuple *)malloc((n ? n : 1) * sizeof(rhj_tuple));
    r->num_tuples = n;
    uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
    for (uint64_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        r->tuples[i].value = x % domain;
        r->tuples[i].row_id = i;
    }
    return r;
}

static uint64_t list_digest(rhj_result *res, uint64_t *count)
{
    uint64_t h = 0, n = 0;
    for (; res; res = res->next) {
        CHECK(res->current_load > 0 || res->next == nullptr);          // no empty node before a non-empty one
        const rhj_result_tuple *p = (const rhj_result_tuple *)res->buff;
        for (uint64_t i = 0; i < res->current_load; ++i, ++n) h = (h ^ (p[i].row_idR * 1315423911ull + p[i].row_idS)) * 0x100000001b3ull;
    }
    *count = n;
    return h;
}

int main()
{
    // 1. result-list API (results.c behaviour): appends across node boundaries, lookups, free
    {
        rhj_result *head = nullptr;
        const int n = 8192 * 2 + 5;
        for (int i = 0; i < n; ++i) { rhj_result_tuple t = {(uint64_t)i, (uint64_t)2 * i}; InsertResult(&head, &t); }
        CHECK(GetResultNum(head) == n);
        CHECK(FindResultTuples(head, 8192)->row_idR == 8192 && FindResultTuples(head, n - 1)->row_idS == 2ull * (n - 1));
        CHECK(FindResultTuples(head, n) == nullptr && FindResultTuples(head, -1) == nullptr);
        FreeResult(head);
        rhj_result *ids = nullptr;
        for (uint64_t i = 0; i < 131072 + 7; ++i) InsertRowIdResult(&ids, &i);
        CHECK(GetResultNum(ids) == 131072 + 7 && FindResultRowId(ids, 131072 + 3) == 131072 + 3);
        FreeResult(ids);
    }
    // 2. RadixHashJoin() through the real list assembly, node pool and mover threads, twice (the second list takes recycled nodes)
    uint64_t want_count = 0, want_digest = 0;
    for (int round = 0; round < 2; ++round) {
        rhj_relation *R = make_relation(60000, 20000, 1), *S = make_relation(90000, 20000, 2);
        rhj_result *res = RadixHashJoin(R, S, nullptr);
        uint64_t cnt = 0;
        const uint64_t d = list_digest(res, &cnt);
        CHECK(cnt > 200000 && (uint64_t)GetResultNum(res) == cnt);
        if (round == 0) { want_count = cnt; want_digest = d; } else CHECK(cnt == want_count && d == want_digest);
        FreeResult(res);
        FreeRelation(R); FreeRelation(S);
    }
    // 3. the same from four host threads at once: API lock, node pool and mover threads under contention
    {
        std::vector<std::thread> th;
        std::vector<int> ok(4, 0);
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                for (int k = 0; k < 3; ++k) {
                    rhj_relation *R = make_relation(60000, 20000, 1), *S = make_relation(90000, 20000, 2);
                    rhj_result *res = RadixHashJoin(R, S, nullptr);
                    uint64_t cnt = 0;
                    const uint64_t d = list_digest(res, &cnt);
                    if (cnt == want_count && d == want_digest) ++ok[(size_t)t];
                    FreeResult(res);
                    FreeRelation(R); FreeRelation(S);
                }
            });
        for (auto &x : th) x.join();
        for (int t = 0; t < 4; ++t) CHECK(ok[(size_t)t] == 3);
    }
    // 4. node sizes that do not divide the blocks; one-node lists; empty result (head with load 0 as shipped)
    for (uint64_t np : {(uint64_t)1000, (uint64_t)65535, (uint64_t)0}) {
        g_node_pairs = np;
        rhj_relation *R = make_relation(5000, 300, 3), *S = make_relation(7000, 300, 4);
        rhj_result *res = RadixHashJoin(R, S, nullptr);
        uint64_t cnt = 0;
        list_digest(res, &cnt);
        CHECK(cnt > 50000);
        FreeResult(res); FreeRelation(R); FreeRelation(S);
    }
    g_node_pairs = 65535;
    {
        rhj_relation *R = make_relation(100, 1000, 5), *S = make_relation(100, 1000, 6);
        for (uint64_t i = 0; i < 100; ++i) S->tuples[i].value += 5000;   // disjoint
        rhj_result *res = RadixHashJoin(R, S, nullptr);
        CHECK(res != nullptr && res->current_load == 0 && res->next == nullptr);
        FreeResult(res); FreeRelation(R);
        S->num_tuples = 0;
        R = make_relation(10, 10, 7);
        CHECK(RadixHashJoin(R, S, nullptr) == nullptr);                // rhjoin.c:15-16
        FreeRelation(R); FreeRelation(S);
    }
    // 5. scheduler tokens, pool release
    struct scheduler *sc = nullptr;
    CHECK(SchedulerInit(&sc, 4) == 0 && sc != nullptr && SchedulerDestroy(sc) == 0);
    rhj_host_pool_release();
    printf("host_san_driver ok (%llu pairs per join)\n", (unsigned long long)want_count);
    return 0;
}
