/*
 * ref_wrap.c — flat, ctypes-callable wrappers around the REFERENCE's own
 * RadixHashJoin / ReorderArray / Filter, compiled by oracle/Makefile together
 * with the reference sources where they lie under /root/reference (never copied
 * into this repository).  Output: oracle/_ref/libref_n<N_LSB>_t<THREADS>.so.
 *
 * TEST INFRASTRUCTURE ONLY (golden-vector generation, oracle validation, and the
 * "reference" CPU baseline of bench.py).  This file is mine; it only calls the
 * reference through its public headers.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "structs.h"
#include "rhjoin.h"
#include "preprocess.h"
#include "results.h"
#include "scheduler.h"
#include "filter.h"
#include "inter_res.h"

static scheduler *g_sched = NULL;

static scheduler *sched_get(void)
{
#if THREADS > 1
    if (!g_sched) SchedulerInit(&g_sched, THREADS);
#endif
    return g_sched;
}

int ref_threads(void) { return THREADS; }
int ref_radix_bits(void) { return N_LSB; }

/* Flatten a result list of 16-byte pairs.  *was_null: the call returned NULL.
 * *nodes: number of list nodes (shape is not part of the contract, reported for
 * the record).  *out is malloc'd. */
int ref_join(const tuple *R, uint64_t nR, const tuple *S, uint64_t nS,
             result_tuple **out, uint64_t *count, int *was_null, uint64_t *nodes,
             double *seconds)
{
    relation relR = { (tuple *)R, nR };
    relation relS = { (tuple *)S, nS };
    struct timespec t0, t1;
    scheduler *sc = sched_get();
    clock_gettime(CLOCK_MONOTONIC, &t0);
    result *res = RadixHashJoin(&relR, &relS, sc);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);

    *was_null = (res == NULL);
    *count = 0;
    *out = NULL;
    if (nodes) *nodes = 0;
    uint64_t total = 0, nn = 0;
    for (result *p = res; p; p = p->next) { total += p->current_load; nn++; }
    if (nodes) *nodes = nn;
    if (total) {
        result_tuple *flat = malloc(total * sizeof(result_tuple));
        uint64_t at = 0;
        for (result *p = res; p; p = p->next) {
            memcpy(flat + at, p->buff, p->current_load * sizeof(result_tuple));
            at += p->current_load;
        }
        *out = flat;
        *count = total;
    }
    FreeResult(res);
    return 0;
}

/* The partition as the compiled variant performs it (SerialReorderArray for
 * THREADS 1, ReorderArray over the job pool otherwise; the latter needs both
 * relations, so the same input is passed twice). */
int ref_partition(const tuple *in, uint64_t n, tuple *out, uint64_t *hist, int64_t *psum)
{
    relation rel = { (tuple *)in, n };
    reordered_relation *nr = NULL;
#if THREADS > 1
    reordered_relation *ns = NULL;
    ReorderArray(&rel, &rel, &nr, &ns, sched_get());
    if (ns) FreeReorderRelation(ns);
#else
    SerialReorderArray(&rel, &nr);
#endif
    if (!nr) return 1;
    memcpy(out, nr->rel_array->tuples, n * sizeof(tuple));
    memcpy(hist, nr->hist, nr->hist_size * sizeof(uint64_t));
    memcpy(psum, nr->psum, nr->hist_size * sizeof(int64_t));
    FreeReorderRelation(nr);
    return 0;
}

/* Filter over one column.  sel != NULL puts the relation into the intermediate
 * result first (table[0] = sel, num_tuples = n), which is the indirection mode
 * of filter.c:124-133.  Returns the hit count; out has capacity n. */
uint64_t ref_filter(const uint64_t *col, uint64_t col_rows, const uint64_t *sel, uint64_t n,
                    char op, int value, uint64_t *out, int *was_null)
{
    uint64_t *cols[1] = { (uint64_t *)col };
    relation_map map;
    memset(&map, 0, sizeof(map));
    map.num_tuples = col_rows;
    map.num_columns = 1;
    map.columns = cols;
    int query_relations[1] = { 0 };
    filter_pred fp = { 0, 0, value, op };

    inter_res *ir = NULL;
    InitInterResults(&ir, 1);
    if (sel) {
        ir->data->num_tuples = n;
        ir->data->table[0] = malloc((n ? n : 1) * sizeof(uint64_t));
        memcpy(ir->data->table[0], sel, n * sizeof(uint64_t));
    }
    result *res = Filter(ir, &fp, &map, query_relations);
    *was_null = (res == NULL);
    uint64_t at = 0;
    for (result *p = res; p; p = p->next) {
        memcpy(out + at, p->buff, p->current_load * sizeof(uint64_t));
        at += p->current_load;
    }
    FreeResult(res);
    FreeInterResults(ir);
    return at;
}

uint64_t ref_find_next_prime(uint64_t n) { return FindNextPrime(n); }

void ref_free(void *p) { free(p); }

/* ---- hooks for the full engine build (oracle/_ref/radixhash_dump_*): query.c is
 * compiled with -DRadixHashJoin=ref_hook_join -DFilter=ref_hook_filter so that
 * ExecuteQuery's two boundary calls (query.c:356, :438) land here; the hooks
 * record what crossed the boundary and forward to the real functions. ---- */
#ifdef REF_ENGINE_HOOKS
static FILE *g_dump = NULL;
static FILE *dump_file(void)
{
    if (!g_dump) {
        const char *path = getenv("REF_DUMP");
        g_dump = path ? fopen(path, "wb") : NULL;
    }
    return g_dump;
}

static void put_u64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }

result *ref_hook_join(relation *relR, relation *relS, scheduler *sched)
{
    result *res = RadixHashJoin(relR, relS, sched);
    FILE *f = dump_file();
    if (f) {
        uint64_t total = 0;
        for (result *p = res; p; p = p->next) total += p->current_load;
        put_u64(f, 0x4a4f494e);             /* 'JOIN' */
        put_u64(f, relR->num_tuples);
        put_u64(f, relS->num_tuples);
        put_u64(f, res == NULL);
        put_u64(f, total);
        fwrite(relR->tuples, sizeof(tuple), relR->num_tuples, f);
        fwrite(relS->tuples, sizeof(tuple), relS->num_tuples, f);
        for (result *p = res; p; p = p->next)
            fwrite(p->buff, sizeof(result_tuple), p->current_load, f);
    }
    return res;
}

result *ref_hook_filter(inter_res *head, filter_pred *fp, relation_map *map, int *query_relations)
{
    result *res = Filter(head, fp, map, query_relations);
    FILE *f = dump_file();
    if (f) {
        const relation_map *rm = &map[query_relations[fp->relation]];
        inter_res *n = head;
        while (n && n->data->table[fp->relation] == NULL) n = n->next;
        uint64_t total = 0;
        for (result *p = res; p; p = p->next) total += p->current_load;
        put_u64(f, 0x46494c54);             /* 'FILT' */
        put_u64(f, (uint64_t)query_relations[fp->relation]);
        put_u64(f, (uint64_t)fp->column);
        put_u64(f, (uint64_t)(int64_t)fp->value);
        put_u64(f, (uint64_t)fp->comperator);
        put_u64(f, rm->num_tuples);
        put_u64(f, n ? n->data->num_tuples : UINT64_MAX);   /* indirection length or none */
        put_u64(f, res == NULL);
        put_u64(f, total);
        if (n) fwrite(n->data->table[fp->relation], 8, n->data->num_tuples, f);
        for (result *p = res; p; p = p->next)
            fwrite(p->buff, 8, p->current_load, f);
    }
    return res;
}
#endif
