/*
 * rhj_abi.c — the reference's own entry points, in C, on top of the HIP path.
 *
 *   RadixHashJoin()   rhjoin.h:9   / rhjoin.c:13-111
 *   Filter()          filter.h:9   / filter.c:92-190
 *   result-list API   results.h:7-25 / results.c
 *   FreeRelation()    preprocess.h:18 / preprocess.c:213-218
 *   SchedulerInit/Destroy  scheduler.h:9,50 (inert: the GPU owns the parallelism)
 *
 * Ownership follows the reference: the caller keeps its relations; the returned
 * list is plain malloc memory released by FreeResult() = free(buff); free(node)
 * (results.c:144-153).  A list never contains an empty node before a non-empty one
 * (consumers step one node per index: inter_res.c:51-55, filter.c:31-35).
 */
#include "rhj.h"
#include "rhj_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <malloc.h>

_Static_assert(sizeof(rhj_tuple) == 16, "tuple layout (structs.h:15-19)");
_Static_assert(sizeof(rhj_relation) == 16, "relation layout (structs.h:25-29)");
_Static_assert(sizeof(rhj_result) == 24, "result layout (structs.h:37-43)");
_Static_assert(sizeof(rhj_result_tuple) == 16, "result_tuple layout (structs.h:46-50)");
_Static_assert(sizeof(rhj_filter_pred) == 16, "filter_pred layout (structs.h:141-147)");
_Static_assert(sizeof(rhj_relation_map) == 32, "relation_map layout (structs.h:132-138)");
_Static_assert(sizeof(rhj_inter_res) == 24, "inter_res layout (structs.h:106-111)");

#define RESULT_MAX_BUFFER   131072   /* structs.h:9  */
#define RESULT_FINAL_BUFFER 1048576  /* structs.h:10 */

/* ---- list assembly from device chunks ----------------------------------- */

typedef struct {
    rhj_result *head, *tail;
    size_t      elem;      /* bytes per element */
    int         failed;
} list_sink;

/* Node buffers of the merged size (RESULT_FINAL_BUFFER, rhjoin.c:371: 65535 pairs or 131072 ids) are recycled:
 * FreeResult() parks them here instead of handing them back to the allocator, and the next result list takes them
 * — pages that are already mapped, where a fresh 1 MiB malloc costs 256 page faults to fill (the D2H into fresh
 * nodes measured 5.8 GB/s).  They stay plain malloc memory: a caller that frees a list with its own
 * free(buff); free(node) (results.c:144-153) is as correct as before.  Bounded; rhj_release() empties it. */
#define POOL_MAX 1024                                 /* 1 GiB of parked buffers at most */
static char *pool_buf[POOL_MAX];
static int   pool_n = 0;

static char *pool_take(size_t bytes)
{
    char *p = NULL;
    if (bytes <= RESULT_FINAL_BUFFER && bytes > RESULT_FINAL_BUFFER / 2) {
        rhj_api_lock();
        if (pool_n > 0) p = pool_buf[--pool_n];
        rhj_api_unlock();
        if (!p) p = (char *)malloc(RESULT_FINAL_BUFFER);
        return p;
    }
    return (char *)malloc(bytes ? bytes : 1);
}

static int pool_put(char *buff)
{
    int kept = 0;
    const size_t usable = buff ? malloc_usable_size(buff) : 0;      /* any malloc block that can hold a merged node */
    if (usable >= RESULT_FINAL_BUFFER && usable <= 2 * (size_t)RESULT_FINAL_BUFFER) {
        rhj_api_lock();
        if (pool_n < POOL_MAX) { pool_buf[pool_n++] = buff; kept = 1; }
        rhj_api_unlock();
    }
    return kept;
}

void rhj_host_pool_release(void)
{
    rhj_api_lock();
    while (pool_n > 0) free(pool_buf[--pool_n]);
    rhj_api_unlock();
}

static void *sink_chunk(void *ctx, uint64_t elems)
{
    list_sink *s = (list_sink *)ctx;
    rhj_result *node = (rhj_result *)malloc(sizeof(rhj_result));
    if (!node) { s->failed = 1; return NULL; }
    node->buff = pool_take(elems * s->elem);
    if (!node->buff) { free(node); s->failed = 1; return NULL; }
    node->next = NULL;
    node->current_load = elems;
    if (s->tail) s->tail->next = node; else s->head = node;
    s->tail = node;
    return node->buff;
}

static void fatal(const char *what)
{
    /* the reference has no error channel on this path: fatal paths print and
     * exit(2) (rhjoin.c:285-286, filter.c:185-186) */
    fprintf(stderr, "rhj: %s failed on the device path; there is no CPU fallback\n", what);
    exit(2);
}

rhj_result *RadixHashJoin(rhj_relation *relR, rhj_relation *relS, struct scheduler *sched)
{
    (void)sched;                                     /* may be NULL (handler.c:60-63) */
    if (relR->num_tuples == 0 || relS->num_tuples == 0) return NULL;   /* rhjoin.c:15-16 */
    /* relations materialised by this library's GetRelation() live on the device (rhj_inter.h):
     * the match list then stays there too */
    if (rhj_resident_relation(relR) || rhj_resident_relation(relS)) {
        if (!(rhj_resident_relation(relR) && rhj_resident_relation(relS))) fatal("RadixHashJoin (one device-resident, one host relation)");
        return rhj_resident_join(relR, relS);
    }
    list_sink sink = {NULL, NULL, sizeof(rhj_result_tuple), 0};
    uint64_t matches = 0;
    int rc = rhj_host_join(relR->tuples, relR->num_tuples, relS->tuples, relS->num_tuples, &matches,
                           sink_chunk, &sink, rhj_host_node_pairs());
    if (rc < 0 || sink.failed) { FreeResult(sink.head); fatal("RadixHashJoin"); }
    if (matches == 0) {
        if (rhj_host_null_on_empty()) return NULL;   /* THREADS==1 behaviour              */
        /* as shipped (THREADS 4) MergeResults always returns a head, rhjoin.c:356-359 */
        rhj_result *head = (rhj_result *)malloc(sizeof(rhj_result));
        head->buff = (char *)malloc(RESULT_FINAL_BUFFER);
        head->next = NULL;
        head->current_load = 0;
        return head;
    }
    return sink.head;
}

rhj_result *Filter(rhj_inter_res *head, rhj_filter_pred *filter_p, rhj_relation_map *map, int *query_relations)
{
    if (rhj_resident_inter(head)) return rhj_resident_filter(head, filter_p, map, query_relations);   /* rhj_inter.h */
    const uint64_t relation = (uint64_t)filter_p->relation, column = (uint64_t)filter_p->column;
    const rhj_relation_map *rm = &map[query_relations[relation]];
    const uint64_t *col = rm->columns[column];                         /* filter.c:96 */
    while (head && head->data->table[relation] == NULL) head = head->next;   /* filter.c:98-104 */

    const char op = filter_p->comperator;
    if (op != '<' && op != '>' && op != '=') {                         /* filter.c:184-186 */
        printf("Wrong comperator in filter function\n");
        exit(2);
    }
    const uint64_t *sel = head ? head->data->table[relation] : NULL;
    const uint64_t n = head ? head->data->num_tuples : rm->num_tuples;
    const uint64_t value = (uint64_t)(int64_t)filter_p->value;         /* int -> u64, filter.c:116 */

    list_sink sink = {NULL, NULL, sizeof(uint64_t), 0};
    uint64_t hits = 0;
    int rc = rhj_host_filter(col, rm->num_tuples, sel, n, op, value, &hits, sink_chunk, &sink,
                             RESULT_FINAL_BUFFER / sizeof(uint64_t));
    if (rc < 0 || sink.failed) { FreeResult(sink.head); fatal("Filter"); }
    return sink.head;                                                  /* NULL on zero hits, filter.c:94,189 */
}

/* ---- result-list API, behaviour of results.c ----------------------------- */

static rhj_result *append_elem(rhj_result **head, const void *elem, size_t elem_bytes, size_t node_bytes)
{
    if (*head == NULL) {
        rhj_result *n = (rhj_result *)malloc(sizeof(rhj_result));
        n->buff = (char *)malloc(node_bytes);
        n->current_load = 1;
        n->next = NULL;
        memcpy(n->buff, elem, elem_bytes);
        *head = n;
        return n;
    }
    rhj_result *t = *head;
    while (t->current_load * elem_bytes + elem_bytes > node_bytes) {
        if (t->next) { t = t->next; continue; }
        rhj_result *n = (rhj_result *)malloc(sizeof(rhj_result));
        n->buff = (char *)malloc(node_bytes);
        n->current_load = 1;
        n->next = NULL;
        memcpy(n->buff, elem, elem_bytes);
        t->next = n;
        return t;                /* results.c:30-36 returns the OLD node; callers re-walk one hop */
    }
    memcpy(t->buff + t->current_load * elem_bytes, elem, elem_bytes);
    t->current_load++;
    return t;
}

rhj_result *InsertResult(rhj_result **head, rhj_result_tuple *res_tuple)
{
    return append_elem(head, res_tuple, sizeof(rhj_result_tuple), RESULT_MAX_BUFFER);
}

rhj_result *InsertRowIdResult(rhj_result **head, uint64_t *row_id)
{
    return append_elem(head, row_id, sizeof(uint64_t), RESULT_FINAL_BUFFER);
}

int GetResultNum(rhj_result *res)
{
    uint64_t n = 0;
    for (; res; res = res->next) n += res->current_load;
    return (int)n;               /* results.c:65 returns int */
}

uint64_t FindResultRowId(rhj_result *res, int num)
{
    uint64_t count = 0;
    for (; res; res = res->next) {
        if (res->current_load + count > (uint64_t)num) {
            if (rhj_resident_result(res)) {
                uint64_t v = 0;
                if (rhj_resident_fetch(res, sizeof(uint64_t), (uint64_t)num - count, &v)) fatal("FindResultRowId");
                return v;
            }
            return ((uint64_t *)res->buff)[(uint64_t)num - count];
        }
        count += res->current_load;
    }
    return 0;
}

rhj_result_tuple *FindResultTuples(rhj_result *head, int num)
{
    uint64_t count = 0;
    if (num < 0) return NULL;
    for (; head; head = head->next) {
        if (head->current_load + count > (uint64_t)num) {
            if (rhj_resident_result(head)) {
                /* a copy of the element (the list itself is on the device), kept in a slot of ITS list: valid until the next
                 * call on the same list or its FreeResult — elements of two live results do not alias (results.c:126-142
                 * returns a pointer into the node) */
                rhj_result_tuple *copy = rhj_resident_slot(head);
                if (!copy || rhj_resident_fetch(head, sizeof(rhj_result_tuple), (uint64_t)num - count, copy)) fatal("FindResultTuples");
                return copy;
            }
            return (rhj_result_tuple *)head->buff + ((uint64_t)num - count);
        }
        count += head->current_load;
    }
    return NULL;
}

void FreeResult(rhj_result *head)
{
    if (rhj_resident_result(head)) { rhj_resident_free_result(head); return; }
    while (head) {
        rhj_result *t = head;
        head = head->next;
        if (!pool_put(t->buff)) free(t->buff);
        free(t);
    }
}

void PrintResult(rhj_result *head)
{
    uint64_t total = 0;
    if (rhj_resident_result(head)) {
        fprintf(stderr, "device-resident result list: %lu elements\n", (unsigned long)head->current_load);
        return;
    }
    fprintf(stderr, "------------------------------\nPrinting results:\n");
    for (; head; head = head->next) {
        const rhj_result_tuple *p = (const rhj_result_tuple *)head->buff;
        for (uint64_t i = 0; i < head->current_load; ++i, ++total)
            fprintf(stderr, "row_id R %lu value S %lu || \n", (unsigned long)p[i].row_idR, (unsigned long)p[i].row_idS);
    }
    fprintf(stderr, "Finished printing results!\nNumber of rows in the result: %lu\n-------------------------------\n",
            (unsigned long)total);
}

void FreeRelation(rhj_relation *rel)
{
    if (rel == NULL) return;
    if (rhj_resident_relation(rel)) { rhj_resident_free_relation(rel); return; }
    free(rel->tuples);
    free(rel);
}

int SchedulerInit(struct scheduler **sched, int num_of_threads)
{
    (void)num_of_threads;
    *sched = (struct scheduler *)calloc(1, 64);      /* inert token; never dereferenced here */
    return *sched ? 0 : 1;
}

int SchedulerDestroy(struct scheduler *sched)
{
    free(sched);
    return 0;
}
