/*
 * rhj_inter.h — device-resident intermediate results: the "next" rows of SURVEY.md §8(f).
 *
 * The reference keeps a query's intermediate result (`inter_res`: per active relation one
 * array of row ids, structs.h:97-111) on the host and rebuilds it after every operator with
 * gathers (inter_res.c, the helper half of filter.c): 71 % of the `small` workload's run
 * time.  Around the hot path that means H2D of both join inputs and D2H of every match
 * list per join.  This header is the reference's own inter_res.h / filter.h:15 interface —
 * same names, same signatures, same struct layouts — implemented so that the row-id tables,
 * the materialised join inputs and the result lists never leave the GPU:
 *
 *   inter_data.table[rel]   device pointer (u64[num_tuples]) or NULL       structs.h:97-101
 *   relation.tuples         device pointer, for relations from GetRelation structs.h:25-29
 *   result.buff             device pointer, ONE node holding all elements  structs.h:37-43
 *
 * The structs themselves, `num_tuples`, `current_load`, `next` and the NULL-ness of
 * `table[rel]` stay host-readable, which is all the reference's query.c looks at
 * (query.c:334-465).  A maintainer switches the engine to this mode by NOT compiling
 * inter_res.c and filter.c (and, for device-side loading and statistics, relation_map.c)
 * and linking librhj.so instead; with the
 * reference's own inter_res.c in the link these definitions are simply never called and
 * RadixHashJoin()/Filter() keep their host-memory behaviour.  RadixHashJoin(), Filter(),
 * FreeRelation(), FreeResult() and GetResultNum() recognise device-resident arguments by
 * identity (objects created here are registered), not by a global switch.
 *
 * Semantics follow inter_res.c line by line (cited per function), with two documented
 * departures where the reference's code is broken: SelfJoin uses the
 * mapped relation's tuple count and pushes the inter_res position (inter_res.c:252,259 use
 * the wrong index); InsertSingleRowIdsToInterResult appends a node to the LAST node instead
 * of dereferencing the NULL it just walked to (filter.c:83-88).
 */
#ifndef RHJ_INTER_H
#define RHJ_INTER_H

#include "rhj.h"

#ifdef __cplusplus
extern "C" {
#endif

/* structs.h:178-203: what CalculateQueryResults / PrintNullResults read of a query */
typedef struct rhj_query_string_array {
    char **data;
    int    num_of_elements;
} rhj_query_string_array;

typedef struct rhj_batch_listnode {
    int                         num_of_relations;
    int                        *relations;
    void                       *predicate_list;      /* predicates_listnode*, not used here */
    rhj_query_string_array     *views;
    struct rhj_batch_listnode  *next;
} rhj_batch_listnode;

/* inter_res.h:5-17 */
int  InitInterData(rhj_inter_data **head, int num_of_relations, int num_tuples);     /* inter_res.c:10-15  */
void FreeInterData(rhj_inter_data *head, int num_of_relations);                      /* inter_res.c:17-24  */
int  InitInterResults(rhj_inter_res **head, int num_of_relations);                   /* inter_res.c:26-32  */
void PrintInterResults(rhj_inter_res *head);                                         /* inter_res.c:154-173 */
void FreeInterResults(rhj_inter_res *var);                                           /* inter_res.c:175-180 */

/* inter_res.h:26: rebuild the node after a join; res pairs index the node's rows on the
 * side that was already active (inter_res.c:34-152) */
int  InsertJoinToInterResults(rhj_inter_res *head, int ex_rel_num, int new_rel_num, rhj_result *res);

/* inter_res.h:33,41: materialise {value = col[table[rel][i]] or col[i], row_id = i}
 * (inter_res.c:182-231) */
rhj_relation *GetRelation(int given_rel, int column, rhj_inter_res *inter, rhj_relation_map *map, int *query_relations);
rhj_relation *ScanInterResults(int given_rel, int column, rhj_inter_res *inter, rhj_relation_map *map, int *query_relations);

/* inter_res.h:44 (inter_res.c:234-263, intended semantics, see above) */
rhj_result *SelfJoin(int given_rel, int column1, int column2, rhj_inter_res **inter, rhj_relation_map *map,
                     int *query_relations);

/* inter_res.h:48,51 (inter_res.c:265-318) */
void MergeInterNodes(rhj_inter_res **inter);
void Merge(rhj_inter_res **head, rhj_inter_res **node, int rel_num);

/* inter_res.h:55,58: wrap-around u64 sums of the views, printed as the reference prints them
 * (inter_res.c:320-350) */
void CalculateQueryResults(rhj_inter_res *inter, rhj_relation_map *map, rhj_batch_listnode *query);
void PrintNullResults(rhj_batch_listnode *query);

/* inter_res.h:61,64,68 (inter_res.c:352-428) */
int  AreActiveInInter(rhj_inter_res *inter, int rel1, int rel2);
int  JoinInterNode(rhj_inter_res **inter, rhj_relation_map *rel_map, int relation1, int column1, int relation2,
                   int column2, int *relations);
void CartesianInterResults(rhj_inter_res **inter);

/* filter.h:15 (filter.c:11-89) */
int  InsertSingleRowIdsToInterResult(rhj_inter_res **head, int relation_num, rhj_result *res);

/* relation_map.h:10-16: map the relation files (header u64 tuples, u64 columns,
 * then column-major u64 data, relation_map.c:39-50), copy every column to the device once — the
 * copies Filter()/GetRelation()/CalculateQueryResults() use — and compute the optimiser's column
 * statistics there: l = min, u = max, f = tuples, d = distinct values counted with the reference's
 * flag array including its cap (range above 50 000 000 folds modulo 5 000 000, relation_map.c:66-84). */
typedef struct rhj_relation_listnode {       /* structs.h:86-91 */
    char                         *filename;
    int                           fd;
    struct rhj_relation_listnode *next;
} rhj_relation_listnode;
int  InitRelationMap(rhj_relation_listnode *head, rhj_relation_map *rel_map);   /* relation_map.c:13-88  */
void FreeRelationMap(rhj_relation_map *rel_map, int map_size);                   /* relation_map.c:90-98  */
void PrintRelationMap(rhj_relation_map *rel_map, int map_size);                  /* relation_map.c:100-115 */

/* ---- device entry points behind them (tests and bench call these directly) ---------- */

/* dst[t][i] = src[t][idx[i * idx_stride]] for t < ntab (idx_stride 2 walks one side of a pair
 * list); src[t] == NULL writes the index itself.  All pointers are device pointers. */
int rhj_gather_tables_device(uint64_t *const *dst, const uint64_t *const *src, int ntab, const uint64_t *idx,
                             int idx_stride, uint64_t n);
/* tuples[i] = {col[sel ? sel[i] : i], i} */
int rhj_build_relation_device(const uint64_t *d_col, const uint64_t *d_sel, uint64_t n, rhj_tuple *d_tuples);
/* wrap-around sum of col[sel[i]] (sel may be NULL) */
int rhj_sum_gather_device(const uint64_t *d_col, const uint64_t *d_sel, uint64_t n, uint64_t *sum);
/* up to 8 such sums in one launch (CalculateQueryResults: all views of a query) */
int rhj_sum_views_device(int views, const uint64_t *const *d_cols, const uint64_t *const *d_sels, const uint64_t *ns, uint64_t *sums);
/* ascending i with colA[selA ? selA[i] : i] == colB[selB ? selB[i] : i] */
int rhj_filter_eq2_device(const uint64_t *d_colA, const uint64_t *d_selA, const uint64_t *d_colB, const uint64_t *d_selB,
                          uint64_t n, uint64_t *d_out, uint64_t *hits);
/* min, max and the reference's distinct-value estimate of a device column (n >= 1) */
int rhj_column_stats_device(const uint64_t *d_col, uint64_t n, uint64_t *l, uint64_t *u, double *d);

/* 1 when the object was created by this library's device-resident side */
int rhj_resident_relation(const rhj_relation *rel);
int rhj_resident_result(const rhj_result *res);
int rhj_resident_inter(const rhj_inter_res *head);

#ifdef __cplusplus
}
#endif
#endif /* RHJ_INTER_H */
