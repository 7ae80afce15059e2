/* rhj_internal.h — host-staging helpers shared by rhj_device.hip (definitions) and
 * rhj_abi.c (callers).  Not part of the public C-ABI. */
#ifndef RHJ_INTERNAL_H
#define RHJ_INTERNAL_H
#include "rhj.h"
#include "rhj_inter.h"
#ifdef __cplusplus
extern "C" {
#endif
int rhj_host_join(const rhj_tuple *R, uint64_t nR, const rhj_tuple *S, uint64_t nS, uint64_t *matches,
                  void *(*alloc_chunk)(void *ctx, uint64_t pairs), void *ctx, uint64_t node_pairs);
int rhj_host_filter(const uint64_t *col, uint64_t col_rows, const uint64_t *sel, uint64_t n, char op,
                    uint64_t value, uint64_t *hits, void *(*alloc_chunk)(void *ctx, uint64_t ids), void *ctx,
                    uint64_t node_ids);
void rhj_host_pool_release(void);                  /* rhj_abi.c: parked result-node buffers */
int rhj_host_null_on_empty(void);
uint64_t rhj_host_node_pairs(void);

/* device-side services (rhj_device.hip) used by rhj_inter.hip */
void *rhj_dev_alloc(size_t bytes);                 /* stream-ordered, on the library's stream */
void  rhj_dev_free(void *p);
void *rhj_dev_stream(void);                        /* hipStream_t */
/* device copy of a host column: the REGISTERED copy (*temp = NULL), or a block uploaded for this call that the
 * caller hands back with rhj_dev_free(*temp) once the kernels reading it are queued */
const uint64_t *rhj_dev_column(const uint64_t *host_col, uint64_t rows, void **temp);
int   rhj_dev_reserve(uint64_t rows);
int   rhj_dev_register_column(const uint64_t *host_col, uint64_t rows, const void *pin_base, uint64_t pin_bytes);
void  rhj_dev_unregister_column(const uint64_t *host_col);
int   rhj_dev_join(const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS, rhj_result_tuple **out,
                   uint64_t *matches);             /* *out is library-owned and valid until the next join */

/* One library context per process: every entry point runs under this (recursive) lock, so calls from several
 * host threads are serialised instead of racing on the workspace, the registries and the stream. */
void rhj_api_lock(void);
void rhj_api_unlock(void);
/* rhj_host.cpp: elements that arrive through a ring of staging blocks -> result nodes, by several host threads */
int rhj_move_blocks(uint64_t total, uint64_t elem, uint64_t node_elems, char *const *nodes, uint64_t blk, int ring,
                    char *const *staging, unsigned threads, int (*issue)(void *ctx, uint64_t b),
                    int (*wait)(void *ctx, uint64_t b), void *ctx);
int rhj_move_blocks_at(uint64_t base, uint64_t total, uint64_t elem, uint64_t node_elems, char *const *nodes, uint64_t blk, int ring,
                       char *const *staging, unsigned threads, int (*issue)(void *ctx, uint64_t b),
                       int (*wait)(void *ctx, uint64_t b), void *ctx);

/* device-resident side of the reference's entry points (rhj_inter.hip), called by rhj_abi.c */
rhj_result *rhj_resident_join(rhj_relation *relR, rhj_relation *relS);
rhj_result *rhj_resident_filter(rhj_inter_res *head, rhj_filter_pred *filter_p, rhj_relation_map *map, int *query_relations);
void rhj_resident_free_result(rhj_result *res);
void rhj_resident_free_relation(rhj_relation *rel);
int  rhj_resident_fetch(const rhj_result *res, uint64_t elem_bytes, uint64_t index, void *dst);
rhj_result_tuple *rhj_resident_slot(const rhj_result *res);   /* per-list host copy of the element FindResultTuples returns */
#ifdef __cplusplus
}
struct RhjApiLock {
    RhjApiLock() { rhj_api_lock(); }
    ~RhjApiLock() { rhj_api_unlock(); }
    RhjApiLock(const RhjApiLock &) = delete;
    RhjApiLock &operator=(const RhjApiLock &) = delete;
};
#endif
#endif
