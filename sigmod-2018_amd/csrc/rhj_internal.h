/* rhj_internal.h — host-staging helpers shared by rhj_device.hip (definitions) and
 * rhj_abi.c (callers).  Not part of the public C-ABI. */
#ifndef RHJ_INTERNAL_H
#define RHJ_INTERNAL_H
#include "rhj.h"
#ifdef __cplusplus
extern "C" {
#endif
int rhj_host_join(const rhj_tuple *R, uint64_t nR, const rhj_tuple *S, uint64_t nS, uint64_t *matches,
                  void *(*alloc_chunk)(void *ctx, uint64_t pairs), void *ctx, uint64_t node_pairs);
int rhj_host_filter(const uint64_t *col, uint64_t col_rows, const uint64_t *sel, uint64_t n, char op,
                    uint64_t value, uint64_t *hits, void *(*alloc_chunk)(void *ctx, uint64_t ids), void *ctx,
                    uint64_t node_ids);
int rhj_host_null_on_empty(void);
uint64_t rhj_host_node_pairs(void);
#ifdef __cplusplus
}
#endif
#endif
