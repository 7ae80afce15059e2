/*
 * rhj_oracle.h — CPU restatement of the reference's radix hash join, radix
 * partition and filter scan.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: tests/, the smoke()
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py may call it,
 * as the checker or the timed CPU baseline.  Nothing under sigmod-2018_amd/
 * includes, links or executes it, and the product path has no CPU fallback.
 *
 * Parity pinning: oracle/Makefile compiles the reference's own rhjoin.c /
 * preprocess.c / results.c / filter.c (from /root/reference, THREADS 1 and 4,
 * N_LSB 4/8/12) into oracle/_ref/, oracle/gen_golden.py runs both on the same
 * seeded inputs and commits the digests under tests/golden/; tests/test_oracle.py
 * re-checks this restatement against those fixtures and against the reference's
 * own golden file (small.result) where the path reaches it.
 */
#ifndef RHJ_ORACLE_H
#define RHJ_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_tuple { uint64_t value, row_id; } orc_tuple;   /* structs.h:15-19 */
typedef struct orc_pair  { uint64_t row_idR, row_idS; } orc_pair; /* structs.h:46-50 */

/* SerialReorderArray, preprocess.c:302-362.  out[n]; hist[1<<bits];
 * psum[1<<bits] (-1 where the bucket is empty).  Returns 0. */
int orc_partition(const orc_tuple *in, uint64_t n, int bits,
                  orc_tuple *out, uint64_t *hist, int64_t *psum);

/* RadixHashJoin, THREADS==1 branch, rhjoin.c:69-110 with CreateIndex :219-250
 * and GetResults :141-217.  *pairs is malloc'd (caller frees) and holds
 * *count pairs in the reference's emission order.  Returns 0; *pairs == NULL
 * when there is no match or an input is empty. */
int orc_join(const orc_tuple *R, uint64_t nR, const orc_tuple *S, uint64_t nS,
             int bits, orc_pair **pairs, uint64_t *count);

/* Filter, filter.c:92-190.  sel == NULL: direct scan of col[0..n); else
 * col[sel[i]].  out (capacity n) gets the ascending indices; returns the hit
 * count, or UINT64_MAX for an unknown comparator (the reference exit(2)s). */
uint64_t orc_filter(const uint64_t *col, const uint64_t *sel, uint64_t n,
                    char op, int value, uint64_t *out);

/* FindNextPrime as written (rhjoin.c:327-346), exported so a test can pin the
 * quirk (9, 25, 49 ... pass as prime because the trial loop stops at i*i < n). */
uint64_t orc_find_next_prime(uint64_t num);

/* FNV-1a 64 over a byte range: digest of ordered pair streams in fixtures. */
uint64_t orc_fnv1a64(const void *data, size_t bytes);

/* splitmix64 finaliser (a bijection on u64) and generator step, used by the
 * synthetic workloads of SURVEY.md §8(d) so that C, Python and the reference
 * driver generate identical inputs from a seed. */
uint64_t orc_mix64(uint64_t x);
uint64_t orc_splitmix64_next(uint64_t *state);

/* Synthetic relations (row_id[i] = i).
 *   kind 0: unique keys  mix64(perm(j)),  perm = Fisher–Yates of [0,n) by seed
 *   kind 1: FK keys      mix64(u), u uniform on [0,domain)
 *   kind 2: Zipf keys    mix64(z), z ~ Zipf(theta) over ranks [0,domain)
 *   kind 3: dense keys   j+1 (contest-like)
 *   kind 4: small-domain keys  u uniform on [0,domain)   (heavy duplicates)
 */
int orc_generate(orc_tuple *out, uint64_t n, int kind, uint64_t domain,
                 double theta, uint64_t seed);

void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
