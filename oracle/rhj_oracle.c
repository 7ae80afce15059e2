/*
 * rhj_oracle.c — CPU restatement of the reference algorithm (see rhj_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the parity checker and the timed CPU baseline.
 * Written from the reference's behaviour, not from its text; every function
 * names the reference lines it restates.
 */
#include "rhj_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* HashFunction1, rhjoin.c:311-325: keep the n low bits. */
static inline uint64_t low_bits(uint64_t v, int bits)
{
    return v & ((UINT64_C(1) << bits) - 1);
}

/* ---------------------------------------------------------------- partition */

int orc_partition(const orc_tuple *in, uint64_t n, int bits,
                  orc_tuple *out, uint64_t *hist, int64_t *psum)
{
    const uint64_t nb = UINT64_C(1) << bits;
    int64_t *cursor = (int64_t *)malloc(nb * sizeof(int64_t));
    if (!cursor) return -1;

    /* preprocess.c:320-325 — histogram of the low bits (the reference casts the
     * value to int32 first; the low <=31 bits are unaffected). */
    memset(hist, 0, nb * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; ++i)
        hist[low_bits(in[i].value, bits)]++;

    /* preprocess.c:328-340 — running start per non-empty bucket, -1 otherwise */
    int64_t start = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        if (hist[b] > 0) {
            psum[b] = cursor[b] = start;
            start += (int64_t)hist[b];
        } else {
            psum[b] = cursor[b] = -1;
        }
    }

    /* preprocess.c:349-359 — stable scatter in input order */
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t b = low_bits(in[i].value, bits);
        out[cursor[b]++] = in[i];
    }
    free(cursor);
    return 0;
}

/* --------------------------------------------------------------- bucket index */

uint64_t orc_find_next_prime(uint64_t num)
{
    /* rhjoin.c:327-346 as written: even -> +1, then trial division by every
     * i >= 3 while i*i < num (strict), stepping num by 2 on a hit. */
    if ((num & 1) == 0) num++;
    for (;;) {
        int composite = 0;
        for (uint64_t i = 3; i * i < num; ++i) {
            if (num % i == 0) { composite = 1; break; }
        }
        if (!composite) return num;
        num += 2;
    }
}

typedef struct {
    uint64_t  size;    /* H2 modulus (rhjoin.c:348-351)                       */
    int64_t  *head;    /* -1 empty, else position+1 of the LAST tuple hashed   */
    int64_t  *link;    /* 0 end of chain, else position+1 of the next-lower    */
} bucket_index;

/* InitIndex + CreateIndex, rhjoin.c:253-273 and :219-250.  Tuples of the bucket
 * are visited last to first; the first visitor of a slot becomes its head, each
 * later one is appended at the tail, so a chain lists positions in DESCENDING
 * order.  The reference walks to the tail on every append; a tail cursor gives
 * the same chains without the quadratic walk. */
static int index_build(bucket_index *ix, const orc_tuple *tuples, uint64_t count)
{
    ix->size = orc_find_next_prime(count);
    ix->head = (int64_t *)malloc(ix->size * sizeof(int64_t));
    ix->link = (int64_t *)malloc((count ? count : 1) * sizeof(int64_t));
    int64_t *tail = (int64_t *)malloc(ix->size * sizeof(int64_t));
    if (!ix->head || !ix->link || !tail) return -1;
    for (uint64_t s = 0; s < ix->size; ++s) ix->head[s] = -1;

    for (int64_t i = (int64_t)count - 1; i >= 0; --i) {
        const uint64_t slot = tuples[i].value % ix->size;
        if (ix->head[slot] == -1) ix->head[slot] = i + 1;
        else                      ix->link[tail[slot]] = i + 1;
        ix->link[i] = 0;
        tail[slot] = i;
    }
    free(tail);
    return 0;
}

static void index_free(bucket_index *ix)
{
    free(ix->head);
    free(ix->link);
}

typedef struct { orc_pair *data; uint64_t len, cap; } pair_vec;

static int pair_push(pair_vec *v, uint64_t r, uint64_t s)
{
    if (v->len == v->cap) {
        uint64_t cap = v->cap ? v->cap * 2 : 8192;   /* RESULT_MAX_BUFFER / 16 */
        orc_pair *p = (orc_pair *)realloc(v->data, cap * sizeof(orc_pair));
        if (!p) return -1;
        v->data = p; v->cap = cap;
    }
    v->data[v->len].row_idR = r;
    v->data[v->len].row_idS = s;
    v->len++;
    return 0;
}

/* GetResults, rhjoin.c:141-217: stream the un-indexed side in bucket order,
 * compare the slot head, then every chain element; each equal value emits one
 * pair.  flip != 0 means the streamed side is S (r_s == 1, rhjoin.c:172-176). */
static int bucket_probe(const orc_tuple *stream, uint64_t nstream,
                        const orc_tuple *indexed, const bucket_index *ix,
                        int flip, pair_vec *out)
{
    for (uint64_t i = 0; i < nstream; ++i) {
        const uint64_t v = stream[i].value;
        int64_t at = ix->head[v % ix->size];
        while (at > 0) {
            const orc_tuple *q = &indexed[at - 1];
            if (q->value == v) {
                int rc = flip ? pair_push(out, q->row_id, stream[i].row_id)
                              : pair_push(out, stream[i].row_id, q->row_id);
                if (rc) return rc;
            }
            at = ix->link[at - 1];
        }
    }
    return 0;
}

int orc_join(const orc_tuple *R, uint64_t nR, const orc_tuple *S, uint64_t nS,
             int bits, orc_pair **pairs, uint64_t *count)
{
    *pairs = NULL;
    *count = 0;
    if (nR == 0 || nS == 0) return 0;                 /* rhjoin.c:15-16 */

    const uint64_t nb = UINT64_C(1) << bits;
    orc_tuple *pr = (orc_tuple *)malloc(nR * sizeof(orc_tuple));
    orc_tuple *ps = (orc_tuple *)malloc(nS * sizeof(orc_tuple));
    uint64_t *hr = (uint64_t *)malloc(nb * sizeof(uint64_t));
    uint64_t *hs = (uint64_t *)malloc(nb * sizeof(uint64_t));
    int64_t *sr = (int64_t *)malloc(nb * sizeof(int64_t));
    int64_t *ss = (int64_t *)malloc(nb * sizeof(int64_t));
    int rc = (!pr || !ps || !hr || !hs || !sr || !ss) ? -1 : 0;
    pair_vec out = {0, 0, 0};

    if (!rc) rc = orc_partition(R, nR, bits, pr, hr, sr);   /* rhjoin.c:72 */
    if (!rc) rc = orc_partition(S, nS, bits, ps, hs, ss);   /* rhjoin.c:73 */

    for (uint64_t b = 0; !rc && b < nb; ++b) {              /* rhjoin.c:79-105 */
        if (hr[b] == 0 || hs[b] == 0) continue;
        bucket_index ix;
        if (hr[b] >= hs[b]) {                               /* index S, stream R */
            rc = index_build(&ix, ps + ss[b], hs[b]);
            if (!rc) rc = bucket_probe(pr + sr[b], hr[b], ps + ss[b], &ix, 0, &out);
        } else {                                            /* index R, stream S */
            rc = index_build(&ix, pr + sr[b], hr[b]);
            if (!rc) rc = bucket_probe(ps + ss[b], hs[b], pr + sr[b], &ix, 1, &out);
        }
        index_free(&ix);
    }

    free(pr); free(ps); free(hr); free(hs); free(sr); free(ss);
    if (rc) { free(out.data); return rc; }
    if (out.len == 0) { free(out.data); return 0; }
    *pairs = out.data;
    *count = out.len;
    return 0;
}

/* -------------------------------------------------------------------- filter */

uint64_t orc_filter(const uint64_t *col, const uint64_t *sel, uint64_t n,
                    char op, int value, uint64_t *out)
{
    /* filter.c:116 etc. compare a uint64_t with an int: the int converts to
     * uint64_t (sign-extending), then the comparison is unsigned. */
    const uint64_t k = (uint64_t)(int64_t)value;
    uint64_t hits = 0;
    if (op != '<' && op != '>' && op != '=') return UINT64_MAX;  /* filter.c:184 */
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t v = sel ? col[sel[i]] : col[i];           /* filter.c:126 */
        const int keep = op == '>' ? v > k : op == '<' ? v < k : v == k;
        if (keep) out[hits++] = i;
    }
    return hits;
}

/* --------------------------------------------------------- digests, generators */

uint64_t orc_fnv1a64(const void *data, size_t bytes)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = UINT64_C(0xcbf29ce484222325);
    for (size_t i = 0; i < bytes; ++i) {
        h ^= p[i];
        h *= UINT64_C(0x100000001b3);
    }
    return h;
}

uint64_t orc_mix64(uint64_t x)
{
    x ^= x >> 30; x *= UINT64_C(0xbf58476d1ce4e5b9);
    x ^= x >> 27; x *= UINT64_C(0x94d049bb133111eb);
    x ^= x >> 31;
    return x;
}

uint64_t orc_splitmix64_next(uint64_t *state)
{
    *state += UINT64_C(0x9e3779b97f4a7c15);
    return orc_mix64(*state);
}

static double zeta(uint64_t n, double theta)
{
    double s = 0.0;
    for (uint64_t i = 1; i <= n; ++i) s += pow(1.0 / (double)i, theta);
    return s;
}

int orc_generate(orc_tuple *out, uint64_t n, int kind, uint64_t domain,
                 double theta, uint64_t seed)
{
    uint64_t st = seed;
    if (domain == 0) domain = n ? n : 1;
    switch (kind) {
    case 0: {   /* unique: mix64 of a seeded permutation of [0,n) */
        for (uint64_t i = 0; i < n; ++i) out[i].value = i;
        for (uint64_t i = n; i > 1; --i) {
            uint64_t j = orc_splitmix64_next(&st) % i;
            uint64_t t = out[i - 1].value; out[i - 1].value = out[j].value; out[j].value = t;
        }
        for (uint64_t i = 0; i < n; ++i) out[i].value = orc_mix64(out[i].value);
        break;
    }
    case 1:
        for (uint64_t i = 0; i < n; ++i)
            out[i].value = orc_mix64(orc_splitmix64_next(&st) % domain);
        break;
    case 2: {   /* Gray et al. "Quickly generating billion-record synthetic databases" */
        const double zetan = zeta(domain, theta);
        const double alpha = 1.0 / (1.0 - theta);
        const double eta = (1.0 - pow(2.0 / (double)domain, 1.0 - theta)) /
                           (1.0 - zeta(2, theta) / zetan);
        for (uint64_t i = 0; i < n; ++i) {
            const double u = (double)(orc_splitmix64_next(&st) >> 11) * (1.0 / 9007199254740992.0);
            const double uz = u * zetan;
            uint64_t rank;
            if (uz < 1.0) rank = 0;
            else if (uz < 1.0 + pow(0.5, theta)) rank = 1;
            else rank = (uint64_t)((double)domain * pow(eta * u - eta + 1.0, alpha));
            if (rank >= domain) rank = domain - 1;
            out[i].value = orc_mix64(rank);
        }
        break;
    }
    case 3:
        for (uint64_t i = 0; i < n; ++i) out[i].value = i + 1;
        break;
    case 4:
        for (uint64_t i = 0; i < n; ++i)
            out[i].value = orc_splitmix64_next(&st) % domain;
        break;
    default:
        return -1;
    }
    for (uint64_t i = 0; i < n; ++i) out[i].row_id = i;
    return 0;
}

void orc_free(void *p) { free(p); }
