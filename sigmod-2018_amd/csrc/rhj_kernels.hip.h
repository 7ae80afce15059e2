// rhj_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the radix hash
// join and the filter scan.  Integer / indexing work only: the roofline is HBM.
//
// Reference loops these kernels replace (file:line in VagelisN/Sigmod-2018):
//   k_hist_tiles     HistJob                      preprocess.c:181-195
//   k_scan_*         hist merge + psum            preprocess.c:83-102 / :328-340
//   k_scatter_lds    SerialReorderArray scatter   preprocess.c:349-359 (stable)
//   k_plan           bucket loop / side choice    rhjoin.c:79-102 (>= picks the probe side)
//   k_build_lds / k_build_hbm   InitIndex + CreateIndex   rhjoin.c:253-273, :219-250
//   k_probe          GetResults                   rhjoin.c:141-217
//   k_filter_*       Filter                       filter.c:110-183
//
// Partition.  A pass handles at most 8 digit bits: a workgroup ranks a 4096-tuple
// tile stably (wave match-any ballots + per-wave LDS counters), stages it in LDS in
// digit order and writes every digit run as full 16-byte-per-lane coalesced stores
// (a direct 4096-way scatter of 16-byte tuples measured 2.6x write amplification and
// 834 GB/s on MI355X: profiles/r01a).  Wider radixes (9..12 bits) run two stable LSD
// passes (low half, then high half), which yields the same stable order as one pass.
//
// Hash index.  The reference chains bucket positions in DESCENDING order behind a
// prime-modulus slot (CreateIndex walks last->first and appends at the tail), which
// is what fixes the order of duplicate matches.  Here each bucket's index is an
// ORDERED linear-probing table (Amble & Knuth): an entry is (tag | position+1),
// inserted with atomic max so that along every probe run entries are in descending
// (tag, position) order.  The final table is the same for every insertion
// interleaving (deterministic), a walk from a key's home slot meets that key's
// duplicates in descending position — the reference's chain order — and can stop at
// the first entry whose tag is smaller.  Tags only pre-filter: every candidate is
// verified against the build tuple's full 64-bit key, so results are exact.
// Tables live in HBM (L2-resident while their bucket is being probed):
//   * 32-bit entries (16-bit tag, 16-bit position), built in LDS by one workgroup per
//     bucket and dumped, when the bucket's build side has <= lds_cap tuples;
//   * 64-bit entries (32-bit tag, 32-bit position), built with global atomics, else.
// Probe units are tiles of 2048 probe tuples in canonical order; any number of
// workgroups share a bucket's table, so a hot bucket costs no extra build work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rhj.h"
#include "rhj_common.hip.h"
#include "rhj_partition.hip.h"
#include "rhj_join_tiled.hip.h"
#include "rhj_join_fused.hip.h"
#include "rhj_small.hip.h"
#include "rhj_filter.hip.h"
#include "rhj_diag.hip.h"
