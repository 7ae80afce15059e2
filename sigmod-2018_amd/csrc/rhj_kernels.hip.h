// rhj_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the radix hash
// join and the filter scan.  Integer / indexing work only: the roofline is HBM.
//
// Reference loops these kernels replace (file:line in VagelisN/Sigmod-2018):
//   k_hist_tiles     HistJob                      preprocess.c:181-195
//   k_scan_*         hist merge + psum            preprocess.c:83-102 / :328-340
//   k_scatter_lds    SerialReorderArray scatter   preprocess.c:349-359 (stable)
//   k_plan           bucket loop / side choice    rhjoin.c:79-102 (>= picks the probe side)
//   k_local_part / k_scatter_runs   the same stable scatter in two LSD passes (radix bits 9..15)
//   k_group_scan / k_bucket_psum    hist merge + psum for the two passes   preprocess.c:83-102 / :328-340
//   fj_build / k_build_lds / k_build_hbm   InitIndex + CreateIndex   rhjoin.c:253-273, :219-250
//   k_join_fused / k_join_walk / k_probe   GetResults + MergeResults   rhjoin.c:141-217, :354-392
//   k_filter_*       Filter                       filter.c:110-183
//
// The device code by path (one header each; the design notes sit at the top of every header and in DESIGN.md §4):
//   rhj_partition.hip.h    stable radix partition.  Radix bits <= 8: one pass (k_hist_tiles, k_scan_*, k_scatter_lds: a
//                          4096-tuple tile ranked by wave match-any ballots + per-wave LDS counters, staged in LDS in digit
//                          order, written run by run).  Bits 9..15: two LSD passes in RUN FORM — k_local_part partitions every
//                          tile in place on the low half of the bits (no histogram, no offsets; 12-byte tuples when the row ids
//                          fit 32 bits) and, up to 12 bits, counts pass 2's digits on the way (strips of 4 tiles, cells in LDS;
//                          13..15 bits: k_hist_runs from one byte per tuple); k_group_scan / k_bucket_psum turn the counts into
//                          pass 2's offsets and the bucket histogram; k_scatter_runs moves the runs to their final places
//                          (software-pipelined, persistent, XCD-aware tile order).  Five launches, no memset.
//   rhj_small.hip.h        joins of up to 4 M tuples per side on <= 8 bits: the partition in two launches, the plan riding along.
//   rhj_join_fused.hip.h   the default join: one persistent workgroup per CU takes (bucket, <= 65536 probe tuples) units by
//                          ticket; per unit a CSR slot index of the build side in LDS ((tag16, position) entries sorted
//                          descending per slot = the reference's chain order), the probe side streamed once with one verifying
//                          gather per candidate — or none when the build tuples fit LDS too —, match counts chained through a
//                          decoupled look-back, pairs emitted by a deferred streaming pass; k_join_walk for the rare units
//                          that pass cannot describe.
//   rhj_join_tiled.hip.h   buckets beyond the LDS index: ordered linear-probing tag tables in HBM (Amble-Knuth order by
//                          (tag, position) via atomic max: deterministic, duplicates meet in descending position), count and
//                          emit passes over 1024-tuple probe tiles.
//   rhj_lowradix.hip.h     joins on few radix bits (the reference ships 4) over inputs too big for them: run on r + k bits, emitted
//                          in the canonical order of r bits by replaying pass 2 of the partition over the probe side.
//   rhj_filter.hip.h       predicate -> ballot masks -> ascending index list.
// Tags only pre-filter everywhere: every candidate is verified against the build tuple's full 64-bit key, so results are exact
// for any hash and any tag collision.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rhj.h"
#include "rhj_common.hip.h"
#include "rhj_partition.hip.h"
#include "rhj_join_tiled.hip.h"
#include "rhj_join_fused.hip.h"
#include "rhj_join_exact.hip.h"
#include "rhj_lowradix.hip.h"
#include "rhj_small.hip.h"
#include "rhj_filter.hip.h"
#include "rhj_diag.hip.h"
