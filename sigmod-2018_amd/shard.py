"""Multi-GPU sharding of the join path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Two ways the path shards, both without any collective on the data path of
a single join:

* across joins — independent joins of a plan (best_tree.c starts a new inter_res node for
  predicates that share no relation, inter_res.c:147-150) are dealt to ranks, largest first
  (`assign_joins`, `run_independent_joins`); each rank runs whole joins on its own GPU;
* inside one join — bucket b of R only ever meets bucket b of S (rhjoin.c:42-57), so ranks
  take contiguous BUCKET RANGES balanced by histR+histS (`bucket_ranges`), select their
  tuples with a stable device-side compaction, join them through `rhj_join_device`
  (`sharded_join`), and the concatenation of the per-rank pair lists in rank order IS the
  canonical order.

The one exchange step is the match-list all-gather-v, issued only when the consumer of a
join's result lives on another GPU (`allgatherv_pairs`): an 8-byte count all-gather, then
every rank sends its pairs ONCE to each peer and receives each peer's pairs at their exact
offset of the result (grouped isend/irecv: RCCL has no native allgatherv; every GPU pair of
a node has a direct xGMI link, so the direct exchange moves (world-1)/world of the result
per GPU with no padding and no ring hops).

The device steps are behind a small `ops` object so that the very same sharding code runs
on a GPU (`RhjOps`: librhj.so's C-ABI) and, in the CPU tests, over gloo with the three HIP
calls replaced by the oracle (tests/test_shard_gloo.py: the checker standing in where no
GPU exists; nothing in this package imports it).
"""
import ctypes as C

import numpy as np


# ------------------------------------------------------------------ host-side planning

def assign_joins(sizes, world):
    """Longest-processing-time assignment of independent joins to ranks.
    sizes: per-join cost estimate (e.g. nR + nS).  Returns rank of every join."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0] * world
    owner = [0] * len(sizes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += sizes[i]
    return owner


def bucket_ranges(hist_r, hist_s, world):
    """Contiguous bucket ranges [lo, hi) per rank, balanced by histR + histS."""
    w = np.asarray(hist_r, dtype=np.float64) + np.asarray(hist_s, dtype=np.float64)
    total = w.sum()
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(len(w))
    cuts = [min(max(c, cuts[i - 1] if i else 0), len(w)) for i, c in enumerate(cuts)]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


# ------------------------------------------------------------------ device steps on a GPU

class RhjOps:
    """The three device steps of a sharded join through librhj.so (include/rhj.h).  Relations are
    int64 tensors [n,2] (value,row_id) on the GPU; results int64 tensors [m,2] on the GPU."""

    def __init__(self, rhj):
        self.rhj = rhj
        self.torch = rhj.torch

    def histogram(self, T, bits):
        self.rhj.set_bits(bits)
        h = self.torch.empty(1 << bits, dtype=self.torch.int64, device=T.device)
        rc = self.rhj.lib.rhj_bucket_histogram_device(T.data_ptr(), T.shape[0], h.data_ptr())
        if rc != 0:
            raise RuntimeError("rhj_bucket_histogram_device failed (%d)" % rc)
        return h

    def select(self, T, bits, lo, hi, count):
        """stable selection of the tuples with lo <= bucket < hi; `count` = their number (from the histogram)"""
        self.rhj.set_bits(bits)
        out = self.torch.empty((max(int(count), 1), 2), dtype=self.torch.int64, device=T.device)
        got = C.c_uint64(0)
        rc = self.rhj.lib.rhj_select_bucket_range_device(T.data_ptr(), T.shape[0], int(lo), int(hi), out.data_ptr(),
                                                         int(count), C.byref(got))
        if rc != 0 or got.value != int(count):
            raise RuntimeError("rhj_select_bucket_range_device: rc %d, %d tuples selected, histogram says %d"
                               % (rc, got.value, int(count)))
        return out[:int(count)]

    def join(self, R, S, bits):
        self.rhj.set_bits(bits)
        if R.shape[0] == 0 or S.shape[0] == 0:
            return self.torch.empty((0, 2), dtype=self.torch.int64, device=R.device)
        pairs, m = self.rhj.join_device(R, S, capacity=max(int(S.shape[0]), int(R.shape[0])))
        if m > pairs.shape[0]:                      # fan-out above the guess: the count is known now
            pairs, m = self.rhj.join_device(R, S, capacity=m)
        return pairs


# ------------------------------------------------------------------ the exchange step

def allgatherv_pairs(local, group=None):
    """All-gather-v of [m,2] int64 pair tensors: every rank gets the concatenation in rank order.
    Counts first (8 B per rank), then one grouped round of isend/irecv: each rank's pairs go once
    to every peer and land at their exact offset of the result — no padding to the largest rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    out = torch.empty((int(offs[-1]), 2), dtype=torch.int64, device=local.device)
    local = local.contiguous()
    out[offs[rank]:offs[rank + 1]] = local
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        gpeer = dist.get_global_rank(group, peer) if group is not None else peer
        if counts[rank]:
            ops.append(dist.P2POp(dist.isend, local, gpeer, group))
        if counts[peer]:
            ops.append(dist.P2POp(dist.irecv, out[offs[peer]:offs[peer + 1]], gpeer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts


# ------------------------------------------------------------------ one join over all ranks

def sharded_join(ops, R, S, bits, group=None, gather=True):
    """One RadixHashJoin over the ranks of `group`.  R, S: the whole relations, replicated on every rank
    (device-resident column store per GPU).  Every rank histograms both relations, takes its bucket range,
    selects and joins it; with `gather` the pair lists are exchanged so that every rank holds the canonical
    result.  Returns (pairs, info): pairs = canonical result (gather) or this rank's slice of it."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    hr = ops.histogram(R, bits)
    hs = ops.histogram(S, bits)
    hr_h, hs_h = hr.cpu().numpy(), hs.cpu().numpy()
    ranges = bucket_ranges(hr_h, hs_h, world)
    lo, hi = ranges[rank]
    cr, cs = int(hr_h[lo:hi].sum()), int(hs_h[lo:hi].sum())
    if world == 1:
        Rm, Sm = R, S                               # the whole radix: nothing to select
    else:
        Rm, Sm = ops.select(R, bits, lo, hi, cr), ops.select(S, bits, lo, hi, cs)
    local = ops.join(Rm, Sm, bits)
    info = {"range": (lo, hi), "ranges": ranges, "tuples": (cr, cs), "local_pairs": int(local.shape[0])}
    if not gather or not dist.is_initialized():
        info["counts"] = [int(local.shape[0])]
        return local, info
    full, counts = allgatherv_pairs(local, group)
    info["counts"] = counts
    return full, info


# ------------------------------------------------------------------ independent joins of a plan

def run_independent_joins(ops, joins, bits, group=None, gather=True):
    """joins: list of (R, S) relation pairs that share no intermediate result (each would start an
    inter_res node of its own, inter_res.c:147-150).  They are dealt to the ranks largest first
    (`assign_joins`), every rank runs its joins whole, and with `gather` every match list is sent to all
    ranks (all-gather-v per join: only the owner contributes pairs).  Returns (results, owner):
    results[i] = pairs of join i (None on ranks that neither own it nor gathered)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    owner = assign_joins([int(r.shape[0]) + int(s.shape[0]) for r, s in joins], world)
    results = [None] * len(joins)
    for i, (R, S) in enumerate(joins):
        if owner[i] == rank:
            results[i] = ops.join(R, S, bits)
    if gather and dist.is_initialized() and world > 1:
        for i, (R, S) in enumerate(joins):
            mine = results[i] if owner[i] == rank else torch.empty((0, 2), dtype=torch.int64, device=R.device)
            results[i], _ = allgatherv_pairs(mine, group)
    return results, owner
