"""Multi-GPU sharding of the join path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Two ways the path shards, both without any collective on the data path of
a single join:

* across joins — independent joins of a plan (best_tree.c starts a new inter_res node for
  predicates that share no relation, inter_res.c:147-150) are dealt to ranks, largest first
  (`assign_joins`); each rank runs whole joins on its own GPU;
* inside one join — bucket b of R only ever meets bucket b of S (rhjoin.c:42-57), so ranks
  take contiguous BUCKET RANGES balanced by histR+histS (`bucket_ranges`), join their slice
  (`join_bucket_range`), and the concatenation of the per-rank pair lists in rank order IS
  the canonical order.

The one exchange step is the match-list all-gather-v, issued only when the consumer of a
join's result lives on another GPU (`allgatherv_pairs`): an 8-byte count all-gather, then
one all_gather of pair tensors padded to the largest count (RCCL has no native allgatherv;
every GPU pair has a direct xGMI link, so this is a single direct exchange).
"""
import numpy as np


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


def join_bucket_range(R, S, bits, lo, hi, join_fn):
    """Join the tuples whose bucket lies in [lo, hi).  R, S: [n,2] int64 tensors (value,row_id)
    on any device; join_fn(Rsub, Ssub) -> [m,2] pairs in canonical order.  Boolean selection
    keeps the input order, so the slice's result is the canonical result restricted to those
    buckets."""
    mask = (1 << bits) - 1
    br, bs = R[:, 0] & mask, S[:, 0] & mask
    return join_fn(R[(br >= lo) & (br < hi)], S[(bs >= lo) & (bs < hi)])


def allgatherv_pairs(local, group=None):
    """All-gather-v of [m,2] int64 pair tensors: every rank gets the concatenation in rank
    order.  Two collectives: counts (8 B per rank), then pairs padded to the largest count."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts + [1])
    padded = torch.zeros((mx, 2), dtype=torch.int64, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0), counts
