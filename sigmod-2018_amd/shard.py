"""Multi-GPU sharding of the join path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Two ways the path shards, both without any collective on the data path of
a single join:

* across joins — independent joins of a plan (best_tree.c starts a new inter_res node for
  predicates that share no relation, inter_res.c:147-150) are dealt to ranks, largest first
  (`assign_joins`, `run_independent_joins`); each rank runs whole joins on its own GPU;
* inside one join — bucket b of R only ever meets bucket b of S (rhjoin.c:42-57), so ranks
  take contiguous BUCKET RANGES (equal width, or balanced by histR+histS: `bucket_ranges`) and
  join them with ONE call each — `rhj_join_device_range`: the join's own first partition pass
  drops the other ranks' buckets while it reads the relations — (`sharded_join`), and the
  concatenation of the per-rank pair lists in rank order IS the canonical order.

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


def bucket_slices(hist_r, hist_s, world):
    """Shares (lo, hi, first_skip, last_end) per rank that may be cut INSIDE a hot bucket (a hot bucket's probe side
    across GPUs, its build side replicated): rank r joins the buckets [lo, hi), of the first only the probe tuples from first_skip
    on, of the last only those before last_end (0: all) — `rhj_join_device_slice`.  A cut falls inside a bucket only when the
    bucket holds at least 1 / (2 world) of all tuples and both relations have tuples in it: the rank in front takes the build side
    and the probe tuples up to the cut (a multiple of 256, never a sliver of less than an eighth of the probe side) so that its
    tuples reach r / world of the total; every other cut is a bucket boundary.  `rhj_plan_device_slices` + `rhj_cut_to_slice` in Python, integer arithmetic (the two agree bit for bit)."""
    hr = [int(x) for x in np.asarray(hist_r).tolist()]
    hs = [int(x) for x in np.asarray(hist_s).tolist()]
    bins = len(hr)
    total = sum(hr) + sum(hs)
    cuts = [(0, 0)]
    at, cum = 0, 0
    for d in range(1, world):
        target = total * d // world
        while at < bins and cum + hr[at] + hs[at] <= target:
            cum += hr[at] + hs[at]
            at += 1
        cut = (at, 0)
        if at < bins:
            w = hr[at] + hs[at]
            if hr[at] and hs[at] and w * 2 * world >= total:
                pc, bc = max(hr[at], hs[at]), min(hr[at], hs[at])
                off = (target - cum - bc if target > cum + bc else 0) & ~255
                if off * 8 < pc:                     # (not a sliver: less than an eighth of the probe side on either side
                    off = 0                          #  goes to the boundary — it would cost a second partition of the build side)
                elif (pc - off) * 8 < pc:
                    off = pc
                if off >= pc:
                    cum += w
                    at += 1
                    cut = (at, 0)
                else:
                    cut = (at, off)
            elif cum < target:
                cum += w
                at += 1
                cut = (at, 0)
        cuts.append(cut)
    cuts.append((bins, 0))
    out = []
    for (b0, o0), (b1, o1) in zip(cuts, cuts[1:]):
        out.append((b0, b1 + 1, o0, o1) if o1 else (b0, b1, o0, 0))
    return out


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

    def join(self, R, S, bits, bucket_range=None):
        """the canonical pair list of R x S — of the buckets [lo, hi) only when a range is given (rhj_join_device_range: the
        join's own first partition pass drops the other buckets; no selection pass, one read of each relation)"""
        self.rhj.set_bits(bits)
        if bucket_range is not None and len(bucket_range) == 4 and not bucket_range[2] and not bucket_range[3]:
            bucket_range = bucket_range[:2]
        if R.shape[0] == 0 or S.shape[0] == 0 or (bucket_range is not None and bucket_range[0] >= bucket_range[1]):
            return self.torch.empty((0, 2), dtype=self.torch.int64, device=R.device)
        if self.rhj.lib.rhj_get_order():
            # the concatenation of the ranks' lists is the canonical order of THIS radix: with the order left to the library
            # every rank would pick its own radix from its slice sizes
            raise RuntimeError("sharded joins need the canonical order mode (rhj_set_order(0) / RHJ_ORDER unset)")
        guess = max(int(S.shape[0]), int(R.shape[0]))
        if bucket_range is not None:
            guess = max(guess * (bucket_range[1] - bucket_range[0]) // (1 << bits) * 5 // 4, 1 << 16)
        pairs, m = self.rhj.join_device(R, S, capacity=guess, bucket_range=bucket_range)
        if m > pairs.shape[0]:                      # fan-out above the guess: the count is known now
            pairs, m = self.rhj.join_device(R, S, capacity=m, bucket_range=bucket_range)
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

def equal_ranges(bits, world):
    """contiguous bucket ranges of equal width: no histogram, no read-back (uniform keys: balanced to within a bucket)"""
    bins = 1 << bits
    cuts = [bins * r // world for r in range(world + 1)]
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def sharded_join(ops, R, S, bits, group=None, gather=True, balance="equal"):
    """One RadixHashJoin over the ranks of `group`.  R, S: the whole relations, replicated on every rank
    (device-resident column store per GPU).  Every rank takes a contiguous bucket range and joins it with ONE call
    (`ops.join(..., bucket_range)`: the join's first partition pass drops the other ranks' buckets while it reads the
    relations — one read of each relation per rank, nothing in front of the join); with `gather` the pair lists are
    exchanged so that every rank holds the canonical result.
    balance = "equal": ranges of equal width, no histogram, no host round trip (the default: at one rank the call IS the
    plain join); "hist": ranges balanced by histR + histS (skewed keys) — two histogram launches and a 2^bits-word
    read-back in front of the join; "slice": the same, and a bucket that holds 1 / (2 world) of the tuples or more is shared by
    the ranks around the cut (`bucket_slices`: its probe side split, its build side on each of them).
    Returns (pairs, info): pairs = canonical result (gather) or this rank's slice of it."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if balance in ("hist", "slice") and world > 1:
        hr_h, hs_h = ops.histogram(R, bits).cpu().numpy(), ops.histogram(S, bits).cpu().numpy()
        ranges = bucket_slices(hr_h, hs_h, world) if balance == "slice" else bucket_ranges(hr_h, hs_h, world)
    else:
        ranges = equal_ranges(bits, world)
    local = ops.join(R, S, bits, None if world == 1 else ranges[rank])
    info = {"range": ranges[rank], "ranges": ranges, "local_pairs": int(local.shape[0])}
    if not gather or not dist.is_initialized():
        info["counts"] = [int(local.shape[0])]
        return local, info
    full, counts = allgatherv_pairs(local, group)
    info["counts"] = counts
    return full, info


# ------------------------------------------------------------------ independent joins of a plan

def run_independent_joins(ops, joins, bits, group=None, gather=True, consumers=None):
    """joins: list of (R, S) relation pairs that share no intermediate result (each would start an
    inter_res node of its own, inter_res.c:147-150).  They are dealt to the ranks largest first
    (`assign_joins`) and every rank runs its joins whole.  A match list crosses xGMI only when the
    operator that consumes it lives on another GPU:
      consumers[i] = rank that consumes join i's result (it alone receives the list: one send from the owner),
                     None / -1 = nobody else needs it, "all" = every rank (all-gather-v);
      consumers = None: gather=True means "all" for every join, gather=False means no exchange.
    The match counts of ALL joins go through one all-gather (one collective, one host read-back), the lists through
    one grouped round of isend / irecv.  Returns (results, owner): results[i] = pairs of join i on the ranks that own
    or consume it, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    owner = assign_joins([int(r.shape[0]) + int(s.shape[0]) for r, s in joins], world)
    results = [None] * len(joins)
    for i, (R, S) in enumerate(joins):
        if owner[i] == rank:
            results[i] = ops.join(R, S, bits)
    if consumers is None:
        consumers = ["all" if gather else None] * len(joins)
    if not dist.is_initialized() or world == 1 or not joins:
        return results, owner
    wanted = [c for c in consumers if c == "all" or (c is not None and c >= 0)]
    if not wanted:
        return results, owner
    dev = joins[0][0].device
    # one count exchange for all joins: every rank fills in the joins it owns
    mine = torch.zeros(len(joins), dtype=torch.int64, device=dev)
    for i in range(len(joins)):
        if owner[i] == rank:
            mine[i] = results[i].shape[0]
    counts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    total = torch.stack(counts).sum(dim=0).cpu().tolist()              # the one host read-back
    ops_list = []
    for i, (R, S) in enumerate(joins):
        c = consumers[i]
        if c is None or (c != "all" and c < 0):
            continue
        receivers = [r for r in range(world) if r != owner[i]] if c == "all" else ([c] if c != owner[i] else [])
        n = int(total[i])
        if rank in receivers:
            results[i] = torch.empty((n, 2), dtype=torch.int64, device=dev)
        if n == 0:
            continue
        for peer in receivers:
            gpeer = dist.get_global_rank(group, peer) if group is not None else peer
            gown = dist.get_global_rank(group, owner[i]) if group is not None else owner[i]
            if rank == owner[i]:
                ops_list.append(dist.P2POp(dist.isend, results[i].contiguous(), gpeer, group))
            elif rank == peer:
                ops_list.append(dist.P2POp(dist.irecv, results[i], gown, group))
    if ops_list:
        for req in dist.batch_isend_irecv(ops_list):
            req.wait()
    return results, owner
