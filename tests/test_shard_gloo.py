"""CPU, world_size 2 over gloo: the product's multi-GPU sharding code (sigmod-2018_amd/shard.py:
bucket-range sharding of one join + exact-size all-gather-v; independent joins of a plan dealt to ranks)
reproduces the single-rank canonical result.  The three device steps (histogram, selection, join) are
the oracle's here (tests/shard_ops.py) because this container has no GPU; tests/test_gpu_shard.py runs
the very same functions with RhjOps = librhj.so on the device."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _relations(o, bits):
    R = o.generate(40000, 0, 0, 0.0, 42)
    S = o.generate(70000, 2, 40000, 0.9, 43)          # Zipf: unbalanced buckets
    return R, S


def _worker(rank, world, port, bits, out_q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from pyoracle import Oracle, PAIR
    from shard_ops import OracleOps
    shard = importlib.import_module("sigmod-2018_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    ops = OracleOps()
    R, S = _relations(o, bits)
    tR = torch.from_numpy(R.view(np.int64).reshape(-1, 2).copy())
    tS = torch.from_numpy(S.view(np.int64).reshape(-1, 2).copy())

    def as_pairs(t):
        return t.numpy().view(np.uint64).reshape(-1, 2).copy().view(PAIR).reshape(-1)

    # (1) one join sharded by bucket range — ONE ranged join call per rank, nothing in front of it — pair lists exchanged
    full, info = shard.sharded_join(ops, tR, tS, bits)
    want = o.join(R, S, bits)
    got = as_pairs(full)
    ok1 = len(got) == len(want) and bool((got == want).all()) and sum(info["counts"]) == len(want)
    ok1 = ok1 and ops.calls == {"histogram": 0, "select": 0, "join": 1}
    # (1b) ranges balanced by histR + histS (Zipf keys): two histograms in front, same canonical result
    fullh, infoh = shard.sharded_join(ops, tR, tS, bits, balance="hist")
    ok1 = ok1 and bool((as_pairs(fullh) == want).all()) and ops.calls == {"histogram": 2, "select": 0, "join": 2}
    # (1c) a key that dominates (60 % of S): the cut falls INSIDE its bucket — the bucket's probe side is shared by the two ranks,
    # its build side on both (bucket_slices, rhj_join_device_slice) — same canonical result
    hotS = S.copy()
    hotS["value"][: len(hotS) * 6 // 10] = R["value"][len(R) // 3]
    thS = torch.from_numpy(hotS.view(np.int64).reshape(-1, 2).copy())
    fulls, infos = shard.sharded_join(ops, tR, thS, bits, balance="slice")
    wants = o.join(R, hotS, bits)
    gots = as_pairs(fulls)
    ok1 = ok1 and len(gots) == len(wants) and bool((gots == wants).all()) and ops.calls == {"histogram": 4, "select": 0, "join": 3}
    ok1 = ok1 and any(len(r) == 4 and (r[2] or r[3]) for r in infos["ranges"]) and min(infos["counts"]) * 4 > max(infos["counts"])
    ops.calls = {"histogram": 2, "select": 0, "join": 2}
    # (2) kept sharded (the consumer lives on this rank): this rank's slice only
    local, info2 = shard.sharded_join(ops, tR, tS, bits, gather=False)
    off = sum(info["counts"][:rank])
    ok2 = bool((as_pairs(local) == want[off:off + info["counts"][rank]]).all())
    # (3) independent joins of a plan: three joins of different size, gathered everywhere
    rels = [(o.generate(n, 0, 0, 0.0, 50 + i), o.generate(m, 1, n, 0.0, 60 + i)) for i, (n, m) in enumerate(((9000, 20000), (300, 500), (5000, 5000)))]
    tj = [(torch.from_numpy(a.view(np.int64).reshape(-1, 2).copy()), torch.from_numpy(b.view(np.int64).reshape(-1, 2).copy())) for a, b in rels]
    res, owner = shard.run_independent_joins(ops, tj, bits)
    ok3 = all(bool((as_pairs(r) == o.join(a, b, bits)).all()) for r, (a, b) in zip(res, rels))
    ok3 = ok3 and owner[0] != owner[2] and len(set(owner)) == 2
    # (3b) a match list crosses only to the rank that consumes it: join 0 -> the other rank, join 1 -> nobody, join 2 -> its owner
    cons = [1 - owner[0], None, owner[2]]
    res_c, owner_c = shard.run_independent_joins(ops, tj, bits, consumers=cons)
    for i, (a, b) in enumerate(rels):
        holds = rank == owner_c[i] or (cons[i] is not None and rank == cons[i])
        if holds:
            ok3 = ok3 and res_c[i] is not None and bool((as_pairs(res_c[i]) == o.join(a, b, bits)).all())
        else:
            ok3 = ok3 and res_c[i] is None
    # (4) the recorded joins of `small` (a subset: the oracle is the device here) dealt to the two ranks
    import helpers
    g = helpers.Golden()
    recs = [j for j in g.small["joins"] if j["nR"] + j["nS"] < 60000 and j["matches"] < 200000][:24]
    sj = []
    for j in recs:
        a, b = g.small_join(j["idx"])
        sj.append((torch.from_numpy(a.view(np.int64).reshape(-1, 2).copy()), torch.from_numpy(b.view(np.int64).reshape(-1, 2).copy())))
    res4, owner4 = shard.run_independent_joins(ops, sj, 4)
    ok4 = len(set(owner4)) == 2
    for r, j in zip(res4, recs):
        try:
            helpers.assert_digest(o, as_pairs(r), j, "small join %d" % j["idx"])
        except AssertionError:
            ok4 = False
    out_q.put((rank, ok1 and ok2 and ok3 and ok4, info["counts"], info["range"], (ok1, ok2, ok3, ok4)))
    dist.destroy_process_group()


@pytest.mark.parametrize("bits", [4, 8])
def test_bucket_range_sharding_and_allgatherv(bits):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bits, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    ranges = sorted(tuple(r[3])[:2] for r in res)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 1 << bits
    assert all(c > 0 for c in res[0][2])              # both ranks contributed pairs


def test_assign_joins_balances_largest_first():
    shard = importlib.import_module("sigmod-2018_amd.shard")
    sizes = [318066 + 40363, 270137 + 43131, 84343 + 26388, 5715 + 3754, 787 + 1561, 1 + 1561]
    owner = shard.assign_joins(sizes, 2)
    load = [sum(s for s, o in zip(sizes, owner) if o == r) for r in range(2)]
    assert owner[0] != owner[1] and abs(load[0] - load[1]) <= max(sizes)
    assert shard.assign_joins(sizes, 1) == [0] * len(sizes)
    ranges = shard.bucket_ranges([5, 0, 0, 100], [5, 0, 0, 100], 3)
    assert ranges[0][0] == 0 and ranges[-1][1] == 4 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
