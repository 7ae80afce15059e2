"""GPU: the sharded join (sigmod-2018_amd/shard.py) with its device steps on the MI355X — RhjOps =
rhj_join_device_range (and rhj_bucket_histogram_device, rhj_select_bucket_range_device, rhj_join_device) of librhj.so — and the
exchange step over RCCL (backend "nccl"), at the world size this box offers (1 GPU under gpurun), checked
bit for bit against the oracle.  The rank-count-dependent part (who takes which range, exact-size
all-gather-v between peers) is covered at world_size 2 on gloo in tests/test_shard_gloo.py; here every
RANGE of a 2-, 3- and 8-way split is additionally run on the one GPU and concatenated."""
import importlib
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mod():
    return importlib.import_module("sigmod-2018_amd")


@pytest.fixture(scope="module")
def rhj(mod):
    return mod.RHJ()


@pytest.fixture(scope="module")
def shard():
    return importlib.import_module("sigmod-2018_amd.shard")


@pytest.fixture(scope="module")
def nccl_world():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    yield dist
    dist.destroy_process_group()


def relations(oracle, kind):
    if kind == "zipf":
        return oracle.generate(60000, 0, 0, 0.0, 42), oracle.generate(150000, 2, 60000, 0.9, 43)
    R = oracle.generate(50000, 4, 3000, 0.0, 44)             # duplicates on both sides
    S = oracle.generate(80000, 4, 3000, 0.0, 45)
    R["value"] |= np.uint64(1) << np.uint64(63)              # top bit set: int64 views are negative
    S["value"] |= np.uint64(1) << np.uint64(63)
    R["row_id"] = R["row_id"][::-1].copy() + np.uint64(7)    # arbitrary row ids survive the selection
    return R, S


@pytest.mark.parametrize("kind", ["zipf", "dups_topbit"])
@pytest.mark.parametrize("bits", [4, 8, 12])
def test_sharded_join_on_the_device_under_nccl(rhj, shard, oracle, nccl_world, kind, bits):
    ops = shard.RhjOps(rhj)
    R, S = relations(oracle, kind)
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    want = oracle.join(R, S, bits)
    # the whole path at this box's world size: ranges, (ranged) join, all-gather-v over RCCL
    for balance in ("equal", "hist"):
        full, info = shard.sharded_join(ops, dR, dS, bits, balance=balance)
        assert info["ranges"] == [(0, 1 << bits)] and info["counts"] == [len(want)]
        assert np.array_equal(rhj.pairs_to_numpy(full), want)
    # histogram against numpy
    mask = np.uint64((1 << bits) - 1)
    hr = np.bincount((R["value"] & mask).astype(np.int64), minlength=1 << bits)
    hs = np.bincount((S["value"] & mask).astype(np.int64), minlength=1 << bits)
    assert np.array_equal(ops.histogram(dR, bits).cpu().numpy(), hr)
    assert np.array_equal(ops.histogram(dS, bits).cpu().numpy(), hs)
    # every range of an N-way split through select + join on this GPU; concatenation = canonical order
    for world in (2, 3, 8):
        parts = []
        for lo, hi in shard.bucket_ranges(hr, hs, world):
            cr, cs = int(hr[lo:hi].sum()), int(hs[lo:hi].sum())
            Rm, Sm = ops.select(dR, bits, lo, hi, cr), ops.select(dS, bits, lo, hi, cs)
            sel = (R["value"] & mask >= np.uint64(lo)) & (R["value"] & mask < np.uint64(hi))
            got = Rm.cpu().numpy().view(np.uint64)
            assert np.array_equal(got[:, 0], R["value"][sel]) and np.array_equal(got[:, 1], R["row_id"][sel])
            parts.append(rhj.pairs_to_numpy(ops.join(Rm, Sm, bits)))
        assert np.array_equal(np.concatenate(parts), want), (world, bits, kind)
    # the same ranges through ONE call each — rhj_join_device_range: the join's first partition pass drops the other ranks'
    # buckets — equal-width and histogram-balanced splits; concatenation = canonical order, every slice = the oracle's slice
    keyR = {int(r): int(v) for r, v in zip(R["row_id"], R["value"])} if len(R) < 100000 else None
    for world in (2, 3, 8):
        for ranges in (shard.equal_ranges(bits, world), shard.bucket_ranges(hr, hs, world)):
            parts = [rhj.pairs_to_numpy(ops.join(dR, dS, bits, rng)) for rng in ranges]
            assert np.array_equal(np.concatenate(parts), want), (world, bits, kind, ranges)
            if keyR is not None:                   # every slice holds exactly its own buckets
                for (lo, hi), part in zip(ranges, parts):
                    b = np.array([keyR[int(x)] for x in part["row_idR"][:2000]], dtype=np.uint64) & mask
                    assert ((b >= lo) & (b < hi)).all()


@pytest.mark.parametrize("bits,n,spec", [(8, 5_000_000, 0), (12, 5_000_000, 0), (10, 5_000_000, 1), (10, 8_000_000, 1)])
def test_ranged_join_at_the_sizes_a_sharded_run_has(rhj, shard, oracle, bits, n, spec):
    """rhj_join_device_range where a 100 M-tuple multi-GPU run takes it: more than 1024 tiles a relation (the chunked scan of
    the one-pass partition, k_local_part<RANGED> over strips of several tiles with the counts taken in pass 1, k_group_scan /
    k_bucket_psum over thinned tiles), the fused join and the foreign-key speculation (k_join_spec at 4.9 K tuples a bucket,
    k_join_exact at 7.8 K) on a ranged partition.  Every range of a three-way split, equal-width and histogram-balanced:
    the concatenation is the plain call's result bit for bit, which is the oracle's.
    Regression guard for the fault of profiles/README.md r03i (round 3): thinned tiles (kept < count) in k_scatter_lds<RANGED>
    and S's offsets behind R's KEPT tuples in k_small_scan_plan — the 8-bit case walks both at 1221 tiles a relation."""
    ops = shard.RhjOps(rhj)
    R = oracle.generate(n, 0, 0, 0.0, 71)
    S = oracle.generate(n, 1, n, 0.0, 72)
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    want = oracle.join(R, S, bits)
    rhj.set_bits(bits)
    rhj.lib.rhj_set_spec(1)
    rhj.lib.rhj_set_exact(1)
    plain, m = rhj.join_device(dR, dS, capacity=n)
    assert m == len(want) and np.array_equal(rhj.pairs_to_numpy(plain)[:m], want)
    assert rhj.lib.rhj_last_spec() == spec
    del plain
    mask = np.uint64((1 << bits) - 1)
    hr = np.bincount((R["value"] & mask).astype(np.int64), minlength=1 << bits)
    hs = np.bincount((S["value"] & mask).astype(np.int64), minlength=1 << bits)
    for ranges in (shard.equal_ranges(bits, 3), shard.bucket_ranges(hr, hs, 3)):
        at = 0
        for rng in ranges:
            part = rhj.pairs_to_numpy(ops.join(dR, dS, bits, rng))
            assert rhj.lib.rhj_last_spec() == spec, (bits, rng)       # the speculation is tried on a rank's share as on the whole join
            assert np.array_equal(part, want[at:at + len(part)]), (bits, rng)
            at += len(part)
        assert at == len(want)
    if spec:
        # one S tuple without a partner, in the middle range: that rank's speculation fails (the ordinary kernel takes over in the
        # same call), the other ranks' holds
        S2 = S.copy()
        victim = int(np.nonzero((S2["value"] & mask) == np.uint64((1 << bits) // 2))[0][0])
        S2["value"][victim] = (S2["value"][victim] & mask) | np.uint64(1 << 62)
        want2 = oracle.join(R, S2, bits)
        dS2 = rhj.to_device(S2)
        at, went = 0, []
        for rng in shard.equal_ranges(bits, 3):
            rhj.lib.rhj_set_spec(1)
            part = rhj.pairs_to_numpy(ops.join(dR, dS2, bits, rng))
            went.append(rhj.lib.rhj_last_spec())
            assert np.array_equal(part, want2[at:at + len(part)]), (bits, rng)
            at += len(part)
        assert at == len(want2) and went == [1, 2, 1]


def test_allgatherv_and_independent_joins_on_device_tensors(rhj, shard, oracle, nccl_world):
    import torch
    ops = shard.RhjOps(rhj)
    local = torch.arange(2 * 12345, dtype=torch.int64, device=rhj.dev).reshape(-1, 2)
    full, counts = shard.allgatherv_pairs(local)
    assert counts == [12345] and torch.equal(full, local)
    empty, counts = shard.allgatherv_pairs(torch.empty((0, 2), dtype=torch.int64, device=rhj.dev))
    assert counts == [0] and empty.shape == (0, 2)
    rels = [(oracle.generate(n, 0, 0, 0.0, 50 + i), oracle.generate(m, 1, n, 0.0, 60 + i))
            for i, (n, m) in enumerate(((9000, 20000), (300, 500), (5000, 5000)))]
    joins = [(rhj.to_device(a), rhj.to_device(b)) for a, b in rels]
    res, owner = shard.run_independent_joins(ops, joins, 8)
    assert owner == [0, 0, 0]
    for r, (a, b) in zip(res, rels):
        assert np.array_equal(rhj.pairs_to_numpy(r), oracle.join(a, b, 8))


def test_select_capacity_and_empty_ranges(rhj, oracle):
    import ctypes as C
    import torch
    rhj.set_bits(6)
    R = oracle.generate(10000, 1, 1000, 0.0, 9)
    dR = rhj.to_device(R)
    out = torch.empty((10000, 2), dtype=torch.int64, device=rhj.dev)
    got = C.c_uint64(0)
    lib = rhj.lib
    assert lib.rhj_select_bucket_range_device(dR.data_ptr(), 10000, 5, 5, out.data_ptr(), 10000, C.byref(got)) == 0 and got.value == 0
    assert lib.rhj_select_bucket_range_device(dR.data_ptr(), 10000, 0, 64, out.data_ptr(), 10000, C.byref(got)) == 0 and got.value == 10000
    assert torch.equal(out, dR)
    assert lib.rhj_select_bucket_range_device(dR.data_ptr(), 10000, 0, 64, out.data_ptr(), 100, C.byref(got)) == 1 and got.value == 10000
    assert lib.rhj_select_bucket_range_device(dR.data_ptr(), 0, 0, 64, out.data_ptr(), 100, C.byref(got)) == 0 and got.value == 0


def test_small_workload_joins_as_independent_joins_of_a_plan(rhj, shard, oracle, golden, nccl_world):
    """shard.run_independent_joins over the 88 RadixHashJoin calls of `small` (recorded inputs): dealt with assign_joins,
    run through rhj_join_device, gathered; every pair list against the digest of the compiled reference."""
    import helpers
    ops = shard.RhjOps(rhj)
    recs = golden.small["joins"]
    joins = [tuple(rhj.to_device(x) for x in golden.small_join(j["idx"])) for j in recs]
    res, owner = shard.run_independent_joins(ops, joins, 4)
    assert len(res) == 88 and set(owner) == {0}
    for r, j in zip(res, recs):
        helpers.assert_digest(oracle, rhj.pairs_to_numpy(r), j, "small join %d" % j["idx"])
    # the same deal for 8 ranks: balanced within the largest join, nothing lost
    sizes = [j["nR"] + j["nS"] for j in recs]
    own8 = shard.assign_joins(sizes, 8)
    load = [sum(s for s, o in zip(sizes, own8) if o == r) for r in range(8)]
    assert sorted(set(own8)) == list(range(8)) and max(load) - min(load) <= max(sizes)


@pytest.mark.parametrize("bits,nR,nS,kind", [(4, 200_000, 300_000, "hot"), (8, 400_000, 400_000, "dup"), (12, 3_000_000, 3_000_000, "fk"),
                                             (10, 500_000, 800_000, "hbm"), (14, 2_000_000, 6_000_000, "hot")])
def test_shares_cut_inside_buckets_tile_the_plain_result(rhj, shard, oracle, bits, nR, nS, kind):
    """rhj_join_device_slice (a hot bucket's probe side across GPUs, its build side on each): shares that tile the
    (bucket, position among the bucket's probe tuples) space concatenate to the plain join's list, bit for bit — cuts planned by
    shard.bucket_slices for 2, 3 and 5 ranks and cuts at arbitrary positions of arbitrary buckets; one-pass and two-pass
    partitions, the fused kernels with split units, the tiled path (HBM tables), keys repeated on both sides."""
    import torch
    rng = np.random.default_rng(bits * 1000 + nR % 977)
    R = oracle.generate(nR, 4 if kind == "dup" else 0, nR // 3, 0.0, 21)
    S = oracle.generate(nS, 2 if kind == "dup" else 1, nR // 3 if kind == "dup" else nR, 0.7, 22)
    if kind == "hot":
        S["value"][rng.integers(0, nS, nS * 6 // 10)] = R["value"][5]          # one key holds 45 % of S
    rhj.set_bits(bits)
    rhj.lib.rhj_set_force_hbm_table(1 if kind == "hbm" else 0)
    try:
        dR, dS = rhj.to_device(R), rhj.to_device(S)
        plain, m = rhj.join_device(dR, dS)
        if nR + nS <= 1_000_000:
            assert np.array_equal(rhj.pairs_to_numpy(plain), oracle.join(R, S, bits))
        mask = np.uint64((1 << bits) - 1)
        hr = np.bincount((R["value"] & mask).astype(np.int64), minlength=1 << bits)
        hs = np.bincount((S["value"] & mask).astype(np.int64), minlength=1 << bits)
        plans = [shard.bucket_slices(hr, hs, w) for w in (2, 3, 5)]
        if kind == "hot":
            assert any(any(s_[2] or s_[3] for s_ in p) for p in plans)          # the hot bucket IS cut (where a cut lands well inside it)
        # arbitrary cuts: (bucket, position) pairs in order, positions anywhere inside the bucket's probe side (also beyond it)
        cuts = sorted((int(b), int(rng.integers(0, max(hr[b], hs[b]) + 300))) for b in rng.integers(0, 1 << bits, 4))
        cuts = [(0, 0)] + cuts + [(1 << bits, 0)]
        plans.append([(b0, b1 + 1, o0, o1) if o1 else (b0, b1, o0, 0) for (b0, o0), (b1, o1) in zip(cuts, cuts[1:])])
        for plan in plans:
            parts = []
            for lo, hi, skip, end in plan:
                if lo >= hi:
                    continue
                t, k = rhj.join_device(dR, dS, bucket_range=(lo, hi, skip, end))
                assert rhj.lib.rhj_last_spec() != 1                             # (never the speculation on a cut bucket)
                parts.append(t[:k])
            got = torch.cat(parts) if parts else plain[:0]
            assert got.shape[0] == m and torch.equal(got, plain[:m]), (bits, kind, plan)
    finally:
        rhj.lib.rhj_set_force_hbm_table(0)


def test_a_few_bit_share_takes_the_low_radix_path(rhj, oracle):
    """rhj_join_device_range on the reference's 4 radix bits over relations whose buckets are beyond the fused kernels' index: the
    share runs the low-radix path like the whole join does — its first pass, on the caller's bits, drops the other
    ranks' buckets — instead of HBM tables over build sides of hundreds of thousands of tuples.  The shares' lists concatenate to
    the plain join's, bit for bit (uniform and repeated keys); a share cut inside a bucket is the plan's matter (tiled path)."""
    import torch
    n = 6_000_000
    for kindR, kindS, dom in ((0, 1, n), (0, 1, n // 2)):
        R = oracle.generate(n, kindR, dom, 0.0, 31)
        S = oracle.generate(n + 1000, kindS, dom, 0.0, 32)
        dR, dS = rhj.to_device(R), rhj.to_device(S)
        rhj.set_bits(4)
        plain, m = rhj.join_device(dR, dS)
        assert rhj.stats()["path"] == "lowradix", rhj.stats()
        parts = []
        for lo, hi in ((0, 4), (4, 5), (5, 12), (12, 16)):
            t, k = rhj.join_device(dR, dS, bucket_range=(lo, hi))
            assert rhj.stats()["path"] == "lowradix", (lo, hi, rhj.stats())
            parts.append(t[:k])
        assert sum(p.shape[0] for p in parts) == m and torch.equal(torch.cat(parts), plain[:m])
        t, k = rhj.join_device(dR, dS, bucket_range=(3, 9, 1000, 2000))       # cut inside buckets: the plan's matter
        assert rhj.stats()["path"] != "lowradix" and k > 0
        del dR, dS, plain, parts, t
    for nr, ns, lr in ((200_000, 300_000, False), (700_000, 900_000, True)):   # against the oracle: a share the fused kernels take as it is, and a low-radix one
        sR, sS = R[:nr], S[:ns]
        t, k = rhj.join_device(rhj.to_device(sR), rhj.to_device(sS), bucket_range=(2, 7))
        assert (rhj.stats()["path"] == "lowradix") == lr, (nr, rhj.stats())
        inR, inS = ((x["value"] & np.uint64(15) >= 2) & (x["value"] & np.uint64(15) < 7) for x in (sR, sS))
        assert np.array_equal(rhj.pairs_to_numpy(t[:k]), oracle.join(sR[inR], sS[inS], 4))
