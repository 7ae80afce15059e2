"""GPU parity tests proper: the HIP path, called through the C-ABI of include/rhj.h,
against the oracle on the same seeded inputs and against the committed golden vectors
(bit-exact: integer / index work)."""
import importlib
import os

import numpy as np
import pytest

import helpers
from helpers import assert_digest, make_rel
from pyoracle import PAIR, TUPLE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rhj():
    mod = importlib.import_module("sigmod-2018_amd")
    r = mod.RHJ(device=0)
    yield r
    set_path(r, "fused")
    r.lib.rhj_set_fused(1)
    r.lib.rhj_set_empty_mode(0)


PATHS = ["fused", "fused_gather", "tiled32", "tiled64"]


def set_path(rhj, path):
    rhj.lib.rhj_set_fused(2 if path.startswith("fused") else 0)     # 2: also where the tiny-bucket rule would pick the tiled path
    rhj.lib.rhj_set_resident(0 if path == "fused_gather" else 1)
    rhj.lib.rhj_set_force_hbm_table(1 if path == "tiled64" else 0)


def dev_join(rhj, R, S):
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S))
    out = rhj.pairs_to_numpy(t)
    assert m == len(out)
    return out


@pytest.mark.parametrize("path", PATHS)
def test_synthetic_golden_device(rhj, golden, oracle, path):
    set_path(rhj, path)
    for c in golden.synthetic["cases"]:
        if path == "tiled64" and c["R"]["n"] + c["S"]["n"] > 600000:
            continue
        rhj.set_bits(c["bits"])
        R, S = golden.gen(c["R"]), golden.gen(c["S"])
        assert_digest(oracle, dev_join(rhj, R, S), c, "%s path=%s" % (c["name"], path))
    set_path(rhj, "fused")


def test_edge_cases_host_abi(rhj, golden):
    for c in golden.edges["cases"]:
        rhj.set_bits(c["bits"])
        R = make_rel([int(v) for v in c["R"]]); S = make_rel([int(v) for v in c["S"]])
        for null_mode, key in ((0, "null_t4"), (1, "null_t1")):
            rhj.lib.rhj_set_empty_mode(null_mode)
            got, info = rhj.RadixHashJoin(R, S, with_info=True)
            assert [list(map(int, p)) for p in got.tolist()] == c["pairs"], c["name"]
            assert info["null"] == c[key], (c["name"], key)
            # a list never carries an empty node before a non-empty one (SURVEY finding 8)
            assert all(l > 0 for l in info["loads"]) or info["loads"] == [0]
    rhj.lib.rhj_set_empty_mode(0)


def test_arbitrary_row_ids(rhj, golden, oracle):
    R, S, rec = golden.arbitrary_row_id_inputs()
    rhj.set_bits(4)
    assert_digest(oracle, dev_join(rhj, R, S), rec, "arbitrary_row_ids")
    assert_digest(oracle, rhj.RadixHashJoin(R, S), rec, "arbitrary_row_ids host")


@pytest.mark.parametrize("where", ["none", "ends", "middle_only", "one_in_S"])
@pytest.mark.parametrize("bits", [9, 12, 14])
def test_row_ids_wider_than_32_bits_in_the_two_pass_partition(rhj, oracle, where, bits):
    """The two-pass partition keeps 12-byte tuples between its passes when a sample of the row ids fits
    32 bits; wide row ids the sample does not see (middle of the array) must be caught by pass 1 and the
    join run again with 16-byte intermediates.  Same for the partition entry point."""
    nR, nS = 60_000, 90_000
    R = oracle.generate(nR, 0, 0, 0.0, 71)
    S = oracle.generate(nS, 1, nR, 0.0, 72)
    wide = np.uint64(1) << np.uint64(40)
    if where == "ends":
        R["row_id"][:10] += wide
        S["row_id"][-3:] += wide
    elif where == "middle_only":
        R["row_id"][nR // 2: nR // 2 + 5] += wide
        S["row_id"][nS // 3] += wide
    elif where == "one_in_S":
        S["row_id"][nS // 2] = np.uint64(0xFFFFFFFF00000001)
    set_path(rhj, "fused")
    rhj.set_bits(bits)
    want = oracle.join(R, S, bits)
    got = dev_join(rhj, R, S)
    assert len(got) == len(want) and (got == want).all()
    wpart, whist, wpsum = oracle.partition(S, bits)
    out, hist, psum = rhj.partition_device(rhj.to_device(S), bits)
    part = out.cpu().numpy().view(np.uint64).reshape(-1, 2).copy().view(TUPLE).reshape(-1)
    assert (part == wpart).all() and (hist == whist).all() and (psum == wpsum).all()


def test_last_bucket_skew(rhj, golden, oracle):
    k = golden.edges["last_bucket_skew"]
    vals = np.array(k["values_R"], dtype=np.uint64)
    rhj.set_bits(4)
    assert_digest(oracle, dev_join(rhj, make_rel(vals), make_rel(vals[np.array(k["perm"])])), k["t1"], "skew")


@pytest.mark.parametrize("path", ["fused", "fused_gather", "tiled32"])
def test_small_workload_joins(rhj, golden, oracle, path):
    rhj.set_bits(4)
    set_path(rhj, path)
    for j in golden.small["joins"]:
        R, S = golden.small_join(j["idx"])
        if j["idx"] % 4 == 0:
            got = rhj.RadixHashJoin(R, S)          # every 4th through the host ABI
        else:
            got = dev_join(rhj, R, S)
        assert_digest(oracle, got, j, "small join %d" % j["idx"])
    set_path(rhj, "fused")


def test_small_workload_filters(rhj, golden, oracle):
    for f in golden.small["filters"]:
        rel = golden.small_relations["r%d" % f["rel"]].astype(np.uint64)
        ids, info = rhj.Filter(list(rel), rel.shape[1], f["col"], f["op"], f["value"], with_info=True)
        assert len(ids) == f["hits"] and "%016x" % oracle.fnv(ids) == f["fnv"], f
        assert info["null"] == f["null"]


def test_filter_golden(rhj, golden, oracle):
    torch = rhj.torch
    for c in golden.filters["cases"]:
        col, sel = golden.filter_inputs(c)
        ids = rhj.Filter([col], len(col), 0, c["op"], c["value"], sel=sel)
        assert len(ids) == c["hits"] and "%016x" % oracle.fnv(ids) == c["fnv"], c
        dcol = torch.from_numpy(col.view(np.int64)).to(rhj.dev)
        dsel = torch.from_numpy(sel.view(np.int64)).to(rhj.dev) if sel is not None else None
        d = rhj.filter_device(dcol, c["op"], c["value"], dsel).cpu().numpy().view(np.uint64)
        assert len(d) == c["hits"] and (d == ids).all()


@pytest.mark.parametrize("bits", [1, 2, 5, 8, 9, 11, 12, 13, 14, 15])
def test_partition_matches_oracle(rhj, oracle, bits):
    for n, kind, dom in ((1, 4, 3), (63, 4, 5), (4097, 4, 1 << 30), (300000, 1, 100000), (1000003, 2, 50000)):
        rel = oracle.generate(n, kind, dom, 0.9, 100 + bits)
        want, hist, psum = oracle.partition(rel, bits)
        out, h, p = rhj.partition_device(rhj.to_device(rel), bits)
        got = out.cpu().numpy().view(np.uint64).reshape(-1, 2).copy().view(TUPLE).reshape(-1)
        assert (h == hist).all() and (p == psum).all()
        assert (got == want).all(), "partition differs (n=%d bits=%d)" % (n, bits)


@pytest.mark.parametrize("bits", [9, 10, 12])
def test_partition_counts_from_digit_bytes_match_the_counts_of_pass_1(rhj, oracle, bits):
    """Up to 12 radix bits pass 1 counts pass 2's digits itself (strips of tiles, cells in LDS); the digit-byte kernel that
    serves 13..15 bits can be forced at any width: both must give the oracle's partition, also on skewed keys and on a
    relation that ends inside a strip."""
    try:
        for n, kind, dom in ((4096 * 5 + 17, 4, 1 << 30), (700001, 1, 100000), (1200000, 2, 2000), (300000, 4, 1)):
            rel = oracle.generate(n, kind, dom, 0.9, 300 + bits)
            want, hist, psum = oracle.partition(rel, bits)
            for on in (1, 0):
                rhj.lib.rhj_set_count_in_pass1(on)
                out, h, p = rhj.partition_device(rhj.to_device(rel), bits)
                got = out.cpu().numpy().view(np.uint64).reshape(-1, 2).copy().view(TUPLE).reshape(-1)
                assert (h == hist).all() and (p == psum).all() and (got == want).all(), (n, bits, on)
    finally:
        rhj.lib.rhj_set_count_in_pass1(1)


def test_foreign_key_speculation_holds_or_hands_over(rhj, oracle):
    """Big joins first run k_join_spec on the hypothesis that every tuple of the bigger relation has exactly one match
    (pairs written without stash or chained offsets); one tuple that breaks it — no partner, two partners, on either side —
    and the ordinary kernel does the join in the same call.  Identical pairs in every case; rhj_last_spec() says which way it went."""
    rhj.set_bits(9)                                              # (the speculation wants 4096 tuples of the bigger relation a bucket)
    set_path(rhj, "fused")
    nR, nS = 1_200_000, 2_200_000
    R = oracle.generate(nR, 0, 0, 0.0, 91)                       # unique keys 0..nR-1
    S = oracle.generate(nS, 1, nR, 0.0, 92)                      # every S key has its R tuple
    def run(R, S, expect):
        rhj.lib.rhj_set_spec(1)                                  # (also resets the try-or-not score)
        want = oracle.join(R, S, 9)
        t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), capacity=max(len(R), len(S)) + 16)   # (room for the prediction)
        got = rhj.pairs_to_numpy(t)
        assert m == len(want) and len(got) == len(want) and (got == want).all()
        assert rhj.lib.rhj_last_spec() == expect, (rhj.lib.rhj_last_spec(), expect)
    try:
        run(R, S, 1)                                             # S is the foreign-key side
        run(S, R, 1)                                             # ... and as the first argument (hypothesis on R)
        S2 = S.copy(); S2["value"][nS // 2] = np.uint64(1 << 50)   # one S tuple without a partner
        run(R, S2, 2)
        R2 = R.copy(); R2["value"][nR // 3] = R2["value"][nR // 3 + 1]   # one R key twice (and one missing)
        run(R2, S, 2)
        run(S2, R, 2)
        Rb = oracle.generate(nS + 5, 0, 0, 0.0, 93)              # R bigger than S: the hypothesis is on R, whose tuples match 0..k times
        run(Rb, S, 2)
        rhj.lib.rhj_set_spec(0)
        want = oracle.join(R, S, 9)
        got = dev_join(rhj, R, S)
        assert (got == want).all() and rhj.lib.rhj_last_spec() == 0
    finally:
        rhj.lib.rhj_set_spec(1)


def test_speculation_on_gathered_build_sides_both_kinds_of_units(rhj, oracle):
    """k_join_spec<false> — build sides beyond the LDS-resident size (7.8 K tuples a bucket: 8M x 8M at 10 bits), both relations
    the same size, so that about half of the buckets are probed by S, the hypothesis' relation (pairs written from the probe loop)
    and half by R (fj_group_direct: first candidates four a lane, the others lane by lane, second and later matches as records,
    the groups of a unit chained in LDS).  Uniform draws (0..9 matches an R tuple); draws from half of R's keys (twice the
    matches, ~145 records a group); from an eighth (eight matches a key, ~224 records a group); from the FIRST quarter of R's
    tuples (whole groups of tuples with four matches each: more than the 512 records a wave keeps — such a unit's pairs are written
    again by k_join_walk, from the index alone, and the speculation stands); one R tuple with 700 matches (its group's records
    run out likewise).  The pairs are the oracle's in every case."""
    bits, n = 10, 8_000_000
    rhj.set_bits(bits)
    set_path(rhj, "fused")
    R = oracle.generate(n, 0, 0, 0.0, 191)
    rng = np.random.default_rng(192)
    def run(S, expect):
        rhj.lib.rhj_set_spec(1)
        want = oracle.join(R, S, bits)
        t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), capacity=len(S) + 16)
        got = rhj.pairs_to_numpy(t)[:m]
        assert m == len(want) and (got == want).all()
        assert rhj.lib.rhj_last_spec() == expect, (rhj.lib.rhj_last_spec(), expect)
    try:
        S = helpers.make_rel(R["value"][rng.integers(0, n, n)])
        run(S, 1)
        some = rng.permutation(n)                                    # (a random half / eighth of R's tuples: mixed with the others inside every group)
        run(helpers.make_rel(R["value"][some[rng.integers(0, n // 2, n)]]), 1)
        run(helpers.make_rel(R["value"][some[rng.integers(0, n // 8, n)]]), 1)
        run(helpers.make_rel(R["value"][rng.integers(0, n // 4, n)]), 1)     # the FIRST quarter of R's tuples: whole groups of them in every bucket
        S3 = S.copy()
        hr = np.bincount((R["value"] & np.uint64(1023)).astype(np.int64), minlength=1024)
        hs = np.bincount((S3["value"] & np.uint64(1023)).astype(np.int64), minlength=1024)
        b0 = int(np.nonzero(hr >= hs)[0][0])                         # a bucket R probes (rhjoin.c:86)
        k0 = R["value"][np.nonzero((R["value"] & np.uint64(1023)) == np.uint64(b0))[0][7]]
        same_bucket = np.nonzero((S3["value"] & np.uint64(1023)) == np.uint64(b0))[0]
        S3["value"][same_bucket[:700]] = k0                          # (the bucket's sizes stay what they were)
        run(S3, 1)
    finally:
        rhj.lib.rhj_set_spec(1)


@pytest.mark.parametrize("bits,nR,nS", [(4, 30_000, 50_000), (8, 300_000, 500_000), (9, 700_000, 900_000), (12, 2_000_000, 3_000_000),
                                        (10, 8_000_000, 8_000_000), (14, 1_000_000, 5_000_000)])
def test_join_of_key_columns(rhj, oracle, bits, nR, nS):
    """rhj_join_keys_device: relations given as key columns, row id = position (what GetRelation makes of a base relation,
    inter_res.c:199-204).  On 9..15 bits pass 1 of the partition reads the columns themselves (k_local_part<., ., ., COL>: 8 bytes a
    tuple); elsewhere the tuples are built first.  The pairs are the oracle's on {keys[i], i}."""
    import ctypes as C
    import torch
    R = oracle.generate(nR, 0, 0, 0.0, 301)
    S = oracle.generate(nS, 1, nR, 0.0, 302)
    assert (R["row_id"] == np.arange(nR, dtype=np.uint64)).all()
    want = oracle.join(R, S, bits)
    rhj.set_bits(bits)
    set_path(rhj, "fused")
    kR = torch.from_numpy(R["value"].view(np.int64).copy()).to(rhj.dev)
    kS = torch.from_numpy(S["value"].view(np.int64).copy()).to(rhj.dev)
    out = torch.empty((len(want) + 8, 2), dtype=torch.int64, device=rhj.dev)
    m = C.c_uint64(0)
    for rep in range(2):                                         # (twice: the speculation's score and the workspace carry over)
        assert rhj.lib.rhj_join_keys_device(kR.data_ptr(), nR, kS.data_ptr(), nS, out.data_ptr(), out.shape[0], C.byref(m)) == 0
        assert m.value == len(want) and (rhj.pairs_to_numpy(out)[:m.value] == want).all(), (bits, rep)
    # a buffer that is too small: the count comes back, rc 1
    assert rhj.lib.rhj_join_keys_device(kR.data_ptr(), nR, kS.data_ptr(), nS, out.data_ptr(), 10, C.byref(m)) == 1 and m.value == len(want)


def test_row_id_width_is_speculated_and_a_wrong_guess_runs_again():
    """The 16-byte kernels of the two-pass partition are not launched until a join of the process has needed them: the first
    join with wide row ids is reported as an overflow by the sample and run again wide; later joins launch both widths."""
    import subprocess, sys, os
    code = r'''
import importlib, sys, numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle
o = Oracle()
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
rhj.set_bits(12)
R = o.generate(150000, 0, 0, 0.0, 5); S = o.generate(210000, 1, 150000, 0.0, 6)
wide = np.uint64(1) << np.uint64(45)
def run(R, S):
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), capacity=len(S))
    got = rhj.pairs_to_numpy(t); want = o.join(R, S, 12)
    assert m == len(want) and (got == want).all()
run(R, S)                                    # narrow: the 12-byte kernels alone
Rw = R.copy(); Rw["row_id"][:7] += wide
run(Rw, S)                                   # wide at the ends: the sample sees it, nothing was launched for it -> again, wide
run(R, S)                                    # narrow again, both widths launched from now on
Sm = S.copy(); Sm["row_id"][len(S) // 2] += wide
run(R, Sm)                                   # wide in the middle: pass 1 catches it
run(Rw, Sm)
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0 and b"ok" in res.stdout, res.stderr.decode()[-1500:]


@pytest.mark.parametrize("bits,nR,nS,kind,dom", [
    (3, 1000, 1, 4, 10), (7, 77777, 99999, 1, 50000), (9, 200000, 1000, 4, 1 << 20),
    (12, 2000000, 3000000, 1, 2000000), (12, 100000, 3000000, 2, 100000), (4, 50000, 60000, 4, 11),
    (4, 800000, 900000, 1, 800000), (2, 300000, 500000, 1, 300000),
    (14, 3000000, 2000000, 1, 3000000), (15, 2500000, 3500000, 1, 2500000), (13, 500000, 4000000, 2, 500000), (14, 100, 5000, 4, 17),
    (4, 300000, 350000, 1, 300000), (4, 350000, 300000, 1, 350000), (6, 400000, 400000, 4, 90000),
])
def test_random_joins_vs_oracle(rhj, oracle, bits, nR, nS, kind, dom):
    rhj.set_bits(bits)
    R = oracle.generate(nR, 0 if kind != 4 else 4, dom, 0.0, 5 + bits)
    S = oracle.generate(nS, kind, dom, 0.9, 6 + bits)
    want = oracle.join(R, S, bits)
    for path in ("fused", "fused_gather", "tiled32"):
        set_path(rhj, path)
        got = dev_join(rhj, R, S)
        assert len(got) == len(want) and (got == want).all(), path
    set_path(rhj, "fused")


def test_foreign_tag_hits_between_matches_on_the_gather_path(rhj, oracle):
    """16 M unique R keys probing 8 M S tuples whose keys come in pairs, on 9 radix bits: build sides of 15.6 K tuples are
    gathered (not LDS-resident), 4 M probe tuples have two matches each, and with 16-bit tags about a hundred of them meet a
    FOREIGN key's tag hit between their matches — the tuples whose overflow entries sit in the runs of the rounds they were
    found in (patch list of the fused kernel's streaming emit; before that the whole unit went to the index-walking emit).
    Bit for bit against the oracle."""
    rhj.set_bits(9)
    R = oracle.generate(16_000_000, 0, 0, 0.0, 71)
    rng = np.random.default_rng(72)
    keys = np.repeat(R["value"][:4_000_000], 2)
    rng.shuffle(keys)
    S = np.zeros(len(keys), dtype=R.dtype)
    S["value"] = keys
    S["row_id"] = np.arange(len(keys), dtype=np.uint64)
    want = oracle.join(R, S, 9)
    assert len(want) == 8_000_000
    set_path(rhj, "fused")
    got = dev_join(rhj, R, S)
    assert len(got) == len(want) and (got == want).all()


@pytest.mark.parametrize("bits,nR,nS,kind,dom,path", [
    (4, 3_000_000, 4_000_000, 1, 3_000_000, "lowradix"),      # S probes in most buckets
    (4, 4_000_000, 2_500_000, 1, 4_000_000, "lowradix"),      # R probes: the bigger side, Poisson matches per tuple
    (8, 12_000_000, 16_000_000, 1, 12_000_000, "lowradix"),
    (5, 2_500_000, 2_500_000, 4, 2_000_000, "lowradix"),      # duplicates on both sides (a few matches per tuple)
    (4, 2_000_000, 6_000_000, 2, 2_000_000, "lowradix"),      # Zipf: the hot key's pass-2 tiles are walked in several batches
    (6, 5_000_000, 5_000_000, 4, 1_500_000, None),            # heavy duplicates: tuples with more than 16 matches -> refused
])
def test_low_radix_path_matches_oracle(rhj, oracle, bits, nR, nS, kind, dom, path):
    """Joins on few radix bits whose buckets are beyond the LDS index (the reference ships 4 bits): run on r + k bits, emitted
    in the canonical order of the r bits by replaying pass 2 over the probe side (csrc/rhj_lowradix.hip.h) — bit for bit the
    oracle's result on r bits; what the path refuses comes out of the tiled path, equally exact."""
    rhj.set_bits(bits)
    set_path(rhj, "fused")
    R = oracle.generate(nR, 0 if kind != 4 else 4, dom, 0.0, 15 + bits)
    S = oracle.generate(nS, kind, dom, 0.9, 16 + bits)
    want = oracle.join(R, S, bits)
    got = dev_join(rhj, R, S)
    assert len(got) == len(want) and (got == want).all()
    if path:
        assert rhj.stats()["path"] == path
    rhj.lib.rhj_set_lowradix(0)
    try:
        got = dev_join(rhj, R, S)
        assert rhj.stats()["path"] != "lowradix" and len(got) == len(want) and (got == want).all()
    finally:
        rhj.lib.rhj_set_lowradix(1)


def test_low_radix_multi_match_probe_tuples_and_capacity(rhj, oracle):
    """6 M unique R keys probing 3 M S tuples whose keys come in pairs, on 4 radix bits: every matching R tuple has two
    matches, whose pairs the emit pass copies from the internal join's list; and the capacity protocol (count, too small a
    buffer) on this path."""
    rhj.set_bits(4)
    set_path(rhj, "fused")
    R = oracle.generate(6_000_000, 0, 0, 0.0, 81)
    keys = np.repeat(R["value"][:1_500_000], 2)
    np.random.default_rng(82).shuffle(keys)
    S = np.zeros(len(keys), dtype=R.dtype)
    S["value"] = keys
    S["row_id"] = np.arange(len(keys), dtype=np.uint64)
    want = oracle.join(R, S, 4)
    got = dev_join(rhj, R, S)
    assert rhj.stats()["path"] == "lowradix"
    assert len(got) == len(want) == 3_000_000 and (got == want).all()
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    _, m = rhj.join_device(dR, dS, count_only=True)
    assert m == len(want)
    t, m = rhj.join_device(dR, dS, capacity=1000)
    assert m == len(want) and (rhj.pairs_to_numpy(t) == want[:1000]).all()


def test_full_size_properties_c3_on_the_shipped_4_bits(rhj):
    """100M x 100M on the reference's own 4 radix bits (structs.h:11), canonical order, at full size through the
    size-independent properties — incl. the order inside the 16 buckets (probe side by bucket counts, input order)."""
    import bench
    w = bench.WORKLOADS["c3b4"]
    rhj.set_bits(w["bits"])
    R, S = bench.make_relations(w, rhj.dev, 98)
    t, m = rhj.join_device(R, S, capacity=w["nS"])
    assert rhj.stats()["path"] == "lowradix"
    bench.check_properties(R, S, t, m, w)


def test_capacity_overflow_reports_count(rhj, oracle):
    rhj.set_bits(4)
    R = oracle.generate(5000, 0, 0, 0.0, 1); S = oracle.generate(9000, 1, 5000, 0.0, 2)
    want = oracle.join(R, S, 4)
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), capacity=100)
    assert m == len(want) and t.shape[0] == 100
    assert (rhj.pairs_to_numpy(t) == want[:100]).all()


def test_full_size_properties_c2(rhj, oracle):
    """BASELINE config 2 (1M x 1M uniform FK, 8 bits) at full size against the oracle,
    plus the size-independent properties used at sizes the oracle cannot reach."""
    rhj.set_bits(8)
    R = oracle.generate(1000000, 0, 0, 0.0, 42); S = oracle.generate(1000000, 1, 1000000, 0.0, 43)
    got = dev_join(rhj, R, S)
    want = oracle.join(R, S, 8)
    assert (got == want).all()
    check_join_properties(R, S, got, 8)


def check_join_properties(R, S, pairs, bits):
    """Properties that pin a canonical-order FK join without an oracle run:
    every pair joins equal keys; every S row appears exactly once (FK, unique R);
    buckets ascend; inside a bucket the probe side's row ids ascend."""
    assert (R["value"][pairs["row_idR"]] == S["value"][pairs["row_idS"]]).all()
    assert len(pairs) == len(S) and len(np.unique(pairs["row_idS"])) == len(S)
    b = (S["value"][pairs["row_idS"]] & np.uint64((1 << bits) - 1)).astype(np.int64)
    assert (np.diff(b) >= 0).all()
    cR = np.bincount((R["value"] & np.uint64((1 << bits) - 1)).astype(np.int64), minlength=1 << bits)
    cS = np.bincount((S["value"] & np.uint64((1 << bits) - 1)).astype(np.int64), minlength=1 << bits)
    probe_is_R = (cR >= cS)[b]
    probe_ids = np.where(probe_is_R, pairs["row_idR"], pairs["row_idS"]).astype(np.int64)
    same = np.diff(b) == 0
    assert (np.diff(probe_ids)[same] >= 0).all()


def test_fuzz_shapes_bits_paths(rhj, oracle):
    """Randomised sweep over sizes (ragged, tiny, one side empty), key distributions, radix widths and
    paths in ONE process: also exercises the grow-only workspace across calls of varying size."""
    rng = np.random.RandomState(20181004)
    for it in range(60):
        bits = int(rng.choice([1, 3, 4, 6, 8, 9, 12, 14]))
        nR = int(rng.choice([0, 1, 2, 63, 64, 65, 1000, 4097, 30000, 200000]))
        nS = int(rng.choice([0, 1, 5, 64, 129, 2500, 8191, 50000, 300000]))
        kind = int(rng.choice([1, 2, 3, 4]))
        dom = int(rng.choice([1, 7, 1000, 100000, 1 << 40]))
        R = oracle.generate(nR, 4 if kind == 4 else 0 if kind != 3 else 3, dom, 0.0, 1000 + it)
        S = oracle.generate(nS, kind, min(dom, max(nR, 1)) if kind != 4 else dom, 0.8, 2000 + it)
        if rng.rand() < 0.3 and nR:
            R["row_id"] = R["row_id"] * np.uint64(0x9E3779B97F4A7C15) + np.uint64(it)
        path = PATHS[it % len(PATHS)]
        set_path(rhj, path)
        rhj.set_bits(bits)
        want = oracle.join(R, S, bits)
        got = dev_join(rhj, R, S) if (nR and nS) else rhj.RadixHashJoin(R, S)
        assert len(got) == len(want) and (got == want).all(), (it, bits, nR, nS, kind, dom, path)
    set_path(rhj, "fused")
    rhj.lib.rhj_release()          # drops the workspace; the next call must rebuild it
    R = oracle.generate(5000, 0, 0, 0.0, 1); S = oracle.generate(7000, 1, 5000, 0.0, 2)
    rhj.set_bits(4)
    assert (dev_join(rhj, R, S) == oracle.join(R, S, 4)).all()


def _full_size_both_kernels(rhj, name, seed, want_spec, fk=True):
    """One BASELINE workload at full size through the size-independent properties — twice: with the foreign-key speculation
    (k_join_spec; rhj_last_spec() says whether it held, so a silent fall-through to the ordinary kernel cannot pass) and with
    it switched off (k_join_fused); the two pair lists must be the same bytes."""
    import bench
    torch = rhj.torch
    w = bench.WORKLOADS[name]
    rhj.set_bits(w["bits"])
    R, S = bench.make_relations(w, rhj.dev, seed)
    try:
        rhj.lib.rhj_set_spec(1)
        t, m = rhj.join_device(R, S, capacity=w["nS"])
        assert rhj.lib.rhj_last_spec() == want_spec, "speculation: %d (0 not tried, 1 held, 2 failed)" % rhj.lib.rhj_last_spec()
        bench.check_properties(R, S, t, m, w, fk=fk)
        rhj.lib.rhj_set_spec(0)
        t0, m0 = rhj.join_device(R, S, capacity=w["nS"])
        assert rhj.lib.rhj_last_spec() == 0
        assert m0 == m and torch.equal(t[:m], t0[:m]), "the speculative and the ordinary kernel disagree"
        del t0
    finally:
        rhj.lib.rhj_set_spec(1)
    del R, S, t
    torch.cuda.empty_cache()


def test_full_size_properties_c3(rhj):
    """BASELINE config 3 (100M x 100M uniform FK, 12 radix bits) at full size through the
    size-independent properties (the oracle would need minutes): one pair per S tuple, equal keys,
    S row ids a permutation, buckets ascending, the canonical order inside the buckets — by k_join_spec (the
    speculation must have held) and by k_join_fused, bit for bit the same."""
    _full_size_both_kernels(rhj, "c3", 99, 1)


def test_full_size_properties_c3_where_the_speculation_fails(rhj):
    """The same 100M x 100M join with half of S without a partner (bench.py --workload c3half): the speculation is tried,
    fails its checks on the device and hands over inside the call; the result is the ordinary kernel's."""
    _full_size_both_kernels(rhj, "c3half", 98, 2, fk=False)


def test_first_call_in_fresh_process_takes_the_fallback(oracle):
    """A fresh process whose FIRST join has buckets too large for LDS (fused kernel launched on the
    upper-bound grid, plan rejects it, tiled path takes over) — workspace buffers are at their
    smallest here, which is what exposes out-of-bounds reads that a warmed-up workspace hides."""
    import subprocess, sys, os
    code = r'''
import importlib, sys, numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle
o = Oracle()
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
rhj.set_bits(4)
rhj.lib.rhj_set_lowradix(0)                  # (the low-radix path would take this join: second half below)
R = o.generate(1200000, 0, 0, 0.0, 5); S = o.generate(900000, 1, 1200000, 0.0, 6)
t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), capacity=len(S))
got = rhj.pairs_to_numpy(t); want = o.join(R, S, 4)
assert m == len(want) and (got == want).all()
assert rhj.stats()["hbm_units"] > 0          # 64-bit HBM tables were really used
print("ok")
'''
    code_lr = code.replace("rhj.lib.rhj_set_lowradix(0)", "pass").replace('assert rhj.stats()["hbm_units"] > 0', 'assert rhj.stats()["path"] == "lowradix"')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for c in (code, code_lr):                # ... and the same join as a fresh process's first call on the low-radix path
        res = subprocess.run([sys.executable, "-c", c], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert res.returncode == 0 and b"ok" in res.stdout, res.stderr.decode()[-1500:]


def test_full_size_properties_c4(rhj):
    """BASELINE config 4 (100M x 1B Zipf(0.9) FK, 14 radix bits: LDS-resident units, skewed buckets split
    into many units) at full size through the size-independent properties, by k_join_spec<resident> (held) and by
    k_join_fused<resident>, bit for bit the same."""
    free, _ = rhj.torch.cuda.mem_get_info()
    if free < 130 * (1 << 30):
        pytest.skip("needs ~100 GB of device memory")
    _full_size_both_kernels(rhj, "c4", 7, 1)


def test_above_2_31_tuples(rhj):
    """Maximum sizes: a probe side above 2^31 tuples (the reference's own limit is 2^31 - 1, SURVEY.md
    finding 9; this library's is 2^32 - 1) through the size-independent properties, and the refusal at 2^32."""
    import bench
    torch = rhj.torch
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~170 GB of device memory")
    w = dict(nR=1_000_000, nS=(1 << 31) + (1 << 22) + 12345, bits=12, dist="uniform")
    rhj.set_bits(w["bits"])
    R, S = bench.make_relations(w, rhj.dev, 11)
    t, m = rhj.join_device(R, S, capacity=w["nS"])
    bench.check_properties(R, S, t, m, w)
    # the pairs at the far end of the list, beyond 2^31
    tail = t[m - 4096:m]
    assert bool((R[tail[:, 0], 0] == S[tail[:, 1], 0]).all())
    assert bool(((S[tail[:, 1], 0] & 4095) == 4095).all())
    del t, tail
    torch.cuda.empty_cache()
    import ctypes as C
    mm = C.c_uint64(0)
    rc = rhj.lib.rhj_join_device(R.data_ptr(), 1 << 32, S.data_ptr(), 16, None, 0, C.byref(mm))
    assert rc == -2
    del R, S
    rhj.lib.rhj_release()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n", [1, 63, 127, 128, 129, 4095, 4096, 4097, 8191, 8192, 8193, 12289, 100003, 1 << 20])
def test_filter_sizes_and_densities(rhj, n):
    """The index-list pass works per pair of 4096-element tiles and switches from walking set bits to the
    coalesced round-by-round form above 1024 hits per pair: sizes around the tile / pair / round edges, hit
    densities on both sides of the switch (incl. none, all, exactly 1024 and 1025 per pair), direct and
    through a row-id vector, against numpy (Filter(): filter.c:92-190, ascending indices)."""
    torch = rhj.torch
    rng = np.random.RandomState(n)
    col = rng.randint(0, 1000, n).astype(np.uint64)
    cases = [("<", 0), ("<", 10), ("<", 124), ("<", 126), ("<", 500), ("<", 1000), (">", 998), ("=", 7)]
    dcol = torch.from_numpy(col.view(np.int64)).to(rhj.dev)
    for op, v in cases:
        want = np.nonzero(col < v if op == "<" else col > v if op == ">" else col == v)[0].astype(np.uint64)
        got = rhj.filter_device(dcol, op, v).cpu().numpy().view(np.uint64)
        assert len(got) == len(want) and (got == want).all(), (n, op, v)
    if n >= 8192:
        # exactly 1024 and 1025 hits in the first tile pair, nothing elsewhere
        for k in (1024, 1025):
            c2 = np.full(n, 5, dtype=np.uint64)
            pos = np.sort(rng.choice(8192, k, replace=False))
            c2[pos] = 1
            got = rhj.filter_device(torch.from_numpy(c2.view(np.int64)).to(rhj.dev), "<", 2).cpu().numpy().view(np.uint64)
            assert (got == pos.astype(np.uint64)).all(), (n, k)
    sel = rng.randint(0, n, max(1, n // 2)).astype(np.uint64)
    dsel = torch.from_numpy(sel.view(np.int64)).to(rhj.dev)
    for op, v in (("<", 10), ("<", 500)):
        want = np.nonzero(col[sel] < v)[0].astype(np.uint64)
        got = rhj.filter_device(dcol, op, v, dsel).cpu().numpy().view(np.uint64)
        assert len(got) == len(want) and (got == want).all(), (n, op, v, "sel")


def test_entry_points_from_several_host_threads(rhj, oracle):
    """One library context per process: the entry points serialise under a lock (rhj_internal.h), so callers on
    several host threads get the same pair lists and index lists as one caller.  (ctypes releases the GIL
    during the calls, so the four threads really are inside the library together.)"""
    import threading
    rhj.set_bits(12)
    work = []
    for t in range(4):
        R = oracle.generate(40000 + 1000 * t, 0, 30000, 0.0, 900 + t)
        S = oracle.generate(90000 + 5000 * t, 1, 30000, 0.9, 950 + t)
        col = (np.arange(300000, dtype=np.uint64) * np.uint64(2654435761 + t)) % np.uint64(1000)
        work.append((R, S, oracle.join(R, S, 12), col, np.nonzero(col < 17 + t)[0].astype(np.uint64), 17 + t))
    errors = []

    def run(t):
        R, S, want, col, want_ids, v = work[t]
        try:
            for it in range(12):
                got = rhj.RadixHashJoin(R, S)
                if len(got) != len(want) or not (got == want).all():
                    errors.append("thread %d join %d differs" % (t, it))
                    return
                ids = rhj.Filter([col], len(col), 0, "<", v)
                if len(ids) != len(want_ids) or not (np.asarray(ids, dtype=np.uint64) == want_ids).all():
                    errors.append("thread %d filter %d differs" % (t, it))
                    return
        except Exception as e:                         # noqa: BLE001 - reported below
            errors.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_threads_on_the_second_device_when_there_is_one(oracle):
    """HIP's current device is per thread: with the library on device 1, calls from threads that never touched HIP
    must still allocate and launch there (every entry point selects the library's device).  Needs two visible GPUs:
    skipped on a one-GPU box.  Own process: one library context per process."""
    import subprocess, sys, torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    code = r'''
import importlib, sys, threading
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from pyoracle import Oracle
mod = importlib.import_module("sigmod-2018_amd")
o = Oracle(); rhj = mod.RHJ(device=1); rhj.set_bits(12)
R = o.generate(50000, 0, 0, 0.0, 5); S = o.generate(120000, 1, 50000, 0.0, 6); want = o.join(R, S, 12)
bad = []
def run():
    for _ in range(6):
        got = rhj.RadixHashJoin(R, S)
        if len(got) != len(want) or not (got == want).all(): bad.append(1)
ts = [threading.Thread(target=run) for _ in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
try:
    mod.RHJ(device=0)
    bad.append("second context on another device was not refused")
except RuntimeError:
    pass
print("BAD" if bad else "OK", bad)
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip().startswith("OK"), (r.stdout[-500:], r.stderr[-1500:])


MSD_CHILD = r'''
import importlib, sys
import numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from pyoracle import Oracle, TUPLE
o = Oracle()
rhj = importlib.import_module("sigmod-2018_amd").RHJ(device=0)
for bits in (9, 12, 13, 14, 15):
    for n, kind, dom in ((4097, 4, 1 << 30), (700001, 1, 100000), (1200000, 2, 2000)):
        rel = o.generate(n, kind, dom, 0.9, 500 + bits)
        want, hist, psum = o.partition(rel, bits)
        out, h, p = rhj.partition_device(rhj.to_device(rel), bits)
        got = out.cpu().numpy().view(np.uint64).reshape(-1, 2).copy().view(TUPLE).reshape(-1)
        assert (h == hist).all() and (p == psum).all() and (got == want).all(), (n, bits)
    R, S = o.generate(600000, 0, 0, 0.0, 7), o.generate(900000, 2, 600000, 0.8, 8)
    rhj.set_bits(bits)
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S))
    assert (rhj.pairs_to_numpy(t)[:m] == o.join(R, S, bits)).all(), bits
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S), bucket_range=(3, (1 << bits) - 5))
    sel = lambda x: x[((x["value"] & np.uint64((1 << bits) - 1)) >= 3) & ((x["value"] & np.uint64((1 << bits) - 1)) < (1 << bits) - 5)]
    assert (rhj.pairs_to_numpy(t)[:m] == o.join(sel(R), sel(S), bits)).all(), ("ranged", bits)
print("ok")
'''


def test_two_pass_partition_with_pass_1_on_the_high_bits():
    """RHJ_MSD=1 (A/B knob): pass 1 of the two-pass partition takes the HIGH bits of the radix and pass 2 the low ones —
    same buckets, same order inside them.  Partition, join and ranged join against the oracle at 9..15 bits, in a process of its
    own (the knob is read when the library's context is made)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", MSD_CHILD], cwd=root, env=dict(os.environ, RHJ_MSD="1"), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0 and b"ok" in res.stdout, res.stderr.decode()[-3000:]
