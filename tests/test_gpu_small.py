"""The small-join path (csrc/rhj_small.hip.h: tile histogram, self-scanning scatter with the plan riding along,
fused join — three launches, no host memset, no read-back copy) against the oracle, bit-exact, and against
the path of separate launches on the same inputs.  Called through the C-ABI of include/rhj.h."""
import ctypes as C
import importlib
import zlib

import numpy as np
import pytest

from helpers import make_rel
from pyoracle import PAIR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rhj():
    mod = importlib.import_module("sigmod-2018_amd")
    r = mod.RHJ(device=0)
    yield r
    r.lib.rhj_set_small(1)
    r.lib.rhj_set_timing(2)
    r.lib.rhj_set_resident(1)


def dev_join(rhj, R, S):
    t, m = rhj.join_device(rhj.to_device(R), rhj.to_device(S))
    out = rhj.pairs_to_numpy(t)
    assert m == len(out)
    return out


def same(got, want, what):
    want = np.ascontiguousarray(want, dtype=PAIR)
    assert len(got) == len(want), "%s: %d pairs, oracle %d" % (what, len(got), len(want))
    assert np.array_equal(got["row_idR"], want["row_idR"]) and np.array_equal(got["row_idS"], want["row_idS"]), what


def rel(rng, n, dom, row_ids=None):
    return make_rel(rng.integers(0, dom, size=n, dtype=np.uint64), row_ids)


CASES = [
    # name, nR, nS, key domain, bits
    ("one_tile_each", 3000, 2500, 4000, 4),
    ("tile_boundary", 8192, 8192, 9000, 8),
    ("tile_boundary_plus_one", 8193, 16385, 9000, 6),
    ("ragged", 100003, 49999, 70000, 8),
    ("one_bit", 20000, 30000, 25000, 1),
    ("two_bits", 20000, 30000, 25000, 2),
    ("three_bits", 40000, 10000, 25000, 3),
    ("seven_bits", 250000, 250000, 300000, 7),
    ("c2_like", 1000000, 1000000, 1000000, 8),
    ("many_tiles", 3000000, 1200000, 2000000, 8),
    ("tiny", 7, 5, 4, 4),
    ("single_tuple_each", 1, 1, 1, 8),
    ("heavy_duplicates", 60000, 60000, 300, 8),           # ~200 matches per probe tuple: overflow stash and index walk
    ("one_key", 3000, 2000, 1, 4),                         # one bucket, every tuple matches every tuple
]


@pytest.mark.parametrize("name,nR,nS,dom,bits", CASES, ids=[c[0] for c in CASES])
def test_small_path_matches_oracle(rhj, oracle, name, nR, nS, dom, bits):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    R, S = rel(rng, nR, dom), rel(rng, nS, dom)
    rhj.set_bits(bits)
    rhj.lib.rhj_set_small(1)
    got = dev_join(rhj, R, S)
    st = rhj.stats()
    same(got, oracle.join(R, S, bits), name)
    # the path taken: small unless some bucket's build side is beyond the LDS index (then the tiled path finishes
    # the join on the small path's partition)
    assert st["path"] in ("small", "tiled"), st["path"]
    if name in ("c2_like", "ragged", "seven_bits", "tile_boundary", "tiny"):
        assert st["path"] == "small"


def test_small_path_equals_separate_launches(rhj):
    rng = np.random.default_rng(77)
    R, S = rel(rng, 700000, 500000), rel(rng, 900000, 500000)
    rhj.set_bits(8)
    rhj.lib.rhj_set_small(1)
    a = dev_join(rhj, R, S)
    assert rhj.stats()["path"] == "small"
    rhj.lib.rhj_set_small(0)
    b = dev_join(rhj, R, S)
    assert rhj.stats()["path"] == "fused"
    rhj.lib.rhj_set_small(1)
    assert np.array_equal(a, b)


def test_small_path_gather_variant(rhj, oracle):
    """build tuples gathered from memory instead of LDS-resident (rhj_set_resident(0))"""
    rng = np.random.default_rng(5)
    R, S = rel(rng, 300000, 200000), rel(rng, 200000, 200000)
    rhj.set_bits(6)
    rhj.lib.rhj_set_resident(0)
    got = dev_join(rhj, R, S)
    rhj.lib.rhj_set_resident(1)
    assert rhj.stats()["path"] == "small"
    same(got, oracle.join(R, S, 6), "gather variant")


def test_small_path_wide_row_ids(rhj, oracle):
    """one pass keeps 16-byte tuples: row ids above 2^32 need no second run"""
    rng = np.random.default_rng(9)
    nR, nS = 50000, 70000
    R = rel(rng, nR, 40000, rng.integers(1 << 40, 1 << 63, size=nR, dtype=np.uint64))
    S = rel(rng, nS, 40000, rng.integers(1 << 33, 1 << 62, size=nS, dtype=np.uint64))
    rhj.set_bits(5)
    got = dev_join(rhj, R, S)
    assert rhj.stats()["path"] == "small"
    same(got, oracle.join(R, S, 5), "wide row ids")


def test_small_partition_falls_through_to_the_tiled_path(rhj, oracle):
    """a bucket whose build side is beyond the LDS index: the small path's partition and histograms are kept,
    the plan is made again for the tiled path"""
    rng = np.random.default_rng(11)
    R, S = rel(rng, 400000, 1 << 20), rel(rng, 400000, 1 << 20)
    rhj.set_bits(1)                                    # 200 K tuples per bucket
    want = oracle.join(R, S, 1)
    rhj.lib.rhj_set_lowradix(0)                        # (by default such a join runs on finer buckets: next lines)
    try:
        got = dev_join(rhj, R, S)
        assert rhj.stats()["path"] == "tiled"
        same(got, want, "tiled after small partition")
    finally:
        rhj.lib.rhj_set_lowradix(1)
    got = dev_join(rhj, R, S)                          # 1 + 4 bits internally, emitted in the order of the 1 bit
    assert rhj.stats()["path"] == "lowradix"
    same(got, want, "low-radix path on one radix bit")


def test_small_path_capacity_and_count_only(rhj, oracle):
    rng = np.random.default_rng(13)
    R, S = rel(rng, 20000, 500), rel(rng, 30000, 500)
    rhj.set_bits(4)
    want = oracle.join(R, S, 4)
    dR, dS = rhj.to_device(R), rhj.to_device(S)
    _, m = rhj.join_device(dR, dS, count_only=True)
    assert m == len(want)
    assert rhj.stats()["path"] == "small"
    # an output buffer that is too small: the count is still exact, the pairs that fit are the first ones, rc = 1
    cap = len(want) // 3
    out = rhj.torch.zeros((cap + 16, 2), dtype=rhj.torch.int64, device=rhj.dev)
    mm = C.c_uint64(0)
    rc = rhj.lib.rhj_join_device(dR.data_ptr(), len(R), dS.data_ptr(), len(S), out.data_ptr(), cap, C.byref(mm))
    assert rc == 1 and mm.value == len(want)
    got = rhj.pairs_to_numpy(out[:cap])
    same(got, want[:cap], "truncated output")
    assert int(out[cap:].abs().sum().item()) == 0      # nothing behind the capacity was touched


def test_small_path_back_to_back_joins_of_changing_size(rhj, oracle):
    """the join kernel's ticket / status words are cleared by the histogram launch, not by the host: joins of
    different unit counts one after the other must not see each other's words"""
    rng = np.random.default_rng(17)
    for n, bits in ((900000, 8), (1200, 4), (300000, 8), (50, 2), (700000, 7), (1200, 8)):
        R, S = rel(rng, n, max(n // 2, 2)), rel(rng, n + 17, max(n // 2, 2))
        rhj.set_bits(bits)
        got = dev_join(rhj, R, S)
        same(got, oracle.join(R, S, bits), "n=%d bits=%d" % (n, bits))


@pytest.mark.parametrize("level", [0, 1, 2])
def test_timing_levels(rhj, oracle, level):
    rng = np.random.default_rng(19)
    R, S = rel(rng, 100000, 80000), rel(rng, 100000, 80000)
    rhj.set_bits(8)
    rhj.lib.rhj_set_timing(level)
    got = dev_join(rhj, R, S)
    st = rhj.stats()
    rhj.lib.rhj_set_timing(2)
    same(got, oracle.join(R, S, 8), "timing %d" % level)
    assert st["path"] == "small"
    assert (st["ms_total"] > 0) == (level >= 1)
    assert (st["ms_probe"] > 0) == (level >= 2)


def test_host_abi_takes_the_small_path(rhj, oracle):
    rng = np.random.default_rng(23)
    R, S = rel(rng, 30000, 20000), rel(rng, 25000, 20000)
    rhj.set_bits(4)
    got = rhj.RadixHashJoin(R, S)
    assert rhj.stats()["path"] == "small"
    same(np.ascontiguousarray(got, dtype=PAIR), oracle.join(R, S, 4), "RadixHashJoin()")


def test_tiny_buckets_take_the_tiled_path_unless_told_otherwise(rhj, oracle):
    """4096+ buckets of a few tuples each: a fused unit per bucket has a fixed cost, the tiled path is chosen
    (rhj_set_fused(1) = automatic); rhj_set_fused(2) keeps the fused path; results identical."""
    rng = np.random.default_rng(29)
    R, S = rel(rng, 120000, 1 << 40), rel(rng, 90000, 1 << 40)
    S["value"][:60000] = R["value"][rng.integers(0, len(R), size=60000)]
    rhj.set_bits(13)
    want = oracle.join(R, S, 13)
    try:
        rhj.lib.rhj_set_fused(1)
        a = dev_join(rhj, R, S)
        assert rhj.stats()["path"] == "tiled"
        rhj.lib.rhj_set_fused(2)
        b = dev_join(rhj, R, S)
        assert rhj.stats()["path"] == "fused"
    finally:
        rhj.lib.rhj_set_fused(1)
    same(a, want, "tiny buckets, automatic")
    same(b, want, "tiny buckets, fused kept")


@pytest.mark.parametrize("bits,nR,nS,dom,path", [
    (4, 200_000, 200_000, 2_000, "small"),        # 100 matches per probe tuple on build sides of 12.5 K: gathered, beyond the overflow stash
    (9, 4_200_000, 4_200_000, 230_000, "fused"),  # 18 per tuple on gathered build sides through the two-pass partition (12-byte tuples)
    (2, 20_000, 5_000, 40, "small"),              # resident build sides, runs of 125 entries
])
def test_more_matches_than_the_overflow_stash_describes(rhj, oracle, bits, nR, nS, dom, path):
    """More than 16 matches per probe tuple: on a gathered build side the unit's emit pass walks the index again, on a
    resident one the run of the key's entries is emitted output-centrically: same pairs, same order as the oracle."""
    rng = np.random.default_rng(bits * 1000 + dom)
    R, S = rel(rng, nR, dom), rel(rng, nS, dom)
    rhj.set_bits(bits)
    got = dev_join(rhj, R, S)
    assert rhj.stats()["path"] == path
    same(got, oracle.join(R, S, bits), "bits=%d" % bits)
