"""CPU: the oracle (oracle/rhj_oracle.c) against the committed golden vectors, which
were produced by the reference's own compiled code (oracle/gen_golden.py), against
the reference's golden file small.result where the path reaches it, and against an
independent numpy statement of the canonical order."""
import numpy as np
import pytest

import helpers
from helpers import assert_digest, make_rel, spec_join
from pyoracle import PAIR


def test_synthetic_joins_match_reference_digests(oracle, golden):
    for c in golden.synthetic["cases"]:
        R, S = golden.gen(c["R"]), golden.gen(c["S"])
        assert_digest(oracle, oracle.join(R, S, c["bits"]), c, c["name"])


def test_edge_cases_full_pairs(oracle, golden):
    for c in golden.edges["cases"]:
        R = make_rel([int(v) for v in c["R"]]); S = make_rel([int(v) for v in c["S"]])
        got = oracle.join(R, S, c["bits"]).tolist()
        assert [list(map(int, p)) for p in got] == c["pairs"], c["name"]
        # SURVEY.md finding 5: zero matches -> NULL (THREADS 1) vs empty head (THREADS 4),
        # an empty input -> NULL in both
        if len(c["R"]) == 0 or len(c["S"]) == 0:
            assert c["null_t1"] and c["null_t4"]
        elif not c["pairs"]:
            assert c["null_t1"] and not c["null_t4"]


def test_arbitrary_row_ids_pass_through(oracle, golden):
    R, S, rec = golden.arbitrary_row_id_inputs()
    assert_digest(oracle, oracle.join(R, S, 4), rec, "arbitrary_row_ids")


def test_last_bucket_skew_documents_the_oracle_mode(oracle, golden):
    k = golden.edges["last_bucket_skew"]
    vals = np.array(k["values_R"], dtype=np.uint64)
    R = make_rel(vals); S = make_rel(vals[np.array(k["perm"])])
    assert_digest(oracle, oracle.join(R, S, 4), k["t1"], "skew")
    assert not k["t4_equal"] and k["t4_matches"] != k["t1"]["matches"]   # the shipped threaded partitioner is wrong here


def test_filters_match_reference_digests(oracle, golden):
    for c in golden.filters["cases"]:
        col, sel = golden.filter_inputs(c)
        ids = oracle.filter(col, c["op"], c["value"], sel)
        assert len(ids) == c["hits"] and "%016x" % oracle.fnv(ids) == c["fnv"], c
        assert c["null"] == (c["hits"] == 0)


def test_small_workload_boundary_joins(oracle, golden):
    joins = golden.small["joins"]
    assert len(joins) == 88 and sum(j["matches"] for j in joins) == 19615980
    for j in joins:
        R, S = golden.small_join(j["idx"])
        assert len(R) == j["nR"] and len(S) == j["nS"]
        assert_digest(oracle, oracle.join(R, S, 4), j, "small join %d" % j["idx"])


def test_small_workload_boundary_filters(oracle, golden):
    fl = golden.small["filters"]
    assert len(fl) == 50 and sum(f["hits"] for f in fl) == 303919
    for f in fl:
        col = golden.small_relations["r%d" % f["rel"]][f["col"]].astype(np.uint64)
        ids = oracle.filter(col, f["op"], f["value"])
        assert len(ids) == f["hits"] and "%016x" % oracle.fnv(ids) == f["fnv"]


def test_small_single_join_checksums_reach_small_result(oracle, golden):
    """'0 1|0.0=1.0|0.0 1.1' -> 494 pairs -> '1141020 2134690', through
    nothing but this path (no filter, one join)."""
    rel = golden.small_relations
    r0, r1 = rel["r0"].astype(np.uint64), rel["r1"].astype(np.uint64)
    pairs = oracle.join(make_rel(r0[0]), make_rel(r1[0]), 4)
    assert len(pairs) == 494
    assert int(r0[0][pairs["row_idR"]].sum()) == 1141020 and int(r1[1][pairs["row_idS"]].sum()) == 2134690
    r3 = rel["r3"].astype(np.uint64)
    pairs = oracle.join(make_rel(r3[2]), make_rel(r0[0]), 4)
    assert len(pairs) == 23038
    assert int(r0[2][pairs["row_idS"]].sum()) == 103122577 and int(r3[1][pairs["row_idR"]].sum()) == 132923020


@pytest.mark.parametrize("bits", [1, 4, 7, 12])
def test_oracle_equals_independent_spec(oracle, bits):
    R = oracle.generate(900, 4, 150, 0, 3 + bits)
    S = oracle.generate(1100, 4, 150, 0, 4 + bits)
    a, b = oracle.join(R, S, bits), spec_join(R, S, bits)
    assert len(a) == len(b) and (a == b).all()


def test_partition_is_stable_counting_sort(oracle):
    rel = oracle.generate(5000, 4, 1 << 20, 0, 9)
    for bits in (1, 4, 8, 12):
        out, hist, psum = oracle.partition(rel, bits)
        d = (rel["value"] & np.uint64((1 << bits) - 1)).astype(np.int64)
        order = np.argsort(d, kind="stable")
        assert (out == rel[order]).all()
        assert (hist == np.bincount(d, minlength=1 << bits)).all()
        starts = np.concatenate([[0], np.cumsum(hist)[:-1]]).astype(np.int64)
        assert (psum[hist > 0] == starts[hist > 0]).all() and (psum[hist == 0] == -1).all()


def test_find_next_prime_quirk(oracle):
    # rhjoin.c:333 stops the trial loop at i*i < num, so odd squares pass as prime
    assert [oracle.next_prime(n) for n in (8, 9, 24, 25, 49, 2, 1, 15)] == [9, 9, 25, 25, 49, 3, 1, 17]
