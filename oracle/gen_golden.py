#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE's own compiled code (oracle/_ref).

Run in the build container only (needs /root/reference):
    make -C oracle all ref && python oracle/gen_golden.py

What is written is data, never reference text: seeds/parameters of synthetic
inputs, explicit small inputs, and for every case the reference's output as
(count, FNV-1a-64 of the ordered pair stream, wrap-around sums, first/last pairs).
Every case is produced by the THREADS=1 build (the authoritative mode, SURVEY.md
finding 4), cross-checked against the THREADS=4 build where the shipped
partitioner is correct, and against oracle/rhj_oracle.c; a disagreement aborts.

Also recorded: the `small` workload as it crosses the RadixHashJoin()/Filter()
boundary inside the reference engine (oracle/_ref/radixhash_dump_t1, whose
query.c is compiled with the two calls routed through ref_wrap.c's hooks), and
the workload's relation files re-encoded as a compressed .npz.
"""
import json
import lzma
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
SMALL = os.path.join(REF, "submission", "workloads", "small")
sys.path.insert(0, HERE)
from pyoracle import Oracle, Reference, TUPLE, PAIR  # noqa: E402

o = Oracle()
M64 = (1 << 64) - 1


def digest(pairs):
    pairs = np.ascontiguousarray(pairs, dtype=PAIR)
    return {
        "matches": int(len(pairs)),
        "fnv": "%016x" % o.fnv(pairs),
        "sumR": int(pairs["row_idR"].sum(dtype=np.uint64)) if len(pairs) else 0,
        "sumS": int(pairs["row_idS"].sum(dtype=np.uint64)) if len(pairs) else 0,
        "head": [[int(a), int(b)] for a, b in pairs[:8].tolist()],
        "tail": [[int(a), int(b)] for a, b in pairs[-8:].tolist()],
    }


def gen(spec):
    return o.generate(spec["n"], spec["kind"], spec.get("domain", 0), spec.get("theta", 0.0), spec["seed"])


def run_case(R, S, bits, t4_safe):
    ref1 = Reference(bits, 1)
    a, info1 = ref1.join(R, S, with_info=True)
    b = o.join(R, S, bits)
    assert len(a) == len(b) and (a == b).all(), "oracle != reference(T=1)"
    rec = digest(a)
    rec["null_t1"] = info1["null"]
    if os.path.exists(os.path.join(HERE, "_ref", "libref_n%d_t4.so" % bits)):
        c, info4 = Reference(bits, 4).join(R, S, with_info=True)
        rec["null_t4"] = info4["null"]
        same = len(a) == len(c) and (a == c).all()
        if t4_safe:
            assert same, "reference(T=4) != reference(T=1) on a case marked safe"
        rec["t4_equal"] = bool(same)
    return rec


def synthetic_cases():
    U, FK, ZIPF, DENSE, DUP = 0, 1, 2, 3, 4
    cases = []

    def add(name, bits, R, S, t4_safe=True):
        cases.append({"name": name, "bits": bits, "R": R, "S": S, "t4_safe": t4_safe})

    for bits in (4, 8, 12):
        add("fk_64k_b%d" % bits, bits, dict(n=65536, kind=U, seed=42), dict(n=65536, kind=FK, domain=65536, seed=43))
        add("fk_ragged_b%d" % bits, bits, dict(n=30011, kind=U, seed=7), dict(n=99991, kind=FK, domain=30011, seed=8))
        add("fk_rbig_b%d" % bits, bits, dict(n=120000, kind=U, seed=9), dict(n=17, kind=FK, domain=120000, seed=10))
        add("dense_b%d" % bits, bits, dict(n=50000, kind=DENSE, seed=1), dict(n=50000, kind=DENSE, seed=1))
        add("dups_b%d" % bits, bits, dict(n=20000, kind=DUP, domain=3000, seed=21), dict(n=25000, kind=DUP, domain=3000, seed=22))
        add("heavy_dups_b%d" % bits, bits, dict(n=4000, kind=DUP, domain=37, seed=23), dict(n=6000, kind=DUP, domain=41, seed=24),
            t4_safe=(bits == 4))
        add("zipf_b%d" % bits, bits, dict(n=50000, kind=U, seed=31), dict(n=400000, kind=ZIPF, domain=50000, theta=0.9, seed=44))
        add("disjoint_b%d" % bits, bits, dict(n=5000, kind=DENSE, seed=51), dict(n=6000, kind=U, seed=52))
        add("same_keyset_b%d" % bits, bits, dict(n=5000, kind=U, seed=51), dict(n=5000, kind=U, seed=52))
        add("one_one_b%d" % bits, bits, dict(n=1, kind=DENSE, seed=1), dict(n=1, kind=DENSE, seed=1))
    add("fk_1m_b8", 8, dict(n=1000000, kind=U, seed=42), dict(n=1000000, kind=FK, domain=1000000, seed=43))
    add("fk_big_bucket_b4", 4, dict(n=1500000, kind=U, seed=61), dict(n=1200000, kind=FK, domain=1500000, seed=62))
    add("fk_200k_b1", 1, dict(n=200000, kind=U, seed=71), dict(n=150000, kind=FK, domain=200000, seed=72))
    add("fk_200k_b6", 6, dict(n=200000, kind=U, seed=73), dict(n=250000, kind=FK, domain=200000, seed=74))
    add("fk_200k_b10", 10, dict(n=200000, kind=U, seed=75), dict(n=250000, kind=FK, domain=200000, seed=76))
    add("fk_300k_b14", 14, dict(n=300000, kind=U, seed=77), dict(n=350000, kind=FK, domain=300000, seed=78))
    add("dups_300k_b15", 15, dict(n=300000, kind=DUP, domain=90000, seed=79), dict(n=280000, kind=DUP, domain=90000, seed=80))
    add("all_same_key_b4", 4, dict(n=700, kind=DUP, domain=1, seed=1), dict(n=900, kind=DUP, domain=1, seed=2), t4_safe=False)
    out = []
    for c in cases:
        R, S = gen(c["R"]), gen(c["S"])
        rec = run_case(R, S, c["bits"], c["t4_safe"])
        c.update(rec)
        out.append(c)
        print("  synthetic %-22s matches=%d fnv=%s" % (c["name"], c["matches"], c["fnv"]))
    return out


def edge_cases():
    """SURVEY.md A.2 plus a few more; inputs explicit, expected pairs in full."""
    two63, maxu = 1 << 63, (1 << 64) - 1
    raw = [
        ("disjoint_buckets", [2, 4, 6, 8], [1, 3, 5, 7]),
        ("same_bucket_no_match", [16, 32, 48], [0, 64, 80]),
        ("empty_R", [], [1, 3, 5, 7]),
        ("empty_S", [1, 3, 5, 7], []),
        ("both_empty", [], []),
        ("dups_probe_R", [5, 5, 21, 5], [5, 37, 5]),
        ("dups_probe_S", [5, 5], [5, 5, 5, 21]),
        ("u64_extremes", [maxu, two63, 7], [7, two63, maxu]),
        ("tie_sizes_probe_R", [3, 19, 35], [35, 3, 19]),
        ("zero_key", [0, 0, 16], [0, 32, 0, 16]),
        ("single_match", [9], [9]),
    ]
    out = []
    for name, rv, sv in raw:
        R = np.zeros(len(rv), dtype=TUPLE); S = np.zeros(len(sv), dtype=TUPLE)
        R["value"], R["row_id"] = np.array(rv, dtype=np.uint64), np.arange(len(rv), dtype=np.uint64)
        S["value"], S["row_id"] = np.array(sv, dtype=np.uint64), np.arange(len(sv), dtype=np.uint64)
        for bits in (4, 8):
            a, i1 = Reference(bits, 1).join(R, S, with_info=True)
            c, i4 = Reference(bits, 4).join(R, S, with_info=True)
            b = o.join(R, S, bits)
            assert len(a) == len(b) == len(c) and (a == b).all() and (a == c).all(), name
            out.append({"name": name, "bits": bits, "R": [str(v) for v in rv], "S": [str(v) for v in sv],
                        "pairs": [[int(x), int(y)] for x, y in a.tolist()],
                        "null_t1": i1["null"], "null_t4": i4["null"]})
    # arbitrary (non-identity) row ids must pass through untouched
    R = o.generate(3000, 4, 500, 0, 81); S = o.generate(2000, 4, 500, 0, 82)
    R["row_id"] = (R["row_id"] * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0xABCDEF)
    S["row_id"] = np.uint64(M64) - S["row_id"] * np.uint64(977)
    a = Reference(4, 1).join(R, S); b = o.join(R, S, 4)
    assert (a == b).all()
    rec = digest(a)
    rec.update({"name": "arbitrary_row_ids", "bits": 4, "seedR": 81, "seedS": 82})
    return out, rec


def last_bucket_skew():
    """Finding 4: the shipped THREADS=4 partitioner is wrong when the last active
    bucket holds more than 1/THREADS of an input.  Recorded so the test documents
    which mode is the oracle."""
    rng = np.random.RandomState(5)
    vals = (rng.randint(0, 500, 1000).astype(np.uint64) << np.uint64(4)) | rng.choice([0, 1], 1000, p=[0.4, 0.6]).astype(np.uint64)
    R = np.zeros(1000, dtype=TUPLE); R["value"] = vals; R["row_id"] = np.arange(1000)
    perm = rng.permutation(1000)
    S = np.zeros(1000, dtype=TUPLE); S["value"] = vals[perm]; S["row_id"] = np.arange(1000)
    a = Reference(4, 1).join(R, S); c = Reference(4, 4).join(R, S); b = o.join(R, S, 4)
    assert (a == b).all()
    return {"values_R": [int(v) for v in vals], "perm": [int(p) for p in perm],
            "t1": digest(a), "t4_matches": int(len(c)), "t4_equal": bool(len(a) == len(c) and (a == c).all())}


def filter_cases():
    out = []
    for name, n, dom, seed in [("f_small", 5000, 1000, 5), ("f_mid", 300000, 1 << 17, 6), ("f_wide", 70000, 1 << 40, 7)]:
        col = o.generate(n, 4, dom, 0, seed)["value"]
        sel = o.generate(n // 2 + 3, 4, n, 0, seed + 100)["value"]
        for op in "<>=":
            for v in (0, int(col[n // 3] & 0x7FFFFFFF), -1, dom // 2 if dom < (1 << 31) else 12345, 2147483647):
                for mode, s in (("direct", None), ("indirect", sel)):
                    y, null = Reference(4, 1).filter(col, op, v, s)
                    x = o.filter(col, op, v, s)
                    assert len(x) == len(y) and (x == y).all()
                    out.append({"name": name, "n": n, "domain": dom, "seed": seed, "mode": mode, "op": op, "value": v,
                                "hits": int(len(y)), "null": null, "fnv": "%016x" % o.fnv(y),
                                "sum": int(y.sum(dtype=np.uint64)) if len(y) else 0})
    return out


def small_workload():
    """Run the reference engine on `small` with the boundary hooks and parse the dump."""
    dump = tempfile.NamedTemporaryFile(prefix="rhj_small_", suffix=".bin", delete=False).name
    init = open(os.path.join(SMALL, "small.init")).read()
    work = open(os.path.join(SMALL, "small.work")).read()
    env = dict(os.environ, REF_DUMP=dump)
    res = subprocess.run([os.path.join(HERE, "_ref", "radixhash_dump_t1")], input=(init + "Done\n" + work).encode(),
                         cwd=SMALL, env=env, stdout=subprocess.PIPE, check=True)
    golden = open(os.path.join(SMALL, "small.result")).read()
    assert res.stdout.decode() == golden, "reference engine does not reproduce small.result"

    rels = {}
    for i in range(14):
        raw = np.fromfile(os.path.join(SMALL, "r%d" % i), dtype="<u8")
        nt, nc = int(raw[0]), int(raw[1])
        cols = raw[2:].reshape(nc, nt)
        assert cols.max() < (1 << 32)
        rels["r%d" % i] = cols.astype(np.uint32)

    joins, filters, join_inputs = [], [], {}
    with open(dump, "rb") as f:
        data = memoryview(f.read())
    os.unlink(dump)
    at = 0

    def u64():
        nonlocal at
        v = struct.unpack_from("<Q", data, at)[0]; at += 8
        return v

    def arr(dtype, n):
        nonlocal at
        a = np.frombuffer(data, dtype=dtype, count=n, offset=at); at += a.nbytes
        return a

    while at < len(data):
        magic = u64()
        if magic == 0x4a4f494e:
            nR, nS, null, total = u64(), u64(), u64(), u64()
            R, S, P = arr(TUPLE, nR), arr(TUPLE, nS), arr(PAIR, total)
            assert (R["row_id"] == np.arange(nR, dtype=np.uint64)).all() and (S["row_id"] == np.arange(nS, dtype=np.uint64)).all()
            b = o.join(R, S, 4)
            assert len(b) == total and (b == P).all(), "oracle != reference on small join %d" % len(joins)
            rec = digest(P)
            rec.update({"idx": len(joins), "nR": nR, "nS": nS, "null": bool(null),
                        "fnvR": "%016x" % o.fnv(R["value"]), "fnvS": "%016x" % o.fnv(S["value"])})
            joins.append(rec)
            join_inputs["j%d_R" % rec["idx"]] = R["value"].astype(np.uint32)
            join_inputs["j%d_S" % rec["idx"]] = S["value"].astype(np.uint32)
        elif magic == 0x46494c54:
            rel, col, value, op, rows, sel_len, null, total = (u64() for _ in range(8))
            value = value - (1 << 64) if value >= (1 << 63) else value
            sel = arr("<u8", sel_len) if sel_len != M64 else None
            ids = arr("<u8", total)
            x = o.filter(rels["r%d" % rel][col].astype(np.uint64), chr(op), value, sel)
            assert len(x) == total and (x == ids).all(), "oracle != reference on small filter %d" % len(filters)
            assert sel is None, "small has no indirect filter (would need the sel vector as a fixture)"
            filters.append({"idx": len(filters), "rel": rel, "col": col, "op": chr(op), "value": value, "rows": rows,
                            "hits": total, "null": bool(null), "fnv": "%016x" % o.fnv(ids),
                            "sum": int(ids.sum(dtype=np.uint64)) if total else 0})
        else:
            raise RuntimeError("bad dump magic %x at %d" % (magic, at))
    print("  small: %d joins (%d tuples in, %d pairs out), %d filters" % (
        len(joins), sum(j["nR"] + j["nS"] for j in joins), sum(j["matches"] for j in joins), len(filters)))
    return rels, joins, filters, join_inputs


def save_npz_xz(path, arrays):
    """np.savez into memory, xz on top (the values are < 2^17: xz does far better than deflate)."""
    import io
    bio = io.BytesIO()
    np.savez(bio, **arrays)
    with lzma.open(path, "wb", preset=9 | lzma.PRESET_EXTREME) as f:
        f.write(bio.getvalue())
    return os.path.getsize(path)


def main():
    os.makedirs(GOLD, exist_ok=True)
    print("synthetic joins"); syn = synthetic_cases()
    print("edge cases"); edges, arb = edge_cases()
    skew = last_bucket_skew()
    print("filters"); filt = filter_cases()
    print("small workload"); rels, joins, sfilters, join_inputs = small_workload()

    json.dump({"generator": "oracle/gen_golden.py", "cases": syn, "arbitrary_row_ids": arb}, open(os.path.join(GOLD, "synthetic_joins.json"), "w"), indent=1)
    json.dump({"cases": edges, "last_bucket_skew": skew}, open(os.path.join(GOLD, "edge_cases.json"), "w"), indent=1)
    json.dump({"cases": filt}, open(os.path.join(GOLD, "filters.json"), "w"), indent=1)
    json.dump({"joins": joins, "filters": sfilters,
               "result_lines": open(os.path.join(SMALL, "small.result")).read().splitlines(),
               "work_lines": open(os.path.join(SMALL, "small.work")).read().splitlines()},
              open(os.path.join(GOLD, "small_boundary.json"), "w"), indent=1)
    s1 = save_npz_xz(os.path.join(GOLD, "small_relations.npz.xz"), rels)
    s2 = save_npz_xz(os.path.join(GOLD, "small_join_inputs.npz.xz"), join_inputs)
    print("wrote fixtures: relations %.2f MB, join inputs %.2f MB" % (s1 / 1e6, s2 / 1e6))


if __name__ == "__main__":
    main()
