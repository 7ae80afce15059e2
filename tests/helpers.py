"""Shared helpers of the test-suite: golden-fixture loading and input regeneration."""
import io
import json
import lzma
import os

import numpy as np

from pyoracle import Oracle, TUPLE, PAIR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M64 = (1 << 64) - 1


def load_npz_xz(name):
    with lzma.open(os.path.join(GOLD, name), "rb") as f:
        return dict(np.load(io.BytesIO(f.read())))


def make_rel(values, row_ids=None):
    values = np.asarray(values, dtype=np.uint64)
    rel = np.zeros(len(values), dtype=TUPLE)
    rel["value"] = values
    rel["row_id"] = np.arange(len(values), dtype=np.uint64) if row_ids is None else row_ids
    return rel


def digest(o, pairs):
    pairs = np.ascontiguousarray(pairs, dtype=PAIR)
    return {
        "matches": int(len(pairs)),
        "fnv": "%016x" % o.fnv(pairs),
        "sumR": int(pairs["row_idR"].sum(dtype=np.uint64)) if len(pairs) else 0,
        "sumS": int(pairs["row_idS"].sum(dtype=np.uint64)) if len(pairs) else 0,
        "head": [[int(a), int(b)] for a, b in pairs[:8].tolist()],
        "tail": [[int(a), int(b)] for a, b in pairs[-8:].tolist()],
    }


DIGEST_KEYS = ("matches", "fnv", "sumR", "sumS", "head", "tail")


def assert_digest(o, pairs, rec, what=""):
    got = digest(o, pairs)
    for k in DIGEST_KEYS:
        assert got[k] == rec[k], "%s: %s differs: got %r, golden %r" % (what, k, got[k], rec[k])


class Golden:
    def __init__(self):
        self.o = Oracle()
        self.synthetic = json.load(open(os.path.join(GOLD, "synthetic_joins.json")))
        self.edges = json.load(open(os.path.join(GOLD, "edge_cases.json")))
        self.filters = json.load(open(os.path.join(GOLD, "filters.json")))
        self.small = json.load(open(os.path.join(GOLD, "small_boundary.json")))
        self._rels = None
        self._jin = None

    def gen(self, spec):
        return self.o.generate(spec["n"], spec["kind"], spec.get("domain", 0), spec.get("theta", 0.0), spec["seed"])

    def arbitrary_row_id_inputs(self):
        rec = self.synthetic["arbitrary_row_ids"]
        R = self.o.generate(3000, 4, 500, 0, rec["seedR"])
        S = self.o.generate(2000, 4, 500, 0, rec["seedS"])
        R["row_id"] = (R["row_id"] * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0xABCDEF)
        S["row_id"] = np.uint64(M64) - S["row_id"] * np.uint64(977)
        return R, S, rec

    def filter_inputs(self, c):
        col = self.o.generate(c["n"], 4, c["domain"], 0, c["seed"])["value"]
        sel = self.o.generate(c["n"] // 2 + 3, 4, c["n"], 0, c["seed"] + 100)["value"] if c["mode"] == "indirect" else None
        return col, sel

    @property
    def small_relations(self):
        if self._rels is None:
            self._rels = load_npz_xz("small_relations.npz.xz")
        return self._rels

    @property
    def small_join_inputs(self):
        if self._jin is None:
            self._jin = load_npz_xz("small_join_inputs.npz.xz")
        return self._jin

    def small_join(self, idx):
        j = self.small_join_inputs
        return make_rel(j["j%d_R" % idx]), make_rel(j["j%d_S" % idx])


def spec_join(R, S, bits):
    """Independent numpy statement of SURVEY.md A.1 (not a restatement of the
    reference's code): bucket ascending, probe side = R if cR >= cS else S, probe
    tuples in input order, build matches in descending input position."""
    mask = np.uint64((1 << bits) - 1)
    out = []
    bR, bS = (R["value"] & mask), (S["value"] & mask)
    oR, oS = np.argsort(bR, kind="stable"), np.argsort(bS, kind="stable")
    cR = np.bincount(bR.astype(np.int64), minlength=1 << bits)
    cS = np.bincount(bS.astype(np.int64), minlength=1 << bits)
    sR, sS = np.concatenate([[0], np.cumsum(cR)]), np.concatenate([[0], np.cumsum(cS)])
    for b in range(1 << bits):
        if cR[b] == 0 or cS[b] == 0:
            continue
        r, s = R[oR[sR[b]:sR[b + 1]]], S[oS[sS[b]:sS[b + 1]]]
        probe, build, flip = (r, s, False) if cR[b] >= cS[b] else (s, r, True)
        pos = {}
        for i in range(len(build) - 1, -1, -1):
            pos.setdefault(int(build["value"][i]), []).append(i)
        for p in probe:
            for q in pos.get(int(p["value"]), ()):
                a, c = int(p["row_id"]), int(build["row_id"][q])
                out.append((c, a) if flip else (a, c))
    return np.array(out, dtype=PAIR) if out else np.zeros(0, dtype=PAIR)
