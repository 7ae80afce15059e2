#!/usr/bin/env python3
"""bench.py — probe throughput of the MI355X radix hash join (BASELINE.json metric).

One "step" = one full RadixHashJoin on device-resident inputs (radix partition of both
relations, plan, count pass, probe/emit pass), i.e. rhj_join_device() of include/rhj.h.
value = probe-side tuples (nS) of all ranks / wall time, in 10^9 tuples/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|dense]

Workloads (synthetic, row_id[i] = i, keys through the splitmix64
finaliser so all 64 bits are populated):
    c2    1M  x 1M   uniform FK,  8 radix bits      (BASELINE configs[1])
    c3    100M x 100M uniform FK, 12 radix bits     (BASELINE configs[2]; default: the
          configuration the HBM-roofline target is stated on — c2's 48 MB working set
          lives in the Infinity Cache, so an HBM fraction is meaningless there)
    c4    100M x 1B  Zipf(0.9),   14 radix bits     (BASELINE configs[3]; the config leaves the
          radix free: at 14 bits a bucket's 6.1 K build tuples are LDS-resident)
N > 1: every rank joins its own independent relations of the same size (weak scaling:
independent joins of a plan shard across GPUs with no data-path collective).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    "c2": dict(nR=1_000_000, nS=1_000_000, bits=8, dist="uniform", name="1Mx1M uniform u64 FK, 8 radix bits"),
    "c3": dict(nR=100_000_000, nS=100_000_000, bits=12, dist="uniform", name="100Mx100M uniform u64 FK, 12 radix bits"),
    "c4": dict(nR=100_000_000, nS=1_000_000_000, bits=14, dist="zipf", name="100Mx1B Zipf(0.9) u64 FK, 14 radix bits (config leaves the radix free: 6.1 K build tuples per bucket fit LDS)"),
    "c4b12": dict(nR=100_000_000, nS=1_000_000_000, bits=12, dist="zipf", name="100Mx1B Zipf(0.9) u64 FK, 12 radix bits (experiment)"),
    "c4b13": dict(nR=100_000_000, nS=1_000_000_000, bits=13, dist="zipf", name="100Mx1B Zipf(0.9) u64 FK, 13 radix bits (experiment)"),
    "c4b15": dict(nR=100_000_000, nS=1_000_000_000, bits=15, dist="zipf", name="100Mx1B Zipf(0.9) u64 FK, 15 radix bits"),
    "c3b4": dict(nR=100_000_000, nS=100_000_000, bits=4, dist="uniform", name="100Mx100M uniform u64 FK, 4 radix bits = the reference's N_LSB as shipped (not a BASELINE config: buckets of 6 M tuples, tiled path; with --order any the library's own radix)"),
    "c4b4": dict(nR=100_000_000, nS=1_000_000_000, bits=4, dist="zipf", name="100Mx1B Zipf(0.9) u64 FK, 4 radix bits = the reference's N_LSB as shipped (experiment: low-radix path on skewed keys)"),
    "c3b14": dict(nR=100_000_000, nS=100_000_000, bits=14, dist="uniform", name="100Mx100M uniform u64 FK, 14 radix bits (not a BASELINE config: shows the LDS-resident path)"),
    "c3b13": dict(nR=100_000_000, nS=100_000_000, bits=13, dist="uniform", name="100Mx100M uniform u64 FK, 13 radix bits (experiment)"),
    "c3b15": dict(nR=100_000_000, nS=100_000_000, bits=15, dist="uniform", name="100Mx100M uniform u64 FK, 15 radix bits (experiment)"),
    "m16b8": dict(nR=16_000_000, nS=16_000_000, bits=8, dist="uniform", name="16Mx16M uniform u64 FK, 8 radix bits (experiment: mid-size join on the low-radix path)"),
    "c3half": dict(nR=100_000_000, nS=100_000_000, bits=12, dist="half", name="100Mx100M uniform u64, half of S without a partner, 12 radix bits (experiment: the foreign-key speculation fails)"),
    "dense": dict(nR=1_000_000, nS=1_000_000, bits=8, dist="dense", name="1Mx1M dense keys j+1, 8 radix bits"),
    # BASELINE configs[4]: the SIGMOD'18 `small` workload through the reference's own driver and query executor
    # linked against librhj.so (device-resident configuration); the 50 queries are dealt round-robin to the ranks
    # the 88 RadixHashJoin calls of the `small` workload (inputs recorded at the boundary, tests/golden) as independent
    # joins of a plan: dealt to the ranks largest first (shard.assign_joins), every match list sent to all ranks
    "smalljoins": dict(name="SIGMOD'18 small workload: its 88 RadixHashJoin calls (recorded inputs, 4 radix bits) dealt to the ranks as independent joins"),
    "small": dict(name="SIGMOD'18 small workload (14 relations, 50 queries), reference driver + librhj.so, queries sharded over the ranks"),
}


def source_fingerprint():
    """sha256 (16 hex digits) over the library's sources: what a profile was collected on, comparable where there is no .git (the GPU box)"""
    import glob, hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "sigmod-2018_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "sigmod-2018_amd", "csrc", "*.c*")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        if f.endswith(".o") or f.endswith(".s"):
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def lsr(x, k):
    """logical shift right on int64 tensors"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def mix64(x):
    """splitmix64 finaliser on int64 tensors (wrap-around arithmetic), = oracle orc_mix64"""
    import torch
    c1 = torch.tensor(0xbf58476d1ce4e5b9 - (1 << 64), dtype=torch.int64, device=x.device)
    c2 = torch.tensor(0x94d049bb133111eb - (1 << 64), dtype=torch.int64, device=x.device)
    x = x ^ lsr(x, 30)
    x = x * c1
    x = x ^ lsr(x, 27)
    x = x * c2
    x = x ^ lsr(x, 31)
    return x


def zipf_ranks(n, domain, theta, gen, device):
    """Gray et al. Zipf generator, vectorised (float64)."""
    import torch
    zetan = 0.0
    for a in range(1, domain + 1, 1 << 24):
        b = min(domain + 1, a + (1 << 24))
        zetan += float(torch.arange(a, b, device=device, dtype=torch.float64).pow(-theta).sum())
    alpha = 1.0 / (1.0 - theta)
    eta = (1.0 - (2.0 / domain) ** (1.0 - theta)) / (1.0 - (1.0 + 0.5 ** theta) / zetan)
    out = torch.empty(n, dtype=torch.int64, device=device)
    for a in range(0, n, 1 << 26):
        b = min(n, a + (1 << 26))
        u = torch.rand(b - a, generator=gen, device=device, dtype=torch.float64)
        uz = u * zetan
        r = (domain * (eta * u - eta + 1.0).pow(alpha)).to(torch.int64)
        r = torch.where(uz < 1.0 + 0.5 ** theta, torch.ones_like(r), r)
        r = torch.where(uz < 1.0, torch.zeros_like(r), r)
        out[a:b] = r.clamp_(0, domain - 1)
    return out


def make_relations(w, device, seed):
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    nR, nS = w["nR"], w["nS"]
    R = torch.empty((nR, 2), dtype=torch.int64, device=device)
    S = torch.empty((nS, 2), dtype=torch.int64, device=device)
    if w["dist"] == "dense":
        R[:, 0] = torch.arange(1, nR + 1, device=device)
        S[:, 0] = torch.randint(1, nR + 1, (nS,), generator=gen, device=device)
    else:
        R[:, 0] = mix64(torch.randperm(nR, generator=gen, device=device))
        if w["dist"] == "zipf":
            S[:, 0] = mix64(zipf_ranks(nS, nR, 0.9, gen, device))
        elif w["dist"] == "half":
            S[:, 0] = mix64(torch.randint(0, 2 * nR, (nS,), generator=gen, device=device))
        else:
            for a in range(0, nS, 1 << 27):
                b = min(nS, a + (1 << 27))
                S[a:b, 0] = mix64(torch.randint(0, nR, (b - a,), generator=gen, device=device))
    R[:, 1] = torch.arange(nR, device=device)
    S[:, 1] = torch.arange(nS, device=device)
    return R, S


def check_properties(R, S, out, m, w, bits=None, fk=True):
    """size-independent parity properties at full size (unique R keys; fk: every S tuple has its partner — else the S tuples
    that have one are counted)"""
    import torch
    if fk:
        assert m == w["nS"], "FK join must emit one pair per S tuple (got %d)" % m
    else:
        have = torch.isin(S[:, 0], R[:, 0])
        assert m == int(have.sum()), "one pair per S tuple whose key R holds: %d, got %d" % (int(have.sum()), m)
        assert int(out[:m, 1].sum()) == int(S[have, 1].sum()), "not the S tuples that have a partner"
        del have
    step = max(1, m // (1 << 22))
    idx = torch.arange(0, m, step, device=out.device)
    p = out[idx]
    assert bool((R[p[:, 0], 0] == S[p[:, 1], 0]).all()), "pair joins unequal keys"
    # every S row id appears exactly once: the wrap-around sum is closed-form
    if fk:
        assert int(out[:, 1].sum()) == (w["nS"] * (w["nS"] - 1) // 2) % (1 << 63), "S row ids are not a permutation"
    mask = (1 << (bits or w["bits"])) - 1                       # (order any: the radix width the library used)
    b = S[p[:, 1], 0] & mask
    assert bool((b[1:] >= b[:-1]).all()), "buckets not ascending"
    # the canonical order inside the buckets (row ids are input positions here): R probes where histR >= histS
    # (rhjoin.c:86), probe tuples in input order, a tuple's matches in descending build position — on windows of consecutive pairs
    hR = torch.bincount(R[:, 0] & mask, minlength=mask + 1)
    hS = torch.bincount(S[:, 0] & mask, minlength=mask + 1)
    L = min(m, 1 << 21)
    gen = torch.Generator().manual_seed(7)
    starts = [0, max(0, m // 2 - L // 2), max(0, m - L)] + [int(x) for x in torch.randint(0, max(1, m - L + 1), (5,), generator=gen)]
    for a in starts:
        win = out[a:a + L]
        kb = S[win[:, 1], 0] & mask
        flip = hR[kb] < hS[kb]
        probe = torch.where(flip, win[:, 1], win[:, 0])
        build = torch.where(flip, win[:, 0], win[:, 1])
        same = kb[1:] == kb[:-1]
        ok = (~same) | (probe[1:] > probe[:-1]) | ((probe[1:] == probe[:-1]) & (build[1:] < build[:-1]))
        assert bool(ok.all()), "pairs out of canonical order inside a bucket (window at %d)" % a


def _cpu_row(o, pyoracle, w, threads, budget_s, start_nR):
    """One timed row: the reference's own code at THREADS = `threads` on a sample of the workload that grows
    (x4) until a single RadixHashJoin call takes a few seconds or the row's budget is spent."""
    scale = w["nS"] // w["nR"]
    ref = pyoracle.Reference(w["bits"], threads)
    nR, best, spent = min(w["nR"], start_nR), None, 0.0
    while True:
        nS = nR * scale
        R = o.generate(nR, 3 if w["dist"] == "dense" else 0, 0, 0.0, 42)
        S = o.generate(nS, {"uniform": 1, "zipf": 2, "dense": 4}[w["dist"]], nR, 0.9, 43)
        if w["dist"] == "dense":
            S["value"] += 1
        t = time.time()
        pairs, info = ref.join(R, S, with_info=True)
        spent += time.time() - t
        best = (nR, nS, info["seconds"], len(pairs))
        del R, S, pairs
        if info["seconds"] * 4 > 8.0 or spent >= budget_s or nR >= w["nR"] or nR * 4 * scale * 80 > 40e9:
            break
        nR = min(w["nR"], nR * 4)
    nR, nS, secs, m = best
    return {"value": nS / secs / 1e9, "unit": "10^9 probe tuples/s", "cores": threads, "kind": "reference",
            "sample": "%dx%d %s, %d radix bits, one RadixHashJoin call, %.2f s, %d pairs" % (nR, nS, w["dist"], w["bits"], secs, m)}


def cpu_baseline(w):
    """The reference's own code (oracle/_ref: rhjoin.c / preprocess.c / scheduler.c compiled from /root/reference,
    N_LSB = the workload's radix bits) timed on this box's host cores over a bounded sample of the same workload
    (same distribution and radix bits), about 10-30 s of CPU work in all.  Rows:
      * THREADS 4 — the path as shipped (structs.h:12): scheduler.c's pthread pool running HistJob / PartitionJob /
        JoinJob.  PartitionJob re-scans the input once per bucket it owns (preprocess.c:222-299): O(buckets x N), so
        its RATE does not depend on the sample size — the sample's rate is the extrapolation to the full workload
         — and it is wrong when the last bucket holds more than 1/THREADS of an input
        (SURVEY.md finding 4: uniform and Zipf-over-hashed keys are not affected; the pair count is in the row);
      * THREADS 16 — the same code with as many workers as this box gives one GPU's process (THREADS is a compile-time
        #define: one prebuilt variant per count, so `nproc` itself is not available);
      * THREADS 1 — SerialReorderArray + the bucket loop on one core: the reference's correct and, at 8+ radix bits,
        faster mode; the parity oracle.
    The headline fields are the as-shipped THREADS 4 row (the path north_star names); the fastest row is named too.
    Without oracle/_ref (it cannot be built on the GPU box) the rows fall back to this repo's restatement, kind "port"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    o = pyoracle.Oracle()
    rows = []
    for threads, budget, start in ((4, 8.0, 250_000), (16, 6.0, 250_000), (1, 10.0, 4_000_000 // max(w["nS"] // w["nR"] // 2, 1))):
        if pyoracle.ref_available(w["bits"], threads):
            rows.append(_cpu_row(o, pyoracle, w, threads, budget, start))
    if not rows:                                          # no compiled reference here: time the restatement (1 core)
        scale = w["nS"] // w["nR"]
        nR = min(w["nR"], 4_000_000)
        R = o.generate(nR, 0, 0, 0.0, 42)
        S = o.generate(nR * scale, 1, nR, 0.0, 43)
        t = time.time()
        o.join(R, S, w["bits"])
        secs = time.time() - t
        rows.append({"value": nR * scale / secs / 1e9, "unit": "10^9 probe tuples/s", "cores": 1, "kind": "port",
                     "sample": "%dx%d, %d radix bits, oracle/rhj_oracle.c, %.2f s" % (nR, nR * scale, w["bits"], secs)})
    head = dict(rows[0])
    head["rows"] = rows
    head["best"] = max(rows, key=lambda r: r["value"])
    head["host_cpus"] = os.cpu_count()
    try:
        head["usable_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    head["note"] = ("headline = the reference as shipped when its THREADS-4 variant is built for these radix bits "
                    "(rate independent of n: O(buckets x N) partitioner); `best` = the fastest reference mode")
    return head


class _Engine:
    """One engine process kept alive: relation names + Done once, then batches of queries ended by F — the protocol of the
    reference's own driver (handler.c:66-97 loops over batches until EOF; submission/harness.cpp:206-299 streams them)."""

    def __init__(self, path, cwd, env, head):
        import subprocess, tempfile
        self.err = tempfile.TemporaryFile()
        self.t_start = time.perf_counter()
        self.p = subprocess.Popen([path], cwd=cwd, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=self.err, bufsize=0)
        self.p.stdin.write(head.encode())

    def batch(self, queries):
        """Submits one batch and returns its answer lines (the engine flushes stdout before it reads on, handler.c:68)."""
        self.p.stdin.write(("\n".join(queries) + "\nF\n").encode())
        out = []
        for _ in queries:
            line = self.p.stdout.readline()
            if not line:
                self.err.seek(0)
                raise RuntimeError("engine ended early: " + self.err.read().decode()[-1000:])
            out.append(line.decode().rstrip("\n"))
        return out

    def close(self):
        self.p.stdin.close()
        rc = self.p.wait(timeout=120)
        if rc != 0:
            self.err.seek(0)
            raise RuntimeError("engine exit code %d: %s" % (rc, self.err.read().decode()[-1000:]))


def run_small(args, world, rank, local, dist):
    """BASELINE configs[4] in its query-sharded form.  ONE engine process per rank stays alive for the whole run
    (oracle/_ref/radixhash_rhj_resident: handler.c, query.c, stats.c, best_tree.c, relation_list.c of the reference +
    librhj.so): it loads the relations once, a step is one BATCH of the rank's share of the 50 queries, and the timed
    region is batch submission -> last answer line — HIP start-up and the column upload are reported beside it, not in
    `value` (a process per step measured 0.3 s of start-up around 25 ms of query work).  Rank 0 merges the answers in
    query order and checks them against small.result (tests/golden)."""
    import tempfile
    import numpy as np
    for p in (os.path.join(HERE, "tests"), os.path.join(HERE, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import helpers
    g = helpers.Golden()
    exe = os.path.join(HERE, "oracle", "_ref", "radixhash_rhj_resident")
    ref = os.path.join(HERE, "oracle", "_ref", "radixhash_t4")
    if not os.path.exists(exe):
        raise RuntimeError("oracle/_ref/radixhash_rhj_resident is not built (needs /root/reference at build time)")
    tmp = tempfile.mkdtemp()
    names = []
    for i in range(14):
        cols = g.small_relations["r%d" % i].astype("<u8")
        with open(os.path.join(tmp, "r%d" % i), "wb") as f:
            np.array([cols.shape[1], cols.shape[0]], dtype="<u8").tofile(f)
            cols.tofile(f)
        names.append("r%d" % i)
    queries = [l for l in g.small["work_lines"] if "|" in l]
    mine = [i for i in range(len(queries)) if i % world == rank]
    my_queries = [queries[i] for i in mine]
    head = "\n".join(names) + "\nDone\n"
    env = dict(os.environ, RHJ_DEVICE=str(local))

    def barrier():
        if world > 1:
            dist.barrier()

    eng = _Engine(exe, tmp, env, head)
    lines = eng.batch(my_queries) if my_queries else []        # first batch: start-up, load, first-touch of every workspace
    startup = time.perf_counter() - eng.t_start
    for _ in range(max(args.warmup - 1, 0)):
        eng.batch(my_queries)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if my_queries:
            lines = eng.batch(my_queries)
    barrier()
    elapsed = time.perf_counter() - t0
    eng.close()
    if world > 1:
        import torch
        t = torch.tensor([elapsed, startup], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, startup = float(t[0].item()), float(t[1].item())
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, lines))
    else:
        gathered = [(mine, lines)]
    if rank == 0:
        answers = [None] * len(queries)
        for idx, ls in gathered:
            assert len(idx) == len(ls), "a rank printed %d lines for %d queries" % (len(ls), len(idx))
            for i, l in zip(idx, ls):
                answers[i] = l
        assert answers == g.small["result_lines"], "the merged answers differ from small.result"
        res = {"metric": "queries/s on the SIGMOD'18 small workload (persistent engine per rank; batch submission -> last answer)",
               "value": len(queries) * args.steps / elapsed, "unit": "queries/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "SIGMOD'18 small (fixture)",
               "config": {"workload": WORKLOADS["small"]["name"], "id": "small", "queries": len(queries),
                          "parallelism": "the 50 queries of a batch dealt round-robin over %d rank(s) (a query's joins are a dependent "
                                         "chain, query.c:408-461: the query is the unit that shards); one persistent engine per rank; "
                                         "no collective on the data path" % world,
                          "startup_s": startup,
                          "startup_note": "process start + HIP runtime initialisation + relation load/upload + the first batch; not in `value`",
                          "answers": "identical to small.result"},
               "roofline": None}
        if world == 1 and not args.no_cpu_baseline and os.path.exists(ref):
            cpu = _Engine(ref, tmp, dict(os.environ), head)
            ref_lines = cpu.batch(queries)                    # warm-up batch (page cache, allocator)
            n_ref = 3
            t1 = time.perf_counter()
            for _ in range(n_ref):
                ref_lines = cpu.batch(queries)
            dt = (time.perf_counter() - t1) / n_ref
            cpu.close()
            res["cpu_baseline"] = {"value": len(queries) / dt, "unit": "queries/s", "cores": 4, "kind": "reference",
                                   "sample": "the reference engine as shipped (THREADS 4), kept alive the same way: batches of all 50 queries, "
                                             "%.3f s per batch (mean of %d after one warm-up batch); answers %s" % (
                                                 dt, n_ref, "identical" if ref_lines == g.small["result_lines"] else "DIFFER")}
        print(json.dumps(res))


def run_smalljoins(args, rhj, world, rank, backend, dist):
    """Independent joins of a plan over the ranks (shard.run_independent_joins): the 88 joins of `small`, every pair list
    checked against the digest the compiled reference produced for it (tests/golden/small_boundary.json)."""
    import torch
    for p in (os.path.join(HERE, "tests"), os.path.join(HERE, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import helpers
    from pyoracle import Oracle
    shard = importlib.import_module("sigmod-2018_amd.shard")
    g = helpers.Golden()
    ops = shard.RhjOps(rhj)
    recs = g.small["joins"]
    joins = [tuple(rhj.to_device(x) for x in g.small_join(j["idx"])) for j in recs]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = owner = None
    for _ in range(args.warmup):
        res, owner = shard.run_independent_joins(ops, joins, 4)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, owner = shard.run_independent_joins(ops, joins, 4)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rhj.dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    o = Oracle()
    for r, j in zip(res, recs):                            # every rank holds every list (gathered): all of them check
        helpers.assert_digest(o, rhj.pairs_to_numpy(r), j, "small join %d" % j["idx"])
    if rank == 0:
        tuples = sum(int(a.shape[0]) + int(b.shape[0]) for a, b in joins)
        load = [sum(int(a.shape[0]) + int(b.shape[0]) for (a, b), ow in zip(joins, owner) if ow == r) for r in range(world)]
        print(json.dumps({"metric": "joins/s on the 88 RadixHashJoin calls of the SIGMOD'18 small workload", "value": len(joins) * args.steps / elapsed,
                          "unit": "joins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "SIGMOD'18 small (fixture)",
                          "config": {"workload": WORKLOADS["smalljoins"]["name"], "id": "smalljoins", "joins": len(joins), "input_tuples": tuples,
                                     "pairs": sum(j["matches"] for j in recs), "parallelism": "independent joins dealt largest first over %d rank(s), "
                                     "all-gather-v of every match list" % world, "input_tuples_per_rank": load,
                                     "answers": "every pair list equals the reference's digest"},
                          "roofline": None}))


def run_strong(args, w, rhj, world, rank, backend, dist):
    """ONE join of the workload sharded over the ranks by bucket range: the relations are replicated
    (same seed on every rank: a device-resident column store per GPU), every rank takes a contiguous bucket range of
    equal width and joins it with ONE call — rhj_join_device_range: the join's first partition pass drops the other
    ranks' buckets while it reads the relations, nothing runs in front of the join — and the pair lists are exchanged
    with the exact-size all-gather-v so that every rank ends with the canonical result.  Total work is fixed:
    "scaling": "strong"."""
    import torch
    shard = importlib.import_module("sigmod-2018_amd.shard")
    ops = shard.RhjOps(rhj)
    dev = rhj.dev
    R, S = make_relations(w, dev, 1234)
    gather = not args.no_gather

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    full, info = None, None
    for _ in range(args.warmup):
        full, info = shard.sharded_join(ops, R, S, w["bits"], gather=gather)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        del full
        full, info = shard.sharded_join(ops, R, S, w["bits"], gather=gather)
    barrier()
    elapsed = time.perf_counter() - t0
    st = rhj.stats()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    M = sum(info["counts"]) if gather else None
    if gather:
        check_properties(R, S, full, M, w)                 # canonical order, every S row once, on every rank
    if rank == 0:
        lo, hi = info["range"]
        share = (hi - lo) / float(1 << w["bits"])
        cr, cs = int(w["nR"] * share), int(w["nS"] * share)           # (equal-width ranges of uniform keys)
        lp = info["local_pairs"]
        probe_bytes = 16 * (cr + cs) + 16 * lp
        ms_probe = st["ms_probe"]
        res = {"metric": "probe throughput (10^9 tuples/s) + achieved HBM GB/s",
               "value": w["nS"] * args.steps / elapsed / 1e9,
               "unit": "10^9 probe tuples/s (ONE RadixHashJoin over all ranks, inputs resident in HBM on every rank)",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": w["name"], "id": args.workload, "nR": w["nR"], "nS": w["nS"], "radix_bits": w["bits"],
                          "matches": M, "parallelism": "equal-width bucket ranges over %d rank(s), one ranged join call per rank%s" % (
                              world, ", exact-size all-gather-v of the pair lists" if gather else ", pair lists kept sharded"),
                          "rank0_range": list(info["range"]), "rank0_tuples": [cr, cs], "pairs_per_rank": info["counts"]},
               "roofline": {"bound": "hbm", "kernel": "probe kernel of rank 0's bucket range", "achieved": probe_bytes / (ms_probe * 1e-3) / 1e9 if ms_probe > 0 else 0.0,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (probe_bytes / (ms_probe * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_probe > 0 else 0.0,
                            "traffic": None, "algorithmic_bytes": probe_bytes, "ms": ms_probe,
                            "formula": "16*nS + 16*nR + 16*matches of the rank's range (SURVEY.md 8d)"}}
        print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one independent join per rank (default); strong = ONE join sharded by bucket range "
                         "over the ranks (sigmod-2018_amd/shard.py) with the exact-size all-gather-v of the pair lists")
    ap.add_argument("--no-gather", action="store_true", help="strong scaling: keep the pair lists sharded (no exchange step)")
    ap.add_argument("--input", default="tuples", choices=["tuples", "keys"],
                    help="tuples: rhj_join_device on the ABI's 16-byte {value, row_id} tuples (the reference's relation layout; default); "
                         "keys: rhj_join_keys_device on the key columns alone (row id = position: GetRelation of a base relation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-hbm-table", action="store_true")
    ap.add_argument("--order", default="canonical", choices=["canonical", "any"],
                    help="any: rhj_set_order(1) — same pairs, canonical order of the radix width the library picks from the sizes")
    ap.add_argument("--timing", type=int, default=2, choices=[0, 1, 2],
                    help="HIP events inside rhj_join_device during the timed steps: 2 = per stage (default; the roofline's kernel "
                         "times come from them), 1 = whole join, 0 = none.  Below 2 the stage times are collected in a second, "
                         "untimed pass.  Matters for small joins only: an event between two launches costs them ~6 us")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RHJ_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of N ranks on fewer GPUs
    ngpu = torch.cuda.device_count()
    local = local % max(ngpu, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    if args.workload == "small":
        run_small(args, world, rank, local, dist)
        if world > 1:
            dist.destroy_process_group()
        return
    w = WORKLOADS[args.workload]
    mod = importlib.import_module("sigmod-2018_amd")
    rhj = mod.RHJ(device=local)
    if args.workload == "smalljoins":
        run_smalljoins(args, rhj, world, rank, backend, dist)
        if world > 1:
            dist.destroy_process_group()
        return
    rhj.set_bits(w["bits"])
    if args.force_hbm_table:
        rhj.lib.rhj_set_force_hbm_table(1)
    dev = rhj.dev
    if args.scaling == "strong":
        run_strong(args, w, rhj, world, rank, backend, dist)
        if world > 1:
            dist.destroy_process_group()
        return
    R, S = make_relations(w, dev, 1234 + rank)
    cap = w["nS"]
    out = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    import ctypes as C
    m = C.c_uint64(0)

    keysR = keysS = None
    if args.input == "keys":                    # the key columns alone (contiguous copies; row id = position as in R[:, 1], S[:, 1])
        keysR, keysS = R[:, 0].contiguous(), S[:, 0].contiguous()

    def step():
        if keysR is not None:
            rc = rhj.lib.rhj_join_keys_device(keysR.data_ptr(), w["nR"], keysS.data_ptr(), w["nS"], out.data_ptr(), cap, C.byref(m))
        else:
            rc = rhj.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), cap, C.byref(m))
        if rc != 0:
            raise RuntimeError("rhj_join_device rc=%d" % rc)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rhj.lib.rhj_set_timing(args.timing)
    rhj.lib.rhj_set_order(1 if args.order == "any" else 0)
    for _ in range(args.warmup):
        step()
    if rank == 0 or True:
        check_properties(R, S, out, m.value, w, rhj.stats()["radix_bits"])
    keys = ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_build", "ms_count", "ms_offsets", "ms_probe", "ms_total")
    acc = dict.fromkeys(keys, 0.0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = rhj.stats()
        for k in keys:
            acc[k] += st[k]
    barrier()
    elapsed = time.perf_counter() - t0
    stage_steps = args.steps
    if args.timing < 2:                   # stage times from a second pass with the per-stage events switched on
        rhj.lib.rhj_set_timing(2)
        stage_steps = min(args.steps, 50)
        acc = dict.fromkeys(keys, 0.0)
        for _ in range(stage_steps):
            step()
            st = rhj.stats()
            for k in keys:
                acc[k] += st[k]
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = rhj.stats()
    stage = {k: acc[k] / stage_steps for k in keys}
    spec_state = int(rhj.lib.rhj_last_spec())           # the timed steps' last join: 0 not tried, 1 the speculation held, 2 failed
    nospec = None
    if spec_state == 1:
        # For the record, behind the timed region: the same steps with the foreign-key speculation switched off
        #  — identical pairs from the general kernel alone.
        rhj.lib.rhj_set_spec(0)
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pm = 0.0
        for _ in range(args.steps):
            step()
            pm += rhj.stats()["ms_probe"]
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        rhj.lib.rhj_set_spec(1)
        nospec = {"ms_per_step": e1 / args.steps * 1e3, "value": w["nS"] * args.steps / e1 / 1e9, "probe_kernel": "k_join_fused",
                  "probe_ms": pm / args.steps, "note": "rank 0's own clock, not barriered; same inputs, same pairs"}

    if rank == 0:
        nR, nS, M = w["nR"], w["nS"], m.value
        probe_bytes = 16 * nS + 16 * nR + 16 * M          # probe kernel
        count_bytes = 16 * nS + 16 * nR
        scatter_bytes = 32 * (nR + nS)                    # 16 B read + 16 B written per AoS tuple
        hist_bytes = 16 * (nR + nS)                       # AoS key read (stride-16)
        if args.input == "keys":                          # SURVEY.md 8(d), column form: 8*n histogram read, 8*n read + 16*n written by the scatter
            scatter_bytes = 24 * (nR + nS)
            hist_bytes = 8 * (nR + nS)

        def gbs(b, ms):
            return b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        part_ms = stage["ms_hist"] + stage["ms_scan"] + stage["ms_scatter"]
        # SURVEY.md 8(d) with the ABI's AoS tuples: 16*n read by the histogram + 16*n read + 16*n written by the
        # scatter = 48*n per relation, WHATEVER the number of passes the implementation takes (a second pass is
        # this implementation's choice, not algorithmic work)
        part_bytes = hist_bytes + scatter_bytes
        part_formula = ("48*n, both relations: 16*n histogram read + 16*n read + 16*n written by the scatter (SURVEY.md 8d, AoS)" if args.input == "tuples"
                        else "32*n, both relations: 8*n histogram read + 8*n read + 16*n written by the scatter (SURVEY.md 8d, key columns)")
        if w["bits"] <= 8:        # one pass: per-tile histogram, scan, LDS-staged scatter
            part = {"ms": part_ms, "GBps": gbs(part_bytes, part_ms), "algorithmic_bytes": part_bytes,
                    "formula": part_formula,
                    "histogram": {"ms": stage["ms_hist"], "GBps": gbs(hist_bytes, stage["ms_hist"])},
                    "scan": {"ms": stage["ms_scan"]},
                    "scatter": {"ms": stage["ms_scatter"], "GBps": gbs(scatter_bytes, stage["ms_scatter"])}}
        else:                     # two passes in run form (k_local_part, k_hist_runs + scan, k_scatter_runs)
            part = {"ms": part_ms, "GBps": gbs(part_bytes, part_ms), "algorithmic_bytes": part_bytes,
                    "formula": part_formula,
                    "pass1_tile_local": {"ms": stage["ms_hist"], "moved_GBps": gbs(scatter_bytes, stage["ms_hist"])},
                    "pass2_histogram_scan": {"ms": stage["ms_scan"]},
                    "pass2_scatter_runs": {"ms": stage["ms_scatter"], "moved_GBps": gbs(scatter_bytes, stage["ms_scatter"])}}
        small = st["path"] == "small"
        fused = st["path"] in ("fused", "small")
        lowradix = st["path"] == "lowradix"
        join_ms = stage["ms_build"] + stage["ms_count"] + stage["ms_offsets"] + stage["ms_probe"]
        spec = spec_state if fused and not small else 0
        probe_kernel = ("k_join_spec (the fused join kernel on the foreign-key speculation, which held: LDS index build + probe, pairs of the "
                        "foreign-key side's units written from the probe loop, no stash, no chained offsets; k_join_fused and k_join_walk "
                        "behind it return at once)" if fused and spec == 1
                        else "k_join_fused (LDS index build + probe + emit, one kernel; k_join_walk behind it returns at once on foreign-key joins)" if fused
                        else "k_join_fused on the finer buckets + k_lr_totals + k_lr_emit (internal join, then the pairs in the canonical order "
                             "of the caller's radix: the whole probe phase of the low-radix path)" if lowradix
                        else "k_probe<WRITE> (emit pass of the tiled path)")
        # HBM-side bytes of the dominant kernel per launch: rocprofv3 --pmc passes of this same command, committed under profiles/
        # (tools/pmc.sh; counters cannot be read from inside this process), RAW and CORRECTED by the calibration of
        # profiles/r04_calibration.json (tools/micro/calib.hip: known byte counts in this kernel's own access patterns):
        #   streaming reads of 8, 12 and 16 B a lane: FETCH_SIZE = exactly half of the lines they touch  -> x 2
        #   stores: WRITE_SIZE exact
        #   the 12-byte sc1 gathers: every one that misses the L2s is one 64-byte request, counted as it is
        # so  corrected = stream lines (known by construction: both partitioned relations once) + (raw FETCH - stream lines / 2) + WRITE.
        traffic, traffic_src, traffic_detail = None, None, None
        try:
            import glob
            cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.json" % args.workload)) if "subsplit" not in f)
            if cands and fused:
                pm = json.load(open(cands[-1]))
                want = ("k_join_spec",) if spec == 1 else ("k_join_fused",)
                per = {}
                for kname, kv in pm.items():
                    if isinstance(kv, dict) and any(wk in kname for wk in want) and "FETCH_SIZE" in kv and "WRITE_SIZE" in kv:
                        key = [wk for wk in want if wk in kname][0]
                        if (kv["FETCH_SIZE"] + kv["WRITE_SIZE"]) > sum(per.get(key, (0, 0))):                # the instantiation that ran
                            per[key] = (kv["FETCH_SIZE"] * 1024.0, kv["WRITE_SIZE"] * 1024.0)
                if per:
                    raw_fetch = sum(v[0] for v in per.values())
                    raw_write = sum(v[1] for v in per.values())
                    tuple_bytes = 12 if (w["bits"] > 8 and not small) else 16                              # the partitioned tuples this kernel streams
                    stream_lines = tuple_bytes * (nR + nS)
                    cal = None
                    cal_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_calibration.json")))
                    if cal_file:
                        cal = json.load(open(cal_file[-1]))
                    f_stream = (cal or {}).get("calib_stream_read12", {}).get("fetch_true_over_raw", 2.0)
                    gather_raw = max(0.0, raw_fetch - stream_lines / f_stream)
                    traffic = int(stream_lines + gather_raw + raw_write)
                    traffic_detail = {"raw_FETCH_SIZE": int(raw_fetch), "raw_WRITE_SIZE": int(raw_write), "raw_sum": int(raw_fetch + raw_write),
                                      "stream_lines": int(stream_lines), "stream_factor": f_stream, "gather_requests_bytes": int(gather_raw),
                                      "corrected": traffic, "corrected_over_algorithmic": traffic / probe_bytes,
                                      "calibration": os.path.basename(cal_file[-1]) if cal_file else None,
                                      "pmc_file": os.path.basename(cands[-1]),
                                      "pmc_collected_at": {"git_head": pm.get("_collected_at_git_head", "unknown"), "source_fingerprint": pm.get("_source_fingerprint", "unknown")},
                                      "running": {"source_fingerprint": source_fingerprint()}}
                    traffic_src = ("%s: FETCH_SIZE + WRITE_SIZE of the kernel's last dispatch (separate --pmc passes; not measured in this run: counters "
                                   "cannot be read from inside the process), CORRECTED by %s: streaming reads count half (x2 on the %d-byte tuples "
                                   "both relations are streamed as), gather requests and stores count as they are; raw and corrected bytes in "
                                   "traffic_detail" % (os.path.basename(cands[-1]), os.path.basename(cal_file[-1]) if cal_file else "the guide's rule", tuple_bytes))
        except Exception:
            traffic, traffic_src, traffic_detail = None, None, None
        res = {
            "metric": "probe throughput (10^9 tuples/s) + achieved HBM GB/s",
            "value": world * nS * args.steps / elapsed / 1e9,
            "unit": "10^9 probe tuples/s (whole RadixHashJoin, inputs resident in HBM)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": w["name"], "id": args.workload, "nR": nR, "nS": nS, "radix_bits": w["bits"],
                       "matches": M, "parallelism": "independent join per GPU" if world > 1 else "1 GPU",
                       "path": st["path"], "sub_bits": st["sub_bits"], "pass1_bits": st["pass1_bits"], "timing": args.timing,
                       "order": args.order, "radix_bits_used": st["radix_bits"], "input": ("16-byte {value, row_id} tuples (the ABI's relation layout)" if args.input == "tuples" else "key columns, row id = position (rhj_join_keys_device)"),
                       "stage_times": ("events of the timed steps" if args.timing == 2 else
                                       "second pass of %d steps with per-stage events (the timed steps ran with timing %d)" % (stage_steps, args.timing)),
                       "units": st["units"], "max_build_side": st["max_build"],
                       "fk_speculation": ("held" if spec == 1 else "failed: the ordinary kernel did the join" if spec == 2 else "not tried"),
                       "without_fk_speculation": nospec},
            "roofline": {"bound": "hbm", "kernel": probe_kernel,
                         "achieved": gbs(probe_bytes, stage["ms_probe"]), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs(probe_bytes, stage["ms_probe"]) / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_detail": traffic_detail,
                         "algorithmic_bytes": probe_bytes, "ms": stage["ms_probe"],
                         "formula": "16*nS + 16*nR + 16*matches (SURVEY.md 8d)"},
            "roofline_partition": {"bound": "hbm", "kernel": "radix partition of both relations (all passes: tile-local pass, "
                                   "pass-2 histogram + scans, run scatter)" if w["bits"] > 8 else
                                   "radix partition of both relations (k_small_hist, k_small_scatter: tile histogram, self-scanning scatter)" if small
                                   else "radix partition of both relations (histogram, scans, scatter)",
                                   "achieved": gbs(part_bytes, part_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gbs(part_bytes, part_ms) / HBM_PEAK_GBS, "algorithmic_bytes": part_bytes, "ms": part_ms,
                                   "formula": part_formula},
            "kernels": {
                "probe_tuples_per_s_e9": nS / (stage["ms_probe"] * 1e-3) / 1e9 if stage["ms_probe"] > 0 else 0.0,
                "join_phase_ms": join_ms,
                "join_phase_GBps": gbs(probe_bytes, join_ms),
                "partition": part,
                "plan": {"ms": stage["ms_plan"]},
                "build_tables": {"ms": stage["ms_build"]}, "count": {"ms": stage["ms_count"]},
                "offsets": {"ms": stage["ms_offsets"]},
                "gpu_total_ms": stage["ms_total"],
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
