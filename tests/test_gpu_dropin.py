"""GPU: the reference's own engine (its caller side compiled unchanged in the build
container and linked against librhj.so, see oracle/Makefile and INTEGRATION.md) answers the
SIGMOD'18 `small` workload and must reproduce the reference's golden file small.result line
by line.  Two link configurations:
  radixhash_rhj           our RadixHashJoin()/Filter() behind the reference's inter_res.c
                          (host-resident intermediate results, H2D/D2H per operator);
  radixhash_rhj_resident  inter_res.c and filter.c left out: librhj.so supplies their whole
                          interface with the intermediate results kept on the device
                          (include/rhj_inter.h).
The relation files are rebuilt from the committed fixture."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINES = {name: os.path.join(ROOT, "oracle", "_ref", name) for name in ("radixhash_rhj", "radixhash_rhj_resident")}


@pytest.mark.parametrize("engine,devices", [(e, 0) for e in sorted(ENGINES)] + [("radixhash_rhj", 1), ("radixhash_rhj", 3)])
@pytest.mark.parametrize("empty_mode", ["head", "null"])
def test_reference_engine_on_small_workload(golden, tmp_path, empty_mode, engine, devices):
    """devices > 0: the same unchanged engine with RHJ_DEVICES set — every RadixHashJoin() of the 50 queries sharded by bucket
    range over that many device contexts inside the one process (include/rhj.h; RHJ_DEVICES_SAME=1: all of them on the box's
    one GPU), every device's pairs moved to their place in the one result list: the golden file again, line by line."""
    ENGINE = ENGINES[engine]
    if not os.path.exists(ENGINE):
        pytest.skip("oracle/_ref/%s not built (needs /root/reference at build time)" % engine)
    rels = golden.small_relations
    names = []
    for i in range(14):
        cols = rels["r%d" % i].astype("<u8")
        path = tmp_path / ("r%d" % i)
        with open(path, "wb") as f:
            np.array([cols.shape[1], cols.shape[0]], dtype="<u8").tofile(f)     # tuples, columns
            cols.tofile(f)                                                       # column-major
        names.append("r%d" % i)
    stdin = "\n".join(names) + "\nDone\n" + "\n".join(golden.small["work_lines"]) + "\n"
    env = dict(os.environ, RHJ_EMPTY=empty_mode, RHJ_RADIX_BITS="4")
    if devices:
        env.update(RHJ_DEVICES=str(devices), RHJ_DEVICES_SAME="1")
    res = subprocess.run([ENGINE], input=stdin.encode(), cwd=str(tmp_path), env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    got = res.stdout.decode().splitlines()
    want = golden.small["result_lines"]
    assert len(got) == len(want) == 50
    bad = [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w]
    assert not bad, bad[:5]


def test_bench_small_workload_mode_checks_and_reports(tmp_path):
    """bench.py --workload small (BASELINE configs[4], query-sharded): ONE engine process per rank stays alive, a step is a
    batch of the rank's queries and only batch submission -> last answer is timed (start-up is reported beside it); the run
    itself asserts that the merged answers equal small.result.  Here one rank, three timed batches."""
    import json
    import sys
    if not os.path.exists(ENGINES["radixhash_rhj_resident"]):
        pytest.skip("oracle/_ref/radixhash_rhj_resident not built")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    line = json.loads(res.stdout.decode().strip().splitlines()[-1])
    assert line["config"]["answers"] == "identical to small.result" and line["config"]["queries"] == 50 and line["value"] > 0
    # query work, not process start-up: a batch of the 50 queries takes tens of milliseconds, HIP start-up alone 0.1-0.3 s
    assert line["ms_per_step"] < 150.0, line
    assert line["config"]["startup_s"] > line["ms_per_step"] / 1e3, line
    assert "round-robin" in line["config"]["parallelism"]


@pytest.mark.parametrize("bits", ["4", "10"])
def test_scaled_workload_matches_the_reference_engine(golden, tmp_path, bits):
    """`small` with every relation four copies of itself (copy c shifted by c * 2^24 in every column, so joins
    match inside a copy only): the reference engine as compiled from its own sources (THREADS 1, the
    authoritative mode) and the device-resident configuration answer the same 50 queries and must print the
    same 50 lines — at the engine's 4 radix bits and at 10, where the joins take the two-pass partition and the
    view sums must not change (they do not depend on the radix)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "radixhash_t1")
    eng = ENGINES["radixhash_rhj_resident"]
    if not (os.path.exists(ref) and os.path.exists(eng)):
        pytest.skip("oracle/_ref engines not built (need /root/reference at build time)")
    K, stride = 4, np.uint64(1 << 24)
    names = []
    for i in range(14):
        cols = golden.small_relations["r%d" % i].astype("<u8")
        assert int(cols.max()) < int(stride)
        big = np.concatenate([cols + np.uint64(c) * stride for c in range(K)], axis=1)
        with open(tmp_path / ("r%d" % i), "wb") as f:
            np.array([big.shape[1], big.shape[0]], dtype="<u8").tofile(f)
            np.ascontiguousarray(big).tofile(f)
        names.append("r%d" % i)
    stdin = ("\n".join(names) + "\nDone\n" + "\n".join(golden.small["work_lines"]) + "\n").encode()
    outs = {}
    for name, exe, env in (("reference", ref, dict(os.environ)), ("resident", eng, dict(os.environ, RHJ_RADIX_BITS=bits))):
        res = subprocess.run([exe], input=stdin, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             timeout=900)
        assert res.returncode == 0, (name, res.stderr.decode()[-2000:])
        outs[name] = res.stdout.decode().splitlines()
    assert len(outs["reference"]) == len(outs["resident"]) == 50
    bad = [(i, a, b) for i, (a, b) in enumerate(zip(outs["reference"], outs["resident"])) if a != b]
    assert not bad, bad[:5]
