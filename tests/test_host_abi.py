"""CPU: the C-ABI library loads, exports every symbol include/rhj.h declares, and its
host-only pieces (result lists, scheduler tokens, knobs) behave like results.c /
scheduler.c.  No compute entry point is called here (they need a GPU)."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mod():
    m = importlib.import_module("sigmod-2018_amd")
    if not os.path.exists(m.LIB_PATH):
        m.build()
    return m


@pytest.fixture(scope="module")
def lib(mod):
    return mod.load_library()


def declared_in(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = hdr.split("#ifdef RHJ_REFERENCE_NAMES")[0]
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)                 # comments out
    return set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{()]*\)\s*;", hdr))


def test_exports_every_declared_symbol(mod, lib):
    for header, names in (("rhj.h", mod.ABI_SYMBOLS), ("rhj_inter.h", mod.INTER_SYMBOLS)):
        declared = declared_in(header)
        assert declared == set(names), (header, declared ^ set(names))
        for name in declared:
            assert hasattr(lib, name), name


def test_layouts_match_the_reference_structs(mod):
    assert C.sizeof(mod.Relation) == 16 and C.sizeof(mod.Result) == 24 and C.sizeof(mod.FilterPred) == 16
    assert C.sizeof(mod.RelationMap) == 32 and C.sizeof(mod.InterRes) == 24 and C.sizeof(mod.InterData) == 16
    assert mod.TUPLE.itemsize == 16 and mod.PAIR.itemsize == 16


def test_result_list_api_behaves_like_results_c(mod, lib):
    head = C.POINTER(mod.Result)()
    n = 8192 * 2 + 5                      # RESULT_MAX_BUFFER / 16 = 8192 pairs per node (structs.h:9)
    pair = (C.c_uint64 * 2)()
    for i in range(n):
        pair[0], pair[1] = i, 2 * i
        lib.InsertResult(C.byref(head), pair)
    loads, p = [], head
    while p:
        loads.append(p.contents.current_load)
        p = p.contents.next
    assert loads == [8192, 8192, 5] and lib.GetResultNum(head) == n
    for i in (0, 8191, 8192, n - 1):
        t = C.cast(lib.FindResultTuples(head, i), C.POINTER(C.c_uint64))
        assert (t[0], t[1]) == (i, 2 * i)
    assert lib.FindResultTuples(head, -1) is None
    lib.FreeResult(head)

    ids = C.POINTER(mod.Result)()
    m = 131072 + 3                        # RESULT_FINAL_BUFFER / 8 ids per node (structs.h:10)
    v = C.c_uint64()
    for i in range(m):
        v.value = 3 * i
        lib.InsertRowIdResult(C.byref(ids), C.byref(v))
    assert lib.GetResultNum(ids) == m and ids.contents.current_load == 131072
    assert lib.FindResultRowId(ids, 131073) == 3 * 131073
    lib.FreeResult(ids)
    lib.FreeResult(None)


def test_scheduler_tokens_and_knobs(lib):
    sched = C.c_void_p()
    lib.SchedulerInit.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.SchedulerDestroy.argtypes = [C.c_void_p]
    assert lib.SchedulerInit(C.byref(sched), 4) == 0 and sched.value
    assert lib.SchedulerDestroy(sched) == 0
    keep = lib.rhj_get_radix_bits()
    assert lib.rhj_set_radix_bits(0) == -1 and lib.rhj_set_radix_bits(16) == -1
    assert lib.rhj_set_radix_bits(12) == 0 and lib.rhj_get_radix_bits() == 12
    lib.rhj_set_radix_bits(keep)
    assert b"gfx950" in lib.rhj_version()


def test_device_ranges_are_the_sharding_modules_equal_ranges(lib):
    """rhj_device_range (the planning half of rhj_join_devices / RHJ_DEVICES: which buckets device d of n joins) against
    shard.equal_ranges, the ranges the torch.distributed path and its gloo tests use: contiguous, complete, equal width."""
    shard = importlib.import_module("sigmod-2018_amd.shard")
    lo, hi = C.c_uint32(0), C.c_uint32(0)
    for bits in range(1, 16):
        for n in range(1, 9):
            got = []
            for d in range(n):
                assert lib.rhj_device_range(bits, n, d, C.byref(lo), C.byref(hi)) == 0
                got.append((lo.value, hi.value))
            assert got == shard.equal_ranges(bits, n), (bits, n)
            assert got[0][0] == 0 and got[-1][1] == 1 << bits and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    # the histogram-balanced plan against shard.bucket_ranges: uniform, Zipf-like, one hot bucket, empty buckets, all in one
    rng = np.random.default_rng(7)
    for bits in (1, 4, 8, 12):
        bins = 1 << bits
        for hist in (rng.integers(0, 1000, (2, bins)), (rng.zipf(1.3, (2, bins)) % 100000), np.zeros((2, bins), dtype=np.int64),
                     np.vstack([np.eye(1, bins, bins // 3, dtype=np.int64)[0] * 10**9 + 3, np.ones(bins, dtype=np.int64)])):
            hr, hs = (np.ascontiguousarray(h, dtype=np.uint64) for h in hist)
            for n in range(1, 9):
                cuts = (C.c_uint32 * (n + 1))()
                assert lib.rhj_plan_device_ranges(hr.ctypes.data_as(C.POINTER(C.c_uint64)), hs.ctypes.data_as(C.POINTER(C.c_uint64)), bits, n, cuts) == 0
                assert [(cuts[d], cuts[d + 1]) for d in range(n)] == [(int(a), int(b)) for a, b in shard.bucket_ranges(hr, hs, n)], (bits, n)
    # the plan that may cut INSIDE a hot bucket against shard.bucket_slices: one dominant bucket, two, Zipf-like, uniform, empty
    for bits in (1, 4, 8, 12):
        bins = 1 << bits
        hot1 = np.vstack([np.full(bins, 1000, dtype=np.int64), np.full(bins, 1000, dtype=np.int64)]); hot1[1, bins // 3] = 5_000_000
        hot2 = hot1.copy(); hot2[0, bins - 1] = 3_000_000; hot2[1, bins - 1] = 2_999_999
        for hist in (hot1, hot2, rng.zipf(1.2, (2, bins)) % 10**7, rng.integers(0, 1000, (2, bins)), np.zeros((2, bins), dtype=np.int64),
                     np.vstack([np.eye(1, bins, 0, dtype=np.int64)[0] * 10**9, np.eye(1, bins, 0, dtype=np.int64)[0] * 7])):
            hr, hs = (np.ascontiguousarray(h, dtype=np.uint64) for h in hist)
            for n in range(1, 9):
                cb, co = (C.c_uint32 * (n + 1))(), (C.c_uint64 * (n + 1))()
                assert lib.rhj_plan_device_slices(hr.ctypes.data_as(C.POINTER(C.c_uint64)), hs.ctypes.data_as(C.POINTER(C.c_uint64)), bits, n, cb, co) == 0
                cuts = [(cb[d], co[d]) for d in range(n + 1)]
                assert cuts[0] == (0, 0) and cuts[-1] == (bins, 0) and cuts == sorted(cuts), (bits, n, cuts)
                got = []
                for d in range(n):
                    skip, end = C.c_uint64(0), C.c_uint64(0)
                    lib.rhj_cut_to_slice(cb[d], co[d], cb[d + 1], co[d + 1], C.byref(lo), C.byref(hi), C.byref(skip), C.byref(end))
                    got.append((lo.value, hi.value, skip.value, end.value))
                assert got == shard.bucket_slices(hr, hs, n), (bits, n)
                for b, off in cuts:                                  # a cut inside a bucket: a hot one, in front of its last probe tuple
                    if off:
                        w, tot = int(hr[b]) + int(hs[b]), int(hr.sum()) + int(hs.sum())
                        assert off % 256 == 0 and off < max(int(hr[b]), int(hs[b])) and w * 2 * n >= tot and hr[b] and hs[b]
            if hist is hot1 and bits >= 4:                           # the dominant bucket is shared: no device gets more than ~2 / n
                sl = shard.bucket_slices(hr, hs, 4)
                assert sum(1 for s_ in sl if s_[2] or s_[3]) >= 2
    assert lib.rhj_device_range(4, 9, 0, C.byref(lo), C.byref(hi)) == -1       # at most eight devices
    assert lib.rhj_device_range(4, 2, 2, C.byref(lo), C.byref(hi)) == -1
    assert lib.rhj_device_range(16, 2, 0, C.byref(lo), C.byref(hi)) == -1


def test_order_mode_and_its_radix_rule(lib):
    """rhj_set_order / rhj_get_order and the width order mode "any" picks (include/rhj.h): build sides of ~16 K tuples per
    bucket on comparable sizes (12 bits instead of 13 while a bucket stays below 28 K), ~6.5 K when the probe side is four times the build side or more, and for small relations
    up to 8 bits while an average bucket keeps 512 tuples.  Pure host logic: no device is touched."""
    assert lib.rhj_get_order() == 0
    lib.rhj_set_order(1)
    assert lib.rhj_get_order() == 1
    lib.rhj_set_order(0)
    assert lib.rhj_get_order() == 0
    rule = lib.rhj_auto_radix_bits
    assert rule(100_000_000, 100_000_000) == 12
    assert rule(100_000_000, 1_000_000_000) == 14 and rule(1_000_000_000, 100_000_000) == 14
    assert rule(1_000_000, 1_000_000) == 8 and rule(100_000, 100_000) == 7
    assert rule(1, 1) == 1 and rule(1000, 5) == 1
    assert rule(1 << 31, 1 << 31) == 15                     # never beyond the widest partition
    for n in (3, 1000, 77_777, 5_000_000, 300_000_000):
        for m in (n, 5 * n):
            assert 1 <= rule(n, m) <= 15 and rule(n, m) == rule(m, n)


@pytest.mark.parametrize("target", ["asan", "tsan"])
def test_host_side_is_clean_under_the_sanitizers(target):
    """`make asan` / `make tsan` (sigmod-2018_amd/Makefile): rhj_abi.c (result lists, node pool, RadixHashJoin()'s list
    assembly) and rhj_host.cpp (API lock, the ring-of-staging-blocks mover threads) built with -fsanitize=address,undefined /
    thread and driven by tests/host_san_driver.cpp — the device replaced by a host stand-in, the reference entry points called
    from four threads at once.  A sanitizer report ends the driver with a non-zero exit code."""
    import subprocess
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "sigmod-2018_amd"), target], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode()
    assert res.returncode == 0 and "host_san_driver ok" in out, out[-3000:]
    assert "Sanitizer" not in out and "runtime error" not in out, out[-3000:]
