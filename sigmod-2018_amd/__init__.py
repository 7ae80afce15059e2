"""sigmod-2018_amd — MI355X-native radix hash join / filter scan behind the C-ABI of
include/rhj.h (librhj.so: hand-written gfx950 kernels + the reference's own C entry
points RadixHashJoin()/Filter()).

This module is a thin ctypes binding used by tests/ and bench.py; PyTorch supplies
device memory and streams only.  There is no CPU path: loading fails loudly when
librhj.so is missing, and every call fails when no GPU is visible.

Import with importlib (the directory name carries a hyphen):
    rhj = importlib.import_module("sigmod-2018_amd")
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librhj.so")

TUPLE = np.dtype([("value", "<u8"), ("row_id", "<u8")])       # structs.h:15-19
PAIR = np.dtype([("row_idR", "<u8"), ("row_idS", "<u8")])     # structs.h:46-50

# every symbol include/rhj.h declares; tests check the library exports all of them
ABI_SYMBOLS = [
    "RadixHashJoin", "Filter", "InsertResult", "InsertRowIdResult", "GetResultNum", "FindResultRowId",
    "FindResultTuples", "FreeResult", "PrintResult", "FreeRelation", "SchedulerInit", "SchedulerDestroy",
    "rhj_set_radix_bits", "rhj_get_radix_bits", "rhj_set_empty_mode", "rhj_set_node_pairs", "rhj_set_device", "rhj_get_device",
    "rhj_set_stream", "rhj_set_force_hbm_table", "rhj_set_fused", "rhj_set_resident", "rhj_set_small", "rhj_set_lowradix", "rhj_set_count_in_pass1", "rhj_set_spec", "rhj_last_spec", "rhj_set_exact", "rhj_last_exact", "rhj_set_devices", "rhj_get_devices", "rhj_device_range", "rhj_set_devices_balance", "rhj_plan_device_ranges", "rhj_plan_device_slices", "rhj_cut_to_slice", "rhj_join_devices", "rhj_gather_pairs_devices", "rhj_set_order", "rhj_get_order", "rhj_auto_radix_bits", "rhj_set_timing", "rhj_join_device", "rhj_join_keys_device", "rhj_partition_device", "rhj_filter_device",
    "rhj_register_relation_map", "rhj_unregister_relation_map", "rhj_registered_columns", "rhj_pinned_ranges",
    "rhj_bucket_histogram_device", "rhj_select_bucket_range_device", "rhj_join_device_range", "rhj_join_device_slice", "rhj_pin_refusals",
    "rhj_release", "rhj_last_stats", "rhj_version",
]
# every symbol include/rhj_inter.h declares (device-resident intermediate results)
INTER_SYMBOLS = [
    "InitInterData", "FreeInterData", "InitInterResults", "PrintInterResults", "FreeInterResults",
    "InsertJoinToInterResults", "GetRelation", "ScanInterResults", "SelfJoin", "MergeInterNodes", "Merge",
    "CalculateQueryResults", "PrintNullResults", "AreActiveInInter", "JoinInterNode", "CartesianInterResults",
    "InsertSingleRowIdsToInterResult", "rhj_gather_tables_device", "rhj_build_relation_device", "rhj_sum_gather_device", "rhj_sum_views_device",
    "rhj_filter_eq2_device", "rhj_resident_relation", "rhj_resident_result", "rhj_resident_inter",
    "InitRelationMap", "FreeRelationMap", "PrintRelationMap", "rhj_column_stats_device",
]


class Relation(C.Structure):
    _fields_ = [("tuples", C.c_void_p), ("num_tuples", C.c_uint64)]


class Result(C.Structure):
    pass


Result._fields_ = [("buff", C.c_void_p), ("next", C.POINTER(Result)), ("current_load", C.c_uint64)]


class InterData(C.Structure):
    _fields_ = [("num_tuples", C.c_uint64), ("table", C.POINTER(C.c_void_p))]


class InterRes(C.Structure):
    pass


InterRes._fields_ = [("data", C.POINTER(InterData)), ("num_of_relations", C.c_int), ("next", C.POINTER(InterRes))]


class ColumnStats(C.Structure):
    _fields_ = [("l", C.c_uint64), ("u", C.c_uint64), ("f", C.c_double), ("d", C.c_double)]


class RelationMap(C.Structure):
    _fields_ = [("num_tuples", C.c_uint64), ("num_columns", C.c_uint64), ("columns", C.POINTER(C.c_void_p)),
                ("col_stats", C.POINTER(ColumnStats))]


class FilterPred(C.Structure):
    _fields_ = [("relation", C.c_int), ("column", C.c_int), ("value", C.c_int), ("comperator", C.c_char)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("ms_hist", "ms_scan", "ms_scatter", "ms_plan", "ms_build", "ms_count",
                                         "ms_offsets", "ms_probe", "ms_total", "ms_h2d", "ms_d2h")] + \
               [(n, C.c_uint64) for n in ("n_r", "n_s", "matches", "units", "hbm_units", "max_build", "table_slots")] + \
               [("radix_bits", C.c_int), ("reserved", C.c_int)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}
        r = self.reserved                  # path of the last join (include/rhj.h)
        d["path"] = {0: "tiled", 1: "fused", 3: "small", 4: "lowradix"}.get(r & 0xff, "?")
        d["sub_bits"], d["pass1_bits"] = (r >> 8) & 0xff, (r >> 16) & 0xff
        return d


def build():
    """Compile librhj.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def load_library(path=None):
    LIB_PATH = path or os.environ.get("RHJ_LIB") or globals()["LIB_PATH"]      # RHJ_LIB: e.g. the diagnostics build, `make instr`
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librhj.so is not built (run `make -C sigmod-2018_amd` or __graft_entry__.build()); "
                           "this package has no CPU fallback")
    # torch ships its own libamdhip64; load it first so that librhj.so binds to the HIP
    # runtime already in the process instead of pulling a second one from /opt/rocm
    # (two runtimes in one process: the second sees no device)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    u64p = C.POINTER(C.c_uint64)
    L.RadixHashJoin.argtypes = [C.POINTER(Relation), C.POINTER(Relation), C.c_void_p]
    L.RadixHashJoin.restype = C.POINTER(Result)
    L.Filter.argtypes = [C.POINTER(InterRes), C.POINTER(FilterPred), C.POINTER(RelationMap), C.POINTER(C.c_int)]
    L.Filter.restype = C.POINTER(Result)
    L.FreeResult.argtypes = [C.POINTER(Result)]
    L.GetResultNum.argtypes = [C.POINTER(Result)]
    L.InsertResult.argtypes = [C.POINTER(C.POINTER(Result)), C.c_void_p]
    L.InsertResult.restype = C.POINTER(Result)
    L.InsertRowIdResult.argtypes = [C.POINTER(C.POINTER(Result)), u64p]
    L.InsertRowIdResult.restype = C.POINTER(Result)
    L.FindResultRowId.argtypes = [C.POINTER(Result), C.c_int]
    L.FindResultRowId.restype = C.c_uint64
    L.FindResultTuples.argtypes = [C.POINTER(Result), C.c_int]
    L.FindResultTuples.restype = C.c_void_p
    L.rhj_set_radix_bits.argtypes = [C.c_int]
    L.rhj_set_empty_mode.argtypes = [C.c_int]
    L.rhj_set_node_pairs.argtypes = [C.c_uint64]
    L.rhj_set_device.argtypes = [C.c_int]
    L.rhj_set_stream.argtypes = [C.c_void_p]
    L.rhj_set_force_hbm_table.argtypes = [C.c_int]
    L.rhj_set_fused.argtypes = [C.c_int]
    L.rhj_set_resident.argtypes = [C.c_int]
    L.rhj_set_small.argtypes = [C.c_int]
    L.rhj_set_lowradix.argtypes = [C.c_int]
    L.rhj_set_count_in_pass1.argtypes = [C.c_int]
    L.rhj_set_spec.argtypes = [C.c_int]
    L.rhj_last_spec.restype = C.c_int
    if hasattr(L, "rhj_last_exact"):              # (A/B runs load earlier builds through this module too)
        L.rhj_last_exact.restype = C.c_int
    if hasattr(L, "rhj_set_devices"):
        L.rhj_set_devices.argtypes = [C.c_int]
        L.rhj_device_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.rhj_set_devices_balance.argtypes = [C.c_int]
        L.rhj_plan_device_ranges.argtypes = [u64p, u64p, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        if hasattr(L, "rhj_plan_device_slices"):
            L.rhj_plan_device_slices.argtypes = [u64p, u64p, C.c_int, C.c_int, C.POINTER(C.c_uint32), u64p]
            L.rhj_cut_to_slice.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u64p, u64p]
            L.rhj_cut_to_slice.restype = None
            L.rhj_join_device_slice.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                                C.c_void_p, C.c_uint64, u64p]
        L.rhj_join_devices.argtypes = [C.POINTER(C.c_void_p), C.c_uint64, C.POINTER(C.c_void_p), C.c_uint64, C.POINTER(C.c_void_p), u64p, u64p]
        L.rhj_gather_pairs_devices.argtypes = [C.POINTER(C.c_void_p), u64p, C.c_int, C.c_void_p, C.c_uint64, u64p]
    L.rhj_set_order.argtypes = [C.c_int]
    L.rhj_get_order.restype = C.c_int
    L.rhj_auto_radix_bits.argtypes = [C.c_uint64, C.c_uint64]
    L.rhj_auto_radix_bits.restype = C.c_int
    L.rhj_set_timing.argtypes = [C.c_int]
    L.rhj_join_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
    if hasattr(L, "rhj_join_keys_device"):
        L.rhj_join_keys_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]
    L.rhj_partition_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rhj_filter_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char, C.c_uint64, C.c_void_p, u64p]
    L.rhj_register_relation_map.argtypes = [C.POINTER(RelationMap), C.c_int]
    L.rhj_unregister_relation_map.argtypes = [C.POINTER(RelationMap), C.c_int]
    L.rhj_bucket_histogram_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.rhj_select_bucket_range_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, u64p]
    L.rhj_join_device_range.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, u64p]
    L.rhj_pin_refusals.restype = C.c_int
    L.rhj_last_stats.restype = C.POINTER(Stats)
    L.rhj_version.restype = C.c_char_p
    return L


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class RHJ:
    """Device and host entry points of librhj.so."""

    def __init__(self, device=None, use_torch_stream=True, lib_path=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the radix hash join has no CPU path")
        self.lib = load_library(lib_path)
        if device is not None:
            if self.lib.rhj_set_device(int(device)) != 0 and self.lib.rhj_get_device() != int(device):
                raise RuntimeError("librhj.so already runs on device %d: one library context per process "
                                   "(rhj_set_device(%d) refused)" % (self.lib.rhj_get_device(), int(device)))
            torch.cuda.set_device(int(device))
        self.dev = torch.device("cuda", torch.cuda.current_device())
        if use_torch_stream:
            self.lib.rhj_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))

    # ---- knobs
    def set_bits(self, bits):
        if self.lib.rhj_set_radix_bits(int(bits)) != 0:
            raise ValueError("radix bits out of range: %r" % (bits,))

    def stats(self):
        return self.lib.rhj_last_stats().contents.as_dict()

    # ---- device-resident API
    def to_device(self, rel):
        """numpy TUPLE array -> int64 tensor [n,2] on the device (same bytes)."""
        rel = np.ascontiguousarray(rel, dtype=TUPLE)
        t = self.torch.from_numpy(rel.view(np.int64).reshape(-1, 2).copy())
        return t.to(self.dev)

    def join_device(self, dR, dS, capacity=None, count_only=False, bucket_range=None):
        """dR, dS: int64 tensors [n,2] (value,row_id).  Returns (pairs tensor [M,2], matches).
        bucket_range = (lo, hi): only the buckets [lo, hi) of the current radix (rhj_join_device_range: one rank's share of a
        sharded join; the partition drops the other buckets while it reads the relations); (lo, hi, first_skip, last_end): a share
        cut inside its first / last bucket (rhj_join_device_slice)."""
        torch = self.torch
        nR, nS = dR.shape[0], dS.shape[0]
        m = C.c_uint64(0)

        def call(out_ptr, cap):
            if bucket_range is None:
                rc = self.lib.rhj_join_device(dR.data_ptr(), nR, dS.data_ptr(), nS, out_ptr, cap, C.byref(m))
            elif len(bucket_range) == 4:
                rc = self.lib.rhj_join_device_slice(dR.data_ptr(), nR, dS.data_ptr(), nS, int(bucket_range[0]), int(bucket_range[1]),
                                                    int(bucket_range[2]), int(bucket_range[3]), out_ptr, cap, C.byref(m))
            else:
                rc = self.lib.rhj_join_device_range(dR.data_ptr(), nR, dS.data_ptr(), nS, int(bucket_range[0]), int(bucket_range[1]),
                                                    out_ptr, cap, C.byref(m))
            if rc < 0:
                raise RuntimeError("rhj_join_device failed (%d)" % rc)

        if count_only:
            call(None, 0)
            return None, m.value
        if capacity is None:
            call(None, 0)
            capacity = m.value
        out = torch.empty((max(capacity, 1), 2), dtype=torch.int64, device=self.dev)
        call(out.data_ptr(), capacity)
        return out[:min(m.value, capacity)], m.value

    def partition_device(self, d_in, bits=None):
        torch = self.torch
        if bits is not None:
            self.set_bits(bits)
        bits = self.lib.rhj_get_radix_bits()
        out = torch.empty_like(d_in)
        hist = np.zeros(1 << bits, dtype=np.uint64)
        psum = np.zeros(1 << bits, dtype=np.int64)
        rc = self.lib.rhj_partition_device(d_in.data_ptr(), d_in.shape[0], out.data_ptr(), _np_ptr(hist), _np_ptr(psum))
        if rc < 0:
            raise RuntimeError("rhj_partition_device failed (%d)" % rc)
        return out, hist, psum

    def filter_device(self, d_col, op, value, d_sel=None):
        torch = self.torch
        n = d_col.shape[0] if d_sel is None else d_sel.shape[0]
        out = torch.empty(max(n, 1), dtype=torch.int64, device=self.dev)
        hits = C.c_uint64(0)
        k = int(value) & ((1 << 64) - 1)
        rc = self.lib.rhj_filter_device(d_col.data_ptr(), d_sel.data_ptr() if d_sel is not None else None, n,
                                        op.encode(), k, out.data_ptr(), C.byref(hits))
        if rc < 0:
            raise RuntimeError("rhj_filter_device failed (%d)" % rc)
        return out[:hits.value]

    def pairs_to_numpy(self, t):
        a = t.cpu().numpy()
        return np.ascontiguousarray(a).view(np.uint64).reshape(-1, 2).copy().view(PAIR).reshape(-1)

    # ---- host ABI: the reference's own signatures
    def RadixHashJoin(self, R, S, with_info=False):
        R = np.ascontiguousarray(R, dtype=TUPLE)
        S = np.ascontiguousarray(S, dtype=TUPLE)
        relR, relS = Relation(R.ctypes.data, len(R)), Relation(S.ctypes.data, len(S))
        res = self.lib.RadixHashJoin(C.byref(relR), C.byref(relS), None)
        null = not bool(res)
        chunks, loads = [], []
        p = res
        while p:
            n = p.contents.current_load
            loads.append(int(n))
            if n:
                buf = (C.c_char * (n * 16)).from_address(p.contents.buff)
                chunks.append(np.frombuffer(buf, dtype=PAIR).copy())
            p = p.contents.next
        total = self.lib.GetResultNum(res) if not null else 0
        if not null:
            self.lib.FreeResult(res)
        out = np.concatenate(chunks) if chunks else np.zeros(0, dtype=PAIR)
        assert total == len(out)
        return (out, {"null": null, "loads": loads}) if with_info else out

    def Filter(self, columns, rel_rows, column, op, value, sel=None, with_info=False):
        """columns: list of u64 numpy columns of ONE relation (relation_map layout);
        sel: optional row-id vector that puts the relation into the intermediate result."""
        cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
        ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        rm = RelationMap(rel_rows, len(cols), ptrs, None)
        qrel = (C.c_int * 1)(0)
        fp = FilterPred(0, column, int(value), op.encode())
        table = (C.c_void_p * 1)(None)
        if sel is not None:
            sel = np.ascontiguousarray(sel, dtype=np.uint64)
            table[0] = sel.ctypes.data
        data = InterData(len(sel) if sel is not None else 0, table)
        ir = InterRes(C.pointer(data), 1, None)
        res = self.lib.Filter(C.byref(ir), C.byref(fp), C.byref(rm), qrel)
        null = not bool(res)
        chunks, loads = [], []
        p = res
        while p:
            n = p.contents.current_load
            loads.append(int(n))
            buf = (C.c_char * (n * 8)).from_address(p.contents.buff)
            chunks.append(np.frombuffer(buf, dtype=np.uint64).copy())
            p = p.contents.next
        if not null:
            self.lib.FreeResult(res)
        out = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint64)
        return (out, {"null": null, "loads": loads}) if with_info else out
