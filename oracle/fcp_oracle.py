"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``recom_amd/`` imports this.

Two independent restatements live here:

* :class:`COracle` — ctypes binding of ``libfcp_oracle.so`` (``fcp_oracle.c``,
  plain C, every function cites the reference file:line it follows);
* the ``np_*`` functions — a NumPy restatement of the same TF op semantics with
  float64 accumulation ("truth") used to pin the C oracle, together with
  PyTorch-CPU ``embedding_bag`` / ``bucketize`` in ``tests/test_oracle.py``.

Parity pinning: see ``fcp_oracle.h`` — the reference owns no golden vectors for
this path, so the pin is against these independent implementations and the
committed fixtures in ``tests/golden/``.

Plans are consumed as plain dicts (``PlanSpec.to_dict()``) so this module has no
dependency on the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfcp_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "fcp_oracle.c")
    hdr = os.path.join(_HERE, "fcp_oracle.h")
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfcp_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class _Column(C.Structure):
    _fields_ = [
        ("form", C.c_int32), ("combiner", C.c_int32), ("dim", C.c_int32), ("id_source", C.c_int32),
        ("vocab", C.c_int64),
        ("table_input", C.c_int32), ("ids_input", C.c_int32), ("seg_input", C.c_int32),
        ("seg_kind", C.c_int32), ("seg_stride", C.c_int32),
        ("rows_source", C.c_int32), ("rows_arg", C.c_int32),
        ("n_boundaries", C.c_int32),
        ("boundaries", C.POINTER(C.c_float)),
        ("concat_group", C.c_int32), ("concat_slot", C.c_int32),
        ("xform_mode", C.c_int32), ("xform_n", C.c_int32),
        ("xform_lo", C.POINTER(C.c_int64)), ("xform_hi", C.POINTER(C.c_int64)),
        ("xform_substitute", C.c_int64), ("hash_buckets", C.c_int64),
        ("seg_map_n", C.c_int32), ("seg_map_sym", C.c_int32), ("seg_map_sym_slot", C.c_int32), ("seg_map_pad", C.c_int32),
        ("seg_map_mul", C.c_int64 * 4), ("seg_map_div", C.c_int64),
    ]


class _Plan(C.Structure):
    _fields_ = [
        ("n_columns", C.c_int32), ("columns", C.POINTER(_Column)),
        ("n_host_inputs", C.c_int32),
        ("host_input_ranks", C.POINTER(C.c_int32)),
        ("host_input_elem_sizes", C.POINTER(C.c_int32)),
        ("n_groups", C.c_int32),
        ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
    ]


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


class COracle:
    """ctypes binding of libfcp_oracle.so."""

    def __init__(self) -> None:
        self.lib = C.CDLL(build())
        L = self.lib
        L.orc_bucketize.restype = C.c_int32
        L.orc_bucketize.argtypes = [C.c_void_p, C.c_int32, C.c_float]
        L.orc_bucketize_array.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_gather_rows.restype = C.c_int64
        L.orc_gather_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_int64]
        L.orc_gather_scatter_rows.restype = C.c_int64
        L.orc_gather_scatter_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_int64, C.c_void_p, C.c_int64]
        L.orc_segment_offsets.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_sparse_segment_reduce.restype = C.c_int64
        L.orc_sparse_segment_reduce.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                                C.c_int64, C.c_int32, C.c_void_p, C.c_int64]
        L.orc_sparse_segment_reduce_tfcpu.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                                      C.c_void_p, C.c_int64]
        L.orc_sparse_segment_reduce_ref8x8.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                       C.c_int64, C.c_int32, C.c_void_p, C.c_int64]
        L.orc_sparse_segment_reduce_refscan.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                                        C.c_int64, C.c_int32, C.c_void_p, C.c_int64]
        L.orc_sparse_segment_reduce_refscan_assoc.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                                              C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
        L.orc_fingerprint64.restype = C.c_uint64
        L.orc_fingerprint64.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_hash_bucket_int64.restype = C.c_int64
        L.orc_hash_bucket_int64.argtypes = [C.c_int64, C.c_int64]
        L.orc_batch_col_reduction.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                              C.c_int64]
        L.orc_concat_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
        L.orc_concat_inputs.restype = C.c_int64
        L.orc_concat_inputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_process_feature_columns.restype = C.c_int64
        L.orc_process_feature_columns.argtypes = [C.POINTER(_Plan), C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_serve_throughput.restype = C.c_double
        L.orc_serve_throughput.argtypes = [C.POINTER(_Plan), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_int32]
        L.orc_group_rows.restype = C.c_int64
        L.orc_group_rows.argtypes = [C.POINTER(_Plan), C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_group_width.restype = C.c_int32
        L.orc_group_width.argtypes = [C.POINTER(_Plan), C.c_int32]
        L.orc_column_offset.restype = C.c_int32
        L.orc_column_offset.argtypes = [C.POINTER(_Plan), C.c_int32]

    # ---- single functions -----------------------------------------------------
    def bucketize(self, boundaries, values) -> np.ndarray:
        b = np.ascontiguousarray(boundaries, np.float32)
        v = np.ascontiguousarray(values, np.float32)
        out = np.empty(v.shape, np.int32)
        self.lib.orc_bucketize_array(b.ctypes.data, len(b), v.ctypes.data, v.size, out.ctypes.data)
        return out

    def gather_rows(self, table, ids):
        t = np.ascontiguousarray(table, np.float32)
        i = np.ascontiguousarray(ids, np.int64).ravel()
        out = np.empty((i.size, t.shape[1]), np.float32)
        bad = self.lib.orc_gather_rows(t.ctypes.data, t.shape[0], t.shape[1], i.ctypes.data, i.size,
                                       out.ctypes.data, t.shape[1])
        return out, bad

    def gather_scatter_rows(self, table, ids, rows, num_rows):
        t = np.ascontiguousarray(table, np.float32)
        i = np.ascontiguousarray(ids, np.int64).ravel()
        r = np.ascontiguousarray(rows, np.int64).ravel()
        out = np.empty((num_rows, t.shape[1]), np.float32)
        bad = self.lib.orc_gather_scatter_rows(t.ctypes.data, t.shape[0], t.shape[1], i.ctypes.data,
                                               r.ctypes.data, i.size, num_rows, out.ctypes.data,
                                               t.shape[1])
        return out, bad

    def segment_offsets(self, seg_ids, num_segments) -> np.ndarray:
        s = np.ascontiguousarray(seg_ids, np.int64).ravel()
        out = np.full(num_segments + 1, -12345, np.int32)
        self.lib.orc_segment_offsets(s.ctypes.data, s.size, num_segments, out.ctypes.data)
        return out

    def sparse_segment_reduce(self, table, ids, offsets, mean: bool, ref_order: bool = False):
        t = np.ascontiguousarray(table, np.float32)
        i = np.ascontiguousarray(ids, np.int64).ravel()
        o = np.ascontiguousarray(offsets, np.int32)
        S = o.size - 1
        out = np.empty((S, t.shape[1]), np.float32)
        if ref_order:
            self.lib.orc_sparse_segment_reduce_ref8x8(t.ctypes.data, t.shape[1], i.ctypes.data,
                                                      o.ctypes.data, S, int(mean), out.ctypes.data,
                                                      t.shape[1])
            return out, 0
        bad = self.lib.orc_sparse_segment_reduce(t.ctypes.data, t.shape[0], t.shape[1], i.ctypes.data,
                                                 o.ctypes.data, S, int(mean), out.ctypes.data,
                                                 t.shape[1])
        return out, bad

    def fill_parallel(self, a: np.ndarray, value: float) -> None:
        """First touch of a float32 array from all cores (orc_fill_f32): its pages spread over the sockets' memory."""
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        self.lib.orc_fill_f32.argtypes = [C.c_void_p, C.c_int64, C.c_float]
        self.lib.orc_fill_f32(a.ctypes.data, a.size, float(value))

    def sparse_segment_reduce_tfcpu(self, table, ids, offsets, mean: bool) -> np.ndarray:
        """Form 2 in the addition order of TensorFlow 2.6.2's CPU kernel (orc_sparse_segment_reduce_tfcpu); ids valid."""
        t = np.ascontiguousarray(table, np.float32)
        i = np.ascontiguousarray(ids, np.int64).ravel()
        o = np.ascontiguousarray(offsets, np.int32)
        out = np.empty((o.size - 1, t.shape[1]), np.float32)
        self.lib.orc_sparse_segment_reduce_tfcpu(t.ctypes.data, t.shape[1], i.ctypes.data, o.ctypes.data, o.size - 1, int(mean),
                                                 out.ctypes.data, t.shape[1])
        return out

    def sparse_segment_reduce_refscan(self, table, ids, row_ids, num_segments: int, mean: bool, rocprim: bool = False) -> np.ndarray:
        """Form 2 in the reference GPU kernel's own order for dim <= 20 (CUB block scan over 64-id tiles); rocprim=True swaps
        ONLY the order inside the 64-item scan for rocPRIM's (what the template compiled against hipCUB runs on)."""
        t = np.ascontiguousarray(table, np.float32)
        i = np.ascontiguousarray(ids, np.int64).ravel()
        r = np.ascontiguousarray(row_ids, np.int64).ravel()
        assert i.size == r.size
        out = np.empty((num_segments, t.shape[1]), np.float32)
        self.lib.orc_sparse_segment_reduce_refscan_assoc(t.ctypes.data, t.shape[1], i.ctypes.data, r.ctypes.data, i.size,
                                                         num_segments, int(mean), int(rocprim), out.ctypes.data, t.shape[1])
        return out

    def batch_col_reduction(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        B, R, Cc = x.shape
        out = np.empty((B, Cc), np.float32)
        self.lib.orc_batch_col_reduction(x.ctypes.data, B, R, Cc, out.ctypes.data, Cc)
        return out

    def concat_outputs(self, inputs: Sequence[np.ndarray]) -> np.ndarray:
        arrs = [np.ascontiguousarray(a, np.float32) for a in inputs]
        prefix = arrs[0].shape[0]
        dims = _i32([a.shape[1] for a in arrs])
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        out = np.empty((prefix, int(dims.sum())), np.float32)
        self.lib.orc_concat_outputs(ptrs, dims.ctypes.data, len(arrs), prefix, out.ctypes.data)
        return out

    def concat_inputs(self, tensors: Sequence[np.ndarray]):
        arrs = [np.require(np.asarray(a), requirements="C") for a in tensors]  # keeps rank-0
        n = len(arrs)
        datas = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        nbytes = np.asarray([a.nbytes for a in arrs], np.int64)
        ranks = _i32([a.ndim for a in arrs])
        dim_arrs = [np.asarray(a.shape, np.int64) for a in arrs]
        dims = (C.c_void_p * n)(*[d.ctypes.data for d in dim_arrs])
        offsets = np.empty(n, np.int32)
        shapes = np.empty(int(ranks.sum()), np.int32)
        blob = np.empty(int(nbytes.sum()), np.int8)
        total = self.lib.orc_concat_inputs(datas, nbytes.ctypes.data, ranks.ctypes.data, dims, n,
                                           blob.ctypes.data, offsets.ctypes.data, shapes.ctypes.data)
        assert total == blob.size
        return blob, offsets, shapes

    # ---- whole path -------------------------------------------------------------
    def _make_plan(self, plan: dict):
        cols = plan["columns"]
        keep: List[np.ndarray] = []
        arr = (_Column * len(cols))()
        for k, c in enumerate(cols):
            b = c.get("boundaries")
            if b is not None:
                b = np.ascontiguousarray(b, np.float32)
                keep.append(b)
            xlo = np.ascontiguousarray(c.get("xform_lo", ()), np.int64)
            xhi = np.ascontiguousarray(c.get("xform_hi", ()), np.int64)
            keep += [xlo, xhi]
            arr[k] = _Column(
                c["form"], c["combiner"], c["dim"], c["id_source"], c["vocab"], c["table_input"],
                c["ids_input"], c["seg_input"], c["seg_kind"], max(1, c["seg_stride"]),
                c["rows_source"], c["rows_arg"], 0 if b is None else len(b),
                None if b is None else b.ctypes.data_as(C.POINTER(C.c_float)),
                c["concat_group"], c["concat_slot"], c.get("xform_mode", 0), len(xlo),
                xlo.ctypes.data_as(C.POINTER(C.c_int64)) if len(xlo) else None,
                xhi.ctypes.data_as(C.POINTER(C.c_int64)) if len(xhi) else None, int(c.get("xform_substitute", 0)),
                int(c.get("hash_buckets", 0)))
            mul = [int(v) for v in c.get("seg_mul", ())]
            if mul:                                                   # segment ids through a folded SparseReshape
                arr[k].seg_map_n = len(mul)
                arr[k].seg_map_sym = int(c.get("seg_sym", -1))
                arr[k].seg_map_sym_slot = int(c.get("seg_sym_slot", 0))
                for j, v in enumerate(mul):
                    arr[k].seg_map_mul[j] = v
                arr[k].seg_map_div = int(c.get("seg_div", 1))
        ranks = _i32(plan["host_input_ranks"])
        esz = _i32(plan["host_input_elem_sizes"])
        keep += [ranks, esz]
        p = _Plan(len(cols), arr, len(ranks), ranks.ctypes.data_as(C.POINTER(C.c_int32)),
                  esz.ctypes.data_as(C.POINTER(C.c_int32)), plan["n_groups"],
                  plan.get("shard_rank", 0), plan.get("shard_world", 1))
        return p, (arr, keep)

    def serve_throughput(self, plan: dict, requests: Sequence, tables: Sequence[np.ndarray], symbols=None,
                         n_threads: int = 1, calls_per_thread: int = 1) -> float:
        """Elapsed seconds for n_threads x calls_per_thread requests served by independent
        single-threaded workers rotating over `requests` = [(blob, offsets, shapes), ...]
        (equal row counts; orc_serve_throughput)."""
        p, _keep = self._make_plan(plan)
        bl = [np.ascontiguousarray(r[0]).view(np.int8) for r in requests]
        of = [_i32(r[1]) for r in requests]
        sh = [_i32(r[2]) for r in requests]
        n = len(bl)
        bptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bl])
        optrs = (C.c_void_p * n)(*[o.ctypes.data for o in of])
        sptrs = (C.c_void_p * n)(*[x.ctypes.data for x in sh])
        sym = None if symbols is None else _i32(symbols)
        tabs = [np.ascontiguousarray(t, np.float32) for t in tables]
        tptrs = (C.c_void_p * max(1, len(tabs)))(*[t.ctypes.data for t in tabs])
        el = self.lib.orc_serve_throughput(C.byref(p), bptrs, n, optrs, sptrs, tptrs,
                                           None if sym is None else sym.ctypes.data, n_threads, calls_per_thread)
        if el < 0:
            raise ValueError("oracle: serve_throughput failed")
        return float(el)

    def serve_for(self, plan: dict, requests: Sequence, tables: Sequence[np.ndarray], symbols=None,
                  n_threads: int = 1, seconds: float = 1.0, dataflow: int = 0):
        """(requests completed, elapsed seconds): `n_threads` independent single-threaded workers serve requests,
        rotating over `requests`, for `seconds` each (orc_serve_for_dataflow).  dataflow 0: every column straight into
        the concat matrix (the checker's form); 1: TensorFlow-CPU's dataflow for the unrewritten graph — one [rows, dim]
        tensor per column op, then ConcatV2 (orc_process_feature_columns_unfused)."""
        p, _keep = self._make_plan(plan)
        bl = [np.ascontiguousarray(r[0]).view(np.int8) for r in requests]
        of = [_i32(r[1]) for r in requests]
        sh = [_i32(r[2]) for r in requests]
        n = len(bl)
        bptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bl])
        optrs = (C.c_void_p * n)(*[o.ctypes.data for o in of])
        sptrs = (C.c_void_p * n)(*[x.ctypes.data for x in sh])
        sym = None if symbols is None else _i32(symbols)
        tabs = [np.ascontiguousarray(t, np.float32) for t in tables]
        tptrs = (C.c_void_p * max(1, len(tabs)))(*[t.ctypes.data for t in tabs])
        el = C.c_double(0.0)
        self.lib.orc_serve_for_dataflow.restype = C.c_int64
        self.lib.orc_serve_for_dataflow.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double)]
        done = self.lib.orc_serve_for_dataflow(C.byref(p), bptrs, n, optrs, sptrs, tptrs, None if sym is None else sym.ctypes.data,
                                               n_threads, float(seconds), int(dataflow), C.byref(el))
        if done < 0:
            raise ValueError("oracle: serve_for failed")
        return int(done), float(el.value)

    def process_feature_columns(self, plan: dict, blob: np.ndarray, offsets, shapes,
                                tables: Sequence[np.ndarray], symbols=None, n_threads: int = 1,
                                out: Optional[List[np.ndarray]] = None, unfused: bool = False):
        """Returns (list of [rows_g, width_g] float32 matrices, n_bad_ids)."""
        p, _keep = self._make_plan(plan)
        blob = np.ascontiguousarray(blob).view(np.int8)
        offsets = _i32(offsets)
        shapes = _i32(shapes)
        sym = None if symbols is None else _i32(symbols)
        symp = None if sym is None else sym.ctypes.data
        tabs = [np.ascontiguousarray(t, np.float32) for t in tables]
        tptrs = (C.c_void_p * max(1, len(tabs)))(*[t.ctypes.data for t in tabs])
        if out is None:
            out = []
            for g in range(plan["n_groups"]):
                rows = self.lib.orc_group_rows(C.byref(p), g, shapes.ctypes.data, symp)
                if rows < 0:
                    raise ValueError(f"oracle: inconsistent rows in group {g}")
                out.append(np.zeros((rows, self.lib.orc_group_width(C.byref(p), g)), np.float32))
        optrs = (C.c_void_p * len(out))(*[o.ctypes.data for o in out])
        if unfused:   # TF-CPU's dataflow: one tensor per column op, then ConcatV2 (same values: the concat is a copy)
            n_scratch = sum(int(out[c["concat_group"]].shape[0]) * int(c["dim"]) for c in plan["columns"])
            scratch = np.zeros(max(n_scratch, 1), np.float32)
            self.lib.orc_process_feature_columns_unfused.restype = C.c_int64
            self.lib.orc_process_feature_columns_unfused.argtypes = [C.c_void_p] * 8
            bad = self.lib.orc_process_feature_columns_unfused(C.byref(p), blob.ctypes.data, offsets.ctypes.data,
                                                               shapes.ctypes.data, tptrs, symp, optrs, scratch.ctypes.data)
        else:
            bad = self.lib.orc_process_feature_columns(C.byref(p), blob.ctypes.data, offsets.ctypes.data,
                                                       shapes.ctypes.data, tptrs, symp, optrs, n_threads)
        if bad < 0:
            raise ValueError("oracle: malformed plan / shapes")
        return out, int(bad)


# =============================================================================
# NumPy restatement (float64 accumulation) — the independent pin for the C oracle
# =============================================================================

def np_fingerprint64(s: bytes) -> int:
    """TensorFlow's Fingerprint64 (FarmHash farmhashna::Hash64) for len(s) <= 32, in Python integers —
    an independent restatement of the published algorithm next to the C one (orc_fingerprint64)."""
    M = (1 << 64) - 1
    k0, k1, k2 = 0xc3a5c85c97cb3127, 0xb492b66fbe98f273, 0x9ae16a3b2f90404f

    def rot(v, sh):
        return ((v >> sh) | (v << (64 - sh))) & M if sh else v

    def f(i, n):
        return int.from_bytes(s[i:i + n], "little")

    def len16(u, v, mul):
        a = ((u ^ v) * mul) & M
        a ^= a >> 47
        b = ((v ^ a) * mul) & M
        b ^= b >> 47
        return (b * mul) & M

    n = len(s)
    if n <= 16:
        if n >= 8:
            mul, a, b = (k2 + 2 * n) & M, (f(0, 8) + k2) & M, f(n - 8, 8)
            return len16((rot(b, 37) * mul + a) & M, ((rot(a, 25) + b) * mul) & M, mul)
        if n >= 4:
            mul = (k2 + 2 * n) & M
            return len16((n + (f(0, 4) << 3)) & M, f(n - 4, 4), mul)
        if n > 0:
            y, z = (s[0] + (s[n >> 1] << 8)) & 0xFFFFFFFF, (n + (s[n - 1] << 2)) & 0xFFFFFFFF
            h = (y * k2 ^ z * k0) & M
            h ^= h >> 47
            return (h * k2) & M
        return k2
    if n > 32:
        raise NotImplementedError("Fingerprint64 of strings longer than 32 bytes is not on this path")
    mul = (k2 + 2 * n) & M
    a, b, c, d = (f(0, 8) * k1) & M, f(8, 8), (f(n - 8, 8) * mul) & M, (f(n - 16, 8) * k2) & M
    return len16((rot((a + b) & M, 43) + rot(c, 30) + d) & M, (a + rot((b + k2) & M, 18) + c) & M, mul)


def np_bucketize(boundaries, values) -> np.ndarray:
    """TF Bucketize == number of boundaries <= value (cuda_emitter.cc:233-247)."""
    return np.searchsorted(np.asarray(boundaries, np.float32), np.asarray(values, np.float32),
                           side="right").astype(np.int32)


def np_segment_offsets(seg_ids, num_segments) -> np.ndarray:
    """CSR offsets of sorted segment ids (cuda_emitter.cc:768-818)."""
    seg = np.asarray(seg_ids, np.int64)
    return np.searchsorted(seg, np.arange(num_segments + 1), side="left").astype(np.int32)


def np_sparse_segment_reduce(table, ids, offsets, mean: bool) -> np.ndarray:
    """float64-accumulated truth for SparseSegment{Sum,Mean}WithNumSegments."""
    t = np.asarray(table, np.float64)
    ids = np.asarray(ids, np.int64)
    S = len(offsets) - 1
    out = np.zeros((S, t.shape[1]), np.float64)
    for s in range(S):
        lo, hi = int(offsets[s]), int(offsets[s + 1])
        if hi > lo:
            acc = t[ids[lo:hi]].sum(axis=0)
            out[s] = acc / (hi - lo) if mean else acc
    return out


def np_process_feature_columns(plan: dict, blob: np.ndarray, offsets, shapes, tables, symbols=None
                               ) -> List[np.ndarray]:
    """Whole path in NumPy, float64 pooling (unsharded plans only)."""
    blob = np.ascontiguousarray(blob).view(np.uint8)
    ranks = plan["host_input_ranks"]
    so = np.concatenate([[0], np.cumsum(ranks)]).astype(int)

    def tensor(i, dtype):
        shape = tuple(int(x) for x in shapes[so[i]:so[i + 1]])
        n = int(np.prod(shape)) if shape else 1
        nb = n * np.dtype(dtype).itemsize
        return np.frombuffer(blob[int(offsets[i]):int(offsets[i]) + nb].tobytes(), dtype).reshape(shape)

    cols = plan["columns"]
    widths = [sum(c["dim"] for c in cols if c["concat_group"] == g) for g in range(plan["n_groups"])]
    col_off: Dict[int, int] = {}
    for g in range(plan["n_groups"]):
        acc = 0
        for _, k in sorted((c["concat_slot"], k) for k, c in enumerate(cols) if c["concat_group"] == g):
            col_off[k] = acc
            acc += cols[k]["dim"]

    def rows_of(c):
        if c["rows_source"] == 3:   # external slot: rows of the other columns of its group
            return next(rows_of(o) for o in cols if o["concat_group"] == c["concat_group"] and o["rows_source"] != 3)
        if c["rows_source"] == 0:
            return int(np.prod(shapes[so[c["ids_input"]]:so[c["ids_input"] + 1]]))
        if c["rows_source"] == 1:
            return int(symbols[c["rows_arg"]])
        return int(shapes[so[c["rows_arg"]]])

    outs = [None] * plan["n_groups"]
    for k, c in enumerate(cols):
        g = c["concat_group"]
        rows = rows_of(c)
        if outs[g] is None:
            outs[g] = np.zeros((rows, widths[g]), np.float64)
        dst = outs[g][:, col_off[k]:col_off[k] + c["dim"]]
        if c["form"] == 6:          # external slot: left as it is (zeros here)
            continue
        if c["form"] == 4:
            dst[:] = tensor(c["ids_input"], np.float32).reshape(rows, c["dim"])
            continue
        if c["form"] == 5:
            dst[:] = tensor(c["ids_input"], np.float32).astype(np.float64).sum(axis=1)
            continue
        table = np.asarray(tables[c["table_input"]], np.float64)
        if c["id_source"] == 2:
            ids = np_bucketize(c["boundaries"], tensor(c["ids_input"], np.float32).ravel()).astype(np.int64)
        else:
            ids = tensor(c["ids_input"], np.int32 if c["id_source"] == 0 else np.int64).ravel().astype(np.int64)
        if c.get("hash_buckets", 0):
            ids = np.asarray([np_fingerprint64(str(int(v)).encode()) % c["hash_buckets"] for v in ids], np.int64)
        mode = c.get("xform_mode", 0)
        kept = np.ones(ids.size, bool)
        if mode:
            inside = np.zeros(ids.size, bool)
            for lo_, hi_ in zip(c.get("xform_lo", ()), c.get("xform_hi", ())):
                inside |= (ids >= lo_) & (ids <= hi_)
            if mode == 1:
                ids = np.where(inside, ids, c.get("xform_substitute", 0))
            else:
                kept = inside
        ok = (ids >= 0) & (ids < c["vocab"]) & kept
        if c["form"] == 1:
            dst[:] = np.where(ok[:, None], table[np.where(ok, ids, 0)], 0.0)
            continue
        if c["seg_kind"] == 3:
            offs = tensor(c["seg_input"], np.int32).ravel()
        else:
            raw = tensor(c["seg_input"], np.int32 if c["seg_kind"] == 1 else np.int64).ravel()
            seg = raw[::max(1, c["seg_stride"])][:ids.size]
            mul = [int(v) for v in c.get("seg_mul", ())]
            if mul:      # a SparseReshape folded into the index expression (cuda_emitter.cc:1874-1916), row coordinate
                st = c["seg_stride"]
                sym = int(np.asarray(symbols).reshape(-1)[c["seg_sym"]]) if c.get("seg_sym", -1) >= 0 else 1
                slot = c.get("seg_sym_slot", 0) if c.get("seg_sym", -1) >= 0 else -1
                coords = raw[:ids.size * st].reshape(ids.size, st).astype(np.int64)
                lin = np.zeros(ids.size, np.int64)
                for j, v in enumerate(mul):
                    lin += coords[:, j] * (v * sym if slot == j else v)
                seg = np.where((coords[:, :len(mul)] < 0).any(axis=1), -1, lin // (c.get("seg_div", 1) * (sym if slot == 4 else 1)))
            if c["form"] == 3:
                # ScatterNd with row ids as delivered, in any order (GatherScatterRows, cuda_emitter.cc:296-345): a
                # sequential scatter of the ids the filter kept; rows outside [0, rows) are dropped
                for i in range(ids.size):
                    if kept[i] and 0 <= seg[i] < rows:
                        dst[int(seg[i])] = table[ids[i]] if ok[i] else 0.0
                continue
            offs = np_segment_offsets(seg, rows)
        if c["form"] == 2:
            mean = c["combiner"] == 2
            for s in range(rows):
                lo, hi = int(offs[s]), int(offs[s + 1])
                sel = ids[lo:hi][ok[lo:hi]]
                acc = table[sel].sum(axis=0) if sel.size else 0.0
                n_kept = int(kept[lo:hi].sum())                     # dropped ids do not count in a mean
                dst[s] = acc / n_kept if (mean and n_kept > 0) else acc
        else:  # form 3 with row offsets: the last id of the row that the filter kept wins
            for s in range(rows):
                lo, hi = int(offs[s]), int(offs[s + 1])
                live = np.nonzero(kept[lo:hi])[0]
                if live.size and ok[lo + live[-1]]:
                    dst[s] = table[ids[lo + live[-1]]]
    return outs
