"""Python face of the native measurement harness (``libfcp_harness.so``).

The timing loops themselves are C++ (``csrc/fcp_harness.hip``), mirroring the
reference's ``benchmark_multi_thread`` protocol
(``examples/cc/recom_examples.patch:98-263``): a shared model, ``serve_workers``
host threads, warm-up, N timed requests, latency + throughput.  Python only
prepares the model, the requests and the result record.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import lib as _lib
from .ops import Plan, concat_inputs, pack_as_staged
from .synth import Request, SynthModel

_HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS_PATH = os.path.join(os.environ.get("FCP_LIB_DIR", _HERE), "libfcp_harness.so")
_h = None


def load() -> C.CDLL:
    global _h
    if _h is None:
        _lib.load()
        if not os.path.exists(HARNESS_PATH) and "FCP_LIB_DIR" not in os.environ:
            _lib._build_in_tree()
        if not os.path.exists(HARNESS_PATH):
            raise ImportError(f"{HARNESS_PATH} not built; run __graft_entry__.build()")
        H = C.CDLL(HARNESS_PATH)
        H.fcp_harness_create.argtypes = [C.c_void_p, C.POINTER(_lib.ProcessArgs), C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_void_p)]
        H.fcp_harness_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float),
                                      C.c_void_p]
        H.fcp_harness_run_graph.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double),
                                            C.POINTER(C.c_float)]
        H.fcp_harness_destroy.argtypes = [C.c_void_p]
        if hasattr(H, "fcp_harness_stream"):
            H.fcp_harness_stream.argtypes = [C.c_void_p, C.c_int]
            H.fcp_harness_stream.restype = C.c_void_p
        if hasattr(H, "fcp_harness_run_private"):
            H.fcp_harness_run_private.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float)]
        if hasattr(H, "fcp_harness_run_private_threads"):
            H.fcp_harness_run_private_threads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float)]
        H.fcp_harness_copy_probe.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_float)]
        H.fcp_harness_gather_probe.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                               C.POINTER(C.c_double)]
        H.fcp_harness_bw_probe.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
        _h = H
    return _h


class ServingHarness:
    """A plan + resident tables + a set of resident request blobs, driven by the
    native loop.  Inputs are already in HBM when a timed region starts."""

    def __init__(self, model: SynthModel, device: int = 0, n_requests: int = 16, arena_ring: int = 6,
                 n_threads: int = 1, tables=None, spec=None, seed0: int = 0) -> None:
        import torch
        self.torch = torch
        self.model = model
        self.dev = torch.device("cuda", device)
        self.H = load()
        self.spec = spec or model.spec
        self.plan = Plan(self.spec, device)
        self.tables = tables if tables is not None else model.torch_tables(self.dev, self.spec.shard_rank,
                                                                           self.spec.shard_world)
        self.requests: List[Request] = [model.make_request(seed0 + s) for s in range(n_requests)]
        self._keep = []
        self.packed = []
        args = (_lib.ProcessArgs * n_requests)()
        tptrs = (C.c_void_p * max(1, len(self.tables)))(*[t.data_ptr() for t in self.tables])
        self._keep.append(tptrs)
        stage_modes = getattr(model, "stage_modes", None)   # a staged model: the blob as the staged ConcatInputs lays it out
        for i, r in enumerate(self.requests):
            blob, offsets, shapes = pack_as_staged(r.inputs, stage_modes) if stage_modes else concat_inputs(r.inputs)
            d_blob = torch.from_numpy(blob).to(self.dev)
            sym = None if r.symbols is None else np.ascontiguousarray(r.symbols, np.int32)
            self._keep += [d_blob, offsets, shapes, sym]
            self.packed.append((blob, offsets, shapes))
            args[i] = _lib.ProcessArgs(
                d_blob.data_ptr(), blob.nbytes, offsets.ctypes.data_as(C.POINTER(C.c_int32)),
                shapes.ctypes.data_as(C.POINTER(C.c_int32)), tptrs, None,
                None if sym is None else sym.ctypes.data_as(C.POINTER(C.c_int32)), None,
                _lib.ALLOC_FN(), None, _lib.ALLOC_FN(), None)
        self._args = args
        handle = C.c_void_p()
        torch.cuda.synchronize()
        with torch.cuda.device(self.dev):        # the native harness creates its streams and arenas on the CURRENT device
            _lib.check(self.H.fcp_harness_create(self.plan.handle, args, n_requests, arena_ring, n_threads,
                                                 C.byref(handle)), "fcp_harness_create")
        self.handle = handle
        self.n_threads = n_threads

    def run(self, steps: int, per_request: bool = False):
        """Returns (wall_ms, dev_ms, iter_ms or None) for exactly `steps` requests per worker."""
        wall, dev = C.c_double(), C.c_float()
        it = np.zeros(steps, np.float32) if per_request else None
        _lib.check(self.H.fcp_harness_run(self.handle, steps, C.byref(wall), C.byref(dev),
                                          None if it is None else it.ctypes.data), "fcp_harness_run")
        return wall.value, dev.value, it

    def caller_stream(self, worker: int = 0) -> int:
        """The HIP stream worker ``worker`` issues on (``run_private``: worker 0's is the caller's stream)."""
        return int(self.H.fcp_harness_stream(self.handle, worker) or 0)

    def run_private(self, steps: int, depth: int = 3, threads: int = 1):
        """``threads`` host threads and ONE caller stream (the reference's serve workers share one Session, hence one
        compute stream) over a plan with private streams (``self.plan.set_private_streams`` first) — or without, which
        is "stream order + the same consumer": every thread issues ``steps`` requests, request k's consumer —
        ``fcp_result_wait`` + a reader kernel on the caller's stream — is enqueued ``depth - 1`` of that thread's requests
        behind it.  The harness needs ``n_threads >= threads`` (an arena ring per thread).  Returns (wall_ms, dev_ms) over
        all ``threads * steps`` requests."""
        wall, dev = C.c_double(), C.c_float()
        _lib.check(self.H.fcp_harness_run_private_threads(self.handle, steps, depth, threads, C.byref(wall), C.byref(dev)),
                   "fcp_harness_run_private_threads")
        return wall.value, dev.value

    def run_graph(self, steps: int, group: int):
        """`group` requests captured once into a HIP graph, replayed steps / group times
        (fixed-shape models only).  Returns (wall_ms, dev_ms)."""
        wall, dev = C.c_double(), C.c_float()
        _lib.check(self.H.fcp_harness_run_graph(self.handle, steps, group, C.byref(wall), C.byref(dev)),
                   "fcp_harness_run_graph")
        return wall.value, dev.value

    def verify_resident(self, max_columns: int = 48, pooled_rows: int = 12) -> dict:
        """Every resident request is served ONCE through the library and checked against the closed-form table contents
        (``synth.hash_rows``; no oracle involved) before anything is timed: one-hot columns are exact copies of table
        rows, pooled columns exact fp32 sums / means in id order of them (a sample of columns and rows; columns with id
        transforms, hashing or ScatterNd are left to the test suite).  A wrong kernel is refused here, not timed.  It
        also means that every request blob has been read once before the warm-up starts."""
        from . import synth
        from .ops import FeatureColumnProcess
        from .plan import (COMBINER_MEAN, FORM_GATHER, FORM_PASSTHROUGH, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, SEG_CSR_I32,
                           XFORM_NONE)
        torch = self.torch
        spec = self.spec
        if spec.shard_world > 1 or any(not hasattr(t, "seed") for t in self.model.tables):
            return {"checked": 0, "note": "sharded or table contents not closed-form: not verified here"}
        op = FeatureColumnProcess(spec, self.dev.index or 0, plan=self.plan)
        offs = spec.column_offsets()
        plain = [k for k, c in enumerate(spec.columns) if c.xform_mode == XFORM_NONE and not c.hash_buckets and not len(c.seg_mul)]
        dense = [k for k in plain if spec.columns[k].form in (FORM_GATHER, FORM_PASSTHROUGH)]
        pooled = [k for k in plain if spec.columns[k].form == FORM_SEGMENT_REDUCE]
        step = max(1, len(dense) // max_columns)
        dense, pooled = dense[::step][:max_columns], pooled[::max(1, len(pooled) // 8)][:8]
        n_checked = 0
        for (blob, offsets, shapes), r, keep in zip(self.packed, self.requests, self._keep[1::4]):
            out = op(keep, offsets, shapes, self.tables, r.symbols)
            torch.cuda.synchronize()
            groups = [g.cpu().numpy() for g in out.groups]
            for k in dense + pooled:
                c = spec.columns[k]
                got = groups[c.concat_group][:, offs[k]:offs[k] + c.dim]
                raw = np.asarray(r.inputs[c.ids_input])
                if c.form == FORM_PASSTHROUGH:
                    ok = np.array_equal(got, raw.reshape(got.shape))
                else:
                    ids = raw.reshape(-1)
                    if c.id_source == IDS_F32_BUCKETIZE:
                        b = np.asarray(c.boundaries, np.float32)
                        ids = np.where(np.isnan(ids), b.size, np.searchsorted(b, ids, side="right")).astype(np.int64)
                    seed = self.model.tables[c.table_input].seed
                    valid = (ids >= 0) & (ids < c.vocab)
                    if c.form == FORM_GATHER:
                        want = np.where(valid[:, None], synth.hash_rows(seed, np.where(valid, ids, 0), c.dim), np.float32(0))
                        ok = np.array_equal(got, want)
                    else:
                        seg = np.asarray(r.inputs[c.seg_input])
                        if c.seg_kind == SEG_CSR_I32:
                            csr = seg.reshape(-1).astype(np.int64)
                        else:
                            rows_of = seg.reshape(-1)[::max(1, c.seg_stride)][:ids.size]
                            csr = np.searchsorted(rows_of, np.arange(got.shape[0] + 1), side="left")
                        ok = True
                        for b_ in range(min(pooled_rows, got.shape[0])):
                            acc = np.zeros(c.dim, np.float32)
                            sel = ids[csr[b_]:csr[b_ + 1]]
                            for v in sel[(sel >= 0) & (sel < c.vocab)]:
                                acc = acc + synth.hash_rows(seed, np.asarray([v]), c.dim)[0]        # fp32, id order
                            if c.combiner == COMBINER_MEAN and sel.size:
                                acc = acc / np.float32(sel.size)
                            ok = ok and np.array_equal(got[b_], acc)
                if not ok:
                    raise RuntimeError(f"resident request check failed: column {k} (form {c.form}) differs from the closed-form tables")
                n_checked += 1
        return {"checked": n_checked, "requests": len(self.requests), "one_hot_columns": len(dense), "pooled_columns": len(pooled),
                "note": "each resident request served once and compared with closed-form table rows (exact) before the warm-up"}

    def algorithmic_bytes(self) -> dict:
        """Mean algorithmic bytes per request over the resident requests."""
        acc = None
        for (blob, offsets, shapes), r in zip(self.packed, self.requests):
            b = self.spec.algorithmic_bytes(shapes, r.symbols)
            acc = b if acc is None else {k: acc[k] + b[k] for k in b}
        return {k: v / len(self.requests) for k, v in acc.items()}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.H.fcp_harness_destroy(self.handle)
            self.handle = None
        self.plan.close()


def copy_probe(nbytes: int = 1 << 30, iters: int = 20) -> float:
    """Measured float4 copy bandwidth (read + write bytes / s)."""
    ms = C.c_float()
    _lib.check(load().fcp_harness_copy_probe(nbytes, iters, C.byref(ms)), "copy probe")
    return 2.0 * nbytes / (ms.value * 1e-3)


def gather_probe(row_bytes: int, nbytes: int = 8 << 30, depth: int = 8, iters: int = 5) -> float:
    """Useful bytes / s of random `row_bytes`-sized row gathers from an `nbytes` buffer (nothing written)."""
    ms, useful = C.c_float(), C.c_double()
    _lib.check(load().fcp_harness_gather_probe(nbytes, row_bytes, depth, iters, C.byref(ms), C.byref(useful)),
               "gather probe")
    return useful.value / (ms.value * 1e-3)


def write_probe(nbytes: int = 4 << 30, iters: int = 5) -> float:
    """Bytes / s of non-temporal stores in the kernel's store shape (1-KiB chunks per wave)."""
    ms = C.c_float()
    _lib.check(load().fcp_harness_bw_probe(3, nbytes, iters, C.byref(ms)), "write probe")
    return nbytes / (ms.value * 1e-3)


def read_probe(nbytes: int = 4 << 30, iters: int = 5) -> float:
    ms = C.c_float()
    _lib.check(load().fcp_harness_bw_probe(0, nbytes, iters, C.byref(ms)), "read probe")
    return nbytes / (ms.value * 1e-3)



def sla_throughput_search(make_model, sla_ms: float, serve_workers: int = 1, num_iterations: int = 100, device: int = 0,
                          start_batch: int = 16, max_batch: int = 0, arena_budget_bytes: int = 0, log=None) -> dict:
    """The reference's throughput benchmark for THIS path (``benchmark_throughput``, examples/cc/recom_examples.patch:264-465):
    a pressure test that grows the batch until the average latency of a request reaches the service-level agreement and
    reports the largest batch that stayed under it with its throughput.  Same protocol — one warm-up request, then
    ``serve_workers`` threads x ``num_iterations`` requests of the SAME inputs (``:396-420``), latency = wall time /
    ``num_iterations``, throughput = workers x iterations x batch / wall time — and the same search (``:422-463``): start at 16;
    under the SLA the batch grows by itself below 50 % of the SLA, by a 5th / 10th / 30th / 100th of itself below 70 / 80 / 90 /
    100 % (at least 5); at or over the SLA it grows by 50 while it is <= 512 and the search ends beyond that.  The reference's
    thresholds are milliseconds against its 100-ms default for a whole model; here they are fractions of ``sla_ms`` (the
    embedding stage alone is 27 us at batch 512).  ``make_model(batch)`` builds the workload at one batch size (tables are
    shared between the harnesses: same vocabulary); the search also ends when the next arena would not fit
    ``arena_budget_bytes`` (reported as ``ended_by``)."""
    import torch
    batch, tables, steps = start_batch, None, []
    best = {"max_batch_size": 0, "max_throughput": 0.0}
    ended_by = "sla"
    while True:
        model = make_model(batch)
        width = sum(model.spec.group_width(g) for g in range(model.spec.n_groups))
        if arena_budget_bytes and 4 * batch * width * serve_workers > arena_budget_bytes:
            ended_by = "arena memory"
            break
        h = ServingHarness(model, device=device, n_requests=1, arena_ring=1, n_threads=serve_workers, tables=tables)
        tables = h.tables
        h.run(1)                                                  # "Start warmup predict"
        wall_ms, _, _ = h.run(num_iterations)
        h.close()
        latency = wall_ms / num_iterations
        throughput = serve_workers * num_iterations * batch / (wall_ms * 1e-3)
        steps.append({"batch": batch, "avg_latency_ms": latency, "throughput": throughput})
        if log:
            log(f"batch size: {batch}, avg latency: {latency:.4f} ms, throughput: {throughput:.0f}")
        if latency >= sla_ms:
            if batch > 512:
                break
            batch += 50
        else:
            best = {"max_batch_size": batch, "max_throughput": throughput}
            frac = latency / sla_ms
            inc = batch if frac < 0.5 else batch // 5 if frac < 0.7 else batch // 10 if frac < 0.8 else batch // 30 if frac < 0.9 else batch // 100
            batch += max(5, inc)
        if max_batch and batch > max_batch:
            ended_by = "max_batch"
            break
    del tables
    torch.cuda.empty_cache()
    return {**best, "sla_ms": sla_ms, "serve_workers": serve_workers, "num_iterations": num_iterations, "ended_by": ended_by,
            "search": steps}
