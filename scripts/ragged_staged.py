#!/usr/bin/env python3
"""RAGGED (BASELINE configs[3]: SparseTensor multi-hot features) with the request on the HOST when the clock starts: what the
staging step ships and what the device then has to do.  Three ways through fcp_stager_stage_ex, the C entry points called
directly with arguments marshalled once (the loop is Python, ~10 us per iteration, against 100-500 us of work):
  raw     : byte copy of every tensor (= Addons>ConcatInputs); the device runs the segment-offset pre-pass
  narrow  : int64 ids / indices shipped as int32
  staged  : + sorted row ids -> CSR offsets on the host while packing (FCP_STAGE_SEG_TO_CSR): no pre-pass
each with the H2D copy and with FCP_STAGER_ZERO_COPY.  Also the device time of the fused kernel on resident inputs of
each form.  GPU box: [PACK_THREADS=8] python scripts/ragged_staged.py [steps] [ragged|e|f]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recom_amd import lib as _lib  # noqa: E402
from recom_amd import synth  # noqa: E402
from recom_amd.ops import FeatureColumnProcess, RequestStager  # noqa: E402
from recom_amd.plan import STAGE_COPY  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
which = sys.argv[2] if len(sys.argv) > 2 else "ragged"          # ragged | e | f
dev = torch.device("cuda", 0)
m = synth.model_ragged(seg="indices") if which == "ragged" else synth.model_ae(which)
print(f"== {m.name}: {m.description}; pack threads {os.environ.get('PACK_THREADS', '8')}")
L = _lib.load()
tabs = m.torch_tables(dev)
tptrs = (C.c_void_p * len(tabs))(*[t.data_ptr() for t in tabs])
reqs = [m.make_request(s) for s in range(8)]
nspec, nflags = m.spec.narrowed()
sspec, smodes, rows_col = m.spec.staged()
VARIANTS = {"raw": (m.spec, [STAGE_COPY] * m.spec.n_host_inputs), "narrow": (nspec, [1 if f else 0 for f in nflags]), "staged": (sspec, smodes)}
stream = torch.cuda.current_stream(dev).cuda_stream
arena_ring = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(4)]
turn = [0]


@_lib.ALLOC_FN
def alloc(_ctx, nbytes):
    turn[0] += 1
    return arena_ring[turn[0] % 4].data_ptr()


for name, (spec, modes) in VARIANTS.items():
    op = FeatureColumnProcess(spec, 0)
    for zero_copy in (False, True):
        st = RequestStager(32 << 20, m.spec.n_host_inputs, sum(m.spec.host_input_ranks), depth=4, n_threads=int(os.environ.get("PACK_THREADS", "8")), zero_copy=zero_copy)
        pre = []
        for r in reqs:                                         # marshal once
            arrs = [np.require(np.asarray(a), requirements="C") for a in r.inputs]
            dims = [np.asarray(a.shape, np.int64) for a in arrs]
            tens = (_lib.HostTensor * len(arrs))()
            for i, a in enumerate(arrs):
                tens[i] = _lib.HostTensor(a.ctypes.data, a.dtype.itemsize, a.ndim, dims[i].ctypes.data_as(C.POINTER(C.c_int64)))
            mo = np.asarray(modes, np.uint8)
            ma = np.asarray([int(r.symbols[m.spec.columns[k].rows_arg]) if k >= 0 else 0 for k in rows_col], np.int64)
            sym = np.ascontiguousarray(r.symbols, np.int32)
            pre.append((tens, mo, ma, sym, arrs, dims))
        blob, nbytes = C.c_void_p(), C.c_int64()
        offs, shps = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        a = _lib.ProcessArgs(None, 0, None, None, tptrs, None, None, stream, _lib.ALLOC_FN(), None, alloc, None)
        shipped = 0

        def one(k):
            global shipped
            tens, mo, ma, sym, _, _ = pre[k % len(pre)]
            _lib.check(L.fcp_stager_stage_ex(st.handle, tens, len(tens), mo.ctypes.data, ma.ctypes.data, stream, C.byref(blob),
                                             C.byref(nbytes), C.byref(offs), C.byref(shps)), "stage")
            a.concated_inputs, a.concated_bytes, a.concated_offsets, a.concated_shapes = blob.value, nbytes.value, offs, shps
            a.symbols = sym.ctypes.data_as(C.POINTER(C.c_int32))
            _lib.check(L.fcp_process_feature_columns(op.plan.handle, C.byref(a), None), "process")
            shipped = nbytes.value

        for k in range(20):
            one(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            one(k)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / steps
        lat = 0.0
        for k in range(30):
            t1 = time.perf_counter()
            one(k)
            torch.cuda.synchronize()
            lat += (time.perf_counter() - t1) * 1e6
        print(f"{name:6s} {'zero-copy' if zero_copy else 'H2D copy '}: blob {shipped / 1e6:5.2f} MB, {us:6.1f} us per request pipelined "
              f"({m.batch / us:.2f} M inferences/s), lone request {lat / 30:6.1f} us")
        st.close()
    op.plan.close()
