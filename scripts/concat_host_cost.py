#!/usr/bin/env python3
"""Host-input half of Addons>ConcatOutputs (fcp_concat_outputs_host): cost per call for the reference's E / F shape (32 dense
features of [512, 1]) and for a larger payload, scatter reading the pinned slot directly vs through an H2D copy
(FCP_CONCAT_HOST_DIRECT_MAX=0).  GPU box: python scripts/concat_host_cost.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recom_amd import lib as _lib  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
temp = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
alloc = _lib.ALLOC_FN(lambda _ctx, n: temp.data_ptr())
for label, n, rows, dim in (("E / F: 32 x [512, 1]", 32, 512, 1), ("8 x [512, 64]", 8, 512, 64), ("16 x [2048, 64] (8 MB)", 16, 2048, 64)):
    hosts = [np.random.default_rng(k).standard_normal((rows, dim)).astype(np.float32) for k in range(n)]
    width = n * dim + 64
    out = torch.zeros((rows, width), device=dev)
    ptrs = (C.c_void_p * n)(*[h.ctypes.data for h in hosts])
    dims = np.full(n, dim, np.int32)
    offs = (np.arange(n) * dim + 64).astype(np.int32)

    def one():
        _lib.check(L.fcp_concat_outputs_host(ptrs, dims.ctypes.data, offs.ctypes.data, n, rows, width, out.data_ptr(), alloc, None, 0,
                                             stream), "concat_outputs_host")

    for _ in range(50):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        one()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) * 1e6 / 2000
    lat = 0.0
    for _ in range(200):
        t1 = time.perf_counter()
        one()
        torch.cuda.synchronize()
        lat += (time.perf_counter() - t1) * 1e6
    ok = all(np.array_equal(out[:, 64 + k * dim:64 + (k + 1) * dim].cpu().numpy(), hosts[k]) for k in range(n))
    print(f"{label:26s}: {us:6.2f} us per call back to back, {lat / 200:6.2f} us alone (call + sync), correct: {ok}")
