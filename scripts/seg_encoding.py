#!/usr/bin/env python3
"""RAGGED config with the three segment encodings (CSR offsets / SparseTensor indices [nnz,2] /
int32 row ids): what the segment-offset pre-pass costs.  GPU box: python scripts/seg_encoding.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

for seg in ("csr", "indices", "rowids32"):
    m = synth.model_ragged(seg=seg)
    for n_req, label in ((1, "cached shapes"), (16, "new shapes each request")):
        h = ServingHarness(m, n_requests=n_req)
        h.run(50)
        steps = 1000
        wall, dev, _ = h.run(steps)
        b = h.algorithmic_bytes()["total"]
        print(f"RAGGED seg={seg:9s} {label:24s} dev {dev * 1e3 / steps:6.2f} us  wall {wall * 1e3 / steps:6.2f} us  "
              f"{b / 1e6:6.1f} MB/request")
        h.close()
