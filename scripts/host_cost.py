#!/usr/bin/env python3
"""Host-side cost of one request: tiny batches, so the GPU never limits.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

for name, m in (("RAGGED b=2 (512 cols, new shapes)", synth.model_ragged(batch=2)),
                ("S2 b=2 (1000 cols, cached shapes)", synth.model_s2(batch=2, vocab=1000)),
                ("E b=2 (1000 cols, new shapes)", synth.model_ae("E", batch=2, large_rows=1 << 12))):
    for threads in (1, 3):
        h = ServingHarness(m, n_requests=16, n_threads=threads)
        h.run(100)
        wall, dev, _ = h.run(2000)
        print(f"{name:40s} threads {threads}: {wall * 1e3 / (2000 * threads):6.2f} us per request (aggregate)")
        h.close()
