#!/usr/bin/env python3
"""RAGGED (CSR input, 16 distinct requests) on whichever libfcp_hip.so FCP_LIB_DIR selects."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

m = synth.model_ragged()
for n_req in (16, 1):
    h = ServingHarness(m, n_requests=n_req)
    h.run(50)
    wall, dev, _ = h.run(1000)
    print(f"{os.environ.get('FCP_LIB_DIR', 'product'):40s} requests {n_req:2d}  dev {dev:6.2f} us  wall {wall:6.2f} us")
    h.close()
