#!/usr/bin/env python3
"""Launch-bound workloads with and without HIP-graph replay (fixed-shape models).
GPU box:  python scripts/graph_replay.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

for name, model in (("DLRM", synth.model_dlrm()), ("S2-100col", synth.model_s2(columns=100)),
                    ("S2", synth.model_s2(columns=1000))):
    h = ServingHarness(model, n_requests=8, arena_ring=8)
    h.run(50)
    steps = 960
    _, dev, _ = h.run(steps)
    line = f"{name:10s} stream launches {dev * 1e3 / steps:6.2f} us/request"
    for group in (1, 4, 8):
        _, gdev = h.run_graph(steps, group)
        line += f" | graph x{group}: {gdev * 1e3 / steps:6.2f}"
    print(line)
    h.close()
