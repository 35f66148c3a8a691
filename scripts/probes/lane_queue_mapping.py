"""Does the private-stream mode depend on WHICH hardware queues the lanes land on?  N dummy streams (each used once, so the
runtime materialises their hardware queues) are created before the harness and its lanes; S2, 3 lanes, depth 3.
  FCP_LANE_PRIORITY=normal|low GPU_MAX_HW_QUEUES=8 python scripts/probes/lane_queue_mapping.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

steps = 800
model = synth.model_s2()
base = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1)
base.run(100)
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "FCP_LANE_PRIORITY": os.environ.get("FCP_LANE_PRIORITY"),
       "one_stream_us": round(base.run(steps)[0] * 1e3 / steps, 2), "lanes3_us_by_dummy_streams": {}}
dummies = []
for n in range(0, 9):
    while len(dummies) < n:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda").add_(1)
        s.synchronize()
        dummies.append(s)
    hp = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1, tables=base.tables)
    hp.plan.set_private_streams(3)
    hp.run_private(100, 3)
    out["lanes3_us_by_dummy_streams"][n] = round(hp.run_private(steps, 3)[0] * 1e3 / steps, 2)
    hp.close()
print(json.dumps(out))
