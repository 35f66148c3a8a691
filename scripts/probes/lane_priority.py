"""Private lanes created in the runtime's high / low priority queue pool (FCP_LANE_PRIORITY) under the GPU_MAX_HW_QUEUES of
the environment: does a pool of their own make the lanes independent of the process' other streams?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

steps = 1200
model = synth.model_s2()
base = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1)
base.run(200)
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "FCP_LANE_PRIORITY": os.environ.get("FCP_LANE_PRIORITY"),
       "one_stream_us": round(base.run(steps)[0] * 1e3 / steps, 2)}
for lanes, depth in ((2, 3), (3, 3), (3, 4), (4, 4), (5, 5), (6, 6)):
    hp = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1, tables=base.tables)
    hp.plan.set_private_streams(lanes)
    hp.run_private(200, depth)
    out[f"lanes={lanes}x{depth}"] = round(hp.run_private(steps, depth)[0] * 1e3 / steps, 2)
    hp.close()
print(json.dumps(out))
