"""Why are >= 4 private streams slower than one?  Hypothesis: the process' concurrently ACTIVE HIP streams beyond some count
share a hardware queue / pipe.  Independent workers (no events between streams) at 2..8 streams, and private lanes 3 / 4 / 5,
under the GPU_MAX_HW_QUEUES of the environment (run once per value: the variable is read when the runtime starts).
  GPU_MAX_HW_QUEUES=4 python scripts/probes/hw_queue_limit.py"""
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
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "one_stream_us": round(base.run(steps)[0] * 1e3 / steps, 2)}
for workers in (2, 3, 4, 5, 6, 8):
    hw = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=workers, tables=base.tables)
    hw.run(100)
    per = steps // workers
    out[f"workers={workers}"] = round(hw.run(per)[0] * 1e3 / (per * workers), 2)
    hw.close()
for lanes, depth in ((3, 3), (4, 4), (5, 5), (6, 6)):
    hp = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1, tables=base.tables)
    hp.plan.set_private_streams(lanes)
    hp.run_private(200, depth)
    out[f"lanes={lanes}"] = round(hp.run_private(steps, depth)[0] * 1e3 / steps, 2)
    hp.close()
print(json.dumps(out))
