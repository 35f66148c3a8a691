"""Long run of the single-caller-stream loop: 4 x 500 000 S2 requests through the verified private streams; the rate must not
drift (registry, event rings and descriptor slots are all bounded), host RSS must not grow."""
import json
import os
import resource
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

h = ServingHarness(synth.model_s2(), n_requests=16, arena_ring=8, n_threads=1)
h.plan.set_private_streams(3)
h.run_private(500, 3)
out = {"verdict": h.plan.private_streams_verdict(h.caller_stream()), "blocks": []}
for k in range(4):
    wall, dev = h.run_private(500_000, 3)
    out["blocks"].append({"us_per_request": round(wall * 1e3 / 500_000, 3), "rss_MB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024})
assert h.verify_resident()["checked"] > 0
print(json.dumps(out))
