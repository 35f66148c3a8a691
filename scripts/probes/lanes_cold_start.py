"""Fresh process, private streams switched on BEFORE anything else has run on the GPU: is the verification's verdict right?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

warm = len(sys.argv) > 1 and sys.argv[1] == "warm"
h = ServingHarness(synth.model_s2(), n_requests=16, arena_ring=8, n_threads=1)
if warm:
    h.run(3000)
h.plan.set_private_streams(3)
h.run_private(500, 3)
out = {"warm_first": warm, "verdict": h.plan.private_streams_verdict(h.caller_stream()),
       "private_streams_us": [round(h.run_private(5000, 3)[0] * 1e3 / 5000, 2) for _ in range(3)]}
a, b = h.plan.probe_private_streams(h.caller_stream(), 24, 40, 1)
out["probe_after"] = round(a / b, 2)
h.plan.set_private_streams(0)
out["one_stream_us"] = round(h.run(5000)[0] * 1e3 / 5000, 2)
print(json.dumps(out))
