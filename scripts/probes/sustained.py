"""Sustained rates (blocks of 200 000 S2 requests, ~6 s each): one stream in stream order, then the verified private streams,
then one stream again — does a long run hold the short run's figures (clocks / power)?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

N = 200_000
h = ServingHarness(synth.model_s2(), n_requests=16, arena_ring=8, n_threads=1)
h.run(500)
out = {"short_one_stream_us": round(h.run(2000)[0] * 1e3 / 2000, 2), "one_stream_us": [round(h.run(N)[0] * 1e3 / N, 2) for _ in range(2)]}
h.plan.set_private_streams(3)
h.run_private(500, 3)
out["short_private_streams_us"] = round(h.run_private(2000, 3)[0] * 1e3 / 2000, 2)
out["private_streams_us"] = [round(h.run_private(N, 3)[0] * 1e3 / N, 2) for _ in range(3)]
out["verdict"] = h.plan.private_streams_verdict(h.caller_stream())
h.plan.set_private_streams(0)
out["one_stream_again_us"] = [round(h.run(N)[0] * 1e3 / N, 2) for _ in range(2)]
print(json.dumps(out))
