"""One FRESH process per point: N dummy streams, then one harness with 3 (argv[2]) unverified lanes.  S2 at a small vocabulary
(the effect is in the queues, not in the tables): us per request on the caller's stream alone, through the lanes, and the probe."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

n, lanes = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 3
dummies = []
for _ in range(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
    s.synchronize()
    dummies.append(s)
model = synth.model_s2(vocab=20000)
hp = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1)
hp.run(100)
one = hp.run(600)[0] * 1e3 / 600
hp.plan.set_private_streams(lanes, verify=False)
hp.plan.probe_private_streams(hp.caller_stream(), 12, 40, 1)
a, b = hp.plan.probe_private_streams(hp.caller_stream(), 24, 40, 1)
hp.run_private(100, lanes)
real = hp.run_private(600, lanes)[0] * 1e3 / 600
print(json.dumps({"dummies": n, "lanes": lanes, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "one_stream_us": round(one, 2),
                  "lanes_us": round(real, 2), "probe": round(a / b, 2)}))
