"""Does fcp_plan_probe_private_streams (kernels that only wait) tell the good queue mappings from the bad ones?  The real S2
figure and the probe's serial / lanes ratio, per number of dummy streams created before the lanes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

steps = 600
LANES = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model = synth.model_s2()
base = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1)
base.run(100)
out = {"lanes": LANES, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "FCP_LANE_PRIORITY": os.environ.get("FCP_LANE_PRIORITY"),
       "one_stream_us": round(base.run(steps)[0] * 1e3 / steps, 2), "by_dummy_streams": {}}
dummies = []
for n in range(0, 9):
    while len(dummies) < n:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda").add_(1)
        s.synchronize()
        dummies.append(s)
    hp = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1, tables=base.tables)
    hp.plan.set_private_streams(LANES, verify=False)
    caller = hp.caller_stream()
    probes = []
    for spin, blocks in ((80, 1), (40, 1)):
        hp.plan.probe_private_streams(caller, 12, spin, blocks)
        a, b = hp.plan.probe_private_streams(caller, 24, spin, blocks)
        probes.append(round(a / b, 2))
    hp.run_private(100, LANES)
    real = round(hp.run_private(steps, LANES)[0] * 1e3 / steps, 2)
    out["by_dummy_streams"][n] = {"real_us": real, "probe(80us x1, 40us x1)": probes}
    hp.close()
print(json.dumps(out))
