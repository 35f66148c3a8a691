"""Single-caller-stream serving over the plan's private streams: sweep lanes x depth (x NO_CALLER_WAIT) on S2 (or
--workload ragged|e) against the one-stream and the N-worker figures, all in ONE process on one box.
  python scripts/r04_private_sweep.py [--workload s2] [--steps 1500]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="s2")
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--combos", default="2x2,2x3,3x3,3x4")  # the library creates at most three private streams
ap.add_argument("--nowait", default="0,1")
ap.add_argument("--requests", type=int, default=16)
args = ap.parse_args()
model = {"s2": synth.model_s2, "ragged": lambda: synth.staged_model(synth.model_ragged(seg="indices")),
         "ragged_ad": lambda: synth.model_ragged(seg="indices"), "ragged_csr": lambda: synth.model_ragged(seg="csr"),
         "e": lambda: synth.staged_model(synth.model_ae("e"))}[args.workload]()
base = ServingHarness(model, n_requests=args.requests, arena_ring=8, n_threads=1)
base.run(200)
out = {"workload": model.name, "one_stream_us": base.run(args.steps)[0] * 1e3 / args.steps}
for workers in (3,):
    hw = ServingHarness(model, n_requests=args.requests, arena_ring=8, n_threads=workers, tables=base.tables)
    hw.run(100)
    out[f"{workers}_workers_us"] = hw.run(args.steps // workers)[0] * 1e3 / (args.steps // workers * workers)
    hw.close()
for nowait in [bool(int(x)) for x in args.nowait.split(",")]:
    for lanes, depth in [tuple(int(v) for v in c.split("x")) for c in args.combos.split(",")]:
        hp = ServingHarness(model, n_requests=args.requests, arena_ring=8, n_threads=1, tables=base.tables)
        hp.plan.set_private_streams(lanes, no_caller_wait=nowait)
        hp.run_private(200, depth)
        w, d = hp.run_private(args.steps, depth)
        out[f"private lanes={lanes} depth={depth}{' no_caller_wait' if nowait else ''}"] = w * 1e3 / args.steps
        hp.close()
out["env"] = {k: v for k, v in os.environ.items() if k.startswith("FCP_") or k == "GPU_MAX_HW_QUEUES"}
print(json.dumps(out, indent=1))
