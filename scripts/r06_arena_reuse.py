#!/usr/bin/env python3
"""Round 6, VERDICT r05 item 1: what does the OUTPUT cost when the arena is reused?

The reference's op takes its arena from `allocate_output(2)` (feature_column_process_op_gpu.cu.cc:107-111): a one-thread
serving loop gets back the block it just freed, so the reuse distance of an output line is ONE request (S2: 61 MB of output
+ 86 MB of table lines = 147 MB, within the 256-MiB Infinity Cache).  Every earlier record used a ring of 6 arenas.

One process = one output-store policy (FcpLaunch::store_through is decided from FCP_STORE_THROUGH_BYTES once per process;
plain stores need the -DFCP_NO_NT build, FCP_LIB_DIR); inside it: arena ring in {1, 2, 3, 6} x {no consumer, the consumer
kernel right behind every request}, `rounds` interleaved passes, HIP-event time per request on the launch stream.
Prints one JSON line per (workload, policy): {"ring_1": {"plain_us": [..], "consumer_us": [..]}, ...}.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="s2", choices=["s2", "ragged", "ragged-staged", "dlrm", "f"])
ap.add_argument("--policy", default="product", help="label only; the policy itself comes from the environment / FCP_LIB_DIR")
ap.add_argument("--rings", default="1,2,3,6")
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--single", type=int, default=0, help="ring N only, no consumer pass, one round: the cell a rocprofv3 trace is taken of")
args = ap.parse_args()

from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

if args.workload == "s2":
    model, n_req = synth.model_s2(), 16
elif args.workload == "dlrm":
    model, n_req = synth.model_dlrm(), 16
elif args.workload == "f":
    model, n_req = synth.staged_model(synth.model_ae("f")), 64
elif args.workload == "ragged":
    model, n_req = synth.model_ragged(), 64
else:
    model, n_req = synth.staged_model(synth.model_ragged()), 64

rings = [args.single] if args.single else [int(r) for r in args.rings.split(",")]
hs, tables = {}, None
for r in rings:
    hs[r] = ServingHarness(model, n_requests=n_req, arena_ring=r, n_threads=1, tables=tables)
    tables = hs[r].tables
    hs[r].run(200)
res = {f"ring_{r}": {"plain_us": [], "consumer_us": []} for r in rings}
for rnd in range(1 if args.single else args.rounds):
    for r in rings:
        _, dev, _ = hs[r].run(args.steps)
        res[f"ring_{r}"]["plain_us"].append(round(dev * 1e3 / args.steps, 3))
    if args.single:
        break
    for r in rings:
        hs[r].run_private(100, 1)                   # private streams are OFF: stream order + the consumer right behind
        _, dev = hs[r].run_private(args.steps, 1)
        res[f"ring_{r}"]["consumer_us"].append(round(dev * 1e3 / args.steps, 3))
alg = hs[rings[0]].algorithmic_bytes()
print(json.dumps({"workload": args.workload, "policy": args.policy, "steps": args.steps,
                  "FCP_STORE_THROUGH_BYTES": os.environ.get("FCP_STORE_THROUGH_BYTES"), "FCP_LIB_DIR": os.environ.get("FCP_LIB_DIR"),
                  "algorithmic_MB": alg["total"] / 1e6, "out_MB": alg["out"] / 1e6, **res}))
