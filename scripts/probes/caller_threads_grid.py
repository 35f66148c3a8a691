"""The reference's real serving protocol — T host threads (serve_workers) share ONE Session, hence one compute stream
(examples/cc/recom_examples.patch:193-216) — over the plan's private streams: T threads issue on the one caller stream, every
thread keeps `depth` of its own requests in flight before it enqueues their consumers (depth 1 = FeatureColumnProcess and
Addons>ConcatOutputs back to back inside one Session::Run).  us per request over all threads, private streams (3) next to the
same threads in stream order.
  python scripts/probes/caller_threads_grid.py [s2|ragged]"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s2"
model = synth.model_s2() if which == "s2" else synth.staged_model(synth.model_ragged(seg="indices"))
nreq = 16 if which == "s2" else 64
base = ServingHarness(model, n_requests=nreq, arena_ring=6, n_threads=1)
base.run(100)
out = {"workload": which, "one_thread_stream_order_no_consumer_us": round(base.run(1200)[0] * 1e3 / 1200, 2), "grid": {}}
for T in (1, 2, 3, 4, 6, 8):
    for depth in (1, 2, 3):
        h = ServingHarness(model, n_requests=nreq, arena_ring=6, n_threads=T, tables=base.tables)
        per = max(1800 // T, 300)
        h.run_private(100, depth, T)
        so = h.run_private(per, depth, T)[0] * 1e3 / (per * T)
        h.plan.set_private_streams(3)
        h.run(1)
        v = h.plan.verify_private_streams(h.caller_stream(), 400)
        h.run_private(max(1400 // T, 100), depth, T)   # long enough for three evaluations of the supervisor
        pv = h.run_private(per, depth, T)[0] * 1e3 / (per * T)
        st = h.plan.private_streams_stats()
        out["grid"][f"T{T}_d{depth}"] = {"stream_order_us": round(so, 2), "private3_us": round(pv, 2), "verdict": v,
                                         "sup_last_ratio": round(st["last_ratio"], 3), "demoted": st["demoted"], "evaluations": st["evaluations"]}
        h.close()
print(json.dumps(out, indent=1))
