#!/usr/bin/env python3
"""Round 6: why does the CPU baseline (orc_serve_for: independent single-threaded workers)
get SLOWER beyond 16 workers on the GPU box's 2 x 64-core host?  One process = one condition; prints a JSON line.
  --vocab V   table rows (1 000 000: DRAM-resident random rows; 2000: cache-resident tables — separates software contention
              from the memory system)
  environment: OMP_PROC_BIND / OMP_PLACES are read by libgomp at load (set them outside)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--vocab", type=int, default=1_000_000)
ap.add_argument("--columns", type=int, default=200)
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--workers", default="8,16,32,64,128,256")
ap.add_argument("--label", default="")
ap.add_argument("--dataflow", type=int, default=0, help="0: every column straight into the concat matrix; 1: TF-CPU's (column tensors + ConcatV2)")
args = ap.parse_args()
import fcp_oracle
from recom_amd import synth
from recom_amd.ops import concat_inputs
model = synth.model_s2(columns=args.columns, vocab=args.vocab)
orc = fcp_oracle.COracle()
reqs = [model.make_request(12345 + i) for i in range(64)]
packed = [concat_inputs(r.inputs) for r in reqs]
t0 = time.perf_counter()
tables = []
for t in model.tables:
    a = np.empty((t.vocab, t.dim), np.float32)
    orc.fill_parallel(a, 0.5)
    tables.append(a)
fill_s = time.perf_counter() - t0
plan = model.spec.to_dict()
cores = len(os.sched_getaffinity(0))
out = {}
for w in [int(x) for x in args.workers.split(",") if int(x) <= cores]:
    done, sec = orc.serve_for(plan, packed, tables, None, w, args.seconds, args.dataflow)
    out[str(w)] = round(model.batch * done / sec)
thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip() if os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled") else "?"
print(json.dumps({"label": args.label, "dataflow": args.dataflow, "columns": args.columns, "vocab": args.vocab, "table_GB": sum(a.nbytes for a in tables) / 1e9,
                  "fill_s": round(fill_s, 1), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"),
                  "thp": thp, "inferences_per_s_by_workers": out}))
