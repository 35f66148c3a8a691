"""Summarise a rocprofv3 --hip-trace --hsa-trace run (CSV output) of `fcp_bench --h2d 1`: the slowest host-side HIP calls and,
for each stalled one (> 1 ms), the HSA calls the same thread made inside it — what the runtime was waiting in.
  python scripts/probes/stall_trace_summary.py <output dir of rocprofv3>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
hip = [f for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)]
hsa = [f for f in glob.glob(os.path.join(d, "**", "*hsa_api_trace.csv"), recursive=True)]
print("files:", hip, hsa)


def rows(files):
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


hip_calls = []
for r in rows(hip):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    hip_calls.append((e - s, r["Function"], int(r["Thread_Id"]), s, e))
by_fn = defaultdict(list)
for dur, fn, tid, s, e in hip_calls:
    by_fn[fn].append(dur)
print("HIP calls: count, mean us, p99 us, max us")
for fn, v in sorted(by_fn.items(), key=lambda kv: -max(kv[1]))[:12]:
    v.sort()
    print(f"  {fn:34s} {len(v):7d} {sum(v) / len(v) / 1e3:9.2f} {v[int(len(v) * 0.99)] / 1e3:9.2f} {v[-1] / 1e3:10.1f}")
stalled = sorted([c for c in hip_calls if c[0] > 1_000_000], reverse=True)[:12]
print(f"{len([c for c in hip_calls if c[0] > 1_000_000])} HIP calls over 1 ms; the longest:")
hsa_by_tid = defaultdict(list)
for r in rows(hsa):
    hsa_by_tid[int(r["Thread_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
for v in hsa_by_tid.values():
    v.sort()
for dur, fn, tid, s, e in stalled:
    print(f"  {fn} {dur / 1e3:.1f} us on thread {tid}")
    inner = [(b - a, f) for a, b, f in hsa_by_tid.get(tid, []) if a >= s and b <= e]
    agg = defaultdict(lambda: [0, 0])
    for dd, f in inner:
        agg[f][0] += 1
        agg[f][1] += dd
    for f, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"      {f:44s} x{n:<5d} {tot / 1e3:10.1f} us")
