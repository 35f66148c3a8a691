"""Kernel-trace summary: for the last N fcp_dense launches — duration, how many run concurrently, gap between consecutive
launches of one queue, throughput."""
import csv, glob, sys, collections
d, mode = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fcp_dense" in r["Kernel_Name"] or "fcp_consume" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), "dense" if "fcp_dense" in r["Kernel_Name"] else "reader"))
rows.sort()
dense = [r for r in rows if r[3] == "dense"][-300:]
t0, t1 = dense[0][0], dense[-1][1]
dur = [e - s for s, e, _, _ in dense]
# concurrency: time-weighted average number of dense kernels in flight
ev = sorted([(s, 1) for s, e, _, _ in dense] + [(e, -1) for s, e, _, _ in dense])
cur, last, acc = 0, ev[0][0], 0
for t, dlt in ev:
    acc += cur * (t - last); last = t; cur += dlt
byq = collections.defaultdict(list)
for r in dense:
    byq[r[2]].append(r)
gaps = []
for q, rs in byq.items():
    for a, b in zip(rs, rs[1:]):
        gaps.append(b[0] - a[1])
print(f"{mode:13s} per request {(t1 - t0) / len(dense) / 1e3:6.2f} us | kernel duration avg {sum(dur) / len(dur) / 1e3:6.2f} min {min(dur) / 1e3:6.2f} max {max(dur) / 1e3:6.2f} us | "
      f"in flight {acc / (t1 - t0):4.2f} | queues {len(byq)} | end->next start on the same queue avg {sum(gaps) / max(len(gaps), 1) / 1e3:6.2f} us")
