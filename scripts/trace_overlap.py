"""From a rocprofv3 --kernel-trace CSV: how much of the blind-rotation kernels' time runs concurrently with another blind-rotation
kernel (two streams), and the per-kernel durations.  usage: trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "k_blind_rotate" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("k_blind_rotate_")[1].split("<")[0], r.get("Stream_Id") or r.get("Queue_Id")))
rows.sort()
if not rows:
    sys.exit("no blind-rotation kernels in the trace")
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
t_prev, depth, busy, conc = ev[0][0], 0, 0, 0
for t, d in ev:
    if depth >= 1: busy += t - t_prev
    if depth >= 2: conc += t - t_prev
    depth += d; t_prev = t
by = collections.defaultdict(list)
for s, e, k, q in rows:
    by[(k, q)].append(e - s)
print("blind-rotation kernels: %d; time with one in flight %.2f ms, with two or more %.2f ms (%.1f %%); span %.2f ms"
      % (len(rows), busy / 1e6, conc / 1e6, 100.0 * conc / max(1, busy), (rows[-1][1] - rows[0][0]) / 1e6))
for (k, q), v in sorted(by.items()):
    print("  %-8s queue/stream %s: %d launches, mean %.1f us, min %.1f, max %.1f" % (k, q, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
