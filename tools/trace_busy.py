"""GPU busy fraction and per-stream time from a rocprofv3 kernel trace (union of kernel intervals over the timed steps)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?")), r["Kernel_Name"]) for r in rows)
# keep the last 60 % of the trace (steady state)
t0 = iv[0][0]; t1 = max(e for _, e, _, _ in iv)
cut = t0 + (t1 - t0) * 0.4
iv = [x for x in iv if x[0] >= cut]
span = max(e for _, e, _, _ in iv) - iv[0][0]
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]
for s, e, _, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in iv)
print("span %.1f ms  busy(union) %.1f ms = %.1f %%  sum of kernel durations %.1f ms (overlap factor %.2f)  kernels %d" % (span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / busy, len(iv)))
by = {}
for s, e, q, _ in iv:
    by[q] = by.get(q, 0) + (e - s)
print("per queue ms:", {k: round(v / 1e6, 1) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})
# idle gaps histogram
gaps = []
cur_e = iv[0][1]
for s, e, _, _ in iv[1:]:
    if s > cur_e: gaps.append(s - cur_e)
    cur_e = max(cur_e, e)
gaps.sort()
if gaps:
    print("idle gaps: n=%d total %.2f ms median %.1f us p90 %.1f us max %.1f us" % (len(gaps), sum(gaps) / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
