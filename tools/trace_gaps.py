"""Per-queue idle gaps from a rocprofv3 kernel trace: for every queue, the time between the end of a kernel and the start of the next one on
that queue, summed per step and listed by the kernel that FOLLOWS the gap.  usage: trace_gaps.py <dir with *kernel_trace.csv> [steps]"""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
qk = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qk], r["Kernel_Name"]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
iv = [x for x in iv if x[0] >= t0 + (t1 - t0) * 0.5]                    # steady state: the later half
span = max(e for _, e, _, _ in iv) - iv[0][0]
byq = collections.defaultdict(list)
for x in iv: byq[x[2]].append(x)
print("span %.1f ms, kernels %d" % (span / 1e6, len(iv)))
for q, xs in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    busy = sum(e - s for s, e, _, _ in xs)
    gaps = collections.defaultdict(lambda: [0, 0])
    small = 0
    for a, b in zip(xs, xs[1:]):
        g = b[0] - a[1]
        if g > 0:
            k = b[3].split("(")[0][:40]
            gaps[k][0] += 1; gaps[k][1] += g
            if g < 20000: small += g
    tot = sum(v[1] for v in gaps.values())
    print("queue %s: %d kernels, busy %.1f%% of span, gaps %.1f%% (gaps < 20 us: %.1f%%)" % (q, len(xs), 100.0 * busy / span, 100.0 * tot / span, 100.0 * small / span))
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]:
        print("     before %-42s n=%5d  mean gap %6.1f us  total %.2f ms" % (k, v[0], v[1] / v[0] / 1e3, v[1] / 1e6))
