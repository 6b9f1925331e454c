"""One replayed step of a rocprofv3 kernel trace as a timeline: per-queue busy time, how long 0 / 1 / 2 / 3+ kernels run at once, and the
longest stretches during which ONE kernel runs alone (its name, its queue) -- what the step is waiting for when the chip is under-used.
usage: trace_timeline.py <dir with *kernel_trace.csv> [steps back from the end, default 2]"""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
qk = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
short = lambda n: n.replace("void ", "").split("(")[0][:40]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qk], short(r["Kernel_Name"])) for r in rows)
marks = [x[0] for x in iv if x[3].startswith("set_words_kernel")] or [x[0] for x in iv if x[3].startswith("specaugment_sum")]
lo, hi = (marks[-back - 1], marks[-back]) if len(marks) > back + 1 else (iv[len(iv) // 2][0], iv[-1][1])
iv = [x for x in iv if lo <= x[0] < hi]
span = hi - lo
print("step of %.2f ms, %d kernels" % (span / 1e6, len(iv)))
busy = collections.defaultdict(int)
for s, e, q, n in iv: busy[q] += e - s
print("queues busy (ms):", {q: round(v / 1e6, 2) for q, v in sorted(busy.items())}, " sum %.2f" % (sum(busy.values()) / 1e6))
ev = sorted([(s, 1, i) for i, (s, e, q, n) in enumerate(iv)] + [(e, -1, i) for i, (s, e, q, n) in enumerate(iv)])
conc = collections.defaultdict(int); alone = collections.defaultdict(int); live = set(); t = lo
for tt, d, i in ev:
    if tt > t:
        conc[min(len(live), 4)] += tt - t
        if len(live) == 1:
            k = next(iter(live)); alone[(iv[k][3], iv[k][2])] += tt - t
        t = tt
    if d > 0: live.add(i)
    else: live.discard(i)
print("kernels running at once (ms):", {k: round(v / 1e6, 2) for k, v in sorted(conc.items())})
print("running alone (ms per step), top 14:")
for (n, q), v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
    print("   %6.2f  %s [q%s]" % (v / 1e6, n, q))
# coarse lanes: per queue, what runs in each 1-ms slice (the kernel with most time in the slice)
qs = sorted(busy)
nb = int(span // 1e6) + 1
print("timeline, 1 ms per column (a=attn, g=gemm/panel, l=lstm, n=norm, o=other, .=idle):")
for q in qs:
    line = ""
    for b in range(nb):
        a0, a1 = lo + b * 1e6, lo + (b + 1) * 1e6
        acc = collections.defaultdict(int)
        for s, e, qq, n in iv:
            if qq == q and e > a0 and s < a1:
                c = "a" if "attn" in n else "l" if "lstm" in n else "g" if ("gemm" in n or "panel" in n) else "n" if ("norm" in n or "bn_" in n) else "o"
                acc[c] += min(e, a1) - max(s, a0)
        tot = sum(acc.values())
        line += "." if tot < 2e5 else max(acc, key=acc.get)
    print("   q%-3s %s" % (q, line))
