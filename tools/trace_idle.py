"""Chip-idle gaps of a rocprofv3 kernel trace: every interval in the steady state during which NO kernel runs on any queue, with the kernel
that ended last before it and the kernel that starts after it (name, queue), summed by (before -> after) pair.
usage: trace_idle.py <dir with *kernel_trace.csv> <steps in the trace's later half> [min gap us]"""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ming = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 5e3
rows = list(csv.DictReader(open(f)))
qk = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
short = lambda n: n.replace("void ", "").split("(")[0][:44]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qk], short(r["Kernel_Name"])) for r in rows)
# steady state = the last `steps` replays: a replay starts with its set_words_kernel launch (unast_amd.graphed._replay)
marks = [x[0] for x in iv if x[3].startswith("set_words_kernel")]
if len(marks) > steps + 1:
    lo, hi = marks[-int(steps) - 1], marks[-1]
    iv = [x for x in iv if lo <= x[0] < hi]
else:
    t0, t1 = iv[0][0], max(x[1] for x in iv)
    iv = [x for x in iv if x[0] >= t0 + (t1 - t0) * 0.5]
span = max(x[1] for x in iv) - iv[0][0]
pairs = collections.defaultdict(lambda: [0, 0])
tot = 0; n = 0; small = 0
last = iv[0]
for x in iv[1:]:
    if x[0] > last[1]:
        g = x[0] - last[1]
        tot += g; n += 1
        if g >= ming:
            k = "%s [q%s] -> %s [q%s]" % (last[3], last[2], x[3], x[2])
            pairs[k][0] += 1; pairs[k][1] += g
        else:
            small += g
    if x[1] > last[1]: last = x
print("span %.1f ms (%.1f steps): idle %.2f ms in %d gaps = %.2f ms/step; gaps under %.0f us: %.2f ms/step" % (span / 1e6, steps, tot / 1e6, n, tot / 1e6 / steps, ming / 1e3, small / 1e6 / steps))
for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %6.2f ms/step  n/step %5.1f  mean %6.1f us   %s" % (v[1] / 1e6 / steps, v[0] / steps, v[1] / v[0] / 1e3, k))
