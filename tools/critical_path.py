"""Where the main queue of a step spends its wall time: kernels vs the gaps between them (rocprofv3 --kernel-trace
result db).  The main queue is the one with the most launches.  For one steady-state step (between two
consecutive launches of a marker kernel) it prints: wall, sum of kernel durations, sum of gaps, the gap histogram
and the largest gaps with the kernels before / after them and what the other queues ran meanwhile.

    python tools/critical_path.py results.db [marker-substring]
"""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "image_to_c8"
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel" in t.lower())
rows = c.execute("select name,queue_id,start,end from %s order by start" % kt).fetchall()


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:46]


qs = collections.Counter(r[1] for r in rows)
main_q = qs.most_common(1)[0][0]
main = [(short(n), s, e) for n, q, s, e in rows if q == main_q]
others = [(short(n), q, s, e) for n, q, s, e in rows if q != main_q]
marks = [i for i, (n, s, e) in enumerate(main) if marker in n]
print("queues:", dict(qs), "main =", main_q, "; marker launches on main:", len(marks))
if len(marks) < 4:
    sys.exit("marker not found often enough")
lo, hi = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
step = main[lo:hi]
wall = (step[-1][2] - step[0][1]) / 1e3
ksum = sum(e - s for _, s, e in step) / 1e3
gaps = [(step[i + 1][1] - step[i][2]) / 1e3 for i in range(len(step) - 1)]
print("one step on the main queue: %d launches, wall %.1f us (to the last kernel's end), kernels %.1f us, gaps %.1f us"
      % (len(step), wall, ksum, sum(gaps)))
hist = collections.Counter()
for g in gaps:
    b = "<1" if g < 1 else "1-2" if g < 2 else "2-4" if g < 4 else "4-10" if g < 10 else "10-30" if g < 30 else ">30"
    hist[b] += g
print("gap time by gap size (us):", {k: round(v, 1) for k, v in sorted(hist.items())})
big = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:14]
for i in sorted(big):
    a, b = step[i], step[i + 1]
    busy = collections.Counter()
    for n, q, s, e in others:
        ov = min(e, b[1]) - max(s, a[2])
        if ov > 0:
            busy[(q, n)] += ov / 1e3
    top = ", ".join("q%s %s %.0f" % (q, n[:28], t) for (q, n), t in busy.most_common(3))
    print("  gap %6.1f us after %-40s before %-40s | meanwhile: %s" % (gaps[i], a[0][:40], b[0][:40], top))
# per-kernel-name totals on the main queue for this step
tot = collections.OrderedDict()
for n, s, e in step:
    g = tot.setdefault(n, [0, 0.0])
    g[0] += 1
    g[1] += (e - s) / 1e3
print("kernels of the step on the main queue:")
for n, (cnt, t) in sorted(tot.items(), key=lambda x: -x[1][1])[:28]:
    print("   %-48s %3d  %7.1f us  avg %5.1f" % (n, cnt, t, t / cnt))
