"""Per-queue launch census of a rocprofv3 --kernel-trace result db: launches / step, mean duration, us / step."""
import sqlite3, re, collections, sys
db, steps = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name,queue_id,start,end from kernels order by start").fetchall()
qs = collections.Counter(r[1] for r in rows)
for q, _ in qs.most_common():
    d = collections.OrderedDict()
    for n, qq, s, e in rows:
        if qq != q:
            continue
        n = re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:48]
        g = d.setdefault(n, [0, 0.0]); g[0] += 1; g[1] += e - s
    tot_n = sum(v[0] for v in d.values()); tot_t = sum(v[1] for v in d.values())
    print("== queue %s: %.1f launches/step, %.1f us of kernels/step" % (q, tot_n / steps, tot_t / steps / 1e3))
    for k, (n, t) in sorted(d.items(), key=lambda x: -x[1][1])[:45]:
        print("   %-50s %5.1f/step  avg %6.1f us  %7.1f us/step" % (k, n / steps, t / n / 1e3, t / steps / 1e3))
