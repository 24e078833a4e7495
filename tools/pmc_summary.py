"""Per-kernel mean of one rocprofv3 --pmc counter (result db from a --pmc <NAME> --kernel-trace run)."""
import sqlite3, re, collections, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
d = collections.OrderedDict()
cname = None
for name, dur, cn, val in c.execute("select name,duration,counter_name,counter_value from pmc_events"):
    cname = cn
    n = re.sub(r'\(.*', '', name.replace('(anonymous namespace)::', ''))
    g = d.setdefault(n, [0, 0.0, 0.0]); g[0] += 1; g[1] += dur; g[2] += val
rows = sorted(d.items(), key=lambda kv: -kv[1][1])[:20]
with open(out, 'w') as f:
    f.write("kernel,launches,avg_us,%s_mean\n" % cname)
    for k, (n, dur, val) in rows:
        f.write("%s,%d,%.1f,%.3f\n" % (k, n, dur / n / 1e3, val / n))
print(open(out).read())
