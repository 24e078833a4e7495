"""Kernel timeline of ONE steady-state alternated step (torch.profiler chrome trace): every launch with its queue, start,
duration and the idle time in front of it on its own queue -- where the critical queue waits for another one.

    [TL_BS=32] python tools/timeline.py [out.txt]      (default gpurun_out/timeline.txt; summary on stdout)
"""
import collections
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from combat_amd import step as step_mod  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline.txt"
dev = torch.device("cuda", 0)
np.random.seed(0)
torch.manual_seed(0)
BS = int(os.environ.get("TL_BS", "128"))
opt = bench.Opt()
opt.bs = BS
st = step_mod.AlternatedStep(*bench.build_nets(dev), opt)
batches = bench.synth_batches(8, BS, 0, dev)
for i in range(12):
    st.run(*batches[i % 8])
torch.cuda.synchronize()
S = 5
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for i in range(S):
        st.run(*batches[i % 8])
    torch.cuda.synchronize()
tmp = os.path.join(tempfile.gettempdir(), "combat_timeline.json")
prof.export_chrome_trace(tmp)
ev = [e for e in json.load(open(tmp))["traceEvents"] if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e]
ev.sort(key=lambda e: e["ts"])


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


# step boundaries: the generator's first convolution is the c8 kernel right after the image conversion
marks = [i for i, e in enumerate(ev) if "image_to_c8" in e["name"]]
assert len(marks) >= 3, "step marker not found"
lo, hi = marks[2], marks[3]
t0 = ev[lo]["ts"]
step = ev[lo:hi]
queues = collections.OrderedDict()
for e in step:
    queues.setdefault(e["args"].get("stream", e.get("tid")), []).append(e)
names = {q: "q%d" % i for i, q in enumerate(queues)}
last_end = {}
lines = []
busy = collections.Counter()
for e in step:
    q = e["args"].get("stream", e.get("tid"))
    gap = e["ts"] - last_end[q] if q in last_end else 0.0
    last_end[q] = e["ts"] + e["dur"]
    busy[q] += e["dur"]
    lines.append((e["ts"] - t0, e["dur"], names[q], gap, short(e["name"])))
span = max(e["ts"] + e["dur"] for e in step) - t0
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
with open(out_path, "w") as f:
    f.write("# start_us dur_us queue idle_before_us kernel   (step span %.1f us, %d launches)\n" % (span, len(step)))
    for ts, d, q, gap, n in lines:
        f.write("%9.1f %7.1f %-3s %7.1f  %s\n" % (ts, d, q, gap, n))
print("step span %.1f us, %d launches" % (span, len(step)))
for q, es in queues.items():
    print("  %s: %3d launches, busy %7.1f us (%.0f %%)" % (names[q], len(es), busy[q], 100 * busy[q] / span))
# time with k queues busy
pts = []
for e in step:
    pts.append((e["ts"], 1))
    pts.append((e["ts"] + e["dur"], -1))
pts.sort()
occ = collections.Counter()
cur, prev = 0, pts[0][0]
for t, d in pts:
    occ[cur] += t - prev
    prev = t
    cur += d
print("  kernels in flight:", ", ".join("%d: %.0f us" % (k, v) for k, v in sorted(occ.items())))
big = sorted(lines, key=lambda r: -r[3])[:25]
print("largest idle gaps in front of a launch (start, dur, queue, gap, kernel):")
for ts, d, q, gap, n in sorted(big):
    print("  %9.1f %7.1f %-3s %7.1f  %s" % (ts, d, q, gap, n))
