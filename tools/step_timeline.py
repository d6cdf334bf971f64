"""Kernel timeline of one steady-state training step (the shortest of the timed steps: bench.py's last three are issued on an
empty queue to measure the host, and one step in ten is a negative step with its own preamble) in a rocprofv3 run_results.db (`rocprofv3 --kernel-trace -d DIR -- python3 bench.py …`):
start offset, duration and the idle gap before every dispatch, then totals per kernel name. Usage: python tools/step_timeline.py DB"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x,workgroup_x,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "adam_clip" in r[0]]
cand = list(range(1, len(idx) - 3)) or [len(idx) - 1]
best = min(cand, key=lambda i: rows[idx[i]][2] - rows[idx[i - 1]][2])
a, b = idx[best - 1] + 1, idx[best] + 1
t0 = rows[a][1]
last_end, gaps, busy = None, 0.0, 0.0
tot = defaultdict(lambda: [0, 0.0])
for name, st, en, gx, wx, sid in rows[a:b]:
    n = re.sub(r"_ZN\d+_GLOBAL__N_1\d+", "", name)[:70]
    g = (st - last_end) / 1e3 if last_end is not None else 0.0
    gaps += max(g, 0.0)
    busy += (en - st) / 1e3
    tot[n][0] += 1
    tot[n][1] += (en - st) / 1e3
    print("%8.1f %7.1f gap %6.1f  s%d %s grid %d wg %d" % ((st - t0) / 1e3, (en - st) / 1e3, g, sid, n, gx, wx))
    last_end = max(en, last_end or en)
print("span %.1f us, kernel time %.1f us, idle gaps %.1f us, %d dispatches" % ((rows[b - 1][2] - t0) / 1e3, busy, gaps, b - a))
for n, (k, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us %3d x  %s" % (t, k, n))
