"""Timeline of the LAST sampling call in a rocprofv3 run_results.db (`rocprofv3 --kernel-trace -d DIR -o run -- python3 bench.py
--workload sample …`): per stream and kernel family, the intervals in which its dispatches ran (gaps below 200 us merged), offsets in
ms from the call's first dispatch. Usage: python tools/sample_timeline.py DB"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,stream_id from kernels order by start"))
# a call starts with the prep kernels: find the last flow_prep_invconv dispatch
starts = [i for i, r in enumerate(rows) if "flow_prep_invconv" in r[0]]
a = starts[-1]
while a > 0 and rows[a][1] - rows[a - 1][2] < 100e3 and "rev_chain" not in rows[a - 1][0]:
    a -= 1
t0 = rows[a][1]


def fam(n):
    for k, v in (("rev_chain", "chain"), ("sc_cond", "chain"), ("enc_gru", "encoder recurrence"), ("gemm_planes", "planes GEMM"),
                 ("gemm_bf16x3", "x W_ih GEMM"), ("planes_from", "planes split")):
        if k in n:
            return v
    return "other"


iv = {}
for name, st, en, sid in rows[a:]:
    key = (sid, fam(name))
    L = iv.setdefault(key, [])
    if L and st - L[-1][1] < 200e3:
        L[-1][1] = max(L[-1][1], en)
        L[-1][2] += en - st
        L[-1][3] += 1
    else:
        L.append([st, en, en - st, 1])
print("last call: %.2f ms from first to last dispatch" % ((max(r[2] for r in rows[a:]) - t0) / 1e6))
for (sid, f), L in sorted(iv.items(), key=lambda kv: kv[1][0][0]):
    if f == "other" and sum(x[2] for x in L) < 0.3e6:
        continue
    print("stream %d, %s:" % (sid, f))
    for st, en, busy, n in L:
        print("   %7.2f .. %7.2f ms  (%5.2f ms of kernels, %d dispatches)" % ((st - t0) / 1e6, (en - t0) / 1e6, busy / 1e6, n))
