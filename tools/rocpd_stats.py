#!/usr/bin/env python3
"""Per-kernel summary (calls, total ms, average us, %) from a rocprofv3 rocpd SQLite file or a *_kernel_stats.csv."""
import csv
import sqlite3
import sys


def from_db(path):
    con = sqlite3.connect(path)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = con.execute("select s.kernel_name, count(*), sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id "
                       "group by s.kernel_name order by 3 desc" % (disp, sym)).fetchall()
    return [(n, c, t) for n, c, t in rows]


def from_csv(path):
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(path))]


def demangle(names):
    """Itanium-mangled kernel symbols (rocpd keeps them raw) -> readable names, through c++filt when it is there."""
    import shutil
    import subprocess
    if not shutil.which("c++filt"):
        return names
    raw = [n[:-3] if n.endswith(".kd") else n for n in names]
    out = subprocess.run(["c++filt"], input="\n".join(raw), capture_output=True, text=True).stdout.split("\n")
    return [o if o else n for o, n in zip(out, raw)]


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    names = demangle([r[0] for r in rows])
    rows = [(n, r[1], r[2]) for n, r in zip(names, rows)]
    tot = sum(r[2] for r in rows)
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for n, c, t in rows[:top]:
        n = n.replace("(anonymous namespace)::", "")
        print("| %s | %d | %.2f | %.1f | %.1f |" % (n[:110], c, t / 1e6, t / c / 1e3, 100 * t / tot))
    print("\ntotal kernel time %.2f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main()
