#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel means.

HBM bytes per dispatch (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte requests as 64 bytes for wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    bygrid = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            bygrid[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if len(sys.argv) > 2:
        # per (kernel, grid) HBM bytes per dispatch as JSON: bench.py looks its dominant kernel up here for roofline.traffic
        import json
        out = {}
        for (name, grid), cs in bygrid.items():
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                rd = 2.0 * 1024 * sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
                wr = 1024.0 * sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
                out["%s|grid=%d" % (name, grid)] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                                                     "dispatches": len(cs["FETCH_SIZE"])}
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read bytes = 2 x FETCH_SIZE KiB "
                             "(gfx950 128-byte requests tallied as 64), write bytes = WRITE_SIZE KiB; per dispatch means",
                   "kernels": out}, open(sys.argv[2], "w"), indent=1)
    rows = []
    for k, cs in agg.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        n = max(len(v) for v in cs.values())
        rd = 2.0 * m.get("FETCH_SIZE", float("nan")) * 1024
        wr = m.get("WRITE_SIZE", float("nan")) * 1024
        rows.append((k, n, rd, wr, m))
    rows.sort(key=lambda r: -(r[4].get("SQ_BUSY_CYCLES", 0) * r[1]))
    print("| kernel | dispatches/pass | HBM read MB (2x FETCH_SIZE) | HBM write MB | L2 hit % | MFMA busy % of SQ busy | LDS conflict % |")
    print("|---|---|---|---|---|---|---|")
    for k, n, rd, wr, m in rows[:30]:
        hit = 100 * m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES is per-SE... report the raw ratio basis:
        mf = 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(4.0 * m.get("SQ_WAVE_CYCLES", 0), 1)
        ldc = 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1)
        print("| %s | %d | %.1f | %.1f | %.1f | %.1f | %.1f |" % (k[:100], n, rd / 1e6, wr / 1e6, hit, mf, ldc))
    print("\nMFMA column = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES) (wave-cycles are quad-cycles): the share of "
          "wave lifetime with the MFMA pipe busy, not chip utilisation.")


if __name__ == "__main__":
    main()
