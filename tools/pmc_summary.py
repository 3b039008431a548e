#!/usr/bin/env python
"""Per-kernel average of rocprofv3 --pmc counters from a rocpd .db (or counter_collection csv).
    python tools/pmc_summary.py gpurun_out/pmc1 [--match hv_] > profiles/r01_hough_pmc.csv
"""
import argparse
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" in names:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
        ccol = "counter_name" if "counter_name" in cols else None
        vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
        dcol = "dispatch_id" if "dispatch_id" in cols else None
        if ccol and vcol:
            q = "select %s, %s, %s, %s from counters_collection" % (kcol, ccol, vcol, dcol or "0")
            return list(cur.execute(q)), cols
    return [], names


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r.get("Kernel_Name"), r.get("Counter_Name"), float(r.get("Counter_Value", 0)), r.get("Dispatch_Id")))
    return out, []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--match", default="")
    a = ap.parse_args()
    files = [a.path] if os.path.isfile(a.path) else (
        glob.glob(os.path.join(a.path, "**", "*.db"), recursive=True) +
        glob.glob(os.path.join(a.path, "**", "*counter_collection.csv"), recursive=True))
    rows = []
    info = []
    for f in files:
        r, i = from_db(f) if f.endswith(".db") else from_csv(f)
        rows += r
        info += i
    if not rows:
        sys.exit("no counter rows found; tables/cols seen: %s" % info)
    agg = {}
    for k, c, v, d in rows:
        if a.match and a.match not in (k or ""):
            continue
        k = (k or "?").replace("(anonymous namespace)::", "")
        k = k[:k.find("(")] if "(" in k else k
        e = agg.setdefault((k[:80], c), {})
        e[d] = e.get(d, 0.0) + float(v)   # sum over XCDs / instances of one dispatch
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "avg_per_dispatch", "total"])
    for (k, c), e in sorted(agg.items()):
        tot = sum(e.values())
        w.writerow([k, c, len(e), "%.1f" % (tot / max(len(e), 1)), "%.1f" % tot])


if __name__ == "__main__":
    main()
