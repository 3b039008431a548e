#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd SQLite .db or *_kernel_trace.csv) into a small
CSV for profiles/: per-kernel calls / total / avg / min / max, optionally restricted to the last
`--tail-ms` milliseconds of the trace (the steady-state timed region; MIOpen's find/auto-tune
phase at start-up would otherwise dominate).

    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db --tail-ms 400 > profiles/r01_bench_kernel_stats.csv
"""
import argparse
import csv
import glob
import os
import sqlite3
import sys


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    return [(n, s, e) for n, s, e in cur.execute("select name, start, end from kernels")]


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out


def short(name):
    n = name
    for pre in ("void ", "(anonymous namespace)::"):
        n = n.replace(pre, "")
    cut = n.find("(")
    if cut > 0:
        n = n[:cut]
    return n[:96]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--tail-ms", type=float, default=0.0)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--marker", default="", help="kernel launched once per step (e.g. hv_emit_kernel): with --steps, "
                                                  "the window is exactly the last N steps")
    ap.add_argument("--steps", type=int, default=0)
    a = ap.parse_args()
    path = a.path
    if os.path.isdir(path):
        cands = glob.glob(os.path.join(path, "**", "*.db"), recursive=True) + glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        if not cands:
            sys.exit("no rocprofv3 output under %s" % path)
        path = cands[0]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    if not rows:
        sys.exit("empty trace")
    tmax = max(e for _, _, e in rows)
    span_note = ""
    if a.marker and a.steps > 0:
        marks = sorted(e for n, _, e in rows if a.marker in n)
        if len(marks) > a.steps:
            t0, t1 = marks[-a.steps - 1], marks[-1]
            rows = sorted((r for r in rows if r[1] >= t0 and r[2] <= t1), key=lambda r: r[1])
            # GPU busy time = union of kernel intervals; the largest idle gaps and what ran before them
            busy, cur_s, cur_e, gaps = 0, None, None, []
            for n, s_, e_ in rows:
                if cur_e is None:
                    cur_s, cur_e = s_, e_
                elif s_ > cur_e:
                    busy += cur_e - cur_s
                    gaps.append((s_ - cur_e, prev))
                    cur_s, cur_e = s_, e_
                else:
                    cur_e = max(cur_e, e_)
                prev = short(n)[:40]
            busy += cur_e - cur_s
            span = t1 - t0
            gaps.sort(reverse=True)
            span_note = " steps=%d span_ms=%.3f ms_per_step=%.3f gpu_busy=%.3f idle_ms_per_step=%.3f gaps>20us=%d largest_gaps_us=%s" % (
                a.steps, span / 1e6, span / 1e6 / a.steps, busy / span, (span - busy) / 1e6 / a.steps,
                sum(1 for g, _ in gaps if g > 20000), ";".join("%.0f after %s" % (g / 1e3, p) for g, p in gaps[:6]))
    elif a.tail_ms > 0:
        t0 = tmax - int(a.tail_ms * 1e6)
        rows = [r for r in rows if r[1] >= t0]
    agg = {}
    for n, s, e in rows:
        d = (e - s) / 1e3
        k = short(n)
        c = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        c[0] += 1; c[1] += d; c[2] = min(c[2], d); c[3] = max(c[3], d)
    total = sum(v[1] for v in agg.values())
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        w.writerow([k, v[0], "%.1f" % v[1], "%.2f" % (v[1] / v[0]), "%.2f" % v[2], "%.2f" % v[3], "%.2f" % (100 * v[1] / total)])
    w.writerow(["# window_ms=%s kernels_total_us=%.1f source=%s%s" % (a.tail_ms or "all", total, os.path.basename(path), span_note)])


if __name__ == "__main__":
    main()
