#!/usr/bin/env python
"""Summaries of the clock traces dumped by the debug builds of tools/trace/*.patch (see README.md).
    python tools/trace/analyze_trace.py hv gpurun_out/hv_trace.bin
    python tools/trace/analyze_trace.py wm gpurun_out/wm_trace.bin conv2_1
    python tools/trace/analyze_trace.py bp gpurun_out/bp_trace.bin
Times are in 10-ns ticks of the 100 MHz counter in the file, microseconds in the output."""
import sys

import numpy as np

WM_WORKGROUPS = {"conv1_2": 19200, "conv2_1": 9600, "conv2_2": 9600, "conv3_1": 4800, "conv3_2": 4800, "conv3_3": 4800,
                 "conv4_1": 2400, "conv4_2": 2400, "conv4_3": 2400, "conv5_1": 640, "conv5_2": 640, "conv5_3": 640}   # 2 x 16 frames


def timeline(t0, t1, mask, base, span, label, step):
    for t in np.arange(0, span, step):
        tt = base + int(t * 100)
        c = (t0 <= tt) & (t1 > tt)
        print("   t=%6.0f us: resident %5d (%s %5d)" % (t, int(c.sum()), label, int((c & mask).sum())))


def hv(path):
    h = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    t0, t1 = h[:, 0].astype(np.int64), h[:, 1].astype(np.int64)
    ok = t0 > 0
    base = t0[ok].min()
    span = (t1.max() - base) / 100.0
    info = h[:, 2]
    nrec = (info >> np.uint64(40)).astype(np.int64) - 1
    tlook = ((info >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.int64) / 100.0
    dur = (t1 - t0) / 100.0
    live = info > 0
    w = nrec > 0
    print("hv_vote: span %.1f us; %d workgroups, %d live, %d with records in range" % (span, ok.sum(), live.sum(), w.sum()))
    print("   live durations sum %.0f us = %.1f us of 1024 full slots; mean %.2f median %.2f p90 %.2f max %.2f; range lookup %.2f us"
          % (dur[live].sum(), dur[live].sum() / 1024, dur[w].mean(), np.median(dur[w]), np.quantile(dur[w], .9), dur[w].max(), tlook[w].mean()))
    print("   last live workgroup ends at %.1f us; first empty one starts at %.1f" % ((t1[live].max() - base) / 100.0, (t0[~live & ok].min() - base) / 100.0 if (~live & ok).any() else -1))
    timeline(t0, t1, live, base, span, "live", 10.0)
    for lo, hi in ((1, 256), (256, 512), (512, 1024), (1024, 1536), (1536, 4096), (4096, 10 ** 6)):
        m = (nrec >= lo) & (nrec < hi)
        if m.sum():
            print("   records [%d, %d): %5d workgroups, %.2f us each" % (lo, hi, m.sum(), dur[m].mean()))


def wm(path, layer):
    n = WM_WORKGROUPS.get(layer, 65536)
    h = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)[:n]
    t0 = h[:, 0].astype(np.int64)
    ok = t0 > 0
    base = t0[ok].min()
    pro = (h[:, 1] & np.uint64(0xffffffff)).astype(np.int64) / 100.0
    loop = (h[:, 1] >> np.uint64(32)).astype(np.int64) / 100.0 - pro
    tot = h[:, 2].astype(np.int64) / 100.0
    end = t0 + h[:, 2].astype(np.int64)
    span = (end[ok].max() - base) / 100.0
    xcc = (h[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = (h[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    cu = xcc * 64 + ((hwid >> 13) & 7) * 16 + ((hwid >> 12) & 1) * 8 + ((hwid >> 8) & 0xf)
    print("%s: %d workgroups, span %.0f us; duration mean %.1f (p10 %.1f / p90 %.1f / max %.1f); prologue %.2f, K loop %.1f, epilogue %.2f; "
          "slot occupancy sum(dur) / (512 x span) = %.2f" % (layer, ok.sum(), span, tot[ok].mean(), np.quantile(tot[ok], .1), np.quantile(tot[ok], .9),
                                                           tot[ok].max(), pro[ok].mean(), loop[ok].mean(), (tot - pro - loop)[ok].mean(), tot[ok].sum() / (512 * span)))
    cnt = np.bincount(cu[ok]); cnt = cnt[cnt > 0]
    print("   per XCD %s; workgroups per CU min %d max %d over %d CUs; last start at %.0f us" % (np.bincount(xcc[ok], minlength=8).tolist(), cnt.min(), cnt.max(), len(cnt), (t0[ok].max() - base) / 100.0))
    print("   resident at 2 % .. 98 % of the span:", [int(((t0 <= base + int(f * span * 100)) & (end > base + int(f * span * 100)) & ok).sum()) for f in np.linspace(0.02, 0.98, 13)])


def bp(path):
    h = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    t0 = h[:, 0].astype(np.int64)
    ok = t0 > 0
    hit = (h[:, 3] >> np.uint64(63)).astype(bool)
    tA = (h[:, 1] & np.uint64(0xffffffff)).astype(np.int64) / 100.0
    tB = (h[:, 1] >> np.uint64(32)).astype(np.int64) / 100.0
    tend = h[:, 2].astype(np.int64) / 100.0
    nhit = (h[:, 3] & np.uint64(0xffff)).astype(np.int64)
    csum = ((h[:, 3] >> np.uint64(16)) & np.uint64(0xffffffff)).astype(np.int64)
    span = ((t0 + h[:, 2].astype(np.int64))[ok].max() - t0[ok].min()) / 100.0
    m, hh = ok & ~hit, ok & hit
    print("backproject: %d trips, span %.0f us, %.0f resident waves on average" % (ok.sum(), span, tend[ok].sum() / span))
    print("   all-miss trips %d: range trip %.2f us, total %.2f (p90 %.2f); %.0f %% of the wave-time" % (m.sum(), tA[m].mean(), tend[m].mean(), np.quantile(tend[m], .9), 100 * tend[m].sum() / tend[ok].sum()))
    print("   hit trips %d: A %.2f, B %.2f, C %.2f, total %.2f us (p90 %.2f); %.1f hit voxels, %.0f matches per trip; %.0f %% of the wave-time"
          % (hh.sum(), tA[hh].mean(), (tB - tA)[hh].mean(), (tend - tB)[hh].mean(), tend[hh].mean(), np.quantile(tend[hh], .9), nhit[hh].mean(), csum[hh].mean(), 100 * tend[hh].sum() / tend[ok].sum()))


if __name__ == "__main__":
    kind = sys.argv[1]
    if kind == "hv":
        hv(sys.argv[2])
    elif kind == "wm":
        wm(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    elif kind == "bp":
        bp(sys.argv[2])
    else:
        sys.exit(__doc__)
