#!/usr/bin/env python
"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: per kernel name the busy time, and the idle gaps between
consecutive kernels (all streams merged) attributed to the kernel that FOLLOWS the gap.  Usage: trace_gaps.py t_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# the LAST run = everything after the last restart kernel (k_normalize4 with src != dst is the first kernel of hmx_restart)
starts = [i for i, r in enumerate(rows) if "k_normalize4" in r[2]]
if len(starts) >= 2:
    rows = rows[starts[-2]:starts[-1]] if len(starts) >= 3 else rows[starts[-1]:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = defaultdict(float); cnt = defaultdict(int); gap = defaultdict(float)
cover_end = rows[0][0]
union = 0.0
for s, e, n in rows:
    busy[n] += (e - s) / 1e3; cnt[n] += 1
    if s > cover_end:
        gap[n] += (s - cover_end) / 1e3
        cover_end_new = e
    union += max(0, e - max(s, cover_end)) / 1e3
    cover_end = max(cover_end, e)
print("window %.3f ms, kernels busy (union over streams) %.3f ms, idle %.3f ms, %d launches" % ((t1 - t0) / 1e6, union / 1e3, ((t1 - t0) / 1e3 - union) / 1e3, len(rows)))
print("%-62s %6s %10s %10s" % ("kernel", "calls", "busy us", "idle-before us"))
for n in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
    print("%-62s %6d %10.1f %10.1f" % (n, cnt[n], busy[n], gap[n]))

# the sequence of one clustering round in the middle of the run: from the end of a chain launch to the end of the next
ch = [i for i, r in enumerate(rows) if "k_tile<7, 4" in r[2] or ", 4, 2" in r[2]]
if len(ch) > 12:
    a, b = ch[9], ch[10]
    print("\none round, chain launch #9 (end) -> chain launch #10 (end); times in us relative to the end of #9")
    base = rows[a][1]
    for s_, e_, n_ in rows[a + 1:b + 1]:
        print("  %-58s start %8.1f  end %8.1f  (%.1f us)" % (n_, (s_ - base) / 1e3, (e_ - base) / 1e3, (e_ - s_) / 1e3))


# reference-arithmetic runs: one clustering round = k_ref_posord ... k_ref_posord (the round's shuffle opens it)
po = [i for i, r in enumerate(rows) if "k_ref_posord" in r[2]]
if len(po) > 8:
    a, b = po[6], po[7]
    base = rows[a][0]
    print("\nreference arithmetic: round #6 (k_ref_posord -> next k_ref_posord), %d launches, %.1f us" % (b - a, (rows[b][0] - base) / 1e3))
    agg = defaultdict(lambda: [0, 0.0, 0.0]); prev_end = rows[a][0]
    for s_, e_, n_ in rows[a:b]:
        g_ = agg[n_]; g_[0] += 1; g_[1] += (e_ - s_) / 1e3; g_[2] += max(0, s_ - prev_end) / 1e3; prev_end = max(prev_end, e_)
    for n_, (c_, bu_, ga_) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print("  %-58s x%-4d busy %9.1f us  idle-before %9.1f us" % (n_, c_, bu_, ga_))
    print("  first 70 launches of the round:")
    for s_, e_, n_ in rows[a:min(b, a + 70)]:
        print("    %-56s start %8.1f  (%.1f us)" % (n_, (s_ - base) / 1e3, (e_ - s_) / 1e3))
