#!/usr/bin/env python
"""Timeline of one replayed step from a rocprofv3 --kernel-trace csv: per kernel its duration and the idle gap since the previous
kernel ended (one batch in flight: the kernels of a step run back to back on one queue, so gaps are launch / dependency latency).
   python tools/trace_gaps.py <kernel_trace.csv> [marker-substring] [--all]"""
import collections
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "enc_prologue_kernel"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
# the last complete step: between the last two markers
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    steps.append(rows[a:b])
steps = steps[-8:]
agg = collections.OrderedDict()
tot_busy = tot_gap = 0.0
for st in steps:
    prev_end = None
    for s, e, name in st:
        key = name.split("(")[0][-60:]
        d = agg.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        if prev_end is not None:
            g = max(0.0, (s - prev_end) / 1e3)
            d[2] += g
            tot_gap += g
        tot_busy += (e - s) / 1e3
        prev_end = e if prev_end is None else max(prev_end, e)
n = len(steps)
print(f"{n} steps, per step: busy {tot_busy / n:.1f} us, gaps {tot_gap / n:.1f} us, span {(steps[-1][-1][1] - steps[-1][0][0]) / 1e3:.1f} us (last)")
print(f"{'kernel':62s} {'calls':>5s} {'avg us':>8s} {'us/step':>8s} {'gap before/step':>15s} {'avg gap':>8s}")
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k:62s} {c / n:5.1f} {d / c:8.1f} {d / n:8.1f} {g / n:15.1f} {g / c:8.2f}")
if "--all" in sys.argv:
    prev = None
    for s, e, name in steps[-1]:
        print(f"{(s - steps[-1][0][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {0.0 if prev is None else (s - prev) / 1e3:6.1f}  {name[:80]}")
        prev = e
