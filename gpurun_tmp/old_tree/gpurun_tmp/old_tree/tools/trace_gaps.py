#!/usr/bin/env python
"""Timeline of replayed steps from a rocprofv3 --kernel-trace csv (bench.py --inflight 1: one batch in flight, the kernels of a
step run back to back on one queue): per kernel its duration and the idle gap since the previous kernel ended.  The steps used
are the runs of consecutive steps whose start-to-start spacing is within 3 % of the median spacing (the replayed, timed region;
warm-up and the eager measurement legs have other spacings).
   python tools/trace_gaps.py <kernel_trace.csv> [--marker conv_in_multi_kernel] [--md profiles/<tag>_graph_step.md] [--all]"""
import collections
import csv
import statistics
import sys

args = sys.argv[1:]
path = args[0]
marker = args[args.index("--marker") + 1] if "--marker" in args else "conv_in_multi_kernel"
md = args[args.index("--md") + 1] if "--md" in args else None
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
spacing = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(starts[:-1], starts[1:])]
med = statistics.median(spacing)
steps = [rows[a:b] for (a, b), d in zip(zip(starts[:-1], starts[1:]), spacing) if abs(d - med) <= 0.03 * med]
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
lines = [f"{n} replayed steps (median start-to-start spacing {med:.1f} us); per step: kernels {tot_busy / n:.1f} us, idle gaps {tot_gap / n:.1f} us, "
         f"{sum(c for c, _, _ in agg.values()) / n:.0f} launches"]
table = [(k, c / n, d / c, d / n, g / n) for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))]
print(lines[0])
print(f"{'kernel':62s} {'calls':>5s} {'avg us':>8s} {'us/step':>8s} {'gap us/step':>11s}")
for k, c, a, u, g in table:
    print(f"{k:62s} {c:5.1f} {a:8.1f} {u:8.1f} {g:11.1f}")
if md:
    with open(md, "w") as f:
        f.write("# One batch in flight, HIP-graph replay: kernel timeline of a step (rocprofv3 --kernel-trace)\n\n")
        f.write("command: `rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 20 --warmup 3 --inflight 1 --no-cpu-baseline "
                "--no-extras --no-bf16-leg --min-seconds 0`, summarised by `tools/trace_gaps.py`.\n\n" + lines[0] + ".  In a replayed graph the "
                "kernels of the step follow each other without idle time on the queue (the dispatch-to-dispatch cost is inside each kernel's "
                "duration: the smallest kernels of the step take 4.6 us), so the step IS the sum of its kernel durations.\n\n")
        f.write("| kernel | launches / step | avg us | us / step | idle before, us / step |\n|---|---:|---:|---:|---:|\n")
        for k, c, a, u, g in table:
            f.write(f"| `{k}` | {c:.1f} | {a:.1f} | {u:.1f} | {g:.1f} |\n")
if "--all" in args:
    prev = None
    for s, e, name in steps[-1]:
        print(f"{(s - steps[-1][0][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {0.0 if prev is None else (s - prev) / 1e3:6.1f}  {name[:80]}")
        prev = e
