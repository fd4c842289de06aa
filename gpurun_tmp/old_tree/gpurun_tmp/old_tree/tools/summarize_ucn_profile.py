#!/usr/bin/env python
"""Per-kernel table of ONE pass of the UCN path out of a rocprofv3 kernel trace of tools/ucn_step.py.
   python tools/summarize_ucn_profile.py <prof_dir> <tag> <precision> [more "<prof_dir>:<precision>" ...]   ->   profiles/<tag>_kernel_stats_ucn.md"""
import collections
import csv
import glob
import os
import sys


def one_pass(prof):
    tr = sorted(csv.DictReader(open(glob.glob(os.path.join(prof, "**", "*kernel_trace.csv"), recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
    # a pass starts after the marker launch tools/ucn_step.py puts in front of every timed pass (torch.cuda._sleep: spin_kernel)
    first = tr[0]["Kernel_Name"]
    names = [r["Kernel_Name"] for r in tr]
    cut = [i for i, n in enumerate(names) if "spin_kernel" in n]
    segs = [tr[x + 1:y] for x, y in zip(cut, cut[1:] + [len(tr)])]
    modal = collections.Counter(len(s) for s in segs).most_common(1)[0][0]
    segs = [s for s in segs if len(s) == modal]
    segs = segs[1:] if len(segs) > 1 else segs       # drop the first plain pass (cold caches)
    per = collections.OrderedDict()
    for s in segs:
        for r in s:
            per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    busy = sum(sum(v) for v in per.values()) / len(segs)
    span = sum(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in segs) / len(segs)
    return per, len(segs), modal, busy, span, first


prof, tag, prec = sys.argv[1:4]
runs = [(prof, prec)] + [tuple(x.rsplit(":", 1)) for x in sys.argv[4:]]
os.makedirs("profiles", exist_ok=True)
with open(f"profiles/{tag}_kernel_stats_ucn.md", "w") as f:
    f.write(f"# UCN path, one pass of 2 images at 307 200 keys: per-kernel time ({tag})\n\n"
            "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python tools/ucn_step.py --precision <plan> --steps 10` "
            "(eager passes, synchronised one at a time; the trace is cut at a marker launch in front of each pass, the table is the mean over the plain passes)\n")
    for p, pr in runs:
        per, n, modal, busy, span, _ = one_pass(p)
        f.write(f"\n## plan `{pr}`: {busy / 1e3:.0f} us of kernels per pass in {modal} launches (first start to last end {span / 1e3:.0f} us eager; mean of {n} passes)\n\n")
        f.write("| kernel | launches per pass | avg us | us per pass | % of kernel time |\n|---|---:|---:|---:|---:|\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"| `{k[:110]}` | {len(v) / n:.0f} | {sum(v) / len(v) / 1e3:.1f} | {sum(v) / n / 1e3:.1f} | {100 * sum(v) / n / busy:.1f} |\n")
print(open(f"profiles/{tag}_kernel_stats_ucn.md").read()[:600])
