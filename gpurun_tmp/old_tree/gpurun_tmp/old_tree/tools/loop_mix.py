#!/usr/bin/env python
"""Instruction mix of the largest basic blocks of one kernel in a hipcc -S listing (tuning aid):
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s csrc/<file>.hip; python tools/loop_mix.py k.s <mangled-name-prefix> [n_blocks]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l.split(";")[0]][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], ("entry", [])
for l in lines[start:end]:
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
    else:
        cur[1].append(l)
blocks.append(cur)
blocks.sort(key=lambda b: -len(b[1]))
for lab, code in blocks[:int(sys.argv[3]) if len(sys.argv) > 3 else 2]:
    c = collections.Counter()
    for line in code:
        line = line.strip()
        if not line or line[0] in ";.":
            continue
        op = line.split()[0]
        if op.startswith("v_mfma"):
            c["mfma:" + op] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            c["v:" + op] += 1
        elif op.startswith(("buffer_", "global_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    print(lab, len(code), "lines")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:40]:
        print("  ", k, v)
