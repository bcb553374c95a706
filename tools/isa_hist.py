#!/usr/bin/env python3
"""Instruction histogram of the largest loop of one kernel in a hipcc -S listing.

    hipcc ... --cuda-device-only -S stft.hip -o /tmp/stft.s && python tools/isa_hist.py /tmp/stft.s <kernel-substring>
"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0]][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
loops.sort(reverse=True)
n, a, b = loops[0]
c = Counter()
for l in body[a:b + 1]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    c[l.split()[0]] += 1
tot = sum(c.values())
valu = sum(v for k, v in c.items() if k.startswith("v_"))
f64 = sum(v for k, v in c.items() if "f64" in k)
lds = sum(v for k, v in c.items() if k.startswith("ds_"))
print(f"loop lines {a}-{b} of {len(body)}: total {tot}  valu {valu}  f64 {f64}  lds {lds}  waitcnt {c['s_waitcnt']}")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{v:5d} {k}")
