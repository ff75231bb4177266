#!/usr/bin/env python3
"""Which lap of a step grew when a step took longer?  Reads the stderr of a bench run made with AGC_AMD_LAPS=1 (one block of
`lap <name> <ms>` lines per sample on the driving thread, a block begins at `lap group map -> device`) and prints, for every
sample whose laps add up to more than 1.4 x the median, the laps that are more than 1 ms above their own median.
Usage: lap_outliers.py LAPS.txt"""
import re
import statistics
import sys

steps, cur = [], None
other = []
for ln in open(sys.argv[1], errors="replace"):
    m = re.match(r"\s*lap (.+?) ([0-9.e+-]+) ms\s*$", ln)
    if m and ln.startswith("  lap "):
        if m.group(1).startswith("group map"):
            cur = {}
            steps.append((cur, []))
        if cur is not None:
            cur[m.group(1)] = cur.get(m.group(1), 0.0) + float(m.group(2))
    elif steps and re.search(r"ensure:|arena:|pinned|grow", ln):
        steps[-1][1].append(ln.strip()[:160])
if not steps:
    sys.exit("no laps found")
tot = [sum(s.values()) for s, _ in steps]
med = statistics.median(tot)
names = sorted({k for s, _ in steps for k in s})
med_of = {k: statistics.median([s.get(k, 0.0) for s, _ in steps]) for k in names}
print(f"{len(steps)} samples, driving-thread laps add up to a median of {med:.2f} ms")
for i, ((s, notes), t) in enumerate(zip(steps, tot)):
    if t > 1.4 * med or notes:
        fat = {k: round(v, 2) for k, v in s.items() if v > med_of[k] + 1.0}
        print(f"sample {i}: {t:.2f} ms; above their median by > 1 ms: {fat}; notes: {notes}")
