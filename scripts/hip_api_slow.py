#!/usr/bin/env python3
"""scripts/hip_api_slow.py TRACE.csv [MIN_US] -- the slow HIP API calls of a rocprofv3 --hip-trace run, per thread (profiling aid)"""
import csv
import sys
from collections import defaultdict

fn, min_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = list(csv.DictReader(open(fn, newline="")))
per = defaultdict(lambda: [0, 0.0])
slow = []
for r in rows:
    name = r.get("Function") or r.get("Name") or ""
    t0, t1 = int(r.get("Start_Timestamp", 0)), int(r.get("End_Timestamp", 0))
    tid = r.get("Thread_Id", "")
    d = (t1 - t0) / 1e3
    per[(tid, name)][0] += 1
    per[(tid, name)][1] += d
    if d >= min_us:
        slow.append((t0, tid, name, d))
print("per (thread, call): count, total ms")
for (tid, name), (n, tot) in sorted(per.items(), key=lambda x: -x[1][1])[:30]:
    print(f"  {tid} {name:40s} {n:7d} {tot / 1e3:10.2f}")
slow.sort()
# the window of interest ends at the first hipStreamDestroy (teardown) -- or at argv[3] ms before the end of the trace
destroys = [int(r["Start_Timestamp"]) for r in rows if (r.get("Function") or r.get("Name") or "") == "hipStreamDestroy"]
t_end = min(destroys) if destroys else (slow[-1][0] if slow else 0)
win = float(sys.argv[3]) if len(sys.argv) > 3 else 120.0
print(f"calls >= {min_us} us in the {win} ms before the teardown:")
for t0, tid, name, d in slow:
    if t_end - win * 1e6 <= t0 < t_end:
        print(f"  t={(t0 - t_end) / 1e6:9.3f} ms  thread {tid}  {name:36s} {d / 1e3:8.3f} ms")
lk = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if (r.get("Function") or r.get("Name") or "") == "hipLaunchKernel")
if lk:
    print("hipLaunchKernel us: median", lk[len(lk) // 2] / 1e3, "p90", lk[len(lk) * 9 // 10] / 1e3, "p99", lk[len(lk) * 99 // 100] / 1e3, "max", lk[-1] / 1e3)
