"""One steady-state step of the bench as a list of kernel dispatches (start, end, queue) from a rocprofv3 kernel trace:
    python scripts/step_gantt.py KERNEL_TRACE.csv [step_from_the_end=5]
The step = from one agc::group_lookup_kernel (one launch per sample, half a millisecond into its step) to the next.  Dispatches
shorter than 20 us are summed per name.  Shows what the driving chain of a step waits for."""
import csv, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"].replace("void ", "")
        n = n[:n.index("(")] if "(" in n else n
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.replace("agc::", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "")))
rows.sort()
looks = [r[0] for r in rows if r[2] == "group_lookup_kernel"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t0, t1 = looks[-k - 1], looks[-k]
print(f"step of {(t1 - t0) / 1e6:.2f} ms (group look-up to group look-up)")
small = collections.Counter()
for s, e, n, q, st in rows:
    if e <= t0 or s >= t1:
        continue
    if e - s < 20000:
        small[n] += e - s
        continue
    print(f"  {(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:6.3f})  queue {q:>3s} {st:>3s}  {n}")
print("  dispatches under 20 us, summed:", ", ".join(f"{n} {v / 1e3:.0f} us" for n, v in small.most_common(12)))
