"""One timed step of the bench as the runtime saw it: every HIP API call of every thread (with its duration) interleaved with the
kernels and copies that ran, from a `rocprofv3 --hip-trace --kernel-trace --memory-copy-trace` of bench.py.
    python scripts/hip_step_trace.py gpurun_out/r6/a_hiptrace/ht [step_from_the_end=2] [min_us=0]
The step = between the starts of two consecutive agc::group_lookup_kernel launches (one per sample)."""
import csv, sys, collections
pre = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0
def short(n):
    n = n.replace("void ", "")
    return n[:n.index("(")] if "(" in n else n
ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"], r["Correlation_Id"]) for r in csv.DictReader(open(pre + "_kernel_trace.csv"))]
ker.sort()
marks = [k[0] for k in ker if k[2] == "agc::group_lookup_kernel"]
t0, t1 = marks[-back - 1], marks[-back]
api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"], r["Correlation_Id"]) for r in csv.DictReader(open(pre + "_hip_api_trace.csv"))]
cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), r["Stream_Id"]) for r in csv.DictReader(open(pre + "_memory_copy_trace.csv"))]
ev = []
for s, e, n, st, c in ker:
    if e > t0 and s < t1:
        ev.append((s, "K", f"stream {st:>3s}  {n}", e - s))
for s, e, n, st in cp:
    if e > t0 and s < t1:
        ev.append((s, "C", f"stream {st:>3s}  {n}", e - s))
thr = collections.Counter()
for s, e, n, th, c in api:
    if e > t0 and s < t1:
        thr[th] += 1
names = {th: "T%d" % i for i, (th, _) in enumerate(thr.most_common())}
for s, e, n, th, c in api:
    if e > t0 and s < t1 and (e - s) / 1e3 >= min_us:
        ev.append((s, "A", f"{names[th]}  {n}", e - s))
ev.sort()
print(f"step window {(t1 - t0) / 1e6:.3f} ms; threads: {dict((names[t], c) for t, c in thr.items())}")
for s, k, what, d in ev:
    print(f"{(s - t0) / 1e6:9.3f} ms  {k}  {d / 1e3:9.1f} us  {what}")
agg = collections.Counter(); cnt = collections.Counter()
for s, e, n, th, c in api:
    if e > t0 and s < t1:
        agg[(names[th], n)] += e - s; cnt[(names[th], n)] += 1
print("API time by (thread, function):")
for (th, n), v in agg.most_common(30):
    print(f"   {th} {n:40s} {cnt[(th, n)]:5d} calls {v / 1e6:8.3f} ms")
