"""busy / idle accounting of the bench's timed steps from a rocprofv3 kernel trace (start / end of every dispatch):
    python scripts/step_timeline.py gpurun_out/r5/t_ktrace/kt_kernel_trace.csv
The window analysed = from the first timed sample's conversion (agc::pack_fasta_count_kernel, one launch per sample: the last
`steps` of them lie in the timed region) to the start of the Close's entropy kernel.
Prints: wall of the window, time with at least one kernel running, time with NO kernel running, and per kernel its summed
duration and the part of it during which no other kernel ran."""
import csv, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        n = n.replace("void ", "")
        n = n[:n.index("(")] if "(" in n else n
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2] == "agc::scan_packed_kernel"]
zst = [i for i, r in enumerate(rows) if r[2].startswith("agc::zstd_frames_grp_kernel")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# the timed region starts with the conversion of its first sample (no pack is queued across the warm-up / timed boundary): the
# last `steps` launches of the counting pass; the window ends where Close's group kernel starts
# (round 6: the conversions are queued by the compressor in the middle of a step, two samples ahead -- the window starts at the timed
# region's first group look-up instead: one launch per sample, half a millisecond into its step)
looks = [i for i, r in enumerate(rows) if r[2] == "agc::group_lookup_kernel"]
packs = [i for i, r in enumerate(rows) if r[2] == "agc::pack_fasta_count_kernel"]
t0 = rows[looks[-steps]][0] if len(looks) >= steps else (rows[packs[-steps]][0] if len(packs) >= steps else rows[scans[-steps]][0])
t1 = rows[zst[-1]][0]
win = [(max(s, t0), min(e, t1), n) for s, e, n in rows if e > t0 and s < t1]
ev = []
for s, e, n in win:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
busy = idle = 0
alone = collections.Counter()
total = collections.Counter()
for s, e, n in win:
    total[n] += e - s
active = collections.Counter()
prev = t0
for t, d, n in ev:
    k = sum(active.values())
    if t > prev:
        if k == 0:
            idle += t - prev
        else:
            busy += t - prev
            if k == 1:
                alone[next(x for x, c in active.items() if c > 0)] += t - prev
    active[n] += d
    prev = t
if t1 > prev:
    idle += t1 - prev
ms = 1e-6
print(f"window {(t1 - t0) * ms:.2f} ms for {steps} steps = {(t1 - t0) * ms / steps:.2f} ms per step; some kernel running {busy * ms:.2f} ms "
      f"({busy / (t1 - t0):.1%}), none {idle * ms:.2f} ms ({idle / (t1 - t0):.1%})")
print(f"sum of kernel durations {sum(total.values()) * ms:.2f} ms = {sum(total.values()) * ms / steps:.2f} ms per step (kernels overlap)")
print(f"{'kernel':48s} {'ms/step':>8s} {'alone ms/step':>14s} {'launches/step':>14s}")
cnt = collections.Counter(n for _, _, n in win)
for n, v in total.most_common(24):
    print(f"{n[:48]:48s} {v * ms / steps:8.3f} {alone[n] * ms / steps:14.3f} {cnt[n] / steps:14.1f}")
# the longest spans without any kernel
gaps = []
prev_end = t0
for s, e, n in sorted(win):
    if s > prev_end:
        gaps.append((s - prev_end, prev_end - t0))
    prev_end = max(prev_end, e)
gaps.sort(reverse=True)
print("longest idle spans (ms, at ms after the window's start):", [(round(g * ms, 3), round(a * ms, 1)) for g, a in gaps[:12]])
print(f"idle spans: {len(gaps)}; {sum(1 for g, _ in gaps if g > 100000)} longer than 0.1 ms holding {sum(g for g, _ in gaps if g > 100000) * ms:.2f} ms")
