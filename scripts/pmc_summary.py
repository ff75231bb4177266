#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc output directories into the per-kernel summary kept under profiles/.

    python scripts/pmc_summary.py OUT.csv DIR [DIR ...]

Every DIR is the -d directory of one `rocprofv3 --pmc <counters> --kernel-include-regex agc -- <cmd>` pass (separate
passes per counter group, as MI355X_MICROARCH.md's rocprofv3 section prescribes).  A dispatch may be reported in several
rows (one per counter instance / dimension): rows are summed per (dispatch, counter) first, then mean and max are taken
over the dispatches of a kernel.  Units are the profiler's (FETCH_SIZE / WRITE_SIZE: KB per dispatch)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    return name.strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    per = defaultdict(lambda: defaultdict(float))   # (kernel, counter) -> dispatch id -> value
    for d in dirs:
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(fn, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row.get("Kernel_Name", ""))
                    c = row.get("Counter_Name", "")
                    did = (fn, row.get("Dispatch_Id", row.get("Correlation_Id", "")))
                    try:
                        v = float(row.get("Counter_Value", "nan"))
                    except ValueError:
                        continue
                    per[(k, c)][did] += v
    with open(out, "w") as f:
        f.write("kernel,counter,dispatches,mean,max\n")
        for (k, c) in sorted(per):
            vals = list(per[(k, c)].values())
            f.write(f"{k},{c},{len(vals)},{sum(vals) / len(vals):.4f},{max(vals):.4f}\n")
    print(f"{out}: {len(per)} (kernel, counter) rows from {len(dirs)} pass(es)")


if __name__ == "__main__":
    main()
