#!/usr/bin/env python3
"""Effective clock of the long kernels: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, "DVFS give-back": within
3 % of the in-kernel clock on dispatches of 10 ms or more, reads high below ~0.3 ms; msm_accumulate runs 2.4 ms).
usage: tools/summarize_clock.py <rocprofv3 dir with counter_collection + kernel_trace csv> <out.md>"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    src, out = sys.argv[1], sys.argv[2]
    dur = {}
    for f in glob.glob(src + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    acc = defaultdict(list)
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
                continue
            d = dur.get(r["Dispatch_Id"])
            if d and d[0] > 300_000:
                acc[d[1]].append(float(r["Counter_Value"]) / 8.0 / d[0])      # cycles per ns = GHz
    with open(out, "w") as o:
        o.write("# Effective clock per kernel: GRBM_GUI_ACTIVE / 8 / duration (dispatches longer than 0.3 ms)\n\n| kernel | launches | GHz (mean) | min | max |\n|---|---|---|---|---|\n")
        for k, v in sorted(acc.items(), key=lambda kv: -len(kv[1])):
            o.write("| `%s` | %d | %.3f | %.3f | %.3f |\n" % (k[:70], len(v), sum(v) / len(v), min(v), max(v)))
    print(open(out).read())


if __name__ == "__main__":
    main()
