#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc pass of SQ counters (issue / stall accounting of the hot kernels).
usage: tools/summarize_sq.py <rocprofv3 output dir> <out.md> "<command>"
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md):
WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    names = []
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k, c = r["Kernel_Name"], r["Counter_Name"]
            acc[k][c] += float(r["Counter_Value"])
            cnt[k][c] += 1
            if c not in names:
                names.append(c)
    rows = sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0.0))
    with open(out, "w") as o:
        o.write(f"# rocprofv3 --pmc {' '.join(names)} (one pass), per-launch averages\n\ncommand: `{cmd}`\n\n")
        o.write("| kernel | launches | " + " | ".join(names) + " | VALU busy = ACTIVE_INST_VALU / WAVE_CYCLES | parked = WAIT_ANY / WAVE_CYCLES "
                "| issue stall = WAIT_INST_ANY / WAVE_CYCLES | VALU insts per wave |\n")
        o.write("|---|---|" + "---|" * (len(names) + 4) + "\n")
        for k in rows[:14]:
            n = max(cnt[k].values())
            v = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
            wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            o.write("| `%s` | %d | " % (k[:60], n) + " | ".join("%.4g" % v.get(c, 0.0) for c in names)
                    + " | %.3f | %.3f | %.3f | %.0f |\n" % (v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, v.get("SQ_WAIT_ANY", 0.0) / wc,
                                                        v.get("SQ_WAIT_INST_ANY", 0.0) / wc, v.get("SQ_INSTS_VALU", 0.0) / (v.get("SQ_WAVES", 0.0) or 1.0)))
    print(open(out).read())


if __name__ == "__main__":
    main()
