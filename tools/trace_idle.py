#!/usr/bin/env python3
"""Idle time between kernels per proof, from a rocprofv3 --kernel-trace CSV (appended to the bench profile summary).
usage: tools/trace_idle.py <rocprofv3 output dir>
A proof = PER consecutive accumulation launches (5 since round 4: one msm_accumulate_batch launch per group of PC calls; 29 before;
ZK_ACC_LAUNCHES_PER_PROOF overrides) with everything between its first NTT pass and its last reduction kernel."""
import csv
import glob
import os
import sys

PER = int(os.environ.get("ZK_ACC_LAUNCHES_PER_PROOF", "5"))


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    acc = [i for i, e in enumerate(ev) if "msm_accumulate" in e[2]]
    print("\n## Idle time between kernels (same trace)\n")
    print("| proof | dispatches | span ms | kernels busy ms | idle ms | largest gaps ms (each follows `msm_win_finish_q`: stream wait + host combine + next launch) |")
    print("|---|---|---|---|---|---|")
    for p in range(len(acc) // PER):
        first, last = acc[PER * p], acc[PER * p + PER - 1]
        lo = first
        while lo > 0 and ev[lo][0] - ev[lo - 1][1] < 1_000_000 and (p == 0 or lo - 1 > acc[PER * p - 1]):
            lo -= 1
        hi = last
        nxt = acc[PER * (p + 1)] if PER * (p + 1) < len(acc) else len(ev)
        while hi + 1 < nxt and ev[hi + 1][0] - ev[hi][1] < 1_000_000 and "ntt" not in ev[hi + 1][2] and "elementwise" not in ev[hi + 1][2]:
            hi += 1
        span = (ev[hi][1] - ev[lo][0]) / 1e6
        busy = sum(e[1] - e[0] for e in ev[lo:hi + 1]) / 1e6
        gaps = sorted(((ev[i + 1][0] - ev[i][1]) / 1e6 for i in range(lo, hi)), reverse=True)[:6]
        print("| %d | %d | %.2f | %.2f | %.2f | %s |" % (p, hi - lo + 1, span, busy, span - busy, ", ".join("%.2f" % g for g in gaps)))


if __name__ == "__main__":
    main()
