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
    # optional: merge the issue accounting of the NTT passes into profiles/pmc_ntt.json (bench.py prints it inside `roofline_ntt`):
    #   tools/summarize_sq.py <pmc dir> <out.md> "<command>" <pmc_ntt.json> <kernel-trace dir of the same session>
    if len(sys.argv) > 5:
        import json
        import re
        vg = {}
        for f in glob.glob(sys.argv[5] + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                # the trace counts registers of a wave64 kernel in pairs (72 for the 144 the code object declares): x 2 = the allocation
                vg.setdefault(r["Kernel_Name"], 2 * (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])))
        per, num, den = {}, 0.0, 0.0
        for k in rows:
            m = re.search(r"ntt_pass_(mid|final)<.*?, (\d+)>", k)
            if not m:
                continue
            v = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
            n = max(cnt[k].values())
            wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            regs = vg.get(k, 0)
            occ = min(8, 512 // (-(-regs // 8) * 8)) if regs else None
            busy = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
            per[f"{m.group(1)}<{m.group(2)}>"] = {"launches": n, "valu_busy_per_wave": round(busy, 3), "vgprs": regs, "waves_per_simd": occ,
                                                  "valu_busy_per_simd": None if occ is None else round(busy * occ, 2)}
            if occ:
                num += busy * occ * wc * n
                den += wc * n
        d = json.load(open(sys.argv[4]))
        d["issue"] = {"what": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave x resident waves per SIMD (512 / VGPRs): ~1 = the vector pipe never idles "
                              "-- the passes are bound by instruction issue, not by HBM",
                      "valu_busy_per_simd_weighted": round(num / den, 2) if den else None, "kernels": per, "source": out}
        json.dump(d, open(sys.argv[4], "w"), indent=1)
        print(json.dumps(d["issue"], indent=1))


if __name__ == "__main__":
    main()
