#!/usr/bin/env python3
"""Copy a rocprofv3 --kernel-trace --stats run into profiles/ as a small per-kernel summary (CSV + md).
usage: tools/summarize_rocprof.py gpurun_out/prof3 profiles/r01/r01_bench_n20 "command line that was profiled"
"""
import csv
import glob
import sys


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tr = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
    res = {}
    if tr:
        for r in csv.DictReader(open(tr[0])):
            res.setdefault(r["Kernel_Name"], (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                                             r["Workgroup_Size_X"], r["Grid_Size_X"]))
    with open(dst + ".csv", "w") as o:
        o.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,pct,vgpr,agpr,sgpr,lds,scratch,wg,grid\n")
        for r in rows:
            x = res.get(r["Name"], ("",) * 7)
            o.write('"%s",%s,%.4f,%.4f,%.4f,%.4f,%s,%s\n' % (r["Name"].replace('"', "'"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                    float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"], ",".join(x)))
    with open(dst + ".md", "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats\n\ncommand: `{cmd}`\n\n| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:25]:
            o.write("| `%s` | %s | %.3f | %.4f | %s |\n" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6,
                                                       r["Percentage"]))
    print(open(dst + ".md").read())


if __name__ == "__main__":
    main()
