"""Per-kernel register / scratch / LDS / occupancy table of one HIP source, from hipcc's own remarks
(`-Rpass-analysis=kernel-resource-usage`, device code only).  `python tools/kernel_resources.py msm_accumulate.hip [-DZK_CURVE_SEL=0] [filter]`"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "ark_plonk_amd", "csrc")


def main():
    src = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    flt = [a for a in sys.argv[2:] if not a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=1000000", "--cuda-device-only",
           "-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + defs
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"^void ", "", name)
            cur = {"name": name.split("(")[0][:70]}
            rows.append(cur)
            continue
        m = re.search(r"remark: .*?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'scratch':>7s} {'LDS':>7s} {'occ':>3s}")
    for r in rows:
        if flt and not any(f in r["name"] for f in flt):
            continue
        print(f"{r['name']:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('VGPRs Spill', 0):5d} {r.get('ScratchSize', 0):7d} "
              f"{r.get('LDS Size', 0):7d} {r.get('Occupancy', 0):3d}")


if __name__ == "__main__":
    main()
