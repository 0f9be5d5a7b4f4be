"""Is the card at its power cap under the prover's kernels?  Samples the amdgpu hwmon files (socket power, sclk, temperatures, the
cap) every few milliseconds while a child command runs, and prints a summary + a coarse timeline.
usage (GPU box, repository root):  python tools/power_probe.py OUT.txt -- python bench.py --steps 40 ...
The sampler never touches the GPU itself (plain sysfs reads)."""
import glob
import os
import subprocess
import sys
import threading
import time


def find_hwmon():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        files = {f: os.path.join(d, f) for f in os.listdir(d)}
        out.append((d, files))
    return out


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    hw = find_hwmon()
    lines = []
    if not hw:
        lines.append("no amdgpu hwmon directory found")
    all_samples = []
    stop = threading.Event()
    want = ["power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input"]

    def sampler():
        while not stop.is_set():
            t = time.perf_counter()
            row = [t]
            for d, files in hw:
                for k in want:
                    v = read(files[k]) if k in files else None
                    row.append(int(v) if v and v.lstrip("-").isdigit() else None)
            all_samples.append(row)
            time.sleep(0.004)

    for d, files in hw:
        lines.append(f"hwmon: {d}")
        for k in sorted(files):
            if k.startswith(("power1_cap", "power1_label", "freq1_label", "freq2_label", "temp1_label", "temp2_label", "temp3_label", "temp1_crit", "name")):
                lines.append(f"  {k} = {read(files[k])}")
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(0.3)      # idle baseline
    t0 = time.perf_counter()
    rc = subprocess.call(cmd)
    t1 = time.perf_counter()
    time.sleep(0.2)
    stop.set()
    th.join()
    # the card the command ran on = the one whose power moved most (a box shows every card of its host)
    nw = len(want)
    best, best_span = 0, -1.0
    for h in range(len(hw)):
        v = [r[1 + h * nw + want.index(k)] for r in all_samples for k in ("power1_average", "power1_input") if r[1 + h * nw + want.index(k)] is not None]
        span = (max(v) - min(v)) if v else -1.0
        lines.append(f"  card {h} ({hw[h][0]}): power span {span / 1e6:.1f} W")
        if span > best_span:
            best, best_span = h, span
    lines.append(f"reporting card {best}")
    samples = [[r[0]] + r[1 + best * nw: 1 + (best + 1) * nw] for r in all_samples]
    lines.append(f"command: {' '.join(cmd)}  rc={rc}  wall {t1 - t0:.2f} s, {len(samples)} samples")
    col = {k: i + 1 for i, k in enumerate(want)}

    def stats(key, lo, hi, scale):
        v = [r[col[key]] for r in samples if lo <= r[0] <= hi and r[col[key]] is not None]
        if not v:
            return None
        v.sort()
        return (v[0] / scale, v[len(v) // 2] / scale, sum(v) / len(v) / scale, v[-1] / scale, len(v))

    for key, scale, unit in (("power1_average", 1e6, "W"), ("power1_input", 1e6, "W"), ("freq1_input", 1e6, "MHz"), ("freq2_input", 1e6, "MHz"),
                             ("temp1_input", 1e3, "C"), ("temp2_input", 1e3, "C"), ("temp3_input", 1e3, "C")):
        a = stats(key, samples[0][0], t0, scale) if samples else None
        b = stats(key, t0, t1, scale)
        if b:
            lines.append(f"{key:16s} [{unit}] before: {a}  during (min, median, mean, max, n): {tuple(round(x, 1) for x in b)}")
    # timeline: 40 bins over the run
    nb = 40
    lines.append("timeline (bin start s: mean power W, mean sclk MHz)")
    for b in range(nb):
        lo = t0 + (t1 - t0) * b / nb
        hi = t0 + (t1 - t0) * (b + 1) / nb
        pw = stats("power1_average", lo, hi, 1e6) or stats("power1_input", lo, hi, 1e6)
        fq = stats("freq1_input", lo, hi, 1e6)
        lines.append(f"  {lo - t0:7.2f}: {pw[2] if pw else float('nan'):8.1f} {fq[2] if fq else float('nan'):8.1f}")
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    sys.exit(rc)


if __name__ == "__main__":
    main()
