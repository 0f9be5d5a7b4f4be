#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately) per kernel.
usage: tools/summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out prefix> "<command>"
Writes <out prefix>.md (all kernels) and profiles/pmc_msm_accumulate.json (read by bench.py for roofline.traffic).
Units and correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950
FETCH_SIZE counts 128-B requests at 64 B, so it is doubled; WRITE_SIZE is taken as read."""
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            s, c = acc.get(k, (0.0, 0))
            acc[k] = (s + float(r["Counter_Value"]), c + 1)
    return acc


def main():
    dfetch, dwrite, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "")
    fe, wr = per_kernel(dfetch, "FETCH_SIZE"), per_kernel(dwrite, "WRITE_SIZE")
    rows = []
    for k, (s, c) in fe.items():
        ws, wc = wr.get(k, (0.0, 0))
        rows.append((k, c, s / c, ws / wc if wc else 0.0))
    rows.sort(key=lambda r: -(2 * r[2] + r[3]) * r[1])
    with open(out + ".md", "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-launch averages, KiB\n\n")
        o.write(f"command: `{cmd}`\n\n| kernel | launches | FETCH_SIZE raw | x2 (gfx950 correction) | WRITE_SIZE | HBM bytes per launch |\n|---|---|---|---|---|---|\n")
        for k, c, f, w in rows[:16]:
            o.write("| `%s` | %d | %.0f | %.0f | %.0f | %.3e |\n" % (k[:70], c, f, 2 * f, w, (2 * f + w) * 1024))
    here = os.path.dirname(os.path.abspath(__file__))
    top = os.path.join(here, "..", "profiles")          # the two JSON files bench.py and tests/test_abi.py read live at the top of profiles/
    commit = "unknown"
    try:
        commit = open(os.path.join(here, "..", ".git_head")).read().strip() or "unknown"
    except OSError:
        pass
    import datetime
    collected = f"{datetime.date.today().isoformat()}, 1x MI355X (gpurun box), tools/collect_profiles.sh {os.path.basename(out).split('_')[0]}: the same session as " \
                f"{os.path.basename(out).split('_')[0]}_bench_n20.md / _pmc_sq.md / _pmc_clock.md (one discarded run first)"
    for k, c, f, w in rows:
        if "msm_accumulate" in k:
            # since round 4 one launch (msm_accumulate_batch) sums every MSM of a prover round: 5 launches and 29 MSMs per proof at the headline
            # schedule, so the per-launch average below goes with bench.py's per-launch average of algorithmic bytes (the same launch mix)
            json.dump({"kernel": (re.search(r"msm_accumulate\w*", k) or re.search(r"\w+", k)).group(0), "launches": c, "fetch_size_kib_raw": f, "write_size_kib": w,
                       "hbm_bytes_per_launch": (2 * f + w) * 1024,
                       "note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `{cmd}`; per-launch averages; "
                               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as read.",
                       "collected": collected, "commit": commit},
                      open(os.path.join(top, "pmc_msm_accumulate.json"), "w"), indent=1)
            # the NTT passes of the same run, per proof: proofs = accumulate launches / launches per proof (ZK_ACC_LAUNCHES_PER_PROOF, default 5)
            per_proof = int(os.environ.get("ZK_ACC_LAUNCHES_PER_PROOF", "5"))
            proofs = max(1, c // per_proof)
            tot = sum((2 * ff + ww) * 1024 * cc for kk, cc, ff, ww in rows if "ntt_pass" in kk)
            json.dump({"kernels": "ntt_pass_mid<S> / ntt_pass_final<S>, every launch of the run", "proofs_in_run": proofs,
                       "hbm_bytes_per_proof": tot / proofs,
                       "alg_bytes_per_proof_n20": 17 * 2 * 32 * (1 << 20) + 14 * 2 * 32 * (1 << 22),
                       "note": f"sum over the NTT pass kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB x launches, divided by the {proofs} proofs of `{cmd}` "
                               "(counted from the accumulate launches); separate --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md",
                       "collected": collected, "commit": commit},
                      open(os.path.join(top, "pmc_ntt.json"), "w"), indent=1)
            break
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
